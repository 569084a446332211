"""Static description of the models on the recognition hot path.

Pure data: configurations, the state_dict key -> shape tables of the reference
checkpoints, and seeded synthetic parameter generation.  No arithmetic of the
path lives here, so both the product (``mtgv``) and the checker (``oracle/``)
may import it.

Reference surfaces described here (paths relative to /root/reference):
  * AE encoder key layout  - mtgvision/models/convnextv2ae.py:193-250
  * plain ConvNeXtV2 keys   - mtgvision/models/convnextv2.py:250-281
  * model-name -> factory   - mtgvision/encoder_train.py:52-67 (`_MODELS`),
    depths/dims tables       mtgvision/models/convnextv2ae.py:484-541
  * Z_SIZE = 768            - mtgvision/encoder_train.py:41
"""

from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field, asdict
from typing import Dict, Tuple

import numpy as np

Z_SIZE = 768  # encoder_train.py:41

HEAD_TYPES = ("conv+linear", "conv+mlp", "conv+act+mlp", "pool+linear", "pool+mlp", "plain")
ACTS = ("gelu", "mish")


@dataclass(frozen=True)
class EncoderConfig:
    """One ConvNeXt-V2 encoder variant.

    kind == "ae":    ConvNeXtV2Encoder (convnextv2ae.py:159-266) - Mish blocks,
                     x*2-1 input scaling, conv/pool heads.
    kind == "plain": ConvNeXtV2 (convnextv2.py:227-303) - GELU blocks, GAP ->
                     nn.LayerNorm -> Linear head (head_type is "plain").
    """

    kind: str = "ae"
    image_hw: Tuple[int, int] = (192, 128)
    in_chans: int = 3
    z_size: int = Z_SIZE
    depths: Tuple[int, int, int, int] = (3, 3, 9, 3)
    dims: Tuple[int, int, int, int] = (96, 192, 384, 768)
    head_type: str = "conv+linear"
    scale_io: bool = True

    def __post_init__(self):
        # same failure modes as the reference constructors
        if self.kind not in ("ae", "plain"):
            raise KeyError(f"kind={self.kind} not recognized")
        if self.kind == "ae":
            if self.head_type not in HEAD_TYPES[:5]:
                raise KeyError(f"head_type={self.head_type} not recognized")  # convnextv2ae.py:249-250
            h, w = self.image_hw
            assert h % 32 == 0 and w % 32 == 0  # convnextv2ae.py:137-139
            assert self.z_size % self.internal_num == 0  # convnextv2ae.py:126
        else:
            object.__setattr__(self, "head_type", "plain")
        assert len(self.depths) == 4 and len(self.dims) == 4

    @property
    def act(self) -> str:
        return "mish" if self.kind == "ae" else "gelu"

    @property
    def internal_hw(self) -> Tuple[int, int]:
        return self.image_hw[0] // 32, self.image_hw[1] // 32

    @property
    def internal_num(self) -> int:
        ih, iw = self.internal_hw
        return ih * iw

    @property
    def stage_hw(self):
        h, w = self.image_hw
        return [(h // s, w // s) for s in (4, 8, 16, 32)]

    def to_dict(self):
        return asdict(self)


_AE_SIZES = {  # convnextv2ae.py:484-541
    "atto": ((2, 2, 6, 2), (40, 80, 160, 320)),
    "femto": ((2, 2, 6, 2), (48, 96, 192, 384)),
    "pico": ((2, 2, 6, 2), (64, 128, 256, 512)),
    "nano": ((2, 2, 8, 2), (80, 160, 320, 640)),
    "tiny": ((3, 3, 9, 3), (96, 192, 384, 768)),
    "tiny_9_128": ((3, 3, 9, 3), (128, 256, 384, 768)),
    "tiny_12_128": ((3, 3, 12, 3), (128, 256, 384, 768)),
    "base_9": ((3, 3, 9, 3), (128, 256, 512, 1024)),
    "base_12": ((3, 3, 12, 3), (128, 256, 512, 1024)),
    "base": ((3, 3, 27, 3), (128, 256, 512, 1024)),
    "large": ((3, 3, 27, 3), (192, 384, 768, 1536)),
    "huge": ((3, 3, 27, 3), (352, 704, 1408, 2816)),
}

_PLAIN_SIZES = {  # convnextv2.py:306-343
    "atto": _AE_SIZES["atto"],
    "femto": _AE_SIZES["femto"],
    "pico": _AE_SIZES["pico"],
    "nano": _AE_SIZES["nano"],
    "tiny": _AE_SIZES["tiny"],
    "base": _AE_SIZES["base"],
    "large": _AE_SIZES["large"],
    "huge": _AE_SIZES["huge"],
}


def encoder_config(model_name: str, x_size_hw=(192, 128), head_type: str = "conv+linear", z_size: int = Z_SIZE) -> EncoderConfig:
    """`_MODELS[model_name](x_size_hw, head_type=...)` of encoder_train.py:52-67, 268-285.

    Names: "cnvnxt2ae_<size>" (AE encoder) and "convnextv2_<size>" (plain net,
    ``num_classes=z_size``).  Unknown names raise KeyError like the reference.
    """
    if model_name.startswith("cnvnxt2ae_"):
        size = model_name[len("cnvnxt2ae_"):]
        depths, dims = _AE_SIZES[size]
        return EncoderConfig("ae", tuple(x_size_hw), 3, z_size, depths, dims, head_type, True)
    if model_name.startswith("convnextv2_"):
        size = model_name[len("convnextv2_"):]
        depths, dims = _PLAIN_SIZES[size]
        return EncoderConfig("plain", tuple(x_size_hw), 3, z_size, depths, dims, "plain", False)
    raise KeyError(model_name)


# ----------------------------------------------------------------------------
# state_dict key tables
# ----------------------------------------------------------------------------


def _block_keys(prefix: str, c: int) -> "OrderedDict[str, tuple]":
    # Block, convnextv2.py:198-207 (+ GRN :168-169)
    return OrderedDict(
        [
            (f"{prefix}.dwconv.weight", (c, 1, 7, 7)),
            (f"{prefix}.dwconv.bias", (c,)),
            (f"{prefix}.norm.weight", (c,)),
            (f"{prefix}.norm.bias", (c,)),
            (f"{prefix}.pwconv1.weight", (4 * c, c)),
            (f"{prefix}.pwconv1.bias", (4 * c,)),
            (f"{prefix}.grn.gamma", (1, 1, 1, 4 * c)),
            (f"{prefix}.grn.beta", (1, 1, 1, 4 * c)),
            (f"{prefix}.pwconv2.weight", (c, 4 * c)),
            (f"{prefix}.pwconv2.bias", (c,)),
        ]
    )


def encoder_param_shapes(cfg: EncoderConfig) -> "OrderedDict[str, tuple]":
    """state_dict keys (in module order) and shapes of the encoder `cfg` describes.

    For kind "ae" these are the keys of ``ConvNeXtV2Encoder`` (strip the
    ``model.encoder.`` prefix of a Lightning checkpoint, encoder_train.py:263-288);
    for "plain" the keys of ``ConvNeXtV2``.
    """
    d, c = cfg.depths, cfg.dims
    out: "OrderedDict[str, tuple]" = OrderedDict()
    if cfg.kind == "ae":
        out["block0.0.weight"] = (c[0], cfg.in_chans, 4, 4)
        out["block0.0.bias"] = (c[0],)
        out["block0.1.weight"] = (c[0],)
        out["block0.1.bias"] = (c[0],)
        for j in range(d[0]):
            out.update(_block_keys(f"block0.2.{j}", c[0]))
        for s in (1, 2, 3):
            out[f"block{s}.0.weight"] = (c[s - 1],)
            out[f"block{s}.0.bias"] = (c[s - 1],)
            out[f"block{s}.1.weight"] = (c[s], c[s - 1], 2, 2)
            out[f"block{s}.1.bias"] = (c[s],)
            for j in range(d[s]):
                out.update(_block_keys(f"block{s}.2.{j}", c[s]))
        z = cfg.z_size
        if cfg.head_type.startswith("conv"):
            zc = z // cfg.internal_num
            out["pool.0.weight"] = (zc, c[3], 1, 1)
            out["pool.0.bias"] = (zc,)
            out["pool.2.weight"] = (zc,)
            out["pool.2.bias"] = (zc,)
            head_in = z
        else:
            out["pool.1.weight"] = (c[3],)
            out["pool.1.bias"] = (c[3],)
            head_in = c[3]
        if cfg.head_type.endswith("+mlp"):
            out["head.layers.0.weight"] = (z, head_in)
            out["head.layers.0.bias"] = (z,)
            out["head.layers.2.weight"] = (z, z)
            out["head.layers.2.bias"] = (z,)
        else:
            out["head.weight"] = (z, head_in)
            out["head.bias"] = (z,)
    else:
        out["downsample_layers.0.0.weight"] = (c[0], cfg.in_chans, 4, 4)
        out["downsample_layers.0.0.bias"] = (c[0],)
        out["downsample_layers.0.1.weight"] = (c[0],)
        out["downsample_layers.0.1.bias"] = (c[0],)
        for s in (1, 2, 3):
            out[f"downsample_layers.{s}.0.weight"] = (c[s - 1],)
            out[f"downsample_layers.{s}.0.bias"] = (c[s - 1],)
            out[f"downsample_layers.{s}.1.weight"] = (c[s], c[s - 1], 2, 2)
            out[f"downsample_layers.{s}.1.bias"] = (c[s],)
        for s in range(4):
            for j in range(d[s]):
                out.update(_block_keys(f"stages.{s}.{j}", c[s]))
        out["norm.weight"] = (c[3],)
        out["norm.bias"] = (c[3],)
        out["head.weight"] = (cfg.z_size, c[3])
        out["head.bias"] = (cfg.z_size,)
    return out


def _fan_in(shape) -> int:
    n = 1
    for s in shape[1:]:
        n *= s
    return max(n, 1)


def random_encoder_state(cfg: EncoderConfig, seed: int) -> Dict[str, np.ndarray]:
    """Seeded synthetic parameters, every tensor non-trivial.

    The reference zero-initialises GRN gamma/beta and all biases
    (convnextv2.py:168-169, convnextv2ae.py:144-147), which would leave GRN and
    the bias adds untested, so all of them are randomised here.  Drawn with
    ``numpy.random.default_rng(seed)`` in key order so that any machine
    regenerates the same tensors.
    """
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = OrderedDict()
    for key, shape in encoder_param_shapes(cfg).items():
        leaf = key.rsplit(".", 1)[-1]
        is_matrix = len(shape) >= 2 and leaf == "weight"
        if is_matrix:
            a = rng.standard_normal(shape) / np.sqrt(_fan_in(shape))
        elif leaf == "weight":  # norm scale
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        elif leaf == "gamma":
            a = 0.3 * rng.standard_normal(shape)
        else:  # biases, norm shift, grn beta
            a = 0.1 * rng.standard_normal(shape)
        sd[key] = np.ascontiguousarray(a, dtype=np.float32)
    return sd


def strip_checkpoint_prefix(state_dict, prefix: str = "model.encoder."):
    """Lightning ``ckpt["state_dict"]`` -> encoder keys (encoder_train.py:263-288)."""
    out = OrderedDict()
    for k, v in state_dict.items():
        if k.startswith(prefix):
            out[k[len(prefix):]] = v
    return out if out else OrderedDict(state_dict)


# ----------------------------------------------------------------------------
# Detector: YOLOv8n-seg (ultralytics 8.3.x, third-party to the reference; call
# sites mtgvision/od_export.py:141-160, od_train.py:46-70).
# ----------------------------------------------------------------------------


@dataclass(frozen=True)
class DetectorConfig:
    # "v8": yolov8n-seg (BASELINE.json); "11": yolo11n-seg, what the reference trains by default (od_train.py:20, :55-56)
    arch: str = "v8"
    nc: int = 3  # od_train.py:46-50
    nm: int = 32  # mask coefficients
    npr: int = 64  # proto channels (256 * width 0.25)
    reg_max: int = 16
    imgsz: int = 640
    width: float = 0.25
    depth: float = 0.33
    max_ch: int = 1024
    conf: float = 0.25
    iou: float = 0.7
    max_det: int = 300
    max_wh: float = 7680.0
    max_nms: int = 30000
    bn_eps: float = 1e-3

    def ch(self, c: int) -> int:
        # make_divisible(min(c, max_ch) * width, 8)
        v = min(c, self.max_ch) * self.width
        return int(-(-v // 8) * 8)

    def rep(self, n: int) -> int:
        return max(round(n * self.depth), 1) if n > 1 else n

    @property
    def head_index(self) -> int:
        return 23 if self.arch == "11" else 22

    @property
    def num_anchors(self) -> int:
        return sum((self.imgsz // s) ** 2 for s in (8, 16, 32))

    @property
    def no(self) -> int:
        return 4 + self.nc + self.nm


# ---- YOLOv8n-seg graph (ultralytics cfg/models/v8/yolov8-seg.yaml, scale "n") ----------
# (index, kind, args): Conv(cout,k,s) | C2f(cout,n,shortcut) | SPPF(cout) | Upsample | Concat(srcs)
def yolov8_seg_graph(cfg: DetectorConfig):
    c, r = cfg.ch, cfg.rep
    return [
        (0, "Conv", (c(64), 3, 2)),
        (1, "Conv", (c(128), 3, 2)),
        (2, "C2f", (c(128), r(3), True)),
        (3, "Conv", (c(256), 3, 2)),
        (4, "C2f", (c(256), r(6), True)),
        (5, "Conv", (c(512), 3, 2)),
        (6, "C2f", (c(512), r(6), True)),
        (7, "Conv", (c(1024), 3, 2)),
        (8, "C2f", (c(1024), r(3), True)),
        (9, "SPPF", (c(1024),)),
        (10, "Upsample", ()),
        (11, "Concat", (10, 6)),
        (12, "C2f", (c(512), r(3), False)),
        (13, "Upsample", ()),
        (14, "Concat", (13, 4)),
        (15, "C2f", (c(256), r(3), False)),
        (16, "Conv", (c(256), 3, 2)),
        (17, "Concat", (16, 12)),
        (18, "C2f", (c(512), r(3), False)),
        (19, "Conv", (c(512), 3, 2)),
        (20, "Concat", (19, 9)),
        (21, "C2f", (c(1024), r(3), False)),
    ]


def yolo11_config(**kw) -> DetectorConfig:
    """yolo11n-seg: scale "n" of ultralytics cfg/models/11/yolo11-seg.yaml = depth 0.50, width 0.25, max_channels 1024"""
    return DetectorConfig(arch="11", depth=0.50, **kw)


def detector_config_for_state(state_dict, **kw) -> DetectorConfig:
    """The scale-"n" family a ultralytics `state_dict` belongs to, from its key set: YOLO11-seg has its Segment head at
    index 23 (`model.23.*`; C2PSA at 10), YOLOv8-seg at 22.  `nc` is read from the class branch's last conv."""
    keys = list(state_dict.keys())
    head = 23 if any(k.startswith("model.23.") for k in keys) else 22
    w = state_dict.get(f"model.{head}.cv3.0.2.weight")
    if w is not None:
        kw.setdefault("nc", int(w.shape[0]))
    return yolo11_config(**kw) if head == 23 else DetectorConfig(**kw)


# ---- YOLO11n-seg graph (ultralytics cfg/models/11/yolo11-seg.yaml, scale "n") ----------
# C3k2(cout, n, c3k, e) | C2PSA(cout, n); the rest as above.  [external - recalled from ultralytics 8.3.x]
def yolo11_seg_graph(cfg: DetectorConfig):
    c, r = cfg.ch, cfg.rep
    return [
        (0, "Conv", (c(64), 3, 2)),
        (1, "Conv", (c(128), 3, 2)),
        (2, "C3k2", (c(256), r(2), False, 0.25)),
        (3, "Conv", (c(256), 3, 2)),
        (4, "C3k2", (c(512), r(2), False, 0.25)),
        (5, "Conv", (c(512), 3, 2)),
        (6, "C3k2", (c(512), r(2), True, 0.5)),
        (7, "Conv", (c(1024), 3, 2)),
        (8, "C3k2", (c(1024), r(2), True, 0.5)),
        (9, "SPPF", (c(1024),)),
        (10, "C2PSA", (c(1024), r(2))),
        (11, "Upsample", ()),
        (12, "Concat", (11, 6)),
        (13, "C3k2", (c(512), r(2), False, 0.5)),
        (14, "Upsample", ()),
        (15, "Concat", (14, 4)),
        (16, "C3k2", (c(256), r(2), False, 0.5)),
        (17, "Conv", (c(256), 3, 2)),
        (18, "Concat", (17, 13)),
        (19, "C3k2", (c(512), r(2), False, 0.5)),
        (20, "Conv", (c(512), 3, 2)),
        (21, "Concat", (20, 10)),
        (22, "C3k2", (c(1024), r(2), True, 0.5)),
    ]


def detector_graph(cfg: DetectorConfig):
    """(graph, indices of the P3 / P4 / P5 feature maps the Segment head reads)"""
    if cfg.arch == "11":
        return yolo11_seg_graph(cfg), (16, 19, 22)
    return yolov8_seg_graph(cfg), (15, 18, 21)


def _conv_bn_keys(prefix: str, cout: int, cin: int, k: int):
    return OrderedDict(
        [
            (f"{prefix}.conv.weight", (cout, cin, k, k)),
            (f"{prefix}.bn.weight", (cout,)),
            (f"{prefix}.bn.bias", (cout,)),
            (f"{prefix}.bn.running_mean", (cout,)),
            (f"{prefix}.bn.running_var", (cout,)),
        ]
    )


def _c3k_keys(prefix: str, c1: int, c2: int, n: int = 2):
    """C3k(c1, c2, n, e=0.5, k=3): cv1, cv2 (1x1 c1 -> c_), cv3 (1x1 2c_ -> c2), n Bottlenecks(c_, c_, 3x3, 3x3, e=1)"""
    c_ = c2 // 2
    out = OrderedDict()
    out.update(_conv_bn_keys(f"{prefix}.cv1", c_, c1, 1))
    out.update(_conv_bn_keys(f"{prefix}.cv2", c_, c1, 1))
    out.update(_conv_bn_keys(f"{prefix}.cv3", c2, 2 * c_, 1))
    for j in range(n):
        out.update(_conv_bn_keys(f"{prefix}.m.{j}.cv1", c_, c_, 3))
        out.update(_conv_bn_keys(f"{prefix}.m.{j}.cv2", c_, c_, 3))
    return out


def detector_param_shapes(cfg: DetectorConfig) -> "OrderedDict[str, tuple]":
    """ultralytics state_dict keys of YOLOv8n-seg / YOLO11n-seg (``num_batches_tracked`` buffers omitted).

    Third-party layout, recalled from ultralytics 8.3.x (pyproject.toml:32 pins ~=8.3.80); it
    cannot be checked against the package in this environment.
    """
    out: "OrderedDict[str, tuple]" = OrderedDict()
    chans = {-1: 3}
    prev = 3
    graph, feats = detector_graph(cfg)
    for idx, kind, a in graph:
        p = f"model.{idx}"
        if kind == "Conv":
            cout, k, _ = a
            out.update(_conv_bn_keys(p, cout, prev, k))
            prev = cout
        elif kind == "C2f":
            cout, n, _ = a
            ch = cout // 2
            out.update(_conv_bn_keys(f"{p}.cv1", 2 * ch, prev, 1))
            out.update(_conv_bn_keys(f"{p}.cv2", cout, (2 + n) * ch, 1))
            for j in range(n):
                out.update(_conv_bn_keys(f"{p}.m.{j}.cv1", ch, ch, 3))
                out.update(_conv_bn_keys(f"{p}.m.{j}.cv2", ch, ch, 3))
            prev = cout
        elif kind == "C3k2":
            # C2f skeleton with hidden width int(cout * e); the inner modules are Bottleneck(c, c, e=0.5) or C3k(c, c, 2)
            cout, n, c3k, e = a
            ch = int(cout * e)
            out.update(_conv_bn_keys(f"{p}.cv1", 2 * ch, prev, 1))
            out.update(_conv_bn_keys(f"{p}.cv2", cout, (2 + n) * ch, 1))
            for j in range(n):
                if c3k:
                    out.update(_c3k_keys(f"{p}.m.{j}", ch, ch, 2))
                else:
                    out.update(_conv_bn_keys(f"{p}.m.{j}.cv1", ch // 2, ch, 3))
                    out.update(_conv_bn_keys(f"{p}.m.{j}.cv2", ch, ch // 2, 3))
            prev = cout
        elif kind == "C2PSA":
            # cv1 (1x1 c1 -> 2c), n PSABlocks on one half (attention with num_heads = c // 64, then a 2-layer FFN), cv2
            cout, n = a
            ch = cout // 2
            out.update(_conv_bn_keys(f"{p}.cv1", 2 * ch, prev, 1))
            out.update(_conv_bn_keys(f"{p}.cv2", cout, 2 * ch, 1))
            nh = max(ch // 64, 1)
            kd = (ch // nh) // 2  # key_dim = head_dim * attn_ratio (0.5)
            for j in range(n):
                q = f"{p}.m.{j}"
                out.update(_conv_bn_keys(f"{q}.attn.qkv", ch + 2 * nh * kd, ch, 1))
                out.update(_conv_bn_keys(f"{q}.attn.proj", ch, ch, 1))
                out.update(_conv_bn_keys(f"{q}.attn.pe", ch, 1, 3))  # depthwise
                out.update(_conv_bn_keys(f"{q}.ffn.0", 2 * ch, ch, 1))
                out.update(_conv_bn_keys(f"{q}.ffn.1", ch, 2 * ch, 1))
            prev = cout
        elif kind == "SPPF":
            (cout,) = a
            out.update(_conv_bn_keys(f"{p}.cv1", prev // 2, prev, 1))
            out.update(_conv_bn_keys(f"{p}.cv2", cout, prev // 2 * 4, 1))
            prev = cout
        elif kind == "Concat":
            prev = sum(chans[s] for s in a)
        chans[idx] = prev
    ch = [chans[f] for f in feats]
    p = f"model.{cfg.head_index}"
    c2 = max(16, ch[0] // 4, cfg.reg_max * 4)
    c3 = max(ch[0], min(cfg.nc, 100))
    c4 = max(ch[0] // 4, cfg.nm)
    for l, cl in enumerate(ch):
        out.update(_conv_bn_keys(f"{p}.cv2.{l}.0", c2, cl, 3))
        out.update(_conv_bn_keys(f"{p}.cv2.{l}.1", c2, c2, 3))
        out[f"{p}.cv2.{l}.2.weight"] = (4 * cfg.reg_max, c2, 1, 1)
        out[f"{p}.cv2.{l}.2.bias"] = (4 * cfg.reg_max,)
    for l, cl in enumerate(ch):
        if cfg.arch == "11":  # Detect(legacy=False): Sequential(DWConv(x, x, 3), Conv(x, c3, 1)), Sequential(DWConv(c3, c3, 3), Conv(c3, c3, 1))
            out.update(_conv_bn_keys(f"{p}.cv3.{l}.0.0", cl, 1, 3))
            out.update(_conv_bn_keys(f"{p}.cv3.{l}.0.1", c3, cl, 1))
            out.update(_conv_bn_keys(f"{p}.cv3.{l}.1.0", c3, 1, 3))
            out.update(_conv_bn_keys(f"{p}.cv3.{l}.1.1", c3, c3, 1))
        else:
            out.update(_conv_bn_keys(f"{p}.cv3.{l}.0", c3, cl, 3))
            out.update(_conv_bn_keys(f"{p}.cv3.{l}.1", c3, c3, 3))
        out[f"{p}.cv3.{l}.2.weight"] = (cfg.nc, c3, 1, 1)
        out[f"{p}.cv3.{l}.2.bias"] = (cfg.nc,)
    out[f"{p}.dfl.conv.weight"] = (1, cfg.reg_max, 1, 1)
    out.update(_conv_bn_keys(f"{p}.proto.cv1", cfg.npr, ch[0], 3))
    out[f"{p}.proto.upsample.weight"] = (cfg.npr, cfg.npr, 2, 2)  # ConvTranspose2d: (in, out, kh, kw)
    out[f"{p}.proto.upsample.bias"] = (cfg.npr,)
    out.update(_conv_bn_keys(f"{p}.proto.cv2", cfg.npr, cfg.npr, 3))
    out.update(_conv_bn_keys(f"{p}.proto.cv3", cfg.nm, cfg.npr, 1))
    for l, cl in enumerate(ch):
        out.update(_conv_bn_keys(f"{p}.cv4.{l}.0", c4, cl, 3))
        out.update(_conv_bn_keys(f"{p}.cv4.{l}.1", c4, c4, 3))
        out[f"{p}.cv4.{l}.2.weight"] = (cfg.nm, c4, 1, 1)
        out[f"{p}.cv4.{l}.2.bias"] = (cfg.nm,)
    return out


def random_detector_state(cfg: DetectorConfig, seed: int, cls_bias: float = -2.1) -> Dict[str, np.ndarray]:
    """Seeded synthetic detector weights (no trained weights exist offline).

    Conv weights are fan-in scaled so activations stay O(1) through the SiLU stack; the
    class-logit bias is shifted by `cls_bias` so that only a few hundred of the 8400
    anchors pass conf 0.25 on random frames."""
    rng = np.random.default_rng(seed)
    sd: Dict[str, np.ndarray] = OrderedDict()
    for key, shape in detector_param_shapes(cfg).items():
        leaf = key.rsplit(".", 1)[-1]
        if key.endswith("dfl.conv.weight"):
            a = np.arange(cfg.reg_max, dtype=np.float64).reshape(shape)
        elif key.endswith("proto.upsample.weight"):
            a = rng.standard_normal(shape) * (1.6 / np.sqrt(shape[0]))
        elif leaf == "weight" and len(shape) == 4:
            a = rng.standard_normal(shape) * (1.6 / np.sqrt(_fan_in(shape)))
        elif leaf == "weight":  # bn scale
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        elif leaf == "running_var":
            a = 1.0 + 0.2 * np.abs(rng.standard_normal(shape))
        else:  # biases, bn shift, running_mean
            a = 0.1 * rng.standard_normal(shape)
            if ".cv3." in key and key.endswith("2.bias"):
                a = a + cls_bias
        sd[key] = np.ascontiguousarray(a, dtype=np.float32)
    return sd
