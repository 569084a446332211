"""Export surface of the detector: a plain `torch.nn.Module` with the ultralytics state_dict keys (model.<i>....),
built from the same graph tables as the GPU executor (`spec.detector_graph`), for `torch.jit.trace` and - where the
`onnx` package exists - `torch.onnx.export`.

The reference exports its trained detector with `YOLO(pt).export(format="onnx", nms=True)` and
`.export(format="coreml", nms=False)` (mtgvision/od_export.py:163-176); ultralytics and coremltools are absent here, so
this module mirrors the `nms=False` form: frames (B, 3, 640, 640) float in [0, 1] -> (pred (B, 4 + nc + nm, A) decoded
boxes / class scores / mask coefficients, protos (B, nm, 160, 160)).  NMS and mask assembly stay outside the graph
(`mtgv.detector.nms`, `Detector.forward`).

NOT on the recognition path: `mtgv.Detector` never calls it and there is no fallback to it.  Checked on CPU against
oracle/detector_ref.py (tests/test_export_cpu.py).
"""

from __future__ import annotations

from typing import Mapping

import numpy as np
import torch
from torch import nn

from . import spec


class Conv(nn.Module):
    """ultralytics Conv: Conv2d(bias=False, padding=k//2) + BatchNorm2d(eps 1e-3) + SiLU"""

    def __init__(self, c1, c2, k=1, s=1, g=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2, groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)
        self.act = nn.SiLU() if act else nn.Identity()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class Bottleneck(nn.Module):
    def __init__(self, c1, c2, shortcut=True, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 3)
        self.cv2 = Conv(c_, c2, 3)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        return x + self.cv2(self.cv1(x)) if self.add else self.cv2(self.cv1(x))


class C2f(nn.Module):
    def __init__(self, c1, c2, n=1, shortcut=False, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, e=1.0) for _ in range(n))

    def forward(self, x):
        y = list(self.cv1(x).chunk(2, 1))
        for m in self.m:
            y.append(m(y[-1]))
        return self.cv2(torch.cat(y, 1))


class C3k(nn.Module):
    def __init__(self, c1, c2, n=2, shortcut=True, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1)
        self.cv2 = Conv(c1, c_, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, e=1.0) for _ in range(n)))

    def forward(self, x):
        return self.cv3(torch.cat((self.m(self.cv1(x)), self.cv2(x)), 1))


class C3k2(C2f):
    def __init__(self, c1, c2, n=1, c3k=False, e=0.5):
        super().__init__(c1, c2, n, True, e)
        self.m = nn.ModuleList(C3k(self.c, self.c, 2, True) if c3k else Bottleneck(self.c, self.c, True) for _ in range(n))


class SPPF(nn.Module):
    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1)
        self.cv2 = Conv(c_ * 4, c2, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)

    def forward(self, x):
        y = [self.cv1(x)]
        for _ in range(3):
            y.append(self.m(y[-1]))
        return self.cv2(torch.cat(y, 1))


class Attention(nn.Module):
    def __init__(self, dim, num_heads, attn_ratio=0.5):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.key_dim = int(self.head_dim * attn_ratio)
        self.scale = self.key_dim**-0.5
        self.qkv = Conv(dim, dim + 2 * self.key_dim * num_heads, 1, act=False)
        self.proj = Conv(dim, dim, 1, act=False)
        self.pe = Conv(dim, dim, 3, 1, g=dim, act=False)

    def forward(self, x):
        b, c, h, w = x.shape
        n = h * w
        q, k, v = self.qkv(x).view(b, self.num_heads, self.key_dim * 2 + self.head_dim, n).split([self.key_dim, self.key_dim, self.head_dim], dim=2)
        attn = ((q.transpose(-2, -1) @ k) * self.scale).softmax(dim=-1)
        y = (v @ attn.transpose(-2, -1)).view(b, c, h, w) + self.pe(v.reshape(b, c, h, w))
        return self.proj(y)


class PSABlock(nn.Module):
    def __init__(self, c, num_heads):
        super().__init__()
        self.attn = Attention(c, num_heads)
        self.ffn = nn.Sequential(Conv(c, 2 * c, 1), Conv(2 * c, c, 1, act=False))

    def forward(self, x):
        x = x + self.attn(x)
        return x + self.ffn(x)


class C2PSA(nn.Module):
    def __init__(self, c1, c2, n=1, e=0.5):
        super().__init__()
        assert c1 == c2
        self.c = int(c1 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1)
        self.cv2 = Conv(2 * self.c, c1, 1)
        self.m = nn.Sequential(*(PSABlock(self.c, max(self.c // 64, 1)) for _ in range(n)))

    def forward(self, x):
        a, b = self.cv1(x).split((self.c, self.c), 1)
        return self.cv2(torch.cat((a, self.m(b)), 1))


class _DFL(nn.Module):
    def __init__(self, c1=16):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        self.c1 = c1

    def forward(self, x):
        b, _, a = x.shape
        return self.conv(x.view(b, 4, self.c1, a).transpose(2, 1).softmax(1)).view(b, 4, a)


class _Proto(nn.Module):
    def __init__(self, c1, c_, c2):
        super().__init__()
        self.cv1 = Conv(c1, c_, 3)
        self.upsample = nn.ConvTranspose2d(c_, c_, 2, 2, 0, bias=True)
        self.cv2 = Conv(c_, c_, 3)
        self.cv3 = Conv(c_, c2, 1)

    def forward(self, x):
        return self.cv3(self.cv2(self.upsample(self.cv1(x))))


class Segment(nn.Module):
    """Segment head (Detect + prototypes + mask coefficients), inference form"""

    def __init__(self, cfg: spec.DetectorConfig, ch):
        super().__init__()
        self.cfg = cfg
        nc, nm, rm = cfg.nc, cfg.nm, cfg.reg_max
        c2 = max(16, ch[0] // 4, rm * 4)
        c3 = max(ch[0], min(nc, 100))
        c4 = max(ch[0] // 4, nm)
        self.cv2 = nn.ModuleList(nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 4 * rm, 1)) for x in ch)
        if cfg.arch == "11":
            self.cv3 = nn.ModuleList(
                nn.Sequential(nn.Sequential(Conv(x, x, 3, g=x), Conv(x, c3, 1)), nn.Sequential(Conv(c3, c3, 3, g=c3), Conv(c3, c3, 1)), nn.Conv2d(c3, nc, 1))
                for x in ch
            )
        else:
            self.cv3 = nn.ModuleList(nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), nn.Conv2d(c3, nc, 1)) for x in ch)
        self.dfl = _DFL(rm)
        self.proto = _Proto(ch[0], cfg.npr, nm)
        self.cv4 = nn.ModuleList(nn.Sequential(Conv(x, c4, 3), Conv(c4, c4, 3), nn.Conv2d(c4, nm, 1)) for x in ch)
        pts, st = [], []
        for s in (8, 16, 32):
            n = cfg.imgsz // s
            sx = torch.arange(n, dtype=torch.float32) + 0.5
            sy, sxx = torch.meshgrid(sx, sx, indexing="ij")
            pts.append(torch.stack((sxx, sy), -1).view(-1, 2))
            st.append(torch.full((n * n, 1), float(s)))
        self.anchors: torch.Tensor
        self.strides: torch.Tensor
        self.register_buffer("anchors", torch.cat(pts).T.contiguous(), persistent=False)
        self.register_buffer("strides", torch.cat(st).T.contiguous(), persistent=False)

    def forward(self, feats):
        cfg = self.cfg
        b = feats[0].shape[0]
        protos = self.proto(feats[0])
        mc = torch.cat([self.cv4[i](f).view(b, cfg.nm, -1) for i, f in enumerate(feats)], 2)
        x = torch.cat([torch.cat((self.cv2[i](f), self.cv3[i](f)), 1).view(b, 4 * cfg.reg_max + cfg.nc, -1) for i, f in enumerate(feats)], 2)
        box, cls = x.split((4 * cfg.reg_max, cfg.nc), 1)
        dist = self.dfl(box)
        lt, rb = dist.chunk(2, 1)
        x1y1, x2y2 = self.anchors.unsqueeze(0) - lt, self.anchors.unsqueeze(0) + rb
        dbox = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * self.strides
        return torch.cat((dbox, cls.sigmoid(), mc), 1), protos


class DetectorModule(nn.Module):
    """`self.model` is an nn.ModuleList indexed like ultralytics' `model.model`: same state_dict keys."""

    def __init__(self, cfg: spec.DetectorConfig):
        super().__init__()
        self.cfg = cfg
        graph, feats = spec.detector_graph(cfg)
        self.graph, self.feats = graph, feats
        mods, chans, prev = [], {-1: 3}, 3
        for idx, kind, a in graph:
            if kind == "Conv":
                m, prev = Conv(prev, a[0], a[1], a[2]), a[0]
            elif kind == "C2f":
                m, prev = C2f(prev, a[0], a[1], a[2]), a[0]
            elif kind == "C3k2":
                m, prev = C3k2(prev, a[0], a[1], a[2], a[3]), a[0]
            elif kind == "C2PSA":
                m, prev = C2PSA(prev, a[0], a[1]), a[0]
            elif kind == "SPPF":
                m, prev = SPPF(prev, a[0]), a[0]
            elif kind == "Upsample":
                m = nn.Upsample(scale_factor=2, mode="nearest")
            elif kind == "Concat":
                m, prev = nn.Identity(), sum(chans[s] for s in a)
            else:
                raise KeyError(kind)
            chans[idx] = prev
            mods.append(m)
        mods.append(Segment(cfg, [chans[f] for f in feats]))
        self.model = nn.ModuleList(mods)

    def forward(self, x):
        outs = {}
        for (idx, kind, a), m in zip(self.graph, self.model):
            x = torch.cat([outs[s] for s in a], 1) if kind == "Concat" else m(x)
            outs[idx] = x
        return self.model[-1]([outs[f] for f in self.feats])


def to_torch_module(cfg: spec.DetectorConfig, state_dict: Mapping) -> DetectorModule:
    m = DetectorModule(cfg).eval()
    want = spec.detector_param_shapes(cfg)
    sd = {k: torch.as_tensor(np.asarray(state_dict[k]) if not isinstance(state_dict[k], torch.Tensor) else state_dict[k]).float() for k in want}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("num_batches_tracked") for k in missing), (missing, unexpected)
    return m


def export_torchscript(cfg: spec.DetectorConfig, state_dict: Mapping, path: str):
    """`torch.jit.trace` of the detector on a (1, 3, imgsz, imgsz) example, saved to `path`"""
    m = to_torch_module(cfg, state_dict)
    ts = torch.jit.trace(m, torch.rand((1, 3, cfg.imgsz, cfg.imgsz)))
    ts.save(path)
    return ts


def export_onnx(cfg: spec.DetectorConfig, state_dict: Mapping, path: str):
    """ONNX export (od_export.py:167-171) when the `onnx` package is importable (it is not in the build image)."""
    import importlib.util

    if importlib.util.find_spec("onnx") is None:
        raise RuntimeError("the onnx package is not installed")
    m = to_torch_module(cfg, state_dict)
    ex = torch.rand((1, 3, cfg.imgsz, cfg.imgsz))
    torch.onnx.export(m, ex, path, input_names=["images"], output_names=["pred", "protos"], dynamic_axes={"images": {0: "n"}, "pred": {0: "n"}, "protos": {0: "n"}})
