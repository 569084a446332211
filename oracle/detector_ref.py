"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement of the YOLO-seg detector (YOLOv8n-seg, and YOLO11n-seg with C3k2 / C2PSA / DWConv class branch - the
architecture od_train.py:20, :55-56 trains by default) that mtg-vision's `CardSegmenter` delegates to
(mtgvision/od_export.py:141-160: `YOLO(path, task="segment")([rgb_im])[0]`, then
`results.masks.xy` / `results.boxes.conf`; model family fixed by od_train.py:46-70).

PARITY UNPINNED: the arithmetic lives in the third-party package `ultralytics~=8.3.80`
(pyproject.toml:32), which is absent from /root/reference and not installed here, and the
reference holds no tests or golden vectors for it.  What is restated below is the
published YOLOv8-seg algorithm (module graph of yolov8-seg.yaml at scale "n", Conv =
Conv2d(bias=False)+BatchNorm2d(eps=1e-3)+SiLU, C2f, SPPF, Segment/Detect head with DFL
decode, `non_max_suppression` defaults conf=0.25 iou=0.7 max_det=300 max_wh=7680
agnostic=False, `process_mask(..., upsample=True)`), anchored on the reference's call
sites.  The upstream NMS wall-clock `time_limit` break is deliberately not reproduced.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.
"""

from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from mtgv import spec


def _conv(x, p, prefix, k, s=1, eps=1e-3, act=True):
    """ultralytics Conv: Conv2d(bias=False, padding=k//2) + BatchNorm2d + SiLU."""
    x = F.conv2d(x, p[f"{prefix}.conv.weight"], None, stride=s, padding=k // 2)
    x = F.batch_norm(
        x, p[f"{prefix}.bn.running_mean"], p[f"{prefix}.bn.running_var"], p[f"{prefix}.bn.weight"], p[f"{prefix}.bn.bias"], False, 0.0, eps
    )
    return F.silu(x) if act else x


def _c2f(x, p, prefix, n, shortcut, eps):
    y = list(_conv(x, p, f"{prefix}.cv1", 1, eps=eps).chunk(2, 1))
    for j in range(n):
        t = _conv(_conv(y[-1], p, f"{prefix}.m.{j}.cv1", 3, eps=eps), p, f"{prefix}.m.{j}.cv2", 3, eps=eps)
        y.append(y[-1] + t if shortcut else t)
    return _conv(torch.cat(y, 1), p, f"{prefix}.cv2", 1, eps=eps)


def _sppf(x, p, prefix, eps):
    y = [_conv(x, p, f"{prefix}.cv1", 1, eps=eps)]
    for _ in range(3):
        y.append(F.max_pool2d(y[-1], 5, 1, 2))
    return _conv(torch.cat(y, 1), p, f"{prefix}.cv2", 1, eps=eps)


def _bottleneck(x, p, prefix, shortcut, eps):
    """Bottleneck(c1, c2, shortcut, k=(3, 3)): cv2(cv1(x)) (+ x when shortcut and c1 == c2)"""
    t = _conv(_conv(x, p, f"{prefix}.cv1", 3, eps=eps), p, f"{prefix}.cv2", 3, eps=eps)
    return x + t if shortcut else t


def _c3k(x, p, prefix, n, shortcut, eps):
    """C3k = C3 with kernel 3: cv3(cat(m(cv1(x)), cv2(x))), m = n Bottlenecks(c_, c_, e=1.0)"""
    a = _conv(x, p, f"{prefix}.cv1", 1, eps=eps)
    for j in range(n):
        a = _bottleneck(a, p, f"{prefix}.m.{j}", shortcut, eps)
    return _conv(torch.cat((a, _conv(x, p, f"{prefix}.cv2", 1, eps=eps)), 1), p, f"{prefix}.cv3", 1, eps=eps)


def _c3k2(x, p, prefix, n, c3k, eps):
    """C3k2(c1, c2, n, c3k, e): the C2f skeleton (shortcut=True) with Bottleneck(c, c, e=0.5) or C3k(c, c, 2) inside.
    [external: ultralytics 8.3.x nn/modules/block.py]"""
    y = list(_conv(x, p, f"{prefix}.cv1", 1, eps=eps).chunk(2, 1))
    for j in range(n):
        y.append(_c3k(y[-1], p, f"{prefix}.m.{j}", 2, True, eps) if c3k else _bottleneck(y[-1], p, f"{prefix}.m.{j}", True, eps))
    return _conv(torch.cat(y, 1), p, f"{prefix}.cv2", 1, eps=eps)


def _dwconv(x, p, prefix, eps, act=True):
    """DWConv(c, c, 3): depthwise Conv2d(groups=c, bias=False) + BatchNorm2d (+ SiLU)"""
    c = x.shape[1]
    x = F.conv2d(x, p[f"{prefix}.conv.weight"], None, stride=1, padding=1, groups=c)
    x = F.batch_norm(
        x, p[f"{prefix}.bn.running_mean"], p[f"{prefix}.bn.running_var"], p[f"{prefix}.bn.weight"], p[f"{prefix}.bn.bias"], False, 0.0, eps
    )
    return F.silu(x) if act else x


def _attention(x, p, prefix, num_heads, eps):
    """Attention(dim, num_heads, attn_ratio=0.5): qkv 1x1 (no act) -> per head softmax(q^T k * kd^-0.5) -> v attn^T
    + depthwise 3x3 positional encoding of v -> proj 1x1 (no act)"""
    b, c, h, w = x.shape
    n = h * w
    hd = c // num_heads
    kd = hd // 2
    qkv = _conv(x, p, f"{prefix}.qkv", 1, eps=eps, act=False)
    q, k, v = qkv.view(b, num_heads, 2 * kd + hd, n).split([kd, kd, hd], dim=2)
    attn = (q.transpose(-2, -1) @ k) * (kd ** -0.5)
    attn = attn.softmax(dim=-1)
    y = (v @ attn.transpose(-2, -1)).view(b, c, h, w) + _dwconv(v.reshape(b, c, h, w), p, f"{prefix}.pe", eps, act=False)
    return _conv(y, p, f"{prefix}.proj", 1, eps=eps, act=False)


def _c2psa(x, p, prefix, n, eps):
    """C2PSA(c1, c1, n, e=0.5): a, b = cv1(x).split; b through n PSABlocks (x + attn(x); x + ffn(x)); cv2(cat(a, b))"""
    a, b = _conv(x, p, f"{prefix}.cv1", 1, eps=eps).chunk(2, 1)
    nh = max(b.shape[1] // 64, 1)
    for j in range(n):
        q = f"{prefix}.m.{j}"
        b = b + _attention(b, p, f"{q}.attn", nh, eps)
        b = b + _conv(_conv(b, p, f"{q}.ffn.0", 1, eps=eps), p, f"{q}.ffn.1", 1, eps=eps, act=False)
    return _conv(torch.cat((a, b), 1), p, f"{prefix}.cv2", 1, eps=eps)


def _branch(x, p, prefix, eps):
    x = _conv(x, p, f"{prefix}.0", 3, eps=eps)
    x = _conv(x, p, f"{prefix}.1", 3, eps=eps)
    return F.conv2d(x, p[f"{prefix}.2.weight"], p[f"{prefix}.2.bias"])


def _branch_dw(x, p, prefix, eps):
    """YOLO11 class branch (Detect, legacy=False): (DWConv 3x3, Conv 1x1) twice, then Conv2d 1x1"""
    x = _conv(_dwconv(x, p, f"{prefix}.0.0", eps), p, f"{prefix}.0.1", 1, eps=eps)
    x = _conv(_dwconv(x, p, f"{prefix}.1.0", eps), p, f"{prefix}.1.1", 1, eps=eps)
    return F.conv2d(x, p[f"{prefix}.2.weight"], p[f"{prefix}.2.bias"])


def backbone_neck(x, p, cfg: spec.DetectorConfig):
    outs = {}
    graph, feats = spec.detector_graph(cfg)
    for idx, kind, a in graph:
        pre = f"model.{idx}"
        if kind == "Conv":
            x = _conv(x, p, pre, a[1], a[2], cfg.bn_eps)
        elif kind == "C2f":
            x = _c2f(x, p, pre, a[1], a[2], cfg.bn_eps)
        elif kind == "C3k2":
            x = _c3k2(x, p, pre, a[1], a[2], cfg.bn_eps)
        elif kind == "C2PSA":
            x = _c2psa(x, p, pre, a[1], cfg.bn_eps)
        elif kind == "SPPF":
            x = _sppf(x, p, pre, cfg.bn_eps)
        elif kind == "Upsample":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        elif kind == "Concat":
            x = torch.cat([outs[s] for s in a], 1)
        outs[idx] = x
    return [outs[f] for f in feats]


def make_anchors(cfg: spec.DetectorConfig):
    """anchor centres (2, A) and strides (1, A): grid (x+0.5, y+0.5), row-major per level."""
    pts, st = [], []
    for s in (8, 16, 32):
        n = cfg.imgsz // s
        sx = torch.arange(n, dtype=torch.float32) + 0.5
        sy, sxx = torch.meshgrid(sx, sx, indexing="ij")
        pts.append(torch.stack((sxx, sy), -1).view(-1, 2))
        st.append(torch.full((n * n, 1), float(s)))
    return torch.cat(pts).T.contiguous(), torch.cat(st).T.contiguous()


def head(feats, p, cfg: spec.DetectorConfig):
    """Segment head: returns pred (B, 4+nc+nm, A) [xywh px, class sigmoid, coeffs] and protos (B, nm, 160, 160)."""
    pre = f"model.{cfg.head_index}"
    eps = cfg.bn_eps
    b = feats[0].shape[0]
    cls_branch = _branch_dw if cfg.arch == "11" else _branch
    # Proto: Conv3 -> ConvTranspose2d(k2,s2,bias) -> Conv3 -> Conv1
    x = _conv(feats[0], p, f"{pre}.proto.cv1", 3, eps=eps)
    x = F.conv_transpose2d(x, p[f"{pre}.proto.upsample.weight"], p[f"{pre}.proto.upsample.bias"], stride=2)
    x = _conv(x, p, f"{pre}.proto.cv2", 3, eps=eps)
    protos = _conv(x, p, f"{pre}.proto.cv3", 1, eps=eps)
    mc = torch.cat([_branch(f, p, f"{pre}.cv4.{l}", eps).view(b, cfg.nm, -1) for l, f in enumerate(feats)], 2)
    xs = [torch.cat((_branch(f, p, f"{pre}.cv2.{l}", eps), cls_branch(f, p, f"{pre}.cv3.{l}", eps)), 1) for l, f in enumerate(feats)]
    x_cat = torch.cat([xi.view(b, 4 * cfg.reg_max + cfg.nc, -1) for xi in xs], 2)
    box, cls = x_cat.split((4 * cfg.reg_max, cfg.nc), 1)
    # DFL: softmax over the 16 bins, expectation with weights arange(16)
    a = box.shape[-1]
    w = p[f"{pre}.dfl.conv.weight"].view(1, cfg.reg_max, 1, 1)
    dist = (box.view(b, 4, cfg.reg_max, a).transpose(2, 1).softmax(1) * w).sum(1)  # (b, 4, a) l,t,r,b
    anchors, strides = make_anchors(cfg)
    anchors, strides = anchors.to(dist.dtype), strides.to(dist.dtype)
    lt, rb = dist.chunk(2, 1)
    x1y1 = anchors.unsqueeze(0) - lt
    x2y2 = anchors.unsqueeze(0) + rb
    dbox = torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), 1) * strides
    pred = torch.cat((dbox, cls.sigmoid(), mc), 1)
    return pred, protos


def preprocess(frames_u8: np.ndarray, flip_rgb: bool = True, dtype=torch.float32):
    """(B, 640, 640, 3) uint8 letterboxed frames -> (B, 3, 640, 640) float in [0,1].

    ultralytics treats ndarray input as BGR and reverses the channel order first
    (`im[..., ::-1]`), which the reference's callers rely on (server.py:272-274 hands RGB,
    od_cam.py:113-118 hands BGR)."""
    x = np.asarray(frames_u8)
    if flip_rgb:
        x = x[..., ::-1]
    x = np.ascontiguousarray(x.transpose(0, 3, 1, 2))
    return torch.from_numpy(x).to(dtype) / 255.0


def forward(params, cfg: spec.DetectorConfig, frames_u8, flip_rgb=True, dtype=torch.float32):
    p = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))).to(dtype) for k, v in params.items()}
    with torch.no_grad():
        x = preprocess(frames_u8, flip_rgb, dtype)
        return head(backbone_neck(x, p, cfg), p, cfg)


# ---------------------------------------------------------------------------
# NMS - float32 numpy, operation for operation what the HIP kernel does
# ---------------------------------------------------------------------------
def nms_single(pred: np.ndarray, nc: int, conf_thres=0.25, iou_thres=0.7, max_det=300, max_wh=7680.0):
    """pred (4+nc+nm, A) float32 of one image -> dict(keep_idx, boxes xyxy, conf, cls), score-descending.

    Candidates: max class score > conf_thres.  Order: score desc, anchor index asc (stable).
    Boxes are offset by cls*max_wh (per-class NMS), IoU > iou_thres suppresses, first max_det kept.
    """
    pred = np.asarray(pred, np.float32)
    cls_scores = pred[4 : 4 + nc]
    conf = cls_scores.max(0)
    cls = cls_scores.argmax(0).astype(np.int32)  # first maximum on ties
    cand = np.nonzero(conf > np.float32(conf_thres))[0]
    order = cand[np.lexsort((cand, -conf[cand]))]
    half = np.float32(0.5)
    x, y, w, h = (pred[i, order] for i in range(4))
    box = np.stack([x - w * half, y - h * half, x + w * half, y + h * half], 1).astype(np.float32)
    off = (cls[order].astype(np.float32) * np.float32(max_wh))[:, None]
    ob = box + off
    area = (ob[:, 2] - ob[:, 0]) * (ob[:, 3] - ob[:, 1])
    n = len(order)
    suppressed = np.zeros(n, bool)
    keep = []
    thr = np.float32(iou_thres)
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if len(keep) >= max_det:
            break
        j = np.arange(i + 1, n)
        iw = np.maximum(np.float32(0), np.minimum(ob[i, 2], ob[j, 2]) - np.maximum(ob[i, 0], ob[j, 0]))
        ih = np.maximum(np.float32(0), np.minimum(ob[i, 3], ob[j, 3]) - np.maximum(ob[i, 1], ob[j, 1]))
        inter = iw * ih
        iou = inter / (area[i] + area[j] - inter)
        suppressed[j] |= iou > thr
    keep = np.asarray(keep, np.int64)
    return {
        "keep_idx": order[keep].astype(np.int32),
        "boxes": box[keep],
        "conf": conf[order][keep],
        "cls": cls[order][keep],
    }


def mask_logits(pred_img: np.ndarray, protos_img: np.ndarray, det: dict, nc: int, imgsz: int = 640) -> np.ndarray:
    """process_mask up to the crop: (n, 160, 160) float32 logits, zero outside the box.

    coeffs @ protos, then crop_mask with the box scaled to mask units, x in [x1, x2), y in [y1, y2)."""
    c, mh, mw = protos_img.shape
    coef = np.asarray(pred_img, np.float32)[4 + nc :, det["keep_idx"]].T  # (n, nm)
    m = (coef.astype(np.float64) @ protos_img.reshape(c, -1).astype(np.float64)).reshape(-1, mh, mw).astype(np.float32)
    b = det["boxes"].astype(np.float32).copy()
    b[:, [0, 2]] *= np.float32(mw / imgsz)
    b[:, [1, 3]] *= np.float32(mh / imgsz)
    r = np.arange(mw, dtype=np.float32)[None, None, :]
    cc = np.arange(mh, dtype=np.float32)[None, :, None]
    inside = (r >= b[:, 0, None, None]) & (r < b[:, 2, None, None]) & (cc >= b[:, 1, None, None]) & (cc < b[:, 3, None, None])
    return m * inside


def masks_binary(logits: np.ndarray, imgsz: int = 640) -> np.ndarray:
    """F.interpolate(bilinear, align_corners=False) to imgsz, then > 0 (== sigmoid > 0.5)."""
    if logits.shape[0] == 0:
        return np.zeros((0, imgsz, imgsz), bool)
    t = F.interpolate(torch.from_numpy(logits)[None], (imgsz, imgsz), mode="bilinear", align_corners=False)[0]
    return (t > 0).numpy()


def detect(params, cfg: spec.DetectorConfig, frames_u8, flip_rgb=True):
    """Full detector on a batch: list of per-image dicts (keep_idx, boxes, conf, cls, mask_logits)."""
    pred, protos = forward(params, cfg, frames_u8, flip_rgb)
    pred, protos = pred.numpy(), protos.numpy()
    out = []
    for i in range(pred.shape[0]):
        d = nms_single(pred[i], cfg.nc, cfg.conf, cfg.iou, cfg.max_det, cfg.max_wh)
        d["mask_logits"] = mask_logits(pred[i], protos[i], d, cfg.nc, cfg.imgsz)
        out.append(d)
    return out, pred, protos
