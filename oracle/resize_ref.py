"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement of `SyntheticBgFgMtgImages.make_cropped` (mtgvision/encoder_datasets.py:733-753):
`remove_border_resized(img, border_width=ceil(max(0.02*H, 0.02*W)), size_hw)` (util/image.py:337-346) with
`cv2.resize(..., interpolation=cv2.INTER_AREA)` and a clip to [0,1] (util/image.py:322-334).

PARITY UNPINNED: cv2 is a third-party dependency absent here and the reference holds no fixture.  INTER_AREA is
restated as its definition - the exact area integral (coverage-weighted mean of the source pixels under each
output pixel's footprint), float64 accumulation.
"""

from __future__ import annotations

from math import ceil

import numpy as np


def _weights(n_src: int, n_out: int) -> np.ndarray:
    """(n_out, n_src) coverage of source cell j by output footprint [o*s, (o+1)*s)"""
    s = n_src / n_out
    o = np.arange(n_out, dtype=np.float64)[:, None]
    j = np.arange(n_src, dtype=np.float64)[None, :]
    return np.clip(np.minimum(j + 1, (o + 1) * s) - np.maximum(j, o * s), 0.0, None)


def make_cropped(img_u8: np.ndarray, size_hw=(192, 128)) -> np.ndarray:
    """uint8 HWC card image of any size -> float32 (h, w, 3) in [0,1]"""
    H, W = img_u8.shape[:2]
    bw = ceil(max(0.02 * H, 0.02 * W))
    crop = img_u8[bw : H - bw, bw : W - bw].astype(np.float64)
    oh, ow = size_hw
    if crop.shape[0] <= 0 or crop.shape[1] <= 0:
        return np.zeros((oh, ow, 3), np.float32)
    wy, wx = _weights(crop.shape[0], oh), _weights(crop.shape[1], ow)
    sy, sx = crop.shape[0] / oh, crop.shape[1] / ow
    tmp = np.tensordot(wy, crop, axes=(1, 0))  # (oh, W, 3)
    out = np.tensordot(wx, tmp, axes=(1, 1)).transpose(1, 0, 2) / (sx * sy * 255.0)  # (oh, ow, 3)
    return np.clip(out.astype(np.float32), 0.0, 1.0)


def letterbox(frame_u8: np.ndarray, size: int = 640, pad_value: int = 114) -> np.ndarray:
    """ultralytics LetterBox ahead of the detector (behind CardSegmenter.__call__, mtgvision/od_export.py:147-150): scale to
    fit size x size, centre (top = round(dh - 0.1), left = round(dw - 0.1)), fill with 114.

    PARITY UNPINNED: the resample upstream is cv2.resize(INTER_LINEAR) (absent here).  Restated as the align_corners =
    False bilinear form (what torch.nn.functional.interpolate computes), float32, in this order:
    src = scale * (dst + 0.5) - 0.5 clamped at 0; h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11); round half to
    even; clamp to [0, 255]."""
    h, w = frame_u8.shape[:2]
    r = min(size / h, size / w)
    nh, nw = int(round(h * r)), int(round(w * r))
    top, left = int(round((size - nh) / 2 - 0.1)), int(round((size - nw) / 2 - 0.1))
    f32 = np.float32

    def axis(n_in, n_out):
        s = f32(n_in) / f32(n_out)
        src = s * (np.arange(n_out, dtype=np.float32) + f32(0.5)) - f32(0.5)
        src = np.maximum(src, f32(0.0))
        i0 = np.minimum(src.astype(np.int32), n_in - 1)
        i1 = i0 + (i0 < n_in - 1)
        l1 = (src - i0.astype(np.float32)).astype(np.float32)
        return i0, i1, (f32(1.0) - l1).astype(np.float32), l1

    y0, y1, ly0, ly1 = axis(h, nh)
    x0, x1, lx0, lx1 = axis(w, nw)
    src = frame_u8.astype(np.float32)
    lx0, lx1 = lx0[None, :, None], lx1[None, :, None]
    t = lx0 * src[y0][:, x0] + lx1 * src[y0][:, x1]
    b = lx0 * src[y1][:, x0] + lx1 * src[y1][:, x1]
    v = ly0[:, None, None] * t + ly1[:, None, None] * b
    img = np.clip(np.rint(v), 0, 255).astype(np.uint8)
    out = np.full((size, size, 3), pad_value, np.uint8)
    out[top : top + nh, left : left + nw] = img
    return out
