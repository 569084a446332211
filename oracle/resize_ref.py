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
