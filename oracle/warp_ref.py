"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement of `InstanceSeg.extract_dewarped` (mtgvision/od_export.py:95-111):
dst quad = (1+e)*[[0,0],[w,0],[w,h],[0,h]] - 0.5*e*[w,h]; M = cv2.getPerspectiveTransform(src, dst);
cv2.warpPerspective(frame, M, (w,h)) with the defaults INTER_LINEAR / BORDER_CONSTANT(0).

PARITY UNPINNED: OpenCV (cv2) is a third-party dependency that is absent here and the
reference holds no fixture for this step.  Restated from OpenCV's published algorithm:
the homography is solved in float64; every destination pixel is mapped to a source
position quantised to 1/32 pixel (INTER_BITS = 5) and blended from 4 taps with 15-bit
fixed-point weights (INTER_REMAP_COEF_BITS = 15), out-of-frame taps contribute 0.
The homography is solved directly for the destination->source direction with Gaussian
elimination (partial pivoting) - the same single float64 operations as the HIP kernel.
"""

from __future__ import annotations

import numpy as np


def homography_dst_to_src(quad: np.ndarray, out_h: int, out_w: int, expand_ratio: float = 0.05) -> np.ndarray:
    w, h, e = float(out_w), float(out_h), float(expand_ratio)
    d = np.asarray([[0, 0], [out_w, 0], [out_w, out_h], [0, out_h]])
    dst = ((1 + e) * d - (0.5 * e) * np.asarray([out_w, out_h])).astype(np.float32).astype(np.float64)
    src = np.asarray(quad, np.float32).astype(np.float64)
    A = np.zeros((8, 9), np.float64)
    for i in range(4):
        u, v = dst[i]
        x, y = src[i]
        A[i] = [u, v, 1, 0, 0, 0, -u * x, -v * x, x]
        A[i + 4] = [0, 0, 0, u, v, 1, -u * y, -v * y, y]
    for c in range(8):
        piv = c
        best = abs(A[c, c])
        for r in range(c + 1, 8):
            if abs(A[r, c]) > best:
                best, piv = abs(A[r, c]), r
        if best == 0.0:
            return np.zeros(9)
        if piv != c:
            A[[c, piv]] = A[[piv, c]]
        for r in range(c + 1, 8):
            f = A[r, c] / A[c, c]
            for k in range(c, 9):
                A[r, k] = A[r, k] - f * A[c, k]
    sol = np.zeros(8)
    for r in range(7, -1, -1):
        s = A[r, 8]
        for k in range(r + 1, 8):
            s = s - A[r, k] * sol[k]
        sol[r] = s / A[r, r]
    return np.concatenate([sol, [1.0]])


def warp_quad(frame: np.ndarray, quad: np.ndarray, out_size_hw=(192, 128), expand_ratio: float = 0.05) -> np.ndarray:
    oh, ow = out_size_hw
    c = homography_dst_to_src(quad, oh, ow, expand_ratio)
    fh, fw = frame.shape[:2]
    ys, xs = np.meshgrid(np.arange(oh, dtype=np.float64), np.arange(ow, dtype=np.float64), indexing="ij")
    X0 = c[0] * xs + c[1] * ys + c[2]
    Y0 = c[3] * xs + c[4] * ys + c[5]
    W = c[6] * xs + c[7] * ys + c[8]
    with np.errstate(divide="ignore", invalid="ignore"):
        W = np.where(W != 0.0, 32.0 / W, 0.0)
    fX = np.maximum(-2147483648.0, np.minimum(2147483647.0, X0 * W))
    fY = np.maximum(-2147483648.0, np.minimum(2147483647.0, Y0 * W))
    X = np.rint(fX).astype(np.int64)
    Y = np.rint(fY).astype(np.int64)
    sx, sy = X >> 5, Y >> 5
    fx = ((X & 31).astype(np.float32) / np.float32(32.0)).astype(np.float32)
    fy = ((Y & 31).astype(np.float32) / np.float32(32.0)).astype(np.float32)
    one = np.float32(1.0)
    sc = np.float32(32768.0)

    def q(v):
        return np.clip(np.rint(v.astype(np.float32)), -32768, 32767).astype(np.int64)

    wq = np.stack([q((one - fy) * (one - fx) * sc), q((one - fy) * fx * sc), q(fy * (one - fx) * sc), q(fy * fx * sc)], 0)
    big = np.zeros(wq.shape[1:], np.int64)
    for i in range(1, 4):
        cur = np.take_along_axis(wq, big[None], 0)[0]
        big = np.where(wq[i] > cur, i, big)
    fix = 32768 - wq.sum(0)
    for i in range(4):
        wq[i] += np.where(big == i, fix, 0)
    acc = np.zeros((oh, ow, 3), np.int64)
    f = frame.astype(np.int64)
    for tap in range(4):
        px, py = sx + (tap & 1), sy + (tap >> 1)
        ok = (px >= 0) & (px < fw) & (py >= 0) & (py < fh)
        pxc, pyc = np.clip(px, 0, fw - 1), np.clip(py, 0, fh - 1)
        acc += np.where(ok[..., None], wq[tap][..., None] * f[pyc, pxc], 0)
    return np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
