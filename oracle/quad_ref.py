"""CPU oracle of the mask -> oriented card quad step (TEST INFRASTRUCTURE ONLY - never imported by the product).

Restates what `InstanceSeg._orient` does with a detection mask (mtgvision/od_export.py:52-93): close the U-shaped mask,
take four corners, and roll them so that corner 0 is the card's top-left (the missing bottom of the "U" tells which
way is up).  The reference leans on third-party geometry (ultralytics `masks.xy` = cv2.findContours, shapely
buffer(+/-), cv2.approxPolyN - `pyproject.toml:29-42`), none of which is importable here and none of which the
reference tests: PARITY UNPINNED.  This file states the build's own algorithm, the one `mask_quads_kernel`
(mtg-vision_amd/csrc/quads.hip) implements, operation for operation so that the GPU result can be compared bit for bit:

  1. per mask row: leftmost / rightmost foreground pixel, pixel count, sum of x        (integers)
  2. convex hull of those extreme pixels, monotone chain ordered by (y, x)             (integer cross products)
  3. minimum-area rectangle: for every hull edge, extent of the hull along / across it (float64, first minimum wins)
  4. direction "up" = centroid of the mask pixels - area centroid of the hull          (float64)
  5. the rectangle edge that lies furthest along "up" becomes edge (0, 1); corners run clockwise (y down)

Sums run in plain Python loops in the same order as the kernel's sequential sections, so no pairwise-summation or
FMA difference can appear.
"""

from __future__ import annotations

import math

import numpy as np


def row_extents(mask: np.ndarray):
    """(ys, xmin, xmax, count, sumx) over the non-empty rows of a (H, W) boolean / uint8 mask."""
    m = np.asarray(mask) != 0
    ys = np.nonzero(m.any(axis=1))[0]
    xmin = np.asarray([int(np.nonzero(m[y])[0][0]) for y in ys], np.int64)
    xmax = np.asarray([int(np.nonzero(m[y])[0][-1]) for y in ys], np.int64)
    cnt = np.asarray([int(m[y].sum()) for y in ys], np.int64)
    sumx = np.asarray([int(np.nonzero(m[y])[0].sum()) for y in ys], np.int64)
    return ys.astype(np.int64), xmin, xmax, cnt, sumx


def hull_of_extents(ys, xmin, xmax):
    """Monotone chain over the points (xmin[y], y), (xmax[y], y) in (y, x) order -> list of (x, y) integer vertices."""
    pts = []
    for y, a, b in zip(ys.tolist(), xmin.tolist(), xmax.tolist()):
        pts.append((a, y))
        if b != a:
            pts.append((b, y))
    if len(pts) <= 1:
        return pts

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower = []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    upper = []
    for p in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    return lower[:-1] + upper[:-1]


def min_area_rect(hull):
    """((4, 2) float64 corners, edge index) of the smallest rectangle with a side on a hull edge; first minimum wins."""
    h = len(hull)
    best_area, best = math.inf, None
    for i in range(h):
        ex = float(hull[(i + 1) % h][0] - hull[i][0])
        ey = float(hull[(i + 1) % h][1] - hull[i][1])
        n = math.sqrt(ex * ex + ey * ey)
        if n == 0.0:
            continue
        ux, uy = ex / n, ey / n
        a0 = b0 = math.inf
        a1 = b1 = -math.inf
        for k in range(h):
            px, py = float(hull[k][0]), float(hull[k][1])
            a = px * ux + py * uy
            b = py * ux - px * uy  # along (-uy, ux)
            a0, a1 = min(a0, a), max(a1, a)
            b0, b1 = min(b0, b), max(b1, b)
        area = (a1 - a0) * (b1 - b0)
        if area < best_area:
            best_area, best = area, (ux, uy, a0, a1, b0, b1, i)
    if best is None:
        return None, -1
    ux, uy, a0, a1, b0, b1, i = best
    vx, vy = -uy, ux
    corners = np.asarray(
        [[ux * a0 + vx * b0, uy * a0 + vy * b0], [ux * a1 + vx * b0, uy * a1 + vy * b0], [ux * a1 + vx * b1, uy * a1 + vy * b1],
         [ux * a0 + vx * b1, uy * a0 + vy * b1]], np.float64)
    return corners, i


def hull_centroid(hull):
    """Area centroid (shoelace, sequential float64); vertex mean for degenerate hulls."""
    h = len(hull)
    a2 = cx = cy = 0.0
    for i in range(h):
        x0, y0 = float(hull[i][0]), float(hull[i][1])
        x1, y1 = float(hull[(i + 1) % h][0]), float(hull[(i + 1) % h][1])
        cr = x0 * y1 - x1 * y0
        a2 += cr
        cx += (x0 + x1) * cr
        cy += (y0 + y1) * cr
    if abs(a2) < 1e-9:
        sx = sy = 0.0
        for p in hull:
            sx += float(p[0])
            sy += float(p[1])
        return sx / h, sy / h
    return cx / (3.0 * a2), cy / (3.0 * a2)


def mask_quad(mask: np.ndarray, box=None):
    """(quad (4, 2) float32 tl/tr/br/bl, ok).  Empty mask: ok = 0 and the quad is the box (or zeros without a box)."""
    ys, xmin, xmax, cnt, sumx = row_extents(mask)
    n = int(cnt.sum())
    if n == 0:
        if box is None:
            return np.zeros((4, 2), np.float32), 0
        x1, y1, x2, y2 = [np.float32(v) for v in box]
        return np.asarray([[x1, y1], [x2, y1], [x2, y2], [x1, y2]], np.float32), 0
    hull = hull_of_extents(ys, xmin, xmax)
    rect, _ = min_area_rect(hull) if len(hull) >= 3 else (None, -1)
    if rect is None:  # a point or a straight run of pixels: its bounding box
        x1, x2 = float(xmin.min()), float(xmax.max())
        y1, y2 = float(ys.min()), float(ys.max())
        rect = np.asarray([[x1, y1], [x2, y1], [x2, y2], [x1, y2]], np.float64)
    mcx = float(int(sumx.sum())) / float(n)
    mcy = float(int((ys * cnt).sum())) / float(n)
    hcx, hcy = hull_centroid(hull) if len(hull) >= 3 else (mcx, mcy)
    vx, vy = mcx - hcx, mcy - hcy
    nv = math.sqrt(vx * vx + vy * vy)
    if nv > 0.0:
        vx, vy = vx / nv, vy / nv
    else:
        vx, vy = 0.0, -1.0
    ccx = (rect[0][0] + rect[1][0] + rect[2][0] + rect[3][0]) / 4.0
    ccy = (rect[0][1] + rect[1][1] + rect[2][1] + rect[3][1]) / 4.0
    idx, best = 0, -math.inf
    for i in range(4):
        mx = (rect[i][0] + rect[(i + 1) % 4][0]) / 2.0 - ccx
        my = (rect[i][1] + rect[(i + 1) % 4][1]) / 2.0 - ccy
        d = mx * vx + my * vy
        if d > best:
            best, idx = d, i
    q = [rect[(idx + i) % 4] for i in range(4)]
    e0x, e0y = q[1][0] - q[0][0], q[1][1] - q[0][1]
    e1x, e1y = q[2][0] - q[1][0], q[2][1] - q[1][1]
    if e0x * e1y - e0y * e1x < 0.0:
        q = [q[1], q[0], q[3], q[2]]
    return np.asarray(q, np.float64).astype(np.float32), 1


def mask_quads(masks: np.ndarray, boxes=None):
    """masks (N, H, W) -> (quads (N, 4, 2) float32, ok (N,) int32)"""
    qs, oks = [], []
    for i in range(masks.shape[0]):
        q, ok = mask_quad(masks[i], None if boxes is None else boxes[i])
        qs.append(q)
        oks.append(ok)
    if not qs:
        return np.zeros((0, 4, 2), np.float32), np.zeros((0,), np.int32)
    return np.stack(qs), np.asarray(oks, np.int32)
