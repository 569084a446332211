"""CPU oracle of the mask -> oriented card quad step (TEST INFRASTRUCTURE ONLY - never imported by the product).

Restates what `InstanceSeg._orient` does with a detection mask (mtgvision/od_export.py:52-93):

  * `closed_poly = orig_poly.buffer(d).buffer(-d)` closes the U-shaped mask (the card's bottom is missing from it);
    `v = centroid(orig) - centroid(closed)` then points at the card's top (:58-71);
  * `cv2.approxPolyN(points, 4)` reduces the contour to a general 4-vertex polygon (:74) - NOT a rectangle: a card seen
    at an angle is a trapezoid, and `extract_dewarped` (:95-111) exists to undo exactly that;
  * the ray from the quad's centroid along v picks the edge that becomes edge (0, 1), the top of the crop (:76-88);
  * the corners are truncated to integers (`astype(int)`, :90).

The reference leans on third-party geometry (ultralytics `masks.xy` = cv2.findContours, shapely, cv2.approxPolyN -
`pyproject.toml:29-42`), none of which is importable here and none of which the reference tests: PARITY UNPINNED.
This file states the algorithm `mask_quads_kernel` (mtg-vision_amd/csrc/quads.hip) implements, operation for operation,
so that the GPU result can be compared bit for bit:

  1. per mask row: leftmost / rightmost foreground pixel, pixel count, sum of x        (integers)
  2. convex hull of those extreme pixels (= hull of the mask), monotone chain           (integer cross products)
  3. approxPolyN as OpenCV documents it ("greedy contraction of two vertices into one so that the area changes
     minimally; straight lines through the edges are drawn and the areas of the resulting triangles are considered;
     every vertex lies on the contour or outside it"): while more than 4 vertices remain, remove the hull edge whose
     two neighbouring edges, prolonged to their intersection, add the smallest triangle (first minimum wins)
  4. v = centroid of the mask pixels - area centroid of the closed shape.  The closed shape here is the convex hull
     (what the reference itself falls back to when the buffered polygon falls apart, :63-64); only the direction of
     v enters the result.  THIS IS A SUBSTITUTION, not the reference's operation: `buffer(+d).buffer(-d)` with
     d = 0.2 sqrt(area) is a morphological closing, which fills a notch only where it is narrower than 2 d.  Measured
     against a raster closing with the same d (scipy.ndimage.binary_closing, tests/test_oracle_quads_cpu.py): the two
     pick the same top edge for every notch narrower than 2 d, and for wider notches as long as the notch ends below
     the mask's centroid (the closing then only rounds the notch's inner corners, which still lie on the far side of
     the centroid); they differ for notches that are BOTH wider than 2 d and deeper than about 70 % of the card - there
     the closing leaves the notch open, its centroid moves by less than 1.5 px and the reference's v points INTO the
     notch (it would crop the card upside down), while the hull keeps pointing away from it
  5. ray test of :76-88, roll, truncation toward zero

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


def _contract_area(ax, ay, bx, by, cx, cy, dx, dy):
    """Removing edge b->c: prolong a->b and d->c to their intersection P.  Returns (added area, Px, Py); inf when the
    neighbours do not converge beyond the edge."""
    rx, ry = bx - ax, by - ay
    qx, qy = cx - dx, cy - dy
    ex, ey = cx - bx, cy - by
    den = rx * qy - ry * qx
    if den == 0.0:
        return math.inf, 0.0, 0.0
    t = (ex * qy - ey * qx) / den
    s = (ex * ry - ey * rx) / den
    if not (t > 0.0 and s > 0.0):
        return math.inf, 0.0, 0.0
    px, py = bx + t * rx, by + t * ry
    ux, uy = bx - px, by - py
    wx, wy = cx - px, cy - py
    return 0.5 * abs(ux * wy - uy * wx), px, py


def approx_poly_n(hull, nsides=4):
    """Greedy edge contraction of a convex polygon (list of (x, y), any numeric type) down to `nsides` vertices.
    Returns a list of (x, y) floats in the input's orientation, starting at the lowest surviving slot."""
    h = len(hull)
    vx = [float(p[0]) for p in hull]
    vy = [float(p[1]) for p in hull]
    nxt = [(i + 1) % h for i in range(h)]
    prv = [(i - 1) % h for i in range(h)]
    alive = [True] * h

    def area_of(i):
        a, c = prv[i], nxt[i]
        d = nxt[c]
        return _contract_area(vx[a], vy[a], vx[i], vy[i], vx[c], vy[c], vx[d], vy[d])

    ar = [area_of(i)[0] for i in range(h)]
    cnt = h
    while cnt > nsides:
        best, bi = math.inf, -1
        for i in range(h):
            if alive[i] and ar[i] < best:
                best, bi = ar[i], i
        if bi < 0:
            break  # nothing can be contracted (parallel neighbours everywhere)
        _, px, py = area_of(bi)
        c = nxt[bi]
        vx[bi], vy[bi] = px, py
        alive[c] = False
        nxt[bi] = nxt[c]
        prv[nxt[c]] = bi
        cnt -= 1
        for j in (prv[prv[bi]], prv[bi], bi, nxt[bi]):
            ar[j] = area_of(j)[0]
    out = []
    i0 = next(i for i in range(h) if alive[i])
    i = i0
    while True:
        out.append((vx[i], vy[i]))
        i = nxt[i]
        if i == i0:
            break
    return out


def _segments_touch(ax, ay, bx, by, cx, cy, dx, dy):
    """closed segments AB and CD intersect (orientation products; what shapely's intersects() decides here)"""
    d1 = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)
    d2 = (bx - ax) * (dy - ay) - (by - ay) * (dx - ax)
    d3 = (dx - cx) * (ay - cy) - (dy - cy) * (ax - cx)
    d4 = (dx - cx) * (by - cy) - (dy - cy) * (bx - cx)
    return d1 * d2 <= 0.0 and d3 * d4 <= 0.0


def mask_quad(mask: np.ndarray, box=None):
    """(quad (4, 2) float32 tl/tr/br/bl with integer values, ok).  Empty mask: ok = 0 and the quad is the box."""
    ys, xmin, xmax, cnt, sumx = row_extents(mask)
    n = int(cnt.sum())
    if n == 0:
        if box is None:
            return np.zeros((4, 2), np.float32), 0
        x1, y1, x2, y2 = [np.float32(v) for v in box]
        return np.asarray([[x1, y1], [x2, y1], [x2, y2], [x1, y2]], np.float32), 0
    hull = hull_of_extents(ys, xmin, xmax)
    if len(hull) >= 3:
        # clockwise on the screen (y down): positive shoelace sum
        a2 = 0
        for i in range(len(hull)):
            a2 += hull[i][0] * hull[(i + 1) % len(hull)][1] - hull[(i + 1) % len(hull)][0] * hull[i][1]
        if a2 < 0:
            hull = [hull[0]] + hull[:0:-1]
    quad = approx_poly_n(hull, 4) if len(hull) >= 4 else None
    if quad is None or len(quad) != 4:  # fewer than 4 hull vertices (point, line, triangle): the bounding box
        x1, x2 = float(xmin.min()), float(xmax.max())
        y1, y2 = float(ys.min()), float(ys.max())
        quad = [(x1, y1), (x2, y1), (x2, y2), (x1, y2)]
    mcx = float(int(sumx.sum())) / float(n)
    mcy = float(int((ys * cnt).sum())) / float(n)
    hcx, hcy = hull_centroid(hull) if len(hull) >= 3 else (mcx, mcy)
    vx, vy = mcx - hcx, mcy - hcy
    nv = math.sqrt(vx * vx + vy * vy)
    if nv > 0.0:
        vx, vy = vx / nv, vy / nv
    else:
        vx, vy = 0.0, -1.0
    qcx, qcy = hull_centroid(quad)
    ex, ey = qcx + vx * 10000000.0, qcy + vy * 10000000.0
    idx = 0
    for i in range(1, 4):
        c, d = quad[i], quad[(i + 1) % 4]
        if _segments_touch(qcx, qcy, ex, ey, c[0], c[1], d[0], d[1]):
            idx = i
            break
    q = [quad[(idx + i) % 4] for i in range(4)]
    return np.trunc(np.asarray(q, np.float64)).astype(np.float32), 1


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
