"""CPU checks of the mask -> quad oracle (oracle/quad_ref.py): known rectangles, U-shaped masks, degenerate masks,
and agreement with the host statement in mtgv/adapters.py (InstanceSeg._orient) on single-blob masks."""
import numpy as np
import pytest

from oracle import quad_ref as Q


def _rot_rect_mask(h, w, cx, cy, rw, rh, ang_deg, notch=0.0):
    """filled rotated rectangle; `notch` > 0 removes a centred bite from its bottom edge (the U shape of a held card)"""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    t = np.deg2rad(ang_deg)
    u = (xx - cx) * np.cos(t) + (yy - cy) * np.sin(t)  # along the card's width
    v = -(xx - cx) * np.sin(t) + (yy - cy) * np.cos(t)  # along the card's height (down)
    m = (np.abs(u) <= rw / 2) & (np.abs(v) <= rh / 2)
    if notch > 0:
        m &= ~((np.abs(u) <= rw * 0.3) & (v > rh / 2 - notch * rh))
    corners = np.asarray([[-rw / 2, -rh / 2], [rw / 2, -rh / 2], [rw / 2, rh / 2], [-rw / 2, rh / 2]])
    R = np.asarray([[np.cos(t), -np.sin(t)], [np.sin(t), np.cos(t)]])
    return m, corners @ R.T + np.asarray([cx, cy])


@pytest.mark.parametrize("ang", [0, 17, 45, 90, 133, 180, 212, 270, 301])
def test_u_shape_gives_oriented_rectangle(ang):
    m, want = _rot_rect_mask(200, 240, 120, 100, 70, 100, ang, notch=0.35)
    q, ok = Q.mask_quad(m)
    assert ok == 1
    assert np.abs(q - want).max() < 2.0, f"{q} vs {want}"  # corner 0 = top-left of the card, clockwise


def _poly_mask(h, w, poly):
    """pixels whose centre lies inside the convex polygon `poly` ((P, 2) x, y, clockwise on the screen)"""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    m = np.ones((h, w), bool)
    for i in range(len(poly)):
        a, b = poly[i], poly[(i + 1) % len(poly)]
        m &= (b[0] - a[0]) * (yy - a[1]) - (b[1] - a[1]) * (xx - a[0]) >= 0
    return m


@pytest.mark.parametrize("quad", [
    [[100, 80], [300, 110], [280, 420], [60, 380]],      # upright, seen from the right
    [[330, 70], [380, 300], [120, 360], [40, 150]],      # rolled ~100 degrees, strong keystone
    [[250, 400], [60, 330], [110, 90], [300, 60]],       # upside down
])
def test_perspective_card_recovers_generating_quad(quad):
    """A card seen at an angle is a general quadrilateral: the recovered quad must be the generating one (<= 1.5 px),
    corner 0 = the card's top-left.  A minimum-area RECTANGLE cannot pass this (it differs by tens of pixels)."""
    quad = np.asarray(quad, np.float64)
    m = _poly_mask(480, 420, quad)
    # bite the middle of the bottom edge (corners 2 -> 3) out: the U shape of the training masks
    bc = (quad[2] + quad[3]) / 2
    inward = quad[:2].mean(0) - bc
    inward /= np.linalg.norm(inward)
    along = (quad[2] - quad[3]) / np.linalg.norm(quad[2] - quad[3])
    yy, xx = np.mgrid[0:480, 0:420].astype(np.float64)
    du = (xx - bc[0]) * along[0] + (yy - bc[1]) * along[1]
    dv = (xx - bc[0]) * inward[0] + (yy - bc[1]) * inward[1]
    m &= ~((np.abs(du) < 0.3 * np.linalg.norm(quad[2] - quad[3])) & (dv < 60))
    q, ok = Q.mask_quad(m)
    assert ok == 1
    assert np.abs(q - quad).max() <= 1.5, f"{q} vs {quad}"
    # the host statement on the traced contour agrees
    from mtgv.adapters import InstanceSeg, _largest_contour

    host = np.asarray(InstanceSeg(points=_largest_contour(m), label=0, conf=1.0).xyxyxyxy, np.float64)
    assert np.abs(host - quad).max() <= 1.5, f"{host} vs {quad}"


def test_degenerate_masks():
    m = np.zeros((32, 48), bool)
    q, ok = Q.mask_quad(m, box=[1, 2, 30, 20])
    assert ok == 0 and q.tolist() == [[1, 2], [30, 2], [30, 20], [1, 20]]
    assert Q.mask_quad(m)[1] == 0
    m[5, 7] = True
    q, ok = Q.mask_quad(m)
    assert ok == 1 and (q == np.asarray([7, 5], np.float32)).all()
    m[5, 7:20] = True  # one row of pixels
    q, ok = Q.mask_quad(m)
    assert ok == 1 and sorted(map(tuple, q.tolist())) == [(7.0, 5.0), (7.0, 5.0), (19.0, 5.0), (19.0, 5.0)]
    d = np.eye(24, dtype=bool)  # a diagonal: collinear points, hull of two vertices
    q, ok = Q.mask_quad(d)
    assert ok == 1 and q.min() == 0 and q.max() == 23
    full = np.ones((16, 20), bool)
    q, ok = Q.mask_quad(full)
    assert ok == 1 and sorted(map(tuple, q.tolist())) == [(0.0, 0.0), (0.0, 15.0), (19.0, 0.0), (19.0, 15.0)]


def test_hull_matches_scipy_on_random_blobs():
    from scipy.spatial import ConvexHull

    rng = np.random.default_rng(3)
    for _ in range(10):
        m = np.zeros((60, 80), bool)
        for _ in range(4):
            cy, cx, r = rng.integers(10, 50), rng.integers(10, 70), rng.integers(3, 12)
            yy, xx = np.mgrid[0:60, 0:80]
            m |= (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
        ys, xmin, xmax, _, _ = Q.row_extents(m)
        hull = np.asarray(Q.hull_of_extents(ys, xmin, xmax))
        pts = np.argwhere(m)[:, ::-1]
        ref = pts[ConvexHull(pts).vertices]
        assert set(map(tuple, hull.tolist())) == set(map(tuple, ref.tolist()))


@pytest.mark.parametrize("ang", [10, 75, 160, 250, 330])
def test_agrees_with_host_orient(ang):
    """the host classes (mtgv/adapters.py) trace a contour and use scipy; same quad up to a pixel on one blob"""
    from mtgv.adapters import InstanceSeg, _largest_contour

    m, _ = _rot_rect_mask(160, 160, 80, 80, 50, 72, ang, notch=0.3)
    q, ok = Q.mask_quad(m)
    seg = InstanceSeg(points=_largest_contour(m), label=0, conf=1.0)
    host = np.asarray(seg.xyxyxyxy, np.float64)
    assert ok == 1 and np.abs(q - host).max() <= 1.5, f"{q} vs {host}"


def _shoelace(p):
    x, y = p[:, 0].astype(np.float64), p[:, 1].astype(np.float64)
    return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(np.roll(x, -1), y))


def test_mask_segments_are_run_end_points_of_every_blob():
    """`masks.xy` stand-in (adapters._mask_segments): CHAIN_APPROX_SIMPLE keeps run end points only - same polygon as
    the dense trace; strategy "all" concatenates every blob, "largest" keeps one; the hull is that of all mask pixels"""
    from mtgv.adapters import _convex_hull, _largest_contour, _mask_segments

    rect = np.zeros((40, 50), bool)
    rect[5:21, 7:31] = True
    seg = _mask_segments(rect)
    assert sorted(map(tuple, seg.tolist())) == [(7.0, 5.0), (7.0, 20.0), (30.0, 5.0), (30.0, 20.0)]
    m, _ = _rot_rect_mask(160, 160, 80, 80, 50, 72, 33, notch=0.3)
    dense, simple = _largest_contour(m), _mask_segments(m)
    assert len(simple) < len(dense) and set(map(tuple, simple.tolist())) <= set(map(tuple, dense.tolist()))
    assert abs(_shoelace(simple) - _shoelace(dense)) < 1e-9
    two = m.copy()
    two[4:12, 4:9] = True  # a second, small blob
    both, one = _mask_segments(two, "all"), _mask_segments(two, "largest")
    assert len(both) == len(one) + 4 and (both[:4].max(0) <= [8, 11]).all()  # raster order: the small blob comes first
    pix = np.argwhere(two)[:, ::-1].astype(np.float64)
    assert set(map(tuple, _convex_hull(both).tolist())) == set(map(tuple, _convex_hull(pix).tolist()))
    with pytest.raises(ValueError):
        _mask_segments(two, "longest")
    assert _mask_segments(np.zeros((5, 5), bool)).shape == (0, 2)
    # the GPU / oracle quad takes the hull of all mask pixels: same quad as the host statement on masks.xy("all")
    from mtgv.adapters import InstanceSeg

    q, ok = Q.mask_quad(two)
    host = np.asarray(InstanceSeg(points=both, label=0, conf=1.0).xyxyxyxy, np.float64)
    assert ok == 1 and np.abs(q - host).max() <= 1.5


def _u_card(depth, notch_w, rot, W=100, H=140, S=200):
    """a W x H card at (50, 30) of an S x S raster with a notch of notch_w x depth pixels cut into its bottom edge,
    rotated by rot quarter turns"""
    m = np.zeros((S, S), bool)
    x0, y0 = 50, 30
    m[y0 : y0 + H, x0 : x0 + W] = True
    nx0 = x0 + (W - notch_w) // 2
    m[y0 + H - depth : y0 + H, nx0 : nx0 + notch_w] = False
    return np.rot90(m, rot).copy()


def _top_edge(quad, v):
    from oracle import quad_ref as Q

    qcx, qcy = Q.hull_centroid(quad)
    ex, ey = qcx + v[0] * 1e7, qcy + v[1] * 1e7
    for i in range(1, 4):
        c, d = quad[i], quad[(i + 1) % 4]
        if Q._segments_touch(qcx, qcy, ex, ey, c[0], c[1], d[0], d[1]):
            return i
    return 0


def _top_edges_hull_and_closing(m):
    """the edge the ray test of od_export.py:76-88 picks with v from (a) the convex hull - quad_ref's substitution - and
    (b) a raster closing with the reference's d = 0.2 sqrt(area) (od_export.py:58-61)"""
    import math

    from scipy import ndimage

    from oracle import quad_ref as Q

    ys, xmin, xmax, cnt, sumx = Q.row_extents(m)
    hull = Q.hull_of_extents(ys, xmin, xmax)
    a2 = sum(hull[i][0] * hull[(i + 1) % len(hull)][1] - hull[(i + 1) % len(hull)][0] * hull[i][1] for i in range(len(hull)))
    if a2 < 0:
        hull = [hull[0]] + hull[:0:-1]
    quad = Q.approx_poly_n(hull, 4)
    yy, xx = np.nonzero(m)
    mc = np.array([xx.mean(), yy.mean()])
    v_hull = mc - np.array(Q.hull_centroid(hull))
    d = math.sqrt(m.sum()) * 0.2
    r = int(round(d))
    gy, gx = np.mgrid[-r : r + 1, -r : r + 1]
    closed = ndimage.binary_closing(np.pad(m, r + 2), structure=(gy**2 + gx**2) <= d * d)[r + 2 : -r - 2, r + 2 : -r - 2]
    cy, cx = np.nonzero(closed)
    v_close = mc - np.array([cx.mean(), cy.mean()])
    return _top_edge(quad, v_hull / np.linalg.norm(v_hull)), _top_edge(quad, v_close / np.linalg.norm(v_close)), 2 * d


def test_hull_substitution_against_a_raster_closing_of_u_shaped_masks():
    """quad_ref closes the U-shaped mask with its convex hull where the reference uses buffer(+d).buffer(-d)
    (od_export.py:58-64), a morphological closing.  Against scipy's raster closing with the same d, in all four
    orientations: same top edge for notches narrower than 2 d (the closing fills them) and for notches WIDER than 2 d
    that end below the mask centroid; for notches both wider than 2 d and deeper than ~70 % of the card the closing
    leaves the notch open and the reference's direction vector flips into it - the hull does not follow (documented in
    oracle/quad_ref.py)."""
    expected_top = {0: 0, 1: 3, 2: 2, 3: 1}  # the edge opposite the notch, per quarter turn
    for rot in range(4):
        for depth, notch_w in [(20, 30), (70, 30), (125, 30), (20, 55), (45, 75), (70, 55), (70, 75)]:
            h, c, two_d = _top_edges_hull_and_closing(_u_card(depth, notch_w, rot))
            assert (notch_w > two_d) == (notch_w >= 55)  # 55 and 75 px are wider than 2 d (27..47 px here)
            assert h == c == expected_top[rot], (rot, depth, notch_w, h, c)
        for depth, notch_w in [(100, 55), (125, 75)]:  # wide AND deep: the documented divergence
            h, c, two_d = _top_edges_hull_and_closing(_u_card(depth, notch_w, rot))
            assert notch_w > two_d and h == expected_top[rot] and c == (expected_top[rot] + 2) % 4, (rot, depth, notch_w, h, c)
