"""CPU-only host logic of the drop-in classes."""
import numpy as np


def test_orientation_of_u_shape():
    """a U-shaped polygon (bottom part of the card missing): corner 0/1 must be the card's top edge"""
    from mtgv.adapters import InstanceSeg

    # card upright at (100..200, 50..250); the mask covers the top 3/4 and two legs -> centroid above the hull's
    pts = np.array([[100, 50], [200, 50], [200, 250], [180, 250], [180, 200], [120, 200], [120, 250], [100, 250]], float)
    s = InstanceSeg(points=pts, label=0, conf=0.9)
    q = s.xyxyxyxy
    assert sorted(q[:2, 1].tolist()) == [50, 50] and sorted(q[2:, 1].tolist()) == [250, 250]
    assert q[0, 0] < q[1, 0]  # tl then tr
    rot = np.array([[0, -1], [1, 0]])  # rotate the card by 90 degrees: top edge now points along -x ... +x
    s2 = InstanceSeg(points=pts @ rot.T + np.array([400, 0]), label=0, conf=0.9)
    q2 = s2.xyxyxyxy
    top_mid = q2[:2].mean(0)
    assert abs(top_mid[0] - (400 - 50)) < 2  # the top edge is the image of y=50


def _u_mask(orientation: str) -> np.ndarray:
    """a 64 x 64 mask of a card whose bottom part is missing (the reference's U shape, od_export.py:57-60): body
    rows 10..39, two legs rows 40..53 with a 20-pixel notch between them; rotated so that the notch opens
    down / left / up / right"""
    m = np.zeros((64, 64), bool)
    m[10:40, 12:52] = True
    m[40:54, 12:22] = True
    m[40:54, 42:52] = True
    return np.rot90(m, {"down": 0, "right": 1, "up": 2, "left": 3}[orientation]).copy()


def _extents(m: np.ndarray) -> np.ndarray:
    ext = np.full((m.shape[0], 2), -1, np.int64)
    for y in range(m.shape[0]):
        xs = np.nonzero(m[y])[0]
        if len(xs):
            ext[y] = xs[0], xs[-1]
    return ext


def _poly_area(p: np.ndarray) -> float:
    return 0.5 * abs(np.sum(p[:, 0] * np.roll(p[:, 1], -1) - np.roll(p[:, 0], -1) * p[:, 1]))


def test_outline_vs_trace_on_u_shaped_masks():
    """what `InstanceSeg.points` holds for the two `CardSegmenter(contours=...)` forms on a U-shaped mask in all four
    orientations.  "trace" (default, reference-shaped): the blob's boundary, notch corners included, for every orientation.
    "outline" (opt-in fast path): row extremes - the notch survives when it opens sideways and is filled when it opens
    up or down.  The convex hull, and with it the fitted quad, is the same in all eight cases."""
    from mtgv.adapters import _convex_hull, _mask_segments, _outline_from_extents

    for o in ("down", "up", "left", "right"):
        m = _u_mask(o)
        trace = _mask_segments(m).astype(np.float64)
        outline = _outline_from_extents(_extents(m)).astype(np.float64)
        # the traced boundary is the mask's own: its polygon area is the pixel-centre polygon of the U (notch excluded)
        full = _poly_area(_convex_hull(trace))
        assert _poly_area(trace) < 0.85 * full, o                      # the notch is there
        notch_filled = _poly_area(outline) > 0.97 * full
        assert notch_filled == (o in ("down", "up")), (o, _poly_area(outline), full)
        if not notch_filled:
            assert abs(_poly_area(outline) - _poly_area(trace)) <= 0.02 * full
        # same hull either way: the quad (od_export.py:62-64 works on the closed polygon's hull) does not depend on the form
        ht, ho = _convex_hull(trace), _convex_hull(outline)
        assert abs(_poly_area(ht) - _poly_area(ho)) < 1e-9
        assert {tuple(p) for p in ht.tolist()} == {tuple(p) for p in ho.tolist()}
        assert len(trace) >= 8  # the U's eight corners at least (run end points only: CHAIN_APPROX_SIMPLE)
