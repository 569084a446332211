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
