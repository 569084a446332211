"""GPU parity of the perspective crop against oracle/warp_ref.py - integer pixels, bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_warp_bit_exact():
    from mtgv.crop import warp_quads
    from oracle import warp_ref as W

    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, (3, 480, 640, 3), dtype=np.uint8)
    quads, fidx = [], []
    for i in range(12):
        c = rng.uniform([120, 120], [520, 360])
        ang = rng.uniform(0, 2 * np.pi)
        hw, hh = rng.uniform(40, 110), rng.uniform(60, 150)
        base = np.array([[-hw, -hh], [hw, -hh], [hw, hh], [-hw, hh]])
        rot = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
        q = base @ rot.T + c + rng.normal(0, 6, (4, 2))  # perspective-ish distortion
        quads.append(q)
        fidx.append(i % 3)
    quads.append(np.array([[-50, -60], [80, -40], [90, 130], [-40, 150]], float))  # partly outside the frame: border 0
    fidx.append(0)
    quads.append(np.array([[100, 50], [228, 50], [228, 242], [100, 242]], float))  # axis aligned
    fidx.append(1)
    quads.append(np.array([[10, 10], [10, 10], [10, 10], [10, 10]], float))  # degenerate: singular system
    fidx.append(2)
    quads = np.asarray(quads, np.float32)
    out = warp_quads(torch.from_numpy(frames).cuda(), torch.from_numpy(quads), torch.tensor(fidx, dtype=torch.int32)).cpu().numpy()
    assert out.shape == (len(quads), 192, 128, 3)
    for i, q in enumerate(quads):
        ref = W.warp_quad(frames[fidx[i]], q, (192, 128), 0.05)
        np.testing.assert_array_equal(out[i], ref, err_msg=f"quad {i}")
    assert (out[-3][:8, :8] == 0).any()  # border region rendered as constant 0


def test_warp_identity_and_sizes():
    from mtgv.crop import boxes_to_quads, warp_quads

    frames = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (1, 100, 90, 3), dtype=np.uint8)).cuda()
    q = boxes_to_quads(torch.tensor([[10.0, 20.0, 42.0, 68.0]]))
    out = warp_quads(frames, q, torch.zeros(1, dtype=torch.int32), (48, 32), 0.0)
    assert (out[0] == frames[0, 20:68, 10:42]).all()  # unit scale, no expansion: exact copy
    assert warp_quads(frames, torch.zeros((0, 4, 2)), torch.zeros(0, dtype=torch.int32)).shape == (0, 192, 128, 3)
