#!/usr/bin/env python3
"""Debug aid: warp_quads on the GPU against oracle/warp_ref.py, mismatches by column phase and channel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import numpy as np, torch
from mtgv.crop import warp_quads
from oracle import warp_ref as W
rng = np.random.default_rng(0)
frames = rng.integers(0, 256, (3, 480, 640, 3), dtype=np.uint8)
q = np.array([[200, 150], [330, 160], [320, 340], [190, 330]], np.float32)
out = warp_quads(torch.from_numpy(frames).cuda(), torch.from_numpy(q[None]), torch.tensor([1], dtype=torch.int32)).cpu().numpy()[0]
ref = W.warp_quad(frames[1], q, (192, 128), 0.05)
bad = out != ref
print("mismatching elements", bad.sum(), "of", bad.size)
print("by x % 4:", [int(bad[:, j::4].sum()) for j in range(4)], " by channel:", [int(bad[..., c].sum()) for c in range(3)])
ys, xs, cs = np.nonzero(bad)
for y, x, c in list(zip(ys, xs, cs))[:12]:
    print(y, x, c, out[y, x], ref[y, x])
