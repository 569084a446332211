#!/usr/bin/env python3
"""Tile sweep on the detector's 3x3 / 1x1 conv shapes (tuning aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import torch
from mtgv import native as nv
L = nv.lib()
def t(f, it=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it
tiles = [None, (1, 1, 16), (1, 1, 32), (1, 2, 16), (1, 3, 16), (1, 4, 16), (1, 5, 16)]
shapes = [(32, 80, 80, 32, 32, 3, 1), (32, 40, 40, 64, 64, 3, 1), (32, 80, 80, 64, 64, 3, 1), (32, 20, 20, 128, 128, 3, 1), (32, 160, 160, 16, 16, 3, 1),
          (32, 80, 80, 64, 160, 3, 1), (32, 40, 40, 128, 160, 3, 1), (32, 20, 20, 256, 160, 3, 1), (32, 640, 640, 4, 16, 3, 2), (32, 320, 320, 16, 32, 3, 2),
          (32, 80, 80, 64, 64, 1, 1), (32, 40, 40, 192, 128, 1, 1), (32, 20, 20, 384, 256, 1, 1), (32, 160, 160, 48, 32, 1, 1)]
for (n, h, w, cin, cout, k, s) in shapes:
    x = torch.randn((n, h, w, cin), device="cuda"); wt = torch.randn((cout, k, k, cin), device="cuda") * 0.05
    b = torch.randn((cout,), device="cuda")
    oh = (h + 2 * (k // 2) - k) // s + 1
    o = torch.empty((n, oh, oh, cout), device="cuda")
    res = []
    for tl in tiles:
        if tl: os.environ["MTGV_GEMM_TILE"] = "%d,%d,%d" % tl
        else: os.environ.pop("MTGV_GEMM_TILE", None)
        ms = t(lambda: nv.check(L.mtgv_op_conv2d(nv.ptr(x), nv.ptr(wt), nv.ptr(b), nv.ptr(o), n, h, w, cin, cout, k, k, s, k // 2, 3, nv.stream())))
        res.append((2.0 * n * oh * oh * cout * k * k * cin / ms / 1e9, tl))
    print(f"M={n*oh*oh} N={cout} K={k*k*cin} k{k}s{s}: " + " ".join(f"{'auto' if tl is None else '%d.%d' % tl[1:]}:{tf:.0f}" for tf, tl in res), flush=True)
