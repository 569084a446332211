#!/usr/bin/env python3
"""Tile sweep of the f32-MFMA GEMM on the shapes of the path (tuning aid, GPU box only).
usage: python tools/gemm_sweep.py [enc|det|all]"""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import torch
from mtgv import native as nv

L = nv.lib()
def bench_linear(m, n, k, act, tile, it=8):
    a = torch.randn((m, k), device="cuda"); w = torch.randn((n, k), device="cuda") * k ** -0.5
    b = torch.randn((n,), device="cuda"); o = torch.empty((m, n), device="cuda")
    if tile: os.environ["MTGV_GEMM_TILE"] = "%d,%d,%d" % tile
    else: os.environ.pop("MTGV_GEMM_TILE", None)
    f = lambda: nv.check(L.mtgv_op_linear(nv.ptr(a), nv.ptr(w), nv.ptr(b), None, nv.ptr(o), m, n, k, act, nv.stream()))
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    return ms, 2.0 * m * n * k / ms / 1e9

which = sys.argv[1] if len(sys.argv) > 1 else "enc"
shapes = []
if which in ("enc", "all"):
    B = 256
    for c, hw in ((96, 1536), (192, 384), (384, 96), (768, 24)):
        shapes += [(B * hw, 4 * c, c, 2), (B * hw, c, 4 * c, 0)]
    shapes += [(256, 100000, 768, 0)]
if which in ("det", "all"):
    shapes += [(819200, 32, 32, 3), (204800, 64, 64, 3), (51200, 128, 256, 3), (12800, 256, 384, 3), (204800, 64, 192, 3)]
tiles = [None] + [(1, tn, bk) for tn in (1, 2, 3, 4, 5) for bk in (16, 32) if not (tn == 5 and bk == 32)] + [(2, 1, 16), (2, 2, 16), (2, 3, 16), (2, 4, 16), (2, 2, 32), (2, 3, 32)]
for (m, n, k, act) in shapes:
    res = []
    for t in tiles:
        try:
            ms, tf = bench_linear(m, n, k, act, t)
            res.append((tf, ms, t))
        except Exception as e:
            res.append((0.0, 0.0, t))
    best = max(res)
    line = " ".join(f"{'auto' if t is None else '%d.%d.%d' % t}:{tf:.0f}" for tf, ms, t in res)
    print(f"M={m} N={n} K={k} act={act} | best {best[2]} {best[0]:.1f} TF {best[1]:.3f} ms | {line}", flush=True)
