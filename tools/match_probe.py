#!/usr/bin/env python3
"""Bank match timing, two-pass (fp16 first pass + exact re-rank) vs one-pass exact kernel: python tools/match_probe.py [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import torch
from mtgv.matcher import Matcher

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
g = torch.Generator(device="cuda").manual_seed(2)
m = Matcher(768, capacity=rows)
for r0 in range(0, rows, 100_000):
    m.add(torch.randn((min(100_000, rows - r0), 768), generator=g, device="cuda"))

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for nq in (256, 1024, 2048):
    q = torch.randn((nq, 768), generator=g, device="cuda")
    res = {}
    for mode in ("1", "0"):
        os.environ["MTGV_MATCH_PREPASS"] = mode
        res[mode] = (timeit(lambda: m.match(q, 1)), m.match(q, 1))
    same = bool((res["1"][1][0] == res["0"][1][0]).all())
    print(f"fallbacks so far {m.prepass_fallbacks()};", end=" ")
    print(f"bank {rows} x 768, {nq} queries: two-pass {res['1'][0]:.3f} ms, one-pass {res['0'][0]:.3f} ms, speed-up {res['0'][0] / res['1'][0]:.2f}x, ids identical: {same}")
