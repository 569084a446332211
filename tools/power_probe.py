#!/usr/bin/env python3
"""Is a GEMM launch power-limited?  The same linear launch (f32 rows by DMA, split products) on random operands, on
zero activations and on all-zero operands, 300 back-to-back launches each: a data-dependent time says the clock (DVFS),
not the schedule, sets it.   python tools/power_probe.py [M N K]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import torch
from mtgv import native as nv
L = nv.lib()
m, n, k = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (24576, 384, 1536)
o = torch.empty((m, n), device="cuda")
b = torch.zeros((n,), device="cuda")


def run(a, w, reps=300):
    for _ in range(20):
        nv.check(L.mtgv_op_linear(nv.ptr(a), nv.ptr(w), nv.ptr(b), None, nv.ptr(o), m, n, k, 0, nv.stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        nv.check(L.mtgv_op_linear(nv.ptr(a), nv.ptr(w), nv.ptr(b), None, nv.ptr(o), m, n, k, 0, nv.stream()))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


a_r = torch.randn((m, k), device="cuda"); w_r = torch.randn((n, k), device="cuda") * k ** -0.5
a_z = torch.zeros_like(a_r); w_z = torch.zeros_like(w_r)
a_c = torch.ones_like(a_r); w_c = torch.ones_like(w_r)   # constant operands: lo halves are zero, no toggling between lanes
print(f"linear {m} x {n} x {k}, us per launch (300 back to back):")
for name, a, w in (("random A, random W", a_r, w_r), ("zero A, random W", a_z, w_r), ("zero A, zero W", a_z, w_z),
                   ("ones A, ones W", a_c, w_c), ("random A, random W (again)", a_r, w_r)):
    print(f"  {name:28s} {run(a, w):7.1f}")
