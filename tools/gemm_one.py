#!/usr/bin/env python3
"""Run one linear shape a few times (for rocprofv3 --pmc): gemm_one.py M N K act [res] [scale]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import torch
from mtgv import native as nv
L = nv.lib()
m, n, k, act = (int(x) for x in sys.argv[1:5])
res = len(sys.argv) > 5 and sys.argv[5] == "1"
a = torch.randn((m, k), device="cuda"); w = torch.randn((n, k), device="cuda") * k ** -0.5
b = torch.randn((n,), device="cuda"); o = torch.empty((m, n), device="cuda")
r = torch.randn((m, n), device="cuda") if res else None
for _ in range(5):
    nv.check(L.mtgv_op_linear(nv.ptr(a), nv.ptr(w), nv.ptr(b), nv.ptr(r), nv.ptr(o), m, n, k, act, nv.stream()))
torch.cuda.synchronize()
