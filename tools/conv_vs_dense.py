#!/usr/bin/env python3
"""Is the 3x3 implicit-GEMM conv limited by its 9x tap re-reads? Same FLOPs as a dense GEMM (tuning aid)."""
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
for (n, h, w, cin, cout) in ((32, 80, 80, 32, 32), (32, 80, 80, 64, 64), (32, 40, 40, 64, 64), (32, 160, 160, 64, 64), (32, 20, 20, 128, 128), (32, 160, 160, 16, 16)):
    x = torch.randn((n, h, w, cin), device="cuda"); wt = torch.randn((cout, 3, 3, cin), device="cuda") * 0.05
    b = torch.randn((cout,), device="cuda"); o = torch.empty((n, h, w, cout), device="cuda")
    ms_c = t(lambda: nv.check(L.mtgv_op_conv2d(nv.ptr(x), nv.ptr(wt), nv.ptr(b), nv.ptr(o), n, h, w, cin, cout, 3, 3, 1, 1, 3, nv.stream())))
    m, k = n * h * w, 9 * cin
    a = torch.randn((m, k), device="cuda"); w2 = wt.reshape(cout, k).contiguous()
    ms_d = t(lambda: nv.check(L.mtgv_op_linear(nv.ptr(a), nv.ptr(w2), nv.ptr(b), None, nv.ptr(o), m, cout, k, 3, nv.stream())))
    fl = 2.0 * m * cout * k
    print(f"M={m} N={cout} K={k}: conv3x3 {ms_c*1e3:.0f} us {fl/ms_c/1e9:.0f} TF | dense same shape {ms_d*1e3:.0f} us {fl/ms_d/1e9:.0f} TF", flush=True)
