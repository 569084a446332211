#!/usr/bin/env python3
"""Ablation of the GEMM's fused extras on the encoder's pointwise shapes (tuning aid, GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import torch
from mtgv import native as nv
L = nv.lib()

def run(m, n, k, act, hw, res, scale, shift, grn, tile=None, it=8):
    a = torch.randn((m, k), device="cuda"); w = torch.randn((n, k), device="cuda") * k ** -0.5
    b = torch.randn((n,), device="cuda"); o = torch.empty((m, n), device="cuda")
    r = torch.randn((m, n), device="cuda") if res else None
    sc = torch.rand((m // hw, k), device="cuda") + 0.5 if scale else None
    sh = torch.randn((k,), device="cuda") if shift else None
    if tile: os.environ["MTGV_GEMM_TILE"] = "%d,%d,%d" % tile
    else: os.environ.pop("MTGV_GEMM_TILE", None)
    part = torch.empty(int(L.mtgv_op_linear_ex_part_floats(m, n, k, act, hw)) + 4, device="cuda") if grn else None
    f = lambda: nv.check(L.mtgv_op_linear_ex(nv.ptr(a), nv.ptr(w), nv.ptr(b), nv.ptr(r), nv.ptr(o), m, n, k, act, hw, nv.ptr(sc), nv.ptr(sh), nv.ptr(part), nv.stream()))
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    return 2.0 * m * n * k / ms / 1e9

def main():
    B = 256
    for (c, hw) in ((96, 1536), (192, 384), (384, 96), (768, 24)):
        m = B * hw
        print(f"--- C={c} hw={hw} M={m}")
        print("pw1: plain %.0f | +mish %.0f | +mish+grn %.0f | gelu+grn %.0f" % (
            run(m, 4*c, c, 0, hw, 0, 0, 0, 0), run(m, 4*c, c, 2, hw, 0, 0, 0, 0), run(m, 4*c, c, 2, hw, 0, 0, 0, 1), run(m, 4*c, c, 1, hw, 0, 0, 0, 1)), flush=True)
        print("pw2: plain %.0f | +res %.0f | +scale %.0f | +scale+res %.0f | +scale+shift+res %.0f" % (
            run(m, c, 4*c, 0, hw, 0, 0, 0, 0), run(m, c, 4*c, 0, hw, 1, 0, 0, 0), run(m, c, 4*c, 0, hw, 0, 1, 0, 0), run(m, c, 4*c, 0, hw, 1, 1, 0, 0), run(m, c, 4*c, 0, hw, 1, 1, 1, 0)), flush=True)
    

if __name__ == "__main__":
    main()
