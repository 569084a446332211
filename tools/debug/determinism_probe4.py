#!/usr/bin/env python3
"""Debug aid: one ConvNeXt block / conv / dwconv / LN on one stream beside the detector on another."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import numpy as np, torch
from mtgv import native as nv, spec
from mtgv.detector import Detector
L = nv.lib()
mode = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
nv.set_gemm_precision(mode)
det_cfg = spec.DetectorConfig()
det = Detector(det_cfg, spec.random_detector_state(det_cfg, 3), max_batch=2)
g = torch.Generator(device="cuda").manual_seed(11)
frames = torch.randint(0, 256, (2, 640, 640, 3), generator=g, device="cuda", dtype=torch.uint8)
s_det, s_op = torch.cuda.Stream(), torch.cuda.Stream()

def check(name, f, reps=24):
    ref = f()
    torch.cuda.synchronize()
    bad = 0; info = ""
    for t in range(reps):
        with torch.cuda.stream(s_det):
            det.forward(frames, True, mask_rows=4)
        with torch.cuda.stream(s_op):
            o = f()
        torch.cuda.synchronize()
        d = o != ref
        if d.any():
            bad += 1
            if not info:
                idx = d.nonzero()
                info = f" first bad: {d.sum().item()} of {d.numel()} elems; images {sorted(set(idx[:, 0].tolist()))}; maxdiff {(o - ref).abs().max().item():.2e}"
    print(f"{mode} {name}: overlapped mismatches {bad}/{reps}{info}", flush=True)

for (n, h, w, c) in ((8, 48, 32, 80), (8, 24, 16, 160), (8, 12, 8, 320)):
    r = lambda *s: torch.randn(*s, device="cuda")
    X = r(n, h, w, c)
    P = dict(dw=r(49, c) / 7, dwb=0.1 * r(c), lnw=1 + 0.1 * r(c), lnb=0.1 * r(c), w1=r(4 * c, c) / c ** 0.5, b1=0.1 * r(4 * c),
             ga=0.3 * r(4 * c), be=0.1 * r(4 * c), w2=r(c, 4 * c) / (4 * c) ** 0.5, b2=0.1 * r(c))
    ws = torch.empty(int(L.mtgv_op_block_workspace_floats(n, h, w, c)), device="cuda")
    def blk():
        out = torch.empty((n, h, w, c), device="cuda")
        nv.check(L.mtgv_op_block(nv.ptr(X), nv.ptr(out), n, h, w, c, 2, nv.ptr(P["dw"]), nv.ptr(P["dwb"]), nv.ptr(P["lnw"]), nv.ptr(P["lnb"]),
                                 nv.ptr(P["w1"]), nv.ptr(P["b1"]), nv.ptr(P["ga"]), nv.ptr(P["be"]), nv.ptr(P["w2"]), nv.ptr(P["b2"]), nv.ptr(ws), nv.stream()))
        return out
    def blk3():  # three blocks back to back on one workspace, like a stage
        a = blk(); 
        for _ in range(2):
            out = torch.empty((n, h, w, c), device="cuda")
            nv.check(L.mtgv_op_block(nv.ptr(a), nv.ptr(out), n, h, w, c, 2, nv.ptr(P["dw"]), nv.ptr(P["dwb"]), nv.ptr(P["lnw"]), nv.ptr(P["lnb"]),
                                     nv.ptr(P["w1"]), nv.ptr(P["b1"]), nv.ptr(P["ga"]), nv.ptr(P["be"]), nv.ptr(P["w2"]), nv.ptr(P["b2"]), nv.ptr(ws), nv.stream()))
            a = out
        return a
    def dw():
        out = torch.empty((n, h, w, c), device="cuda")
        nv.check(L.mtgv_op_dwconv7(nv.ptr(X), nv.ptr(P["dw"]), nv.ptr(P["dwb"]), nv.ptr(out), n, h, w, c, nv.stream()))
        return out
    def ln():
        out = torch.empty((n, h, w, c), device="cuda")
        nv.check(L.mtgv_op_layernorm(nv.ptr(X), nv.ptr(P["lnw"]), nv.ptr(P["lnb"]), nv.ptr(out), n * h * w, c, 1e-6, nv.stream()))
        return out
    check(f"block   n{n} {h}x{w}x{c}", blk)
    check(f"3blocks n{n} {h}x{w}x{c}", blk3)
    check(f"dwconv7 n{n} {h}x{w}x{c}", dw, 8)
    check(f"ln      n{n} {h}x{w}x{c}", ln, 8)
# stem-like conv: 4x4 stride 4 on 192x128x4 (padded RGB)
X = torch.randn(8, 192, 128, 4, device="cuda"); W = torch.randn(80, 4, 4, 4, device="cuda") / 8; B = torch.randn(80, device="cuda")
def stem():
    out = torch.empty((8, 48, 32, 80), device="cuda")
    nv.check(L.mtgv_op_conv2d(nv.ptr(X), nv.ptr(W), nv.ptr(B), nv.ptr(out), 8, 192, 128, 4, 80, 4, 4, 4, 0, 0, nv.stream()))
    return out
check("conv 4x4 s4", stem)
