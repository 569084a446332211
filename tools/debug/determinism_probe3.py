#!/usr/bin/env python3
"""Debug aid: single GEMM variants (and other encoder ops) on one stream beside the detector on another."""
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

def variant(name, m, n, k, act, hw, res, scale, grn):
    a = torch.randn((m, k), device="cuda"); w = torch.randn((n, k), device="cuda") * k ** -0.5
    b = torch.randn((n,), device="cuda")
    r = torch.randn((m, n), device="cuda") if res else None
    sc = torch.rand((m // hw, k), device="cuda") + 0.5 if scale else None
    npart = int(L.mtgv_op_linear_ex_part_floats(m, n, k, act, hw)) + 4 if grn else 0
    def f():
        o = torch.empty((m, n), device="cuda")
        part = torch.zeros(npart, device="cuda") if grn else None
        nv.check(L.mtgv_op_linear_ex(nv.ptr(a), nv.ptr(w), nv.ptr(b), nv.ptr(r), nv.ptr(o), m, n, k, act, hw, nv.ptr(sc), None, nv.ptr(part), nv.stream()))
        return o, part
    o_ref, p_ref = f()
    torch.cuda.synchronize()
    bad_seq = bad_ovl = 0
    info = ""
    for trial in range(12):
        o, p = f()
        torch.cuda.synchronize()
        bad_seq += int(not torch.equal(o, o_ref) or (grn and not torch.equal(p, p_ref)))
    for trial in range(24):
        with torch.cuda.stream(s_det):
            det.forward(frames, True, mask_rows=4)
        with torch.cuda.stream(s_op):
            o, p = f()
        torch.cuda.synchronize()
        d = o != o_ref
        pb = grn and not torch.equal(p, p_ref)
        if d.any() or pb:
            bad_ovl += 1
            if not info and d.any():
                idx = d.nonzero()
                rows = sorted(set(idx[:, 0].tolist())); cols = sorted(set(idx[:, 1].tolist()))
                info = f" first bad: {d.sum().item()} elems rows {rows[:6]}..{rows[-1]} ({len(rows)}) cols {cols[:4]}..{cols[-1]} ({len(cols)}) maxdiff {(o - o_ref).abs().max().item():.2e} part_bad {pb}"
            elif not info:
                info = " only partials differ"
    print(f"{mode} {name:28s} M={m} N={n} K={k}: seq mismatches {bad_seq}/12, overlapped {bad_ovl}/24{info}", flush=True)

for (c, hw, nimg) in ((80, 1536, 8), (160, 384, 8), (320, 96, 8), (640, 24, 8)):
    m = nimg * hw
    variant("plain", m, 4 * c, c, 0, hw, 0, 0, 0)
    variant("pw1 mish+grn-partials", m, 4 * c, c, 2, hw, 0, 0, 1)
    variant("pw2 plain", m, c, 4 * c, 0, hw, 0, 0, 0)
    variant("pw2 +res", m, c, 4 * c, 0, hw, 1, 0, 0)
    variant("pw2 +scale", m, c, 4 * c, 0, hw, 0, 1, 0)
    variant("pw2 +scale+res", m, c, 4 * c, 0, hw, 1, 1, 0)
