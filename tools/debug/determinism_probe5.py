#!/usr/bin/env python3
"""Debug aid: which workspace region of one ConvNeXt block goes wrong beside the detector?"""
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
n, h, w, c = 8, 48, 32, 80
M = n * h * w
r = lambda *s: torch.randn(*s, device="cuda")
X = r(n, h, w, c)
P = dict(dw=r(49, c) / 7, dwb=0.1 * r(c), lnw=1 + 0.1 * r(c), lnb=0.1 * r(c), w1=r(4 * c, c) / c ** 0.5, b1=0.1 * r(4 * c),
         ga=0.3 * r(4 * c), be=0.1 * r(4 * c), w2=r(c, 4 * c) / (4 * c) ** 0.5, b2=0.1 * r(c))
nws = int(L.mtgv_op_block_workspace_floats(n, h, w, c))
npart = int(L.mtgv_op_linear_ex_part_floats(M, 4 * c, c, 2, h * w))
regions = [("t1", M * c), ("t2(ln out)", M * c), ("hid", M * 4 * c), ("part", npart), ("scale", n * 4 * c), ("bfold", c)]
print("workspace floats", nws, "regions", regions, "sum", sum(s for _, s in regions))
def blk():
    ws = torch.zeros(nws, device="cuda")
    out = torch.empty((n, h, w, c), device="cuda")
    nv.check(L.mtgv_op_block(nv.ptr(X), nv.ptr(out), n, h, w, c, 2, nv.ptr(P["dw"]), nv.ptr(P["dwb"]), nv.ptr(P["lnw"]), nv.ptr(P["lnb"]),
                             nv.ptr(P["w1"]), nv.ptr(P["b1"]), nv.ptr(P["ga"]), nv.ptr(P["be"]), nv.ptr(P["w2"]), nv.ptr(P["b2"]), nv.ptr(ws), nv.stream()))
    return out, ws
o_ref, ws_ref = blk()
torch.cuda.synchronize()
shown = 0
nfail = 0
for t in range(30):
    with torch.cuda.stream(s_det):
        det.forward(frames, True, mask_rows=4)
    with torch.cuda.stream(s_op):
        o, ws = blk()
    torch.cuda.synchronize()
    nfail += int(not torch.equal(o, o_ref))
    if not torch.equal(o, o_ref) and shown < int(os.environ.get('SHOW', '0')):
        shown += 1
        off = 0; msg = []
        for name, size in regions:
            a, b = ws[off:off + size], ws_ref[off:off + size]
            d = (a != b)
            if d.any():
                idx = d.nonzero().flatten()
                msg.append(f"{name}: {d.sum().item()} diff, first idx {idx[0].item()} last {idx[-1].item()}, maxdiff {(a - b).abs().max().item():.3e}, e.g. got {a[idx[0]].item():.6g} want {b[idx[0]].item():.6g}")
            off += size
        a, b = ws[M * c:2 * M * c], ws_ref[M * c:2 * M * c]
        idx = (a != b).nonzero().flatten()
        rows = [(int(i) // c, int(i) % c, round(a[i].item(), 5), round(b[i].item(), 5), round(P["lnb"][int(i) % c].item(), 5)) for i in idx[:24]]
        print("   t2 diffs (pixel, ch, got, want, ln_b[ch]):", rows)
        print(f"trial {t}: out diff {(o != o_ref).sum().item()} | " + " | ".join(msg), flush=True)
print("failing trials", nfail, "of 30")
