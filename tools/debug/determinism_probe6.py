#!/usr/bin/env python3
"""Debug aid: one ConvNeXt block (f32 GEMMs!) beside a synthetic aggressor kernel on another stream."""
import ctypes as C, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import torch
from mtgv import native as nv
L = nv.lib()
so = "/tmp/libaggr.so"
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(ROOT, "tools/micro/aggressor.hip"), "-o", so])
A = C.CDLL(so)
A.aggr_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
nv.set_gemm_precision(prec)
s_ag, s_op = torch.cuda.Stream(), torch.cuda.Stream()
n, h, w, c = 8, 48, 32, 80
r = lambda *s: torch.randn(*s, device="cuda")
X = r(n, h, w, c)
P = dict(dw=r(49, c) / 7, dwb=0.1 * r(c), lnw=1 + 0.1 * r(c), lnb=0.1 * r(c), w1=r(4 * c, c) / c ** 0.5, b1=0.1 * r(4 * c),
         ga=0.3 * r(4 * c), be=0.1 * r(4 * c), w2=r(c, 4 * c) / (4 * c) ** 0.5, b2=0.1 * r(c))
nws = int(L.mtgv_op_block_workspace_floats(n, h, w, c))
sink = torch.zeros(16, device="cuda")
def blk():
    ws = torch.zeros(nws, device="cuda")
    out = torch.empty((n, h, w, c), device="cuda")
    nv.check(L.mtgv_op_block(nv.ptr(X), nv.ptr(out), n, h, w, c, 2, nv.ptr(P["dw"]), nv.ptr(P["dwb"]), nv.ptr(P["lnw"]), nv.ptr(P["lnb"]),
                             nv.ptr(P["w1"]), nv.ptr(P["b1"]), nv.ptr(P["ga"]), nv.ptr(P["be"]), nv.ptr(P["w2"]), nv.ptr(P["b2"]), nv.ptr(ws), nv.stream()))
    return out
ref = blk()
torch.cuda.synchronize()
for mode, name in ((0, "f16 MFMA x3 chains"), (1, "f32 MFMA x3 chains"), (2, "f16 MFMA + LDS reads"), (3, "LDS reads only"), (4, "f16 MFMA 1 chain")):
    bad = 0
    for t in range(40):
        with torch.cuda.stream(s_ag):
            rc = A.aggr_launch(mode, 2048, 3000, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        with torch.cuda.stream(s_op):
            o = blk()
        torch.cuda.synchronize()
        bad += int(not torch.equal(o, ref))
    print(f"library GEMMs {prec}; aggressor {name}: block mismatches {bad}/40", flush=True)
