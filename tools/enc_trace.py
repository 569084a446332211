#!/usr/bin/env python3
"""Per-launch GEMM table of one encoder forward (tuning aid): python tools/enc_trace.py out.csv [model] [H] [W] [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import csv, torch
from mtgv import native, spec
from mtgv.encoder import Encoder

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/enc_trace.csv"
name = sys.argv[2] if len(sys.argv) > 2 else "convnextv2_tiny"
H, W, B = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((3, 224), (4, 224), (5, 256)))
cfg = spec.encoder_config(name, (H, W))
enc = Encoder(cfg, spec.random_encoder_state(cfg, 1), max_batch=B)
x = torch.randint(0, 256, (B, H, W, 3), device="cuda", dtype=torch.uint8)
for _ in range(2): enc.encode(x)
torch.cuda.synchronize()
L = native.lib()
native.check(L.mtgv_profile_gemm(1))
enc.encode(x); torch.cuda.synchronize()
native.check(L.mtgv_profile_gemm_dump(out.encode()))
native.check(L.mtgv_profile_gemm(0))
rows = list(csv.DictReader(open(out)))
print(f"{len(rows)} launches, {sum(float(r['ms']) for r in rows):.2f} ms")
for r in rows:
    print({k: r[k] for k in ("idx", "M", "N", "K", "KH", "act", "apro", "grn", "ms", "tflops", "sp")})
