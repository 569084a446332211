#!/usr/bin/env python3
"""Per-launch GEMM table of one pipeline step (tuning aid): python tools/gemm_trace.py out.csv"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import numpy as np, torch
from mtgv import native, spec
from mtgv.detector import Detector
from mtgv.encoder import Encoder
from mtgv.matcher import Matcher
from mtgv.pipeline import Pipeline

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/gemm_trace.csv"
enc_name = os.environ.get("ENC", "cnvnxt2ae_tiny")
F, K = 32, 8
det_cfg = spec.DetectorConfig(); enc_cfg = spec.encoder_config(enc_name)
det = Detector(det_cfg, spec.random_detector_state(det_cfg, 3), max_batch=F)
enc = Encoder(enc_cfg, spec.random_encoder_state(enc_cfg, 1), max_batch=F * K)
m = Matcher(768, capacity=100_000)
m.add(torch.randn((100_000, 768), device="cuda"))
pipe = Pipeline(det, enc, m, K, 1)
frames = torch.randint(0, 256, (F, 640, 640, 3), device="cuda", dtype=torch.uint8)
for _ in range(2): pipe.run(frames)
torch.cuda.synchronize()
L = native.lib()
native.check(L.mtgv_profile_gemm(1))
pipe.run(frames); torch.cuda.synchronize()
native.check(L.mtgv_profile_gemm_dump(out.encode()))
native.check(L.mtgv_profile_gemm(0))
import csv
rows = list(csv.DictReader(open(out)))
tot = sum(float(r["ms"]) for r in rows)
print(f"{len(rows)} launches, {tot:.2f} ms")
rows.sort(key=lambda r: -float(r["ms"]))
for r in rows[:40]:
    print({k: r[k] for k in ("idx", "M", "N", "K", "KH", "act", "apro", "grn", "tn", "bk", "ms", "tflops")})
