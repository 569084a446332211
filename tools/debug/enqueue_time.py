#!/usr/bin/env python3
"""Host time to ENQUEUE the overlapped schedule against the GPU time it takes: python tools/debug/enqueue_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
os.environ["MTGV_OVERLAP"] = "on"
import torch
from mtgv import spec
from mtgv.detector import Detector
from mtgv.encoder import Encoder
from mtgv.matcher import Matcher
from mtgv.pipeline import Pipeline
F, K = 32, 8
det_cfg = spec.DetectorConfig(); enc_cfg = spec.encoder_config("cnvnxt2ae_tiny")
ORDER = os.environ.get("ORDER", "matcher-first")   # which handle exists (and has launched kernels) before the detector creates its branch streams
if ORDER == "matcher-first":
    m = Matcher(768, capacity=100_000); m.add(torch.randn((100_000, 768), device="cuda"))
det = Detector(det_cfg, spec.random_detector_state(det_cfg, 3), max_batch=F)
enc = Encoder(enc_cfg, spec.random_encoder_state(enc_cfg, 1), max_batch=F * K)
if ORDER != "matcher-first":
    m = Matcher(768, capacity=100_000); m.add(torch.randn((100_000, 768), device="cuda"))
pipe = Pipeline(det, enc, m, K, 1, quad_source="mask")
batches = [torch.randint(0, 256, (F, 640, 640, 3), device="cuda", dtype=torch.uint8) for _ in range(4)]
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seq = [batches[i % 4] for i in range(NS)]
pipe.run_many(seq); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); pipe.run_many(seq); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"overlapped: enqueue {(t1 - t0) / NS * 1e3:.2f} ms per step on the host, {(t2 - t0) / NS * 1e3:.2f} ms per step until the GPU is done", flush=True)
for rep in range(2):
    t0 = time.perf_counter(); [pipe.run(b) for b in seq]; t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"one stream: enqueue {(t1 - t0) / NS * 1e3:.2f} ms per step on the host, {(t2 - t0) / NS * 1e3:.2f} ms per step until the GPU is done", flush=True)
