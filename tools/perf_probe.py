#!/usr/bin/env python3
"""Quick per-stage timing on the GPU box (not the contract bench; see bench.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mtg-vision_amd"))
import numpy as np, torch
from mtgv import spec
from mtgv.encoder import Encoder
from mtgv.matcher import Matcher

def timeit(fn, warm=3, it=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

which = sys.argv[1:] or ["ae_tiny", "ae_nano", "plain_tiny", "match", "detector"]
B = int(os.environ.get("B", 256))
for name in [w for w in which if w != "detector"]:
    if name == "match":
        g = torch.Generator(device="cuda").manual_seed(2)
        bank = torch.randn((100_000, 768), generator=g, device="cuda")
        m = Matcher(768, capacity=100_000); m.add(bank)
        for b in (8, 64, 256, 1024):
            q = torch.randn((b, 768), generator=g, device="cuda")
            ms = timeit(lambda: m.match(q, 1))
            fl = 2.0 * b * 100_000 * 768
            print(f"match b={b}: {ms:.3f} ms  {b/ms*1e3:.0f} q/s  {fl/ms/1e9:.1f} TFLOP/s  bank {307.2/ms:.1f} GB/s... ", flush=True)
        continue
    cfg = {"ae_tiny": spec.encoder_config("cnvnxt2ae_tiny"), "ae_nano": spec.encoder_config("cnvnxt2ae_nano"),
           "plain_tiny": spec.encoder_config("convnextv2_tiny", (224, 224))}[name]
    enc = Encoder(cfg, spec.random_encoder_state(cfg, 1), max_batch=B)
    x = torch.rand((B, 3, *cfg.image_hw), device="cuda")
    ms = timeit(lambda: enc.encode(x))
    gf, df = enc.flops_per_image()
    print(f"{name} b={B}: {ms:.2f} ms  {B/ms*1e3:.0f} img/s  gemm {gf*B/ms/1e9:.1f} TFLOP/s ({gf/1e9:.3f} GF/img + dw {df/1e9:.3f})", flush=True)

if "detector" in which:
    from mtgv.detector import Detector
    dc = spec.DetectorConfig()
    det = Detector(dc, spec.random_detector_state(dc, 3), max_batch=32)
    fr = torch.randint(0, 256, (32, 640, 640, 3), device="cuda", dtype=torch.uint8)
    ms = timeit(lambda: det.forward(fr, True, 8))
    print(f"detector b=32: {ms:.2f} ms  {32/ms*1e3:.0f} frames/s  {det.flops_per_frame()*32/ms/1e9:.1f} TFLOP/s ({det.flops_per_frame()/1e9:.2f} GF/frame)", flush=True)
