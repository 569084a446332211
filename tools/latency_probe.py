#!/usr/bin/env python3
"""Single-frame latency of the serving flow (server.py:133-207: one 640x480 frame, 3 cards, k=3)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import numpy as np, torch
from mtgv import spec
from mtgv.detector import Detector, letterbox
from mtgv.encoder import Encoder
from mtgv.matcher import Matcher
from mtgv.pipeline import Pipeline
dc, ec = spec.DetectorConfig(), spec.encoder_config("cnvnxt2ae_nano")
m = Matcher(768, capacity=100_000); m.add(torch.randn((100_000, 768), device="cuda"))
pipe = Pipeline(Detector(dc, spec.random_detector_state(dc, 3), max_batch=1), Encoder(ec, spec.random_encoder_state(ec, 1), max_batch=3), m, 3, 3)
frame = np.random.default_rng(0).integers(0, 256, (480, 640, 3), dtype=np.uint8)
def once():
    img, _, _ = letterbox(frame)
    x = torch.from_numpy(img)[None].cuda()
    o = pipe.run(x)
    return o["ids"].cpu()
for _ in range(5): once()
ts = []
for _ in range(50):
    torch.cuda.synchronize(); t = time.perf_counter(); once(); ts.append((time.perf_counter() - t) * 1e3)
ts.sort(); print(f"one 640x480 frame -> letterbox -> H2D -> detect -> 3 crops -> embed (nano) -> top-3 of 100k -> D2H: median {ts[25]:.2f} ms, p90 {ts[45]:.2f} ms")
