#!/usr/bin/env python3
"""One pipeline step (bench workload, one stream) under the round-4 schedule switches, A/B in one process:
    python tools/step_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import torch
from mtgv import spec
from mtgv.detector import Detector
from mtgv.encoder import Encoder
from mtgv.matcher import Matcher
from mtgv.pipeline import Pipeline

F, K = 32, 8
det_cfg = spec.DetectorConfig(); enc_cfg = spec.encoder_config("cnvnxt2ae_tiny")
m = Matcher(768, capacity=100_000)
m.add(torch.randn((100_000, 768), generator=torch.Generator(device="cuda").manual_seed(2), device="cuda"))
pipe = Pipeline(Detector(det_cfg, spec.random_detector_state(det_cfg, 3), max_batch=F),
                Encoder(enc_cfg, spec.random_encoder_state(enc_cfg, 1), max_batch=F * K), m, K, 1, quad_source="mask")
g = torch.Generator(device="cuda").manual_seed(4)
batches = [torch.randint(0, 256, (F, 640, 640, 3), generator=g, device="cuda", dtype=torch.uint8) for _ in range(4)]

def timeit(fn, warm=8, it=40):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(it): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

sw = ["MTGV_DET_FORK", "MTGV_PROTO_UP1", "MTGV_SPPF_POOLS1", "MTGV_DW_STREAM"]
cases = [("round-3 schedule (all off)", dict.fromkeys(sw, "0")), ("fork only", {**dict.fromkeys(sw, "0"), "MTGV_DET_FORK": "1"}),
         ("fork + upsample/pools in one launch", {**dict.fromkeys(sw, "1"), "MTGV_DW_STREAM": "0"}), ("all on (default)", dict.fromkeys(sw, "1"))]
for rep in range(2):
    for name, env in cases:
        os.environ.update(env)
        one = timeit(lambda i: pipe.run(batches[i % 4]))
        os.environ["MTGV_OVERLAP"] = "on"
        two = timeit(lambda i: pipe.run_many([batches[(2 * i) % 4], batches[(2 * i + 1) % 4]]), warm=4, it=20) / 2
        os.environ.pop("MTGV_OVERLAP")
        print(f"{name}: one stream {one:.3f} ms ({F * K / one * 1e3:.0f} cards/s)   two streams {two:.3f} ms ({F * K / two * 1e3:.0f} cards/s)", flush=True)
