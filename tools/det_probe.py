#!/usr/bin/env python3
"""Detector b=32 under the schedule switches (MTGV_DET_FORK, MTGV_PROTO_UP1), in one process: ms per forward.
    python tools/det_probe.py [yolov8n-seg|yolo11n-seg]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import torch
from mtgv import spec
from mtgv.detector import Detector

arch = sys.argv[1] if len(sys.argv) > 1 else "yolov8n-seg"
cfg = spec.DetectorConfig() if arch == "yolov8n-seg" else spec.yolo11_config()
det = Detector(cfg, spec.random_detector_state(cfg, 3), max_batch=32)
fr = torch.randint(0, 256, (32, 640, 640, 3), device="cuda", dtype=torch.uint8)

def timeit(fn, warm=5, it=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

for rep in range(2):
    for fork, up1, chain in (("0", "0", "0"), ("0", "1", "0"), ("0", "1", "1"), ("1", "0", "0"), ("1", "1", "0"), ("1", "1", "1")):
        os.environ["MTGV_DET_FORK"], os.environ["MTGV_PROTO_UP1"], os.environ["MTGV_SPPF_POOLS1"] = fork, up1, up1
        os.environ["MTGV_DET_CHAIN"] = chain
        ms = timeit(lambda: det.forward(fr, True, 8))
        print(f"{arch} b=32 fork={fork} up1={up1} chain={chain}: {ms:.3f} ms  {32 / ms * 1e3:.0f} frames/s", flush=True)
