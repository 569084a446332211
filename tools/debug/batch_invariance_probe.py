#!/usr/bin/env python3
"""Debug aid: detector raw outputs of a frame alone vs inside a batch of 32, per precision."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import torch
from mtgv import native, spec
from mtgv.detector import Detector
for mode in ("f32", "f16x3"):
    native.set_gemm_precision(mode)
    cfg = spec.DetectorConfig()
    det = Detector(cfg, spec.random_detector_state(cfg, 3), max_batch=32)
    g = torch.Generator(device="cuda").manual_seed(4)
    frames = torch.randint(0, 256, (32, 640, 640, 3), generator=g, device="cuda", dtype=torch.uint8)
    det.forward(frames, True, 8)
    pred32, prot32 = [t.clone() for t in det.raw_outputs(32)]
    for i in (0, 13, 31):
        det.forward(frames[i:i + 1], True, 8)
        pred1, prot1 = det.raw_outputs(1)
        dp = (pred1[0] != pred32[i]); dq = (prot1[0] != prot32[i])
        rows = sorted(set(dp.nonzero()[:, 0].tolist()))
        print(f"{mode} frame {i}: pred diff {dp.sum().item()} of {dp.numel()} (channels {rows[:8]}{'...' if len(rows) > 8 else ''}), max {((pred1[0]-pred32[i]).abs().max().item()):.3e}; "
              f"protos diff {dq.sum().item()} of {dq.numel()}, max {((prot1[0]-prot32[i]).abs().max().item()):.3e}", flush=True)
