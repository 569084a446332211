#!/usr/bin/env python3
"""CPU study for a later round: would a split-fp16 GEMM (x = hi + lo in fp16, three f16 MFMAs per product with
fp32 accumulation: hi*hi + hi*lo + lo*hi) keep the encoder within the 1e-4 contract?  Emulated in numpy/torch on the
oracle's forward (every pointwise / patchify / head Linear replaced), compared with the fp64 reference.
    python tools/split_precision_study.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import numpy as np, torch
import torch.nn.functional as F
from mtgv import spec
from oracle import encoder_ref as R

MODE = {"mode": "f32"}
def split16(t, dtype):
    hi = t.to(dtype).float()
    lo = (t - hi).to(dtype).float()
    return hi, lo
_orig_linear = F.linear
def linear(x, w, b=None):
    m = MODE["mode"]
    if m == "f32" or x.dtype != torch.float32:
        return _orig_linear(x, w, b)
    dt = torch.bfloat16 if m.startswith("bf16") else torch.float16
    if m in ("f16", "bf16"):
        y = _orig_linear(x.to(dt).float(), w.to(dt).float())
    else:
        xh, xl = split16(x, dt); wh, wl = split16(w, dt)
        y = _orig_linear(xh, wh) + _orig_linear(xh, wl) + _orig_linear(xl, wh)
        if m.endswith("x4"):
            y = y + _orig_linear(xl, wl)
    return y if b is None else y + b
R.F.linear = linear  # the oracle's pointwise / head layers; convs (stem, downsample, depthwise) stay fp32

for name in ("cnvnxt2ae_nano", "cnvnxt2ae_tiny"):
    cfg = spec.encoder_config(name)
    sd = spec.random_encoder_state(cfg, 1)
    x = np.random.default_rng(0).random((4, 3, *cfg.image_hw), dtype=np.float32)
    MODE["mode"] = "f32"
    z64 = R.encoder_forward(sd, cfg, x, dtype=torch.float64).numpy()
    out = []
    for m in ("f32", "f16", "bf16", "f16x3", "f16x4", "bf16x3"):
        MODE["mode"] = m
        z = R.encoder_forward(sd, cfg, x).numpy()
        out.append(f"{m}: {np.abs(z - z64).max():.2e}")
    print(name, "max|z - fp64| with pointwise+head GEMMs in ->", " | ".join(out), flush=True)
