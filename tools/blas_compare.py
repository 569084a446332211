#!/usr/bin/env python3
"""Context for the roofline fractions: what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS, fp16 and bf16 operands, one
MFMA per product, fp16/bf16 output) reaches on the pointwise-layer shapes of the encoder and on a large square, next to this
library's split-precision launches (three MFMAs per product, f32-grade output) taken from a launch trace of one step.
Not part of the product path; run on the GPU box: python tools/blas_compare.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import torch

def timeit(fn, warm=5, it=30):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

shapes = [("s0.pw1", 393216, 384, 96), ("s0.pw2", 393216, 96, 384), ("s1.pw1", 98304, 768, 192), ("s1.pw2", 98304, 192, 768),
          ("s2.pw1", 24576, 1536, 384), ("s2.pw2", 24576, 384, 1536), ("s3.pw1", 6144, 3072, 768), ("s3.pw2", 6144, 768, 3072),
          ("bank", 256, 100000, 768), ("square", 8192, 8192, 4096)]
print(f"{'shape':8s} {'M':>7s} {'N':>6s} {'K':>5s} | fp16 us  TFLOP/s  frac of 2.5 PF | bf16 us  TFLOP/s | f32 (torch) us  TFLOP/s")
for name, M, N, K in shapes:
    row = []
    for dt in (torch.float16, torch.bfloat16, torch.float32):
        a = torch.randn((M, K), device="cuda", dtype=dt)
        w = torch.randn((N, K), device="cuda", dtype=dt)
        out = torch.empty((M, N), device="cuda", dtype=dt)
        ms = timeit(lambda: torch.matmul(a, w.t(), out=out))
        row.append((ms * 1e3, 2.0 * M * N * K / ms / 1e9))
        del a, w, out
    print(f"{name:8s} {M:7d} {N:6d} {K:5d} | {row[0][0]:7.1f} {row[0][1]:8.1f} {row[0][1] / 2500:8.3f}        | {row[1][0]:7.1f} {row[1][1]:8.1f} | {row[2][0]:8.1f} {row[2][1]:8.1f}", flush=True)
