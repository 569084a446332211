#!/usr/bin/env python3
"""Encoder batch 256 on one stream vs split into S sub-batches on S streams (S encoder handles): does the VALU-bound
dwconv7_ln of one sub-batch hide under the MFMA / LDS-fill-bound GEMMs of another?   python tools/enc_split_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import torch
from mtgv import spec
from mtgv.encoder import Encoder

B = int(os.environ.get("B", 256))
cfg = spec.encoder_config(os.environ.get("ENC", "cnvnxt2ae_tiny"))
sd = spec.random_encoder_state(cfg, 1)
x = torch.randint(0, 256, (B, *cfg.image_hw, 3), device="cuda", dtype=torch.uint8)

def timeit(fn, warm=3, it=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

ref = None
for S in (1, 2, 4):
    encs = [Encoder(cfg, sd, max_batch=B // S) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    outs = [None] * S
    def run():
        cur = torch.cuda.current_stream()
        for i in range(S):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                outs[i] = encs[i].encode(x[i * (B // S):(i + 1) * (B // S)])
        for i in range(S):
            cur.wait_stream(streams[i])
    ms = timeit(run)
    z = torch.cat(outs)
    if ref is None: ref = z.clone()
    print(f"{S} stream(s) x {B // S} cards: {ms:.3f} ms  {B / ms * 1e3:.0f} img/s  bit-identical to 1 stream: {bool(torch.equal(z, ref))}", flush=True)
    del encs

# which is it when a split differs: the sub-batch size (deterministic) or the concurrency (a race)?
for S in (4, 8):
    encs = [Encoder(cfg, sd, max_batch=B // S) for _ in range(S)]
    seq = torch.cat([encs[i].encode(x[i * (B // S):(i + 1) * (B // S)]) for i in range(S)])
    torch.cuda.synchronize()
    one = torch.cat([encs[0].encode(x[i * (B // S):(i + 1) * (B // S)]) for i in range(S)])
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(S)]
    outs = [None] * S
    cur = torch.cuda.current_stream()
    for rep in range(3):
        for i in range(S):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                outs[i] = encs[i].encode(x[i * (B // S):(i + 1) * (B // S)])
        for i in range(S):
            cur.wait_stream(streams[i])
        torch.cuda.synchronize()
        par = torch.cat(outs)
        print(f"S={S}: sequential(S handles) == ref {bool(torch.equal(seq, ref))} (max diff {(seq - ref).abs().max().item():.3e}); one handle == ref {bool(torch.equal(one, ref))}; "
              f"concurrent == sequential {bool(torch.equal(par, seq))} (max diff {(par - seq).abs().max().item():.3e}, rows differing {int(((par != seq).any(1)).sum())})", flush=True)
    del encs
