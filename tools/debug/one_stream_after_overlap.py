#!/usr/bin/env python3
"""Does the one-stream schedule (Pipeline.run on the current stream) slow down once the overlapped schedule has run in the
process?  Times 20 x pipe.run at each point."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
m.add(torch.randn((100_000, 768), device="cuda"))
pipe = Pipeline(Detector(det_cfg, spec.random_detector_state(det_cfg, 3), max_batch=F),
                Encoder(enc_cfg, spec.random_encoder_state(enc_cfg, 1), max_batch=F * K), m, K, 1, quad_source="mask")
batches = [torch.randint(0, 256, (F, 640, 640, 3), device="cuda", dtype=torch.uint8) for _ in range(4)]


def t_run(tag):
    for i in range(5): pipe.run(batches[i % 4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): pipe.run(batches[i % 4])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    keep = [pipe.run(batches[i % 4]) for i in range(20)]   # (bench.py keeps the step results of a timed region alive)
    torch.cuda.synchronize(); dk = time.perf_counter() - t0
    del keep
    print(f"{tag}: {dt / 20 * 1e3:.3f} ms per step = {F * K * 20 / dt:.0f} cards/s; results kept: {F * K * 20 / dk:.0f} cards/s; "
          f"reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "overlap-first":   # like bench.py: the first pass over the pipeline is the overlapped schedule
    os.environ["MTGV_OVERLAP"] = "on"
    os.environ["MTGV_STREAM_PRIO"] = sys.argv[2] if len(sys.argv) > 2 else "enc"
    pipe.run_many([batches[i % 4] for i in range(20)]); torch.cuda.synchronize()
    t_run(f"one stream, after a FIRST pass with run_many (MTGV_STREAM_PRIO={os.environ['MTGV_STREAM_PRIO']})")
    os.environ["MTGV_DET_FORK"] = "0"
    t_run("  ... with MTGV_DET_FORK=0")
    sys.exit(0)
t_run("fresh process")
s_hi = torch.cuda.Stream(priority=-1)
t_run("after creating an (idle) high-priority stream")
os.environ["MTGV_OVERLAP"] = "on"
for which in (("MTGV_STREAM_PRIO", "none"), ("MTGV_STREAM_PRIO", "enc")):
    os.environ[which[0]] = which[1]
    for a in ("_s_det", "_s_enc"):
        if hasattr(pipe, a): delattr(pipe, a)
    pipe.run_many([batches[i % 4] for i in range(20)]); torch.cuda.synchronize()
    per = []
    for i in range(24):  # step by step right after the overlapped run: is there a transient?
        t0 = time.perf_counter(); pipe.run(batches[i % 4]); torch.cuda.synchronize(); per.append((time.perf_counter() - t0) * 1e3)
    print("  per-step ms right after run_many:", " ".join(f"{x:.1f}" for x in per), flush=True)
    pipe.run_many([batches[i % 4] for i in range(20)]); torch.cuda.synchronize()
    pipe.run(batches[0])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    keep = [pipe.run(batches[i % 4]) for i in range(20)]
    torch.cuda.synchronize(); dk = time.perf_counter() - t0
    del keep
    print(f"  bench-style (1 warm step, 20 timed, results kept) right after run_many: {F * K * 20 / dk:.0f} cards/s", flush=True)
    t_run(f"after run_many with {which[0]}={which[1]}")
