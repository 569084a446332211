#!/usr/bin/env python3
"""Soak: bench-sized pipeline (mask quads), every output compared bit for bit against the fully serial schedule (one stream,
MTGV_DET_FORK=0): the overlapped run_many (detect + crop / embed / match streams, the latter two at high priority; the detector's fork-join off inside it), and
the same with the frames arriving from pinned host memory on a copy stream (HostFrames).  MTGV_OVERLAP=on is set here."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import numpy as np, torch
from mtgv import native, spec
from mtgv.detector import Detector
from mtgv.encoder import Encoder
from mtgv.matcher import Matcher
from mtgv.pipeline import HostFrames, Pipeline

os.environ["MTGV_OVERLAP"] = "on"

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for mode in ("f16x3", "f32"):
    native.set_gemm_precision(mode)
    F, K = 32, 8
    det_cfg = spec.DetectorConfig(); enc_cfg = spec.encoder_config("cnvnxt2ae_tiny")
    m = Matcher(768, capacity=100_000)
    m.add(torch.randn((100_000, 768), generator=torch.Generator(device="cuda").manual_seed(2), device="cuda"))
    pipe = Pipeline(Detector(det_cfg, spec.random_detector_state(det_cfg, 3), max_batch=F),
                    Encoder(enc_cfg, spec.random_encoder_state(enc_cfg, 1), max_batch=F * K), m, K, 1, quad_source="mask")
    g = torch.Generator(device="cuda").manual_seed(4)
    batches = [torch.randint(0, 256, (F, 640, 640, 3), generator=g, device="cuda", dtype=torch.uint8) for _ in range(6)]
    os.environ["MTGV_DET_FORK"] = "0"
    ref = [pipe.run(b) for b in batches]
    torch.cuda.synchronize()
    os.environ.pop("MTGV_DET_FORK")
    src = HostFrames([b.cpu() for b in batches], "cuda")
    bad = 0
    for r in range(rounds):
        for what, outs in (("two streams + fork-join", pipe.run_many(batches)), ("host frames", pipe.run_many(src.leases(len(batches))))):
            torch.cuda.synchronize()
            for a, b in zip(ref, outs):
                for k in ("ids", "scores", "z", "crops", "boxes", "n_det"):
                    if not torch.equal(a[k], b[k]):
                        bad += 1
                        print(f"{mode} round {r} [{what}]: {k} differs ({(a[k] != b[k]).sum().item()} elements)", flush=True)
    print(f"{mode}: {rounds} rounds x 2 schedules x {len(batches)} batches of {F} frames, mismatching outputs: {bad}", flush=True)
