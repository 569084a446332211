#!/usr/bin/env python3
"""Debug aid: is the pipeline bit-reproducible run to run, sequential vs two-stream, per precision?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import numpy as np, torch
from mtgv import native, spec
from mtgv.detector import Detector
from mtgv.encoder import Encoder
from mtgv.matcher import Matcher
from mtgv.pipeline import Pipeline

KEYS = ("n_det", "boxes", "crops", "z", "ids", "scores")
for mode in ("f16x3", "f32", "f16x3-fresh", "f32-fresh"):
    native.set_gemm_precision(mode.split("-")[0])
    if mode.endswith("fresh"):  # the shape of tests/test_gpu_adapters.py: a new pipeline, its first passes are the reference
        for trial in range(8):
            det_cfg = spec.DetectorConfig(); enc_cfg = spec.encoder_config("cnvnxt2ae_nano")
            F, K = 2, 4
            m = Matcher(768, capacity=3000); m.add(np.random.default_rng(2).standard_normal((3000, 768)).astype(np.float32))
            pipe = Pipeline(Detector(det_cfg, spec.random_detector_state(det_cfg, 3), max_batch=F),
                            Encoder(enc_cfg, spec.random_encoder_state(enc_cfg, 1), max_batch=F * K), m, K, 3)
            g = torch.Generator(device="cuda").manual_seed(11)
            batches = [torch.randint(0, 256, (F, 640, 640, 3), generator=g, device="cuda", dtype=torch.uint8) for _ in range(3)]
            seq = [pipe.run(b) for b in batches]
            ovl = pipe.run_many(batches)
            again = [pipe.run(b) for b in batches]
            torch.cuda.synchronize()
            for name, outs in (("ovl", ovl), ("again", again)):
                for j, (a, b) in enumerate(zip(seq, outs)):
                    bad = [k for k in KEYS if not torch.equal(a[k], b[k])]
                    det_bad = [k for k in a["det"] if a["det"][k] is not None and not torch.equal(a["det"][k], b["det"][k])]
                    if bad or det_bad:
                        dz = (a["z"] - b["z"]).abs().amax(dim=1)
                        print(f"{mode} trial {trial} first-seq vs {name} batch {j}: differ {bad} det {det_bad}; rows dz>0 {(dz > 0).nonzero().flatten().tolist()} "
                              f"max dz {dz.max().item():.2e} crops diff px {(a['crops'] != b['crops']).sum().item()}")
        print(mode, "done", flush=True)
        continue
    det_cfg = spec.DetectorConfig(); enc_cfg = spec.encoder_config("cnvnxt2ae_nano")
    F, K = 2, 4
    m = Matcher(768, capacity=3000); m.add(np.random.default_rng(2).standard_normal((3000, 768)).astype(np.float32))
    pipe = Pipeline(Detector(det_cfg, spec.random_detector_state(det_cfg, 3), max_batch=F),
                    Encoder(enc_cfg, spec.random_encoder_state(enc_cfg, 1), max_batch=F * K), m, K, 3)
    g = torch.Generator(device="cuda").manual_seed(11)
    batches = [torch.randint(0, 256, (F, 640, 640, 3), generator=g, device="cuda", dtype=torch.uint8) for _ in range(3)]
    ref = [pipe.run(b) for b in batches]
    torch.cuda.synchronize()
    for trial in range(6):
        seq = [pipe.run(b) for b in batches]
        ovl = pipe.run_many(batches)
        torch.cuda.synchronize()
        for name, outs in (("seq", seq), ("ovl", ovl)):
            for j, (a, b) in enumerate(zip(ref, outs)):
                bad = [k for k in KEYS if not torch.equal(a[k], b[k])]
                det_bad = [k for k in a["det"] if a["det"][k] is not None and not torch.equal(a["det"][k], b["det"][k])]
                if bad or det_bad:
                    dz = (a["z"] - b["z"]).abs().amax(dim=1)
                    print(f"{mode} trial {trial} {name} batch {j}: differ {bad} det {det_bad}; rows with dz>0: {(dz > 0).nonzero().flatten().tolist()} max dz {dz.max().item():.2e}"
                          f" crops diff px {(a['crops'] != b['crops']).sum().item()}")
    print(mode, "done", flush=True)
