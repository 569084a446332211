#!/usr/bin/env python3
"""Debug aid: encoder on one stream beside the detector on another - where does the first difference appear?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import numpy as np, torch
from mtgv import native, spec
from mtgv.detector import Detector
from mtgv.encoder import Encoder

mode = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
native.set_gemm_precision(mode)
det_cfg = spec.DetectorConfig(); enc_cfg = spec.encoder_config("cnvnxt2ae_nano")
F, N = 2, 8
det = Detector(det_cfg, spec.random_detector_state(det_cfg, 3), max_batch=F)
enc = Encoder(enc_cfg, spec.random_encoder_state(enc_cfg, 1), max_batch=N)
g = torch.Generator(device="cuda").manual_seed(11)
frames = torch.randint(0, 256, (F, 640, 640, 3), generator=g, device="cuda", dtype=torch.uint8)
crops = torch.randint(0, 256, (N, 192, 128, 3), generator=g, device="cuda", dtype=torch.uint8)
enc.set_capture(True)
z_ref = enc.encode(crops).clone()
st_ref = [enc.stage_output(s, N).clone() for s in range(4)]
torch.cuda.synchronize()
s_det, s_enc = torch.cuda.Stream(), torch.cuda.Stream()
nbad = 0
other = os.environ.get("OTHER", "det")
A = torch.randn(4096, 4096, device="cuda"); B = torch.randn(4096, 4096, device="cuda")
for trial in range(30):
    with torch.cuda.stream(s_det):
        if other == "det":
            det.forward(frames, True, mask_rows=4)
        elif other == "mm":
            C_ = A @ B
        elif other == "copy":
            C_ = A.clone(); C_ += 1
    with torch.cuda.stream(s_enc):
        z = enc.encode(crops)
        st = [enc.stage_output(s, N) for s in range(4)]
    torch.cuda.synchronize()
    if not torch.equal(z, z_ref):
        nbad += 1
        for s in range(4):
            d = (st[s] != st_ref[s])
            if d.any():
                idx = d.nonzero()
                cards = sorted(set(idx[:, 0].tolist()))
                h, w, c = st[s].shape[1:]
                pix = (idx[:, 1] * w + idx[:, 2])
                rows = idx[:, 0] * h * w + pix  # GEMM row index m
                tiles = sorted(set((rows // 128).tolist()))
                chans = idx[:, 3]
                print(f"trial {trial}: first diff at stage {s}: {d.sum().item()} elems, cards {cards}, max diff {(st[s]-st_ref[s]).abs().max().item():.3e}, "
                      f"128-row tiles {tiles[:12]}{'...' if len(tiles) > 12 else ''} ({len(tiles)}), channels {chans.min().item()}..{chans.max().item()} "
                      f"({len(set(chans.tolist()))} distinct), rows in first tile: {sorted(set((rows[rows // 128 == tiles[0]] % 128).tolist()))[:20]}")
                break
print(mode, "bad trials:", nbad, "of 30")
