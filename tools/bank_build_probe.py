#!/usr/bin/env python3
"""Bank-build throughput: ragged make_cropped + batched encode + append."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import numpy as np, torch
from mtgv import spec
from mtgv.adapters import VectorStoreQdrant
from mtgv.bank import build_bank, make_cropped
from mtgv.encoder import Encoder

cfg = spec.encoder_config(os.environ.get("ENC", "cnvnxt2ae_nano"))
enc = Encoder(cfg, spec.random_encoder_state(cfg, 1), max_batch=256)
rng = np.random.default_rng(0)
n = 2048
imgs = [rng.integers(0, 256, (680, 488, 3), dtype=np.uint8) for _ in range(64)]
cards = [(f"{i:036d}", imgs[i % 64]) for i in range(n)]
store = VectorStoreQdrant(capacity=4096)
build_bank(cards[:256], enc, store, 256)
store.drop_collection()
torch.cuda.synchronize(); t = time.perf_counter()
added = build_bank(cards, enc, store, 256)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print(f"build_bank: {added} cards (680x488 scans) in {dt:.2f} s = {added / dt:.0f} cards/s end to end incl. host packing / H2D / id bookkeeping")
x = [im for _, im in cards[:256]]
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): make_cropped(x)
torch.cuda.synchronize(); print(f"make_cropped(256 images) incl. H2D: {(time.perf_counter() - t) / 5 * 1e3:.1f} ms")
