#!/usr/bin/env python3
"""Smallest reproducer of a failing launch: one query against a 300-row bank (AMD_LOG_LEVEL=2 shows the runtime's reason)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import numpy as np, torch
from mtgv.matcher import Matcher
m = Matcher(768, capacity=512)
m.add(np.random.default_rng(5).standard_normal((300, 768)).astype(np.float32))
torch.cuda.synchronize()
print("added", flush=True)
ids, sc = m.match(np.ones(768, np.float32), 3)
torch.cuda.synchronize()
print(ids, sc)
