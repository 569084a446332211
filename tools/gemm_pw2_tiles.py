#!/usr/bin/env python3
"""pw2 (scale prologue + residual) and pw1 (mish + grn) across tiles (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_ablate import run
B = 256
tiles = [(1, 1, 16), (1, 1, 32), (1, 2, 16), (1, 3, 16), (1, 4, 16), (1, 5, 16)]
for (c, hw) in ((96, 1536), (192, 384), (384, 96), (768, 24)):
    m = B * hw
    print(f"C={c} pw2 scale+res :", " ".join(f"{t[1]}.{t[2]}:{run(m, c, 4*c, 0, hw, 1, 1, 0, 0, t):.0f}" for t in tiles), "| auto %.0f" % run(m, c, 4*c, 0, hw, 1, 1, 0, 0), flush=True)
    print(f"C={c} pw1 mish+grn  :", " ".join(f"{t[1]}.{t[2]}:{run(m, 4*c, c, 2, hw, 0, 0, 0, 1, t):.0f}" for t in tiles), "| auto %.0f" % run(m, 4*c, c, 2, hw, 0, 0, 0, 1), flush=True)
