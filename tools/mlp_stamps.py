#!/usr/bin/env python3
"""Per-tile clock stamps of the fused MLP output pass (MTGV_MLP_STAMPS=<file>, mlp_fused.hip): python tools/mlp_stamps.py <file>"""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.int64)
i = 0
while i < len(raw):
    M, C, nt = raw[i : i + 3]
    d = raw[i + 3 : i + 3 + nt * 8].reshape(nt, 8)
    i += 3 + nt * 8
    d = d[d[:, 3] > 0]
    pro, loop, epi = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2]
    span_c = d[:, 3].max() - d[:, 0].min()
    span_rt = (d[:, 7].max() - d[:, 7].min()) / 100e6  # s_memrealtime: 100 MHz
    med = lambda x: int(np.median(x))  # noqa: E731
    print(f"M={M} C={C} tiles={nt}: prologue {med(pro)}  loop {med(loop)} (wait A {med(d[:,4])}, wait B {med(d[:,5])})  epilogue {med(epi)}  "
          f"| launch span {span_c} cycles = {span_rt*1e6:.1f} us -> clock {span_c/span_rt/1e9:.2f} GHz")
