#!/usr/bin/env python3
"""Summarise the per-tile clock stamps of the LDS-DMA GEMM (MTGV_SP_STAMPS=1 python tools/gemm_trace.py out.csv writes
out.csv.stamps): per launch, mean cycles of prologue (entry -> first stage in LDS), main loop and epilogue, the
launch's span, and how busy a CU's slots were:  python tools/sp_stamps.py out.csv.stamps"""
import struct, sys
import numpy as np

data = open(sys.argv[1], "rb").read()
off = 0
idx = 0
print("idx  M N K cfg amode act tiles | span_us | pro main epi (cycles/tile, mean) | tiles/CU max | sum(tile)/span/CUslots")
while off < len(data):
    M, N, K, cfg, amode, act, tiles, _ = struct.unpack_from("8i", data, off)
    off += 32
    a = np.frombuffer(data, dtype=np.int64, count=tiles * 8, offset=off).reshape(tiles, 8)
    off += tiles * 64
    a = a[a[:, 3] != 0]
    if len(a) == 0:
        continue
    t0, t1, t2, t3 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    # s_memtime is per XCC; compare only within one XCC
    xcc = a[:, 5]
    hw = a[:, 4]
    cu = ((hw >> 8) & 15) | (((hw >> 13) & 7) << 4) | (xcc << 8)
    spans = []
    for x in np.unique(xcc):
        m = xcc == x
        spans.append(t3[m].max() - t0[m].min())
    span = float(np.mean(spans))
    ncu = len(np.unique(cu))
    percu = np.bincount(np.unique(cu, return_inverse=True)[1])
    busy = float((t3 - t0).sum()) / (span * ncu)
    rt = (a[:, 7] - a[:, 6]).astype(np.float64)
    ok = rt > 0
    clk = float(np.median((t3 - t0)[ok] / rt[ok]) * 100.0) if ok.any() else 0.0  # MHz: s_memtime ticks per 10 ns of s_memrealtime
    print(f"{idx:3d} {M:7d} {N:5d} {K:5d} c{cfg} a{amode} act{act} {len(a):5d} | {span/100:.1f} ticks/100 | {np.mean(t1-t0):8.0f} {np.mean(t2-t1):8.0f} {np.mean(t3-t2):8.0f} | cus {ncu} max {percu.max()} | {busy:.2f} | clk {clk:.0f} MHz tile {np.median(rt[ok]) / 100 if ok.any() else 0:.1f} us")
    idx += 1
