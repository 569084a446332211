#!/usr/bin/env python3
"""GPU idle time and kernel concurrency from a rocprofv3 --kernel-trace CSV, over the last `frac` of the trace
(the timed region of bench.py sits at the end of the process).
    python tools/trace_idle.py <kernel_trace.csv> [frac=0.05]
Prints: the union of the kernel intervals (busy) against the window, the largest gaps and which kernel ended them, the
time spent with 1 / 2 / 3+ kernels in flight, and per kernel name the time during which it ran alone."""
import csv, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t_end = max(e for _, e, _ in iv)
t_beg = t_end - int((t_end - iv[0][0]) * frac)
iv = [(max(s, t_beg), e, n) for s, e, n in iv if e > t_beg]
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e, n in iv:
    if cur_e is None: cur_s, cur_e = s, e
    elif s <= cur_e: cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
busy += cur_e - cur_s
wall = t_end - t_beg
print(f"window {wall / 1e6:.2f} ms, kernels {len(iv)}, busy {busy / 1e6:.2f} ms = {busy / wall:.3f}, idle {1 - busy / wall:.3f}; sum of kernel durations {sum(e - s for s, e, _ in iv) / 1e6:.2f} ms")
gaps.sort(reverse=True)
print("largest gaps (us, kernel that ended the gap):", [(round(g / 1e3, 1), n[:40]) for g, n in gaps[:8]])
big = sum(g for g, _ in gaps if g > 2000)
print(f"gaps > 2 us: {sum(1 for g, _ in gaps if g > 2000)} totalling {big / 1e6:.3f} ms; all gaps {sum(g for g, _ in gaps) / 1e6:.3f} ms")

# concurrency sweep
ev = []
for i, (s, e, n) in enumerate(iv):
    ev.append((s, 1, i))
    ev.append((e, -1, i))
ev.sort(key=lambda x: (x[0], x[1]))
live, last, conc, alone = set(), t_beg, defaultdict(int), defaultdict(int)
for t, d, i in ev:
    if t > last:
        conc[min(len(live), 3)] += t - last
        if len(live) == 1:
            alone[iv[next(iter(live))][2]] += t - last
        last = t
    if d > 0: live.add(i)
    else: live.discard(i)
print("time with k kernels in flight: " + ", ".join(f"{k}{'+' if k == 3 else ''}: {v / 1e6:.2f} ms" for k, v in sorted(conc.items())))
tot = defaultdict(int)
for s, e, n in iv: tot[n] += e - s
print("kernel (time alone on the GPU / its total), ms, by total:")
for n, v in sorted(tot.items(), key=lambda x: -x[1])[:14]:
    print(f"  {alone[n] / 1e6:7.2f} / {v / 1e6:7.2f}  {n[:100]}")
