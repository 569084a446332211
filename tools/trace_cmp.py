#!/usr/bin/env python3
"""Compare two per-launch GEMM traces (tools/gemm_trace.py csv): python tools/trace_cmp.py new.csv [old.csv]"""
import csv, sys

new = list(csv.DictReader(open(sys.argv[1])))
old = {int(r["idx"]): r for r in csv.DictReader(open(sys.argv[2]))} if len(sys.argv) > 2 else {}
groups = {}
for r in new:
    i = int(r["idx"])
    ms = float(r["ms"])
    o = float(old[i]["ms"]) if i in old and old[i]["M"] == r["M"] and old[i]["N"] == r["N"] else float("nan")
    conv = int(r["KH"]) > 1
    grp = "det" if i < 74 else ("enc" if i < 116 else "bank")
    if "-v" in sys.argv:
        print(i, r["M"], r["N"], r["K"], "KH", r["KH"], "act", r["act"], "apro", r["apro"], "ms %.4f (old %.4f)" % (ms, o), "tf", r["tflops"])
    g = groups.setdefault(grp, [0.0, 0.0])
    g[0] += ms
    g[1] += o
for k, (a, b) in groups.items():
    print("%-5s %.3f ms (old %.3f)" % (k, a, b))
print("total %.3f (old %.3f)" % (sum(g[0] for g in groups.values()), sum(g[1] for g in groups.values())))
