#!/usr/bin/env python3
"""Aggregate rocprofv3 FETCH_SIZE / WRITE_SIZE counter CSVs (tools/pmc_traffic.sh) into bytes per kernel and step.

usage: pmc_traffic.py <dir with FETCH_SIZE/ and WRITE_SIZE/> <steps in the run> <precision tag>
Counter values are KiB; FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads - checked on
ln_rows_kernel, whose 151 MB input shows as 75.5 MB raw)."""
import csv, glob, json, os, re, sys
from collections import defaultdict

root, steps, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]


def short(name):
    name = re.sub(r"^void ", "", name)
    if "gemm_f32_kernel" in name:
        return "gemm_f32_kernel"
    if "gemm_sp_kernel" in name:
        return "gemm_sp_kernel"
    return re.sub(r"\(.*$", "", name)[:48]


tot = {"FETCH_SIZE": defaultdict(float), "WRITE_SIZE": defaultdict(float)}
for c in tot:
    for f in glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                tot[c][short(r["Kernel_Name"])] += float(r["Counter_Value"])
kern = {}
for k in sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"]), key=lambda k: -(2 * tot["FETCH_SIZE"][k] + tot["WRITE_SIZE"][k])):
    kern[k] = {"read_bytes_per_step": int(2 * tot["FETCH_SIZE"][k] * 1024 / steps), "write_bytes_per_step": int(tot["WRITE_SIZE"][k] * 1024 / steps)}
# the launches bench.py's roofline covers: both GEMM kernels and the fused MLP passes (which do two GEMM layers' work)
gemm = sum(v["read_bytes_per_step"] + v["write_bytes_per_step"] for k, v in kern.items()
           if k in ("gemm_f32_kernel", "gemm_sp_kernel") or "mlp_fused_kernel" in k)
print(json.dumps({"precision": tag, "kernels": kern, "hbm_bytes_per_step": gemm}, indent=1))
