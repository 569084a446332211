#!/usr/bin/env python3
"""Sum rocprofv3 counter_collection csvs per kernel name: python tools/pmc_sq.py gpurun_out/pmc_sq"""
import csv, glob, os, re, sys
from collections import defaultdict

root = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void mtgv::", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], f)
        if r["Counter_Name"] == "SQ_WAVE_CYCLES" and key not in seen:
            seen.add(key)
            cnt[k] += 1
names = sorted({c for v in tot.values() for c in v})
rows = sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))
for k, v in rows[:24]:
    print(k[:110], "dispatches", cnt[k])
    print("   " + "  ".join(f"{c}={v[c]:.4g}" for c in names if c in v))
    wc = v.get("SQ_WAVE_CYCLES", 0)
    if wc:
        d = lambda c: v.get(c, 0) / wc
        print("   per wave-cycle: mfma_busy %.3f  wait_any %.3f  wait_inst_any %.3f  wait_inst_lds %.3f  active_lds %.3f  active_valu %.3f  vmem_cyc %.3f  active_any %.3f  bank_conflict/active_lds %.3f"
              % (d("SQ_VALU_MFMA_BUSY_CYCLES"), d("SQ_WAIT_ANY"), d("SQ_WAIT_INST_ANY"), d("SQ_WAIT_INST_LDS"), d("SQ_ACTIVE_INST_LDS"), d("SQ_ACTIVE_INST_VALU"),
                 d("SQ_INST_CYCLES_VMEM"), d("SQ_ACTIVE_INST_ANY"), v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_ACTIVE_INST_LDS", 1), 1)))
