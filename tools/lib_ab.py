#!/usr/bin/env python3
"""A/B of alternative library builds (mtg-vision_amd/build.py with MTGV_BUILD_TAG): per-launch GEMM times of one pipeline
step per library, minimum over REPS profiled steps, summed by launch kind.
    python tools/lib_ab.py base prio1 stg ...      ("base" = the product library)"""
import csv, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "lib_ab")
os.makedirs(OUT, exist_ok=True)
REPS = int(os.environ.get("REPS", "5"))
tags = sys.argv[1:] or ["base"]


def kind(r):
    if r["grn"] == "1": return "pwconv1"
    if r["apro"] == "1": return "pwconv2"
    if int(r["N"]) >= 50000: return "bank"
    if int(r["M"]) % 6400 == 0: return "det3x3" if int(r["KH"]) > 1 else "det1x1"   # detector maps: 32 x (80^2, 40^2, 20^2, 160^2)
    return "enc_other"


res = {}
for tag in tags:
    env = dict(os.environ)
    env.setdefault("MTGV_DET_FORK", "0")  # every launch alone on the GPU (the detector's branches in sequence)
    lib, *sets = tag.split("+")  # "base+MTGV_SP_ADIRECT=1": the product library with an environment switch
    for kv in sets:
        k, v = kv.split("=", 1)
        env[k] = v
    if lib != "base":
        env["MTGV_LIB_PATH"] = os.path.join(ROOT, "mtg-vision_amd", "mtgv", f"libmtgv_{lib}.so")
    best = {}
    for rep in range(REPS):
        out = os.path.join(OUT, f"{tag.replace('+', '_').replace('=', '')}_{rep}.csv")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_trace.py"), out], env=env, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for r in csv.DictReader(open(out)):
            i = int(r["idx"])
            if i not in best or float(r["ms"]) < float(best[i]["ms"]): best[i] = r
    g = {}
    for r in best.values(): g[kind(r)] = g.get(kind(r), 0.0) + float(r["ms"])
    res[tag] = (g, best)
    print(tag, " ".join(f"{k} {v:.3f}" for k, v in sorted(g.items())), "total %.3f" % sum(g.values()), flush=True)
if len(tags) > 1:
    b = res[tags[0]][1]
    for tag in tags[1:]:
        o = res[tag][1]
        d = sorted(((float(o[i]["ms"]) - float(b[i]["ms"])) * 1e3, i) for i in b if i in o)
        print(f"{tag} vs {tags[0]}: largest per-launch changes (us):",
              [(round(x, 1), f"{b[i]['M']}x{b[i]['N']}x{b[i]['K']}") for x, i in d[:5] + d[-5:]])
