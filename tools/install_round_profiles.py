#!/usr/bin/env python3
"""Copy what tools/prof_round.sh left under gpurun_out/prof_<tag>/ (and the bench lines of a plain run) into profiles/, merge the
HBM traffic of this round into profiles/gemm_traffic.json and print the per-kernel / per-group numbers profiles/README.md
quotes:  python tools/install_round_profiles.py r04 gpurun_out/r4f/bench.json [gpurun_out/r4f/bench_f32.json]"""
import csv, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, bench = sys.argv[1], sys.argv[2]
bench32 = sys.argv[3] if len(sys.argv) > 3 else None
src, dst = os.path.join(ROOT, "gpurun_out", f"prof_{tag}"), os.path.join(ROOT, "profiles")
for m in ("no_overlap", "two_streams", "nccl1"):
    shutil.copy(os.path.join(src, f"{tag}_f16x3_kernel_stats_bench_{m}.csv"), dst)
    shutil.copy(os.path.join(src, f"{tag}_bench_{m}.json"), os.path.join(dst, f"{tag}_bench_f16x3_{m}_under_rocprof.json"))
for f in (f"{tag}_sq_counters_step.txt", f"{tag}_perf_probe.txt", f"{tag}_idle_two_streams.txt", f"{tag}_idle_no_overlap.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), dst)
line = open(os.path.join(ROOT, bench)).read().strip().splitlines()[-1]
open(os.path.join(dst, f"{tag}_bench_f16x3.json"), "w").write(line + "\n")
if bench32:
    open(os.path.join(dst, f"{tag}_bench_f32.json"), "w").write(open(os.path.join(ROOT, bench32)).read().strip().splitlines()[-1] + "\n")
old = json.load(open(os.path.join(dst, "gemm_traffic.json")))
new = json.load(open(os.path.join(src, f"{tag}_traffic_f16x3.json")))
old["f16x3"] = {"kernels": new["kernels"], "hbm_bytes_per_step": new["hbm_bytes_per_step"]}
head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT).decode().strip()
old["collected_at"] = re.sub(r"commit \w+", "commit " + head, old["collected_at"], 1)
json.dump(old, open(os.path.join(dst, "gemm_traffic.json"), "w"), indent=1)
d = json.loads(line)
r = d["roofline"]
print("bench:", d["value"], d["ms_per_step"], {k: d["config"].get(k) for k in ("unsettled_value", "sustained_value", "one_stream_value", "with_h2d_value")})
print("roofline:", r["bound"], r["frac"], r["gemm_ms_per_step"], r["mfma_view"]["issued_frac"], r["algorithmic_gbyte_per_step"], "traffic", new["hbm_bytes_per_step"] / 1e9)
for k, v in r["groups"].items():
    w = r.get("f32_mode", {}).get("groups", {}).get(k, {})
    print(f"| {k} | {v['launches_per_step']} | {v['ms_per_step']:.2f} | {v['algorithmic_tflops']:.0f} | {v['issued_mfma_frac']:.2f} | {v['hbm_frac']:.2f} | {w.get('ms_per_step', 0):.2f} / {w.get('issued_mfma_frac', 0):.2f} |")
rows = list(csv.DictReader(open(os.path.join(dst, f"{tag}_f16x3_kernel_stats_bench_no_overlap.csv"))))
steps = [int(x["Calls"]) for x in rows if "nms_kernel" in x["Name"]][0]
agg = {}
for x in rows:
    n = re.sub(r"<.*", "", re.sub(r"\(.*", "", x["Name"].replace("void mtgv::", "").replace("mtgv::", "")))
    agg[n] = agg.get(n, 0) + int(x["TotalDurationNs"])
tot = sum(v for k, v in agg.items() if not k.startswith("__amd") and "at::" not in k and "pack" not in k) / steps / 1e6
print(f"steps {steps}; kernel ms/step {tot:.2f}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:22]:
    print(f"  {k}: {v / steps / 1e6:.3f} ms/step")
for x in rows:
    if any(s in x["Name"] for s in ("dwconv7_ln", "mlp_fused", "gemm_sp_kernel<4, 1, 1, 3, 2, 2, 3, 0, 2>", "sppf_pools", "mask_quads")):
        print("  avg", x["Name"][:70], int(x["Calls"]) / steps, float(x["AverageNs"]) / 1e3)
tk = new["kernels"]
print("traffic per kernel (GB/step):", {k[:40]: (round(v["read_bytes_per_step"] / 1e9, 2), round(v["write_bytes_per_step"] / 1e9, 2)) for k, v in list(tk.items())[:8]})
