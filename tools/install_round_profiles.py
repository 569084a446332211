#!/usr/bin/env python3
"""Copy what tools/prof_round.sh left under gpurun_out/prof_<tag>/ (and a bench line) into profiles/, merge the HBM traffic
of this round into profiles/gemm_traffic.json and regenerate the per-group table and the per-kernel sentence of
profiles/README.md from those files:  python tools/install_round_profiles.py r03 gpurun_out/r3/bench15.json"""
import csv, json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, bench = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", f"prof_{tag}"), os.path.join(ROOT, "profiles")
for m in ("no_overlap", "two_streams"):
    shutil.copy(os.path.join(src, f"{tag}_f16x3_kernel_stats_bench_{m}.csv"), dst)
    shutil.copy(os.path.join(src, f"{tag}_bench_{m}.json"), os.path.join(dst, f"{tag}_bench_f16x3_{m}_under_rocprof.json"))
line = open(os.path.join(ROOT, bench)).read().strip().splitlines()[-1]
open(os.path.join(dst, f"{tag}_bench_f16x3.json"), "w").write(line + "\n")
old = json.load(open(os.path.join(dst, "gemm_traffic.json")))
new = json.load(open(os.path.join(src, f"{tag}_traffic_f16x3.json")))
old["f16x3"] = {"kernels": new["kernels"], "hbm_bytes_per_step": new["hbm_bytes_per_step"]}
head = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT).decode().strip()
old["collected_at"] = re.sub(r"commit \w+", "commit " + head, old["collected_at"], 1)
json.dump(old, open(os.path.join(dst, "gemm_traffic.json"), "w"), indent=1)
d = json.loads(line)
r = d["roofline"]; g = r["groups"]; f = r["f32_mode"]["groups"]
r2 = {"pwconv1": 2.36, "pwconv2": 2.52, "det3x3": 1.80, "det1x1": 0.78, "bank": 0.17, "enc_other": 0.32}
label = {"pwconv1": "pwconv1 (+bias+Mish+GRN sums; stage 0 = fused statistics pass)", "pwconv2": "pwconv2 (GRN multipliers + residual; stage 0 = fused output pass)",
         "det3x3": "det3x3", "det1x1": "det1x1", "bank": "bank (two-pass: fp16 first pass; re-rank kernel not in the group)", "enc_other": "enc_other (stem, downsample convs, head)"}
note = {"pwconv2": " (the fused pass issues 6 MFMA FLOPs per credited FLOP)", "bank": " (1 MFMA per product)"}
lines = []
for k in ("pwconv1", "pwconv2", "det3x3", "det1x1", "bank", "enc_other"):
    v, w = g[k], f[k]
    lines.append(f"| {label[k]} | {v['launches_per_step']} | {v['ms_per_step']:.2f} ({r2[k]:.2f}) | {v['algorithmic_tflops']:.0f} | {v['issued_mfma_frac']:.2f}{note.get(k, '')} | {v['hbm_frac']:.2f} | {w['ms_per_step']:.2f} / {w['issued_mfma_frac']:.2f} |")
p = os.path.join(dst, "README.md")
s = open(p).read()
a = s.index("| pwconv1 (+bias+Mish+GRN sums"); b = s.index("\n\n", a)
s = s[:a] + "\n".join(lines) + s[b:]
rows = list(csv.DictReader(open(os.path.join(dst, f"{tag}_f16x3_kernel_stats_bench_no_overlap.csv"))))
steps = [int(x["Calls"]) for x in rows if "mlp_fused_kernel<6, 2, 2>" in x["Name"]][0] / 3
agg, avg = {}, {}
for x in rows:
    n = re.sub(r"<.*", "", re.sub(r"\(.*", "", x["Name"].replace("void mtgv::", "").replace("mtgv::", "")))
    agg[n] = agg.get(n, 0) + int(x["TotalDurationNs"])
    for key in ("mlp_fused_kernel<6, 2, 2>", "mlp_fused_kernel<6, 2, 1>", "dwconv7_ln_rows_kernel<4, 3, true, true>", "dwconv7_ln_rows_kernel<8, 3, true, false>", "dwconv7_ln_rows_kernel<4, 3, true, false>"):
        if key in x["Name"]: avg[key] = float(x["AverageNs"]) / 1e3
A = lambda k: agg.get(k, 0) / steps / 1e6
tot = sum(agg.values()) / steps / 1e6
txt = (f"per-kernel time, every kernel alone on the GPU ({steps:.0f} steps: 160 settle + 2 warm-up + 10 timed): {tot:.2f} ms of kernel time per step - `gemm_sp_kernel<...>` {A('gemm_sp_kernel'):.2f}, "
       f"`dwconv7_ln_rows_kernel` {A('dwconv7_ln_rows_kernel'):.2f} (4 x 3 strips with the tap table in LDS {avg['dwconv7_ln_rows_kernel<4, 3, true, true>']:.1f} us x 6 [stage 0 and 1], 8 x 3 strips {avg['dwconv7_ln_rows_kernel<8, 3, true, false>']:.1f} us x 9, 4 x 3 {avg['dwconv7_ln_rows_kernel<4, 3, true, false>']:.1f} x 3; round 2: 1.09 ms), "
       f"`mlp_fused_kernel` {A('mlp_fused_kernel'):.2f} (output pass {avg['mlp_fused_kernel<6, 2, 2>']:.1f} us x 3, statistics pass {avg['mlp_fused_kernel<6, 2, 1>']:.1f} us x 3), `nms_kernel` {A('nms_kernel'):.2f}, `ln_rows` {A('ln_rows_kernel'):.2f}, "
       f"`mask_quads_kernel<true>` {A('mask_quads_kernel'):.2f}, `grn_finalize_kernel` {A('grn_finalize_kernel'):.2f}, `gemm_f32_kernel` (stem + LayerNorm, mask) {A('gemm_f32_kernel'):.2f}, `conv0_u8` {A('conv0_u8_kernel'):.2f} (round 2: 0.14), decode {A('decode_kernel'):.2f}, re-rank {A('rerank_kernel'):.2f}, warp {A('warp_kernel'):.2f} (round 2: 0.08) ")
a = s.index("per-kernel time, every kernel alone on the GPU ("); b = s.index("|", a)
s = s[:a] + txt + s[b:]
a = s.index("Taken at the round's last code commit:"); b = s.index("|", a)
s = s[:a] + (f"Taken at the round's last code commit: {d['value'] / 1e3:.1f}k cards/s ({d['ms_per_step']:.2f} ms), one stream {d['config']['one_stream_value'] / 1e3:.1f}k ({d['config']['one_stream_ms_per_step']:.2f} ms); "
             "other boxes gave 29.7-30.9k for the same command earlier in the round (box-to-box spread of the same build 2-3 %) and one 27.6k (its two-stream region gained nothing) ") + s[b:]
s = re.sub(r"frac 0\.\d+ \(15\.1 GB of compulsory bytes in \d\.\d+ ms of\nGEMM-class kernel time; `mfma_view\.issued_frac` 0\.\d+\)",
           f"frac {r['frac']:.3f} (15.1 GB of compulsory bytes in {r['gemm_ms_per_step']:.2f} ms of\nGEMM-class kernel time; `mfma_view.issued_frac` {r['mfma_view']['issued_frac']:.2f})", s)
open(p, "w").write(s)
print(f"value {d['value']} ms {d['ms_per_step']} one-stream {d['config']['one_stream_value']} frac {r['frac']} gemm_ms {r['gemm_ms_per_step']} issued {r['mfma_view']['issued_frac']} hbm {new['hbm_bytes_per_step']} kernel ms/step {tot:.2f}")
