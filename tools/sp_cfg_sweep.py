#!/usr/bin/env python3
"""Per-launch tile-configuration sweep of the LDS-DMA GEMM over one whole pipeline step: every launch timed with the planner's
choice and with each configuration forced (MTGV_SP_CFG, read per plan), detector branches in sequence.  Prints the launches
where a forced configuration beats the planner by more than 3 % and the total it would save.
    python tools/sp_cfg_sweep.py [out.csv]"""
import csv, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
os.environ["MTGV_DET_FORK"] = "0"
import torch
from mtgv import native, spec
from mtgv.detector import Detector
from mtgv.encoder import Encoder
from mtgv.matcher import Matcher
from mtgv.pipeline import Pipeline

F, K = 32, 8
det_cfg = spec.DetectorConfig(); enc_cfg = spec.encoder_config("cnvnxt2ae_tiny")
m = Matcher(768, capacity=100_000)
m.add(torch.randn((100_000, 768), device="cuda"))
pipe = Pipeline(Detector(det_cfg, spec.random_detector_state(det_cfg, 3), max_batch=F),
                Encoder(enc_cfg, spec.random_encoder_state(enc_cfg, 1), max_batch=F * K), m, K, 1, quad_source="mask")
frames = torch.randint(0, 256, (F, 640, 640, 3), device="cuda", dtype=torch.uint8)
L = native.lib()

def trace(reps=3):
    best = None
    for _ in range(reps):
        native.check(L.mtgv_profile_gemm(1))
        pipe.run(frames); torch.cuda.synchronize()
        with tempfile.NamedTemporaryFile(suffix=".csv", delete=False) as tf: p = tf.name
        native.check(L.mtgv_profile_gemm_dump(p.encode())); native.check(L.mtgv_profile_gemm(0))
        rows = list(csv.DictReader(open(p))); os.unlink(p)
        if best is None: best = rows
        else:
            for a, b in zip(best, rows):
                if float(b["ms"]) < float(a["ms"]): a["ms"] = b["ms"]
    return best

pipe.run(frames); torch.cuda.synchronize()
res = {}
for cfg in (None, 0, 1, 2, 3, 4):
    if cfg is None: os.environ.pop("MTGV_SP_CFG", None)
    else: os.environ["MTGV_SP_CFG"] = str(cfg)
    try:
        pipe.run(frames); torch.cuda.synchronize()
        res[cfg] = trace()
    except Exception as e:  # a forced configuration some launch cannot take
        print(f"cfg {cfg}: {str(e)[:120]}")
os.environ.pop("MTGV_SP_CFG", None)
base = res[None]
tot, save = 0.0, 0.0
for i, r in enumerate(base):
    t0 = float(r["ms"]); tot += t0
    alts = {c: float(res[c][i]["ms"]) for c in res if c is not None and len(res[c]) == len(base)}
    if not alts: continue
    c, t = min(alts.items(), key=lambda kv: kv[1])
    if t < 0.97 * t0:
        save += t0 - t
        print(f"launch {i}: M={r['M']} N={r['N']} K={r['K']} k{r['KH']} s{r['stride']} grn={r['grn']} apro={r['apro']}: planner {t0 * 1e3:.1f} us, cfg {c} {t * 1e3:.1f} us  " +
              " ".join(f"{k}:{v * 1e3:.1f}" for k, v in sorted(alts.items())))
print(f"GEMM time per step {tot:.3f} ms; per-launch best of the forced configurations would save {save:.3f} ms")
