#!/usr/bin/env python3
"""pwconv2 (GRN-scaled Linear + residual) variants in one process: A mode 3 (multipliers staged through LDS) vs A mode 7
(multipliers in registers), and the tile configuration of those launches (MTGV_SP_PW2_CFG: 2 = 128x96, 1 = 128x192, 0 = 128x128).
Encoder b=256 time and bit-identity of the embeddings; per-launch table of the pwconv2 launches.
    python tools/pw2_probe.py [encoder]"""
import csv, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import torch
from mtgv import native, spec
from mtgv.encoder import Encoder

name = sys.argv[1] if len(sys.argv) > 1 else "cnvnxt2ae_tiny"
cfg = spec.encoder_config(name)
B = 256
enc = Encoder(cfg, spec.random_encoder_state(cfg, 1), max_batch=B)
x = torch.randint(0, 256, (B, *cfg.image_hw, 3), device="cuda", dtype=torch.uint8)
L = native.lib()

def timeit(fn, warm=3, it=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

def pw2_rows():
    native.check(L.mtgv_profile_gemm(1))
    enc.encode(x); torch.cuda.synchronize()
    with tempfile.NamedTemporaryFile(suffix=".csv", delete=False) as tf: p = tf.name
    native.check(L.mtgv_profile_gemm_dump(p.encode())); native.check(L.mtgv_profile_gemm(0))
    rows = [r for r in csv.DictReader(open(p)) if int(r["apro"])]
    os.unlink(p)
    agg = {}
    for r in rows:
        k = (int(r["M"]), int(r["N"]), int(r["K"]))
        agg.setdefault(k, []).append(float(r["ms"]))
    return {k: (len(v), sum(v) / len(v) * 1e3) for k, v in agg.items()}

ref = None
for rep in range(2):
    for ascr, pcfg in (("0", None), ("1", None), ("1", "2"), ("1", "1"), ("1", "0")):
        os.environ["MTGV_SP_ASCR"] = ascr
        if pcfg is None: os.environ.pop("MTGV_SP_PW2_CFG", None)
        else: os.environ["MTGV_SP_PW2_CFG"] = pcfg
        ms = timeit(lambda: enc.encode(x))
        z = enc.encode(x); torch.cuda.synchronize()
        if ref is None: ref = z.clone()
        per = pw2_rows() if rep == 0 else {}
        print(f"ascr={ascr} pw2_cfg={pcfg}: {ms:.3f} ms  {B / ms * 1e3:.0f} img/s  bit-identical {bool(torch.equal(z, ref))}  " +
              "  ".join(f"{k[0]}x{k[1]}x{k[2]}: {v[0]} x {v[1]:.1f} us" for k, v in sorted(per.items(), reverse=True)), flush=True)
