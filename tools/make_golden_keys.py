#!/usr/bin/env python3
"""Pin the checkpoint surface: state_dict key/shape tables of every reference encoder variant.

Builds each model of `encoder_train._MODELS` (names reproduced from mtgvision/encoder_train.py:52-67, factories
imported from /root/reference/mtgvision/models) on the meta device for every head type, plus the plain
ConvNeXtV2 factories (on the CPU), and stores a digest of (key, shape) pairs in tests/golden/encoder_key_tables.json.
Only names, shapes and hashes are stored."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mtg-vision_amd")); sys.path.insert(0, "/root/reference"); sys.dont_write_bytecode = True
import torch
from mtgvision.models import convnextv2 as ref_plain, convnextv2ae as ref_ae
from mtgv import spec

AE = {"atto": ref_ae.convnextv2_atto, "femto": ref_ae.convnextv2_femto, "pico": ref_ae.convnextv2ae_pico, "nano": ref_ae.convnextv2ae_nano,
      "tiny": ref_ae.convnextv2ae_tiny, "tiny_9_128": ref_ae.convnextv2ae_tiny_9_128, "tiny_12_128": ref_ae.convnextv2ae_tiny_12_128,
      "base_9": ref_ae.convnextv2ae_base_9, "base_12": ref_ae.convnextv2ae_base_12, "base": ref_ae.convnextv2ae_base,
      "large": ref_ae.convnextv2ae_large, "huge": ref_ae.convnextv2ae_huge}
PLAIN = {"atto": ref_plain.convnextv2_atto, "femto": ref_plain.convnextv2_femto, "pico": ref_plain.convnextv2_pico, "nano": ref_plain.convnextv2_nano,
         "tiny": ref_plain.convnextv2_tiny, "base": ref_plain.convnextv2_base, "large": ref_plain.convnextv2_large, "huge": ref_plain.convnextv2_huge}

def digest(items):
    h = hashlib.sha256()
    for k, s in items: h.update(f"{k}:{tuple(s)};".encode())
    return h.hexdigest()

out = {}
with torch.device("meta"):
    for size, fn in AE.items():
        for ht in spec.HEAD_TYPES[:5]:
            m = fn(image_wh=(128, 192), z_size=768, head_type=ht, encoder_enabled=True, decoder_enabled=False)
            ref = [(k, tuple(v.shape)) for k, v in m.encoder.state_dict().items()]
            cfg = spec.encoder_config(f"cnvnxt2ae_{size}", (192, 128), ht)
            mine = [(k, tuple(v)) for k, v in spec.encoder_param_shapes(cfg).items()]
            assert ref == mine, (size, ht)
            out[f"cnvnxt2ae_{size}|{ht}"] = {"n": len(ref), "sha256": digest(ref)}
for size, fn in PLAIN.items():  # (dp_rates uses .item(): not constructible on the meta device)
    if True:
        m = fn(num_classes=768)
        ref = [(k, tuple(v.shape)) for k, v in m.state_dict().items()]
        cfg = spec.encoder_config(f"convnextv2_{size}", (224, 224))
        mine = [(k, tuple(v)) for k, v in spec.encoder_param_shapes(cfg).items()]
        assert ref == mine, size
        out[f"convnextv2_{size}|plain"] = {"n": len(ref), "sha256": digest(ref)}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "encoder_key_tables.json"), "w"), indent=1, sort_keys=True)
print(len(out), "tables pinned")
