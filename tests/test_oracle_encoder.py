"""Pin the CPU oracle (oracle/encoder_ref.py) to golden vectors captured from the
reference's own modules (tools/make_golden.py).  CPU only."""

import ast
import os

import numpy as np
import pytest
import torch

from mtgv import spec
from oracle import encoder_ref as R

from conftest import GOLDEN


def _load(name):
    return np.load(os.path.join(GOLDEN, name))


def _cfg(npz):
    d = ast.literal_eval(str(npz["cfg"]))
    return spec.EncoderConfig(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items()})


def _digest(sd):
    acc = []
    for k, v in sd.items():
        acc.append(float(np.asarray(v, np.float64).sum()))
        acc.append(float(np.abs(np.asarray(v, np.float64)).sum()))
    return np.asarray(acc, np.float64)


MICRO = [
    "micro_ae_conv_linear",
    "micro_ae_conv_mlp",
    "micro_ae_conv_act_mlp",
    "micro_ae_pool_linear",
    "micro_ae_pool_mlp",
    "micro_plain",
    "micro_ae_ragged",
]
FULL = ["ae_nano_192x128", "ae_tiny_192x128", "ae_tiny_224_z784", "plain_tiny_224"]


def test_ops_layernorm_grn_block():
    g = _load("encoder_ops.npz")
    x = torch.from_numpy(g["ln_x_cl"])
    w, b = torch.from_numpy(g["ln_w"]), torch.from_numpy(g["ln_b"])
    np.testing.assert_allclose(R.layernorm_channels_last(x, w, b).numpy(), g["ln_cl_out"], atol=1e-6)
    np.testing.assert_allclose(R.layernorm_channels_first(x.permute(0, 3, 1, 2), w, b).numpy(), g["ln_cf_out"], atol=1e-6)
    np.testing.assert_allclose(
        R.grn(x, torch.from_numpy(g["grn_gamma"]), torch.from_numpy(g["grn_beta"])).numpy(), g["grn_out"], atol=1e-6
    )
    for act in ("gelu", "mish"):
        p = {k.split("_p.", 1)[1]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"block_{act}_p.")}
        p = {f"b.{k}": v for k, v in p.items()}
        out = R.block(torch.from_numpy(g[f"block_{act}_x"]), p, "b", act)
        np.testing.assert_allclose(out.numpy(), g[f"block_{act}_out"], atol=2e-6)


@pytest.mark.parametrize("name", MICRO)
def test_micro_matches_reference(name):
    g = _load(f"encoder_{name}.npz")
    cfg = _cfg(g)
    sd = spec.random_encoder_state(cfg, seed=1)
    np.testing.assert_allclose(_digest(sd), g["params_digest"], rtol=0, atol=0)
    z, stages = R.encoder_forward(sd, cfg, g["x"], return_stages=True)
    for i, s in enumerate(stages):
        np.testing.assert_allclose(s.numpy(), g[f"stage{i}"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(z.numpy(), g["z_fp32"], atol=1e-5)
    z64 = R.encoder_forward(sd, cfg, g["x"], dtype=torch.float64)
    np.testing.assert_allclose(z64.numpy(), g["z_fp64"], atol=1e-10)


@pytest.mark.parametrize("name", FULL)
def test_full_matches_reference(name):
    g = _load(f"encoder_{name}.npz")
    cfg = _cfg(g)
    sd = spec.random_encoder_state(cfg, seed=1)
    np.testing.assert_allclose(_digest(sd), g["params_digest"], rtol=0, atol=0)
    x = np.random.default_rng(0).random((4, 3, *cfg.image_hw), dtype=np.float32)
    np.testing.assert_allclose(
        np.asarray([x.sum(dtype=np.float64), np.abs(x - 0.5).sum(dtype=np.float64)]), g["x_digest"], rtol=0, atol=0
    )
    torch.set_num_threads(8)
    z, stages = R.encoder_forward(sd, cfg, x, return_stages=True)
    assert z.shape == (4, cfg.z_size)
    np.testing.assert_allclose(z.numpy(), g["z_fp32"], atol=2e-5)
    np.testing.assert_allclose(z.numpy(), g["z_fp64"], atol=5e-5)
    for i, s in enumerate(stages):
        s = s.numpy().astype(np.float64)
        got = np.asarray([s.sum(), np.abs(s).sum(), (s * s).sum()])
        np.testing.assert_allclose(got, g[f"stage{i}_sum"], rtol=1e-5)
        np.testing.assert_allclose(
            s.reshape(-1)[:: max(1, s.size // 257)][:257], g[f"stage{i}_sample"], atol=5e-5, rtol=1e-5
        )


def test_reference_failure_modes():
    # 224x224 cannot give z=768 in the AE encoder: convnextv2ae.py:126
    with pytest.raises(AssertionError):
        spec.encoder_config("cnvnxt2ae_tiny", (224, 224))
    with pytest.raises(KeyError):
        spec.EncoderConfig(head_type="nope")  # convnextv2ae.py:249-250
    with pytest.raises(KeyError):
        spec.encoder_config("not_a_model")  # encoder_train.py:269


def test_predict_hwc_contract():
    cfg = spec.EncoderConfig("ae", (96, 64), 3, 48, (1, 1, 2, 1), (8, 16, 32, 64), "conv+linear", True)
    sd = spec.random_encoder_state(cfg, seed=1)
    im = np.random.default_rng(5).integers(0, 256, (96, 64, 3), dtype=np.uint8)
    z = R.predict_hwc(sd, cfg, im)
    assert z.shape == (48,) and z.dtype == np.float32
    z2 = R.predict_hwc(sd, cfg, im.astype(np.float32) / 255.0)
    np.testing.assert_allclose(z, z2, atol=1e-6)


def test_key_tables_of_every_variant():
    """mtgv.spec's state_dict key/shape tables == the reference modules' for all 12 AE sizes x 5 heads and the 8
    plain sizes (digests captured by tools/make_golden_keys.py from the reference's own factories)."""
    import hashlib
    import json

    pinned = json.load(open(os.path.join(GOLDEN, "encoder_key_tables.json")))
    assert len(pinned) == 12 * 5 + 8
    for name, rec in pinned.items():
        model, ht = name.split("|")
        cfg = spec.encoder_config(model, (224, 224) if ht == "plain" else (192, 128), "conv+linear" if ht == "plain" else ht)
        items = [(k, tuple(v)) for k, v in spec.encoder_param_shapes(cfg).items()]
        h = hashlib.sha256()
        for k, s in items:
            h.update(f"{k}:{tuple(s)};".encode())
        assert len(items) == rec["n"] and h.hexdigest() == rec["sha256"], name
