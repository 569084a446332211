"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement, in plain PyTorch fp32/fp64 functional ops, of the ConvNeXt-V2
encoder forward that mtg-vision serves embeddings with.  Each function cites
the reference lines it follows (paths relative to /root/reference).

Pinned: `tests/test_oracle_encoder.py` checks every function here against
golden vectors captured from the reference's own modules by
`tools/make_golden.py` (fixtures in `tests/golden/encoder_*.npz`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this module.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F


def _t(a, dtype):
    return a.to(dtype) if isinstance(a, torch.Tensor) else torch.as_tensor(a, dtype=dtype)


def layernorm_channels_last(x, w, b, eps=1e-6):
    """mtgvision/models/convnextv2.py:150-154 - F.layer_norm over the last dim."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def layernorm_channels_first(x, w, b, eps=1e-6):
    """mtgvision/models/convnextv2.py:155-160 - per-pixel LN over dim 1, biased variance."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def grn(x, gamma, beta):
    """mtgvision/models/convnextv2.py:171-174 - x is (N,H,W,C); gamma/beta (1,1,1,C)."""
    gx = torch.norm(x, p=2, dim=(1, 2), keepdim=True)
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    return gamma * (x * nx) + beta + x


def activation(x, act: str):
    """GELU(erf) - convnextv2.py:192-193 (nn.GELU default); Mish - convnextv2ae.py:17-18."""
    if act == "gelu":
        return F.gelu(x)
    if act == "mish":
        return F.mish(x)
    raise KeyError(act)


def block(x, p, prefix: str, act: str):
    """mtgvision/models/convnextv2.py:212-224 (Block.forward); x is NCHW.

    drop_path is nn.Identity at inference (convnextv2.py:208-210).
    """
    c = x.shape[1]
    inp = x
    x = F.conv2d(x, p[f"{prefix}.dwconv.weight"], p[f"{prefix}.dwconv.bias"], padding=3, groups=c)
    x = x.permute(0, 2, 3, 1)
    x = layernorm_channels_last(x, p[f"{prefix}.norm.weight"], p[f"{prefix}.norm.bias"])
    x = F.linear(x, p[f"{prefix}.pwconv1.weight"], p[f"{prefix}.pwconv1.bias"])
    x = activation(x, act)
    x = grn(x, p[f"{prefix}.grn.gamma"], p[f"{prefix}.grn.beta"])
    x = F.linear(x, p[f"{prefix}.pwconv2.weight"], p[f"{prefix}.pwconv2.bias"])
    x = x.permute(0, 3, 1, 2)
    return inp + x


def stem(x, p, conv_key: str, norm_key: str):
    """Conv2d(k4,s4) + LayerNorm(channels_first): convnextv2.py:253-256, convnextv2ae.py:193-196."""
    x = F.conv2d(x, p[f"{conv_key}.weight"], p[f"{conv_key}.bias"], stride=4)
    return layernorm_channels_first(x, p[f"{norm_key}.weight"], p[f"{norm_key}.bias"])


def downsample(x, p, norm_key: str, conv_key: str):
    """LayerNorm(channels_first) + Conv2d(k2,s2): convnextv2.py:258-263, convnextv2ae.py:199-214."""
    x = layernorm_channels_first(x, p[f"{norm_key}.weight"], p[f"{norm_key}.bias"])
    return F.conv2d(x, p[f"{conv_key}.weight"], p[f"{conv_key}.bias"], stride=2)


def mlp(x, p, prefix: str):
    """MLP(in, hidden, out, act=Mish, act_out=False): convnextv2ae.py:59-72."""
    x = F.linear(x, p[f"{prefix}.layers.0.weight"], p[f"{prefix}.layers.0.bias"])
    x = F.mish(x)
    return F.linear(x, p[f"{prefix}.layers.2.weight"], p[f"{prefix}.layers.2.bias"])


def ae_head(x, p, cfg):
    """pool + head of ConvNeXtV2Encoder: convnextv2ae.py:219-250, forward :263-265."""
    ht = cfg.head_type
    z = cfg.z_size
    if ht.startswith("conv"):
        x = F.conv2d(x, p["pool.0.weight"], p["pool.0.bias"])
        if "+act" in ht:
            x = F.mish(x)
        x = layernorm_channels_first(x, p["pool.2.weight"], p["pool.2.bias"])
        x = x.reshape(-1, z)  # NCHW-flat: (c, h, w) order
    else:
        x = x.mean([-2, -1])[:, :, None, None]  # convnextv2ae.py:38-41
        x = layernorm_channels_first(x, p["pool.1.weight"], p["pool.1.bias"])
        x = x.reshape(-1, cfg.dims[3])
    if ht.endswith("+mlp"):
        x = mlp(x, p, "head")
    else:
        x = F.linear(x, p["head.weight"], p["head.bias"])
    return x.reshape(x.size(0), z)


def encoder_forward(params, cfg, x, dtype=torch.float32, return_stages: bool = False):
    """Full encoder forward.

    kind "ae":    ConvNeXtV2Encoder.forward, convnextv2ae.py:256-266
    kind "plain": ConvNeXtV2.forward, convnextv2.py:292-303

    params: dict key -> array/tensor with the reference's state_dict key names.
    x: (N,3,H,W) float in [0,1].  Returns (N, z_size) [and the 4 stage outputs].
    """
    p = {k: _t(v, dtype) for k, v in params.items()}
    x = _t(x, dtype)
    stages = []
    with torch.no_grad():
        if cfg.kind == "ae":
            if cfg.scale_io:
                x = (x * 2) - 1
            x = stem(x, p, "block0.0", "block0.1")
            for j in range(cfg.depths[0]):
                x = block(x, p, f"block0.2.{j}", "mish")
            stages.append(x)
            for s in (1, 2, 3):
                x = downsample(x, p, f"block{s}.0", f"block{s}.1")
                for j in range(cfg.depths[s]):
                    x = block(x, p, f"block{s}.2.{j}", "mish")
                stages.append(x)
            z = ae_head(x, p, cfg)
        else:
            x = stem(x, p, "downsample_layers.0.0", "downsample_layers.0.1")
            for j in range(cfg.depths[0]):
                x = block(x, p, f"stages.0.{j}", "gelu")
            stages.append(x)
            for s in (1, 2, 3):
                x = downsample(x, p, f"downsample_layers.{s}.0", f"downsample_layers.{s}.1")
                for j in range(cfg.depths[s]):
                    x = block(x, p, f"stages.{s}.{j}", "gelu")
                stages.append(x)
            # forward_features: GAP -> nn.LayerNorm(eps=1e-6) -> head, convnextv2.py:292-303
            x = x.mean([-2, -1])
            x = F.layer_norm(x, (x.shape[-1],), p["norm.weight"], p["norm.bias"], 1e-6)
            z = F.linear(x, p["head.weight"], p["head.bias"])
    if return_stages:
        return z, stages
    return z


def img_float32(img):
    """mtgvision/util/image.py:220-237 - uint8 -> /255, floats clipped to [0,1], float32."""
    import numpy as np

    img = np.asarray(img)
    if img.dtype in (np.uint8, np.int32):
        img = np.divide(img, 255.0, dtype=np.float32)
    elif img.dtype in (np.float16, np.float64, np.float32):
        img = img.astype(np.float32)
    else:
        raise Exception(f"Unsupported Numpy Type: {img.dtype}")
    return np.clip(img, 0.0, 1.0)


def predict_hwc(params, cfg, rgb_im):
    """CoreMlEncoder.predict: mtgvision/encoder_export.py:91-101 (batch of one, returns (z,))."""
    im = img_float32(rgb_im)
    assert im.ndim == 3 and im.shape[-1] == 3
    x = torch.from_numpy(im.transpose(2, 0, 1)[None].copy())
    z = encoder_forward(params, cfg, x)
    assert z.ndim == 2 and z.shape[0] == 1
    return z[0].numpy()
