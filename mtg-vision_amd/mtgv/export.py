"""Export surface: a plain `torch.nn.Module` with the reference's state_dict keys, for `torch.jit.trace`
(what `ConvNeXtV2Encoder.to_coreml` traces before handing to coremltools, mtgvision/models/convnextv2ae.py:268-278)
and `torch.onnx.export` where the `onnx` package exists.

This module is NOT on the recognition path: `mtgv.Encoder` never calls it and there is no fallback to it.
It exists so that artefacts for other runtimes (TorchScript / ONNX / CoreML) can still be produced from the
same checkpoint; it is checked on CPU against the golden vectors like the oracle (tests/test_export_cpu.py).
"""

from __future__ import annotations

from typing import Mapping

import numpy as np
import torch
from torch import nn

from . import spec


class _LN(nn.Module):
    """LayerNorm over channels; data_format as in convnextv2.py:133-160"""

    def __init__(self, c, channels_first):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.cf = channels_first

    def forward(self, x):
        if not self.cf:
            return nn.functional.layer_norm(x, (x.shape[-1],), self.weight, self.bias, 1e-6)
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        return self.weight[:, None, None] * ((x - u) / torch.sqrt(s + 1e-6)) + self.bias[:, None, None]


class _GRN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.gamma = nn.Parameter(torch.zeros(1, 1, 1, c))
        self.beta = nn.Parameter(torch.zeros(1, 1, 1, c))

    def forward(self, x):
        gx = torch.norm(x, p=2, dim=(1, 2), keepdim=True)
        return self.gamma * (x * (gx / (gx.mean(dim=-1, keepdim=True) + 1e-6))) + self.beta + x


class _Block(nn.Module):
    def __init__(self, c, act):
        super().__init__()
        self.dwconv = nn.Conv2d(c, c, 7, padding=3, groups=c)
        self.norm = _LN(c, False)
        self.pwconv1 = nn.Linear(c, 4 * c)
        self.act = nn.Mish() if act == "mish" else nn.GELU()
        self.grn = _GRN(4 * c)
        self.pwconv2 = nn.Linear(4 * c, c)

    def forward(self, x):
        y = self.dwconv(x).permute(0, 2, 3, 1)
        y = self.pwconv2(self.grn(self.act(self.pwconv1(self.norm(y)))))
        return x + y.permute(0, 3, 1, 2)


class _MLP(nn.Module):
    def __init__(self, i, h, o):
        super().__init__()
        self.layers = nn.Sequential(nn.Linear(i, h), nn.Mish(), nn.Linear(h, o), nn.Identity())

    def forward(self, x):
        return self.layers(x)


class _Flatten(nn.Module):
    def __init__(self, n):
        super().__init__()
        self.n = n

    def forward(self, x):
        return x.reshape(-1, self.n)


class _GAP(nn.Module):
    def forward(self, x):
        return x.mean([-2, -1])[:, :, None, None]


class EncoderModule(nn.Module):
    """Same sub-module names, hence the same state_dict keys, as the reference encoder `cfg` describes."""

    def __init__(self, cfg: spec.EncoderConfig):
        super().__init__()
        self.cfg = cfg
        d, c = cfg.depths, cfg.dims
        act = cfg.act
        if cfg.kind == "ae":
            self.block0 = nn.Sequential(nn.Conv2d(cfg.in_chans, c[0], 4, 4), _LN(c[0], True), nn.Sequential(*[_Block(c[0], act) for _ in range(d[0])]))
            for s in (1, 2, 3):
                setattr(self, f"block{s}", nn.Sequential(_LN(c[s - 1], True), nn.Conv2d(c[s - 1], c[s], 2, 2), nn.Sequential(*[_Block(c[s], act) for _ in range(d[s])])))
            z, ht = cfg.z_size, cfg.head_type
            if ht.startswith("conv"):
                zc = z // cfg.internal_num
                self.pool = nn.Sequential(nn.Conv2d(c[3], zc, 1), nn.Mish() if "+act" in ht else nn.Identity(), _LN(zc, True), _Flatten(z))
                head_in = z
            else:
                self.pool = nn.Sequential(_GAP(), _LN(c[3], True), _Flatten(c[3]))
                head_in = c[3]
            self.head = _MLP(head_in, z, z) if ht.endswith("+mlp") else nn.Linear(head_in, z)
        else:
            self.downsample_layers = nn.ModuleList([nn.Sequential(nn.Conv2d(cfg.in_chans, c[0], 4, 4), _LN(c[0], True))])
            for s in (1, 2, 3):
                self.downsample_layers.append(nn.Sequential(_LN(c[s - 1], True), nn.Conv2d(c[s - 1], c[s], 2, 2)))
            self.stages = nn.ModuleList([nn.Sequential(*[_Block(c[s], act) for _ in range(d[s])]) for s in range(4)])
            self.norm = nn.LayerNorm(c[3], eps=1e-6)
            self.head = nn.Linear(c[3], cfg.z_size)

    def forward(self, x):
        cfg = self.cfg
        if cfg.kind == "ae":
            if cfg.scale_io:
                x = (x * 2) - 1
            x = self.block3(self.block2(self.block1(self.block0(x))))
            return self.head(self.pool(x)).reshape(x.size(0), cfg.z_size)
        for i in range(4):
            x = self.stages[i](self.downsample_layers[i](x))
        return self.head(self.norm(x.mean([-2, -1])))


def to_torch_module(cfg: spec.EncoderConfig, state_dict: Mapping) -> EncoderModule:
    m = EncoderModule(cfg).eval()
    sd = spec.strip_checkpoint_prefix(state_dict)
    want = spec.encoder_param_shapes(cfg)
    m.load_state_dict({k: torch.as_tensor(np.asarray(sd[k]) if not isinstance(sd[k], torch.Tensor) else sd[k]).float() for k in want}, strict=True)
    return m


def export_torchscript(cfg: spec.EncoderConfig, state_dict: Mapping, path: str):
    """`torch.jit.trace` of the encoder on a (1, 3, H, W) example, saved to `path` (what to_coreml feeds coremltools)."""
    m = to_torch_module(cfg, state_dict)
    ex = torch.randn((1, cfg.in_chans, *cfg.image_hw))
    ts = torch.jit.trace(m, ex)
    ts.save(path)
    return ts


def export_onnx(cfg: spec.EncoderConfig, state_dict: Mapping, path: str):
    """ONNX export when the `onnx` package is importable (it is not in the build image)."""
    import importlib.util

    if importlib.util.find_spec("onnx") is None:
        raise RuntimeError("the onnx package is not installed")
    m = to_torch_module(cfg, state_dict)
    ex = torch.randn((1, cfg.in_chans, *cfg.image_hw))
    torch.onnx.export(m, ex, path, input_names=["x"], output_names=["z"], dynamic_axes={"x": {0: "n"}, "z": {0: "n"}})
