"""Host side of the embedding stage.

`Encoder.encode(batch)` is the north-star name; `encode(x)` with an NCHW float
tensor has the semantics of `MtgVisionEncoder.encode` (mtgvision/encoder_train.py:356-358)
and `predict(rgb_im)` those of `CoreMlEncoder.predict` (mtgvision/encoder_export.py:91-110).
All arithmetic happens in libmtgv.so on the GPU; this file only moves tensors.
"""

from __future__ import annotations

import ctypes as C
from typing import Mapping, Optional, Sequence, Union

import numpy as np
import torch

from . import native, spec

_HEAD_CODE = {h: i for i, h in enumerate(spec.HEAD_TYPES)}
_LAYOUT_NCHW_F32, _LAYOUT_NHWC_F32, _LAYOUT_NHWC_U8 = 0, 1, 2


class Encoder:
    def __init__(
        self,
        cfg: spec.EncoderConfig,
        state_dict: Optional[Mapping[str, Union[np.ndarray, torch.Tensor]]] = None,
        max_batch: int = 256,
        device: Optional[Union[int, str, torch.device]] = None,
    ):
        native.require_gpu()
        self.cfg = cfg
        self.max_batch = int(max_batch)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        c = native.EncoderCfg()
        c.kind = 0 if cfg.kind == "ae" else 1
        c.image_h, c.image_w = cfg.image_hw
        c.in_chans = cfg.in_chans
        c.z_size = cfg.z_size
        for i in range(4):
            c.depths[i] = cfg.depths[i]
            c.dims[i] = cfg.dims[i]
        c.head_type = _HEAD_CODE[cfg.head_type]
        c.scale_io = 1 if (cfg.kind == "ae" and cfg.scale_io) else 0
        c.max_batch = self.max_batch
        self._h = native.c_vp(0)
        with torch.cuda.device(self.device):
            native.check(native.lib().mtgv_encoder_create(C.byref(c), C.byref(self._h)))
        if state_dict is not None:
            self.load_state_dict(state_dict)

    # ---- checkpoint surface -------------------------------------------------
    def load_state_dict(self, state_dict: Mapping[str, Union[np.ndarray, torch.Tensor]], strict: bool = True):
        """Accepts the reference's keys: encoder keys, or a Lightning ``state_dict`` with the
        ``model.encoder.`` prefix (mtgvision/encoder_train.py:263-288).  Decoder keys are ignored."""
        sd = spec.strip_checkpoint_prefix(state_dict)
        want = spec.encoder_param_shapes(self.cfg)
        L = native.lib()
        with torch.cuda.device(self.device):
            for key, shape in want.items():
                if key not in sd:
                    if strict:
                        raise KeyError(f"missing parameter {key}")
                    continue
                a = sd[key]
                a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
                a = np.ascontiguousarray(a, dtype=np.float32)
                assert tuple(a.shape) == tuple(shape), f"{key}: shape {tuple(a.shape)} != {tuple(shape)}"
                native.check(L.mtgv_encoder_set_param(self._h, key.encode(), a.ctypes.data_as(native.c_vp), a.size))
        if strict:
            missing = L.mtgv_encoder_missing_params(self._h)
            if missing:
                raise KeyError(f"{missing} encoder parameters were not provided")
        return self

    @classmethod
    def from_checkpoint(cls, path, model_name: str, x_size_hw=(192, 128), head_type="conv+linear", **kw):
        """Load ``ckpt["state_dict"]`` of a Lightning checkpoint without Lightning installed."""
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
        return cls(spec.encoder_config(model_name, x_size_hw, head_type), sd, **kw)

    # ---- forward --------------------------------------------------------------
    @property
    def input_hwc(self):
        h, w = self.cfg.image_hw
        return h, w, self.cfg.in_chans

    def _forward_dev(self, x: torch.Tensor, layout: int) -> torch.Tensor:
        n = x.shape[0]
        z = torch.empty((n, self.cfg.z_size), dtype=torch.float32, device=self.device)
        L = native.lib()
        with torch.cuda.device(self.device):
            for i in range(0, n, self.max_batch):
                xb = x[i : i + self.max_batch]
                zb = z[i : i + self.max_batch]
                native.check(L.mtgv_encoder_forward(self._h, native.ptr(xb), layout, xb.shape[0], native.ptr(zb), native.stream()))
        return z

    def encode(self, batch) -> torch.Tensor:
        """(N,3,H,W) float tensor in [0,1] -> (N, z_size) float32 on the GPU.

        Also accepts an (N,H,W,3) uint8 tensor / array or a list of HWC images.
        Empty batches return an empty (0, z_size) tensor."""
        h, w = self.cfg.image_hw
        if isinstance(batch, (list, tuple)):
            batch = np.stack([np.asarray(b) for b in batch]) if len(batch) else np.zeros((0, h, w, 3), np.uint8)
        if isinstance(batch, np.ndarray):
            batch = torch.from_numpy(np.ascontiguousarray(batch))
        assert isinstance(batch, torch.Tensor) and batch.ndim == 4, f"{getattr(batch, 'shape', None)}"
        if batch.shape[0] == 0:
            return torch.empty((0, self.cfg.z_size), dtype=torch.float32, device=self.device)
        if batch.dtype == torch.uint8:
            assert tuple(batch.shape[1:]) == (h, w, 3), f"{tuple(batch.shape)}"
            return self._forward_dev(batch.to(self.device).contiguous(), _LAYOUT_NHWC_U8)
        if batch.shape[-1] == 3 and batch.shape[1] != 3:
            assert tuple(batch.shape[1:]) == (h, w, 3), f"{tuple(batch.shape)}"
            return self._forward_dev(batch.to(self.device, torch.float32).contiguous(), _LAYOUT_NHWC_F32)
        assert tuple(batch.shape[1:]) == (3, h, w), f"{tuple(batch.shape)}"
        return self._forward_dev(batch.to(self.device, torch.float32).contiguous(), _LAYOUT_NCHW_F32)

    __call__ = encode

    def predict(self, rgb_im: np.ndarray) -> np.ndarray:
        """CoreMlEncoder.predict: one HWC image (uint8, or float in [0,1]) -> (z_size,) float32."""
        rgb_im = np.asarray(rgb_im)
        assert rgb_im.ndim == 3, f"{rgb_im.shape}"
        assert rgb_im.shape[-1] == 3, f"{rgb_im.shape}"
        if rgb_im.dtype in (np.uint8,):
            x = torch.from_numpy(np.ascontiguousarray(rgb_im))[None]
        elif rgb_im.dtype in (np.int32,):
            x = torch.from_numpy(np.divide(rgb_im, 255.0, dtype=np.float32))[None]
        elif rgb_im.dtype in (np.float16, np.float32, np.float64):
            x = torch.from_numpy(np.ascontiguousarray(rgb_im, dtype=np.float32))[None]
        else:
            raise Exception(f"Unsupported Numpy Type: {rgb_im.dtype}")
        z = self.encode(x)
        assert z.ndim == 2
        assert z.shape[0] == 1
        return z[0].cpu().numpy()

    def ran_forward(self):
        return self.predict(np.random.rand(*self.input_hwc))

    def set_graph(self, mode: int = 1, max_n: int = 0):
        """mode 1: small batches (<= max_n images) replay a captured hipGraph; mode 0: eager launches only."""
        native.check(native.lib().mtgv_encoder_set_graph(self._h, int(mode), int(max_n)))

    # ---- introspection ----------------------------------------------------------
    def set_capture(self, on: bool = True):
        native.check(native.lib().mtgv_encoder_set_capture(self._h, 1 if on else 0))

    def stage_output(self, stage: int, n: int) -> torch.Tensor:
        h, w = self.cfg.stage_hw[stage]
        out = torch.empty((n, h, w, self.cfg.dims[stage]), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            native.check(native.lib().mtgv_encoder_stage_output(self._h, stage, n, native.ptr(out), native.stream()))
        return out

    def flops_per_image(self):
        g, d = C.c_double(0), C.c_double(0)
        native.check(native.lib().mtgv_encoder_flops(self._h, C.byref(g), C.byref(d)))
        return g.value, d.value

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                native.lib().mtgv_encoder_destroy(self._h)
                self._h = native.c_vp(0)
        except Exception:
            pass
