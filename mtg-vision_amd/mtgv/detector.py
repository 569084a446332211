"""Host side of the detect stage: YOLOv8n-seg forward + decode + NMS + mask logits on the GPU.

`Detector.detect(frame)` is the north-star name; the reference's boundary is
`CardSegmenter(model_path)(rgb_im) -> list[InstanceSeg]` (mtgvision/od_export.py:141-160), which
`mtgv.adapters.CardSegmenter` provides on top of this class.  All arithmetic is in libmtgv.so.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Mapping, Optional, Union

import numpy as np
import torch

from . import native, spec


@dataclass
class Detections:
    """Raw per-frame detector output (score-descending)."""

    boxes_xyxy: torch.Tensor  # (n, 4) float32, pixels of the letterboxed 640x640 frame
    conf: torch.Tensor  # (n,) float32
    cls: torch.Tensor  # (n,) int64
    keep_idx: torch.Tensor  # (n,) int64 anchor index in [0, 8400)
    mask_logits: Optional[torch.Tensor]  # (n, 160, 160) float32, zero outside the box


def letterbox_geometry(h: int, w: int, size: int = 640):
    """ultralytics LetterBox geometry: (ratio, nh, nw, top, left) - scale to fit, centre (round(d - 0.1) like upstream)"""
    r = min(size / h, size / w)
    nh, nw = int(round(h * r)), int(round(w * r))
    dh, dw = (size - nh) / 2, (size - nw) / 2
    return r, nh, nw, int(round(dh - 0.1)), int(round(dw - 0.1))


def letterbox_device(frame: torch.Tensor, size: int = 640, pad_value: int = 114):
    """(H, W, 3) uint8 frame on the GPU -> ((1, size, size, 3) uint8 letterboxed image on the GPU, ratio, (left, top)): one
    library kernel (resize.hip: letterbox_u8_kernel) instead of a host resample + pad."""
    native.require_gpu()
    assert frame.is_cuda and frame.dtype == torch.uint8 and frame.ndim == 3 and frame.shape[-1] == 3, f"{tuple(frame.shape)} {frame.dtype}"
    h, w = int(frame.shape[0]), int(frame.shape[1])
    r, nh, nw, top, left = letterbox_geometry(h, w, size)
    out = torch.empty((1, size, size, 3), dtype=torch.uint8, device=frame.device)
    with torch.cuda.device(frame.device):
        native.check(native.lib().mtgv_letterbox_u8(native.ptr(frame.contiguous()), h, w, native.ptr(out), size, nh, nw, top, left, pad_value,
                                                    native.stream()))
    return out, r, (left, top)


def letterbox(frame: np.ndarray, size: int = 640, pad_value: int = 114):
    """ultralytics LetterBox for non-.pt backends: scale to fit, centre, pad to size x size with 114.

    Returns (image (size,size,3) uint8, ratio, (pad_left, pad_top)).  Frames that already fit
    (e.g. the 640x480 webcam frames of server.py / od_cam.py) are only padded; other sizes are
    resized bilinearly on the host (cv2.resize is not available here: that resample is unpinned).
    """
    h, w = frame.shape[:2]
    r = min(size / h, size / w)
    nh, nw = int(round(h * r)), int(round(w * r))
    img = frame
    if (nh, nw) != (h, w):
        t = torch.from_numpy(np.ascontiguousarray(frame)).permute(2, 0, 1)[None].float()
        t = torch.nn.functional.interpolate(t, (nh, nw), mode="bilinear", align_corners=False)
        img = t[0].permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).numpy()
    dh, dw = (size - nh) / 2, (size - nw) / 2
    top, left = int(round(dh - 0.1)), int(round(dw - 0.1))
    out = np.full((size, size, 3), pad_value, np.uint8)
    out[top : top + nh, left : left + nw] = img
    return out, r, (left, top)


class Detector:
    def __init__(
        self,
        cfg: Optional[spec.DetectorConfig] = None,
        state_dict: Optional[Mapping[str, Union[np.ndarray, torch.Tensor]]] = None,
        max_batch: int = 32,
        device=None,
    ):
        native.require_gpu()
        self.cfg = cfg or spec.DetectorConfig()
        self.max_batch = int(max_batch)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        c = native.DetectorCfg()
        c.nc, c.imgsz, c.max_batch = self.cfg.nc, self.cfg.imgsz, self.max_batch
        c.conf, c.iou, c.max_det = self.cfg.conf, self.cfg.iou, self.cfg.max_det
        c.arch = 11 if self.cfg.arch == "11" else 8
        self._h = native.c_vp(0)
        with torch.cuda.device(self.device):
            native.check(native.lib().mtgv_detector_create(C.byref(c), C.byref(self._h)))
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, state_dict: Mapping[str, Union[np.ndarray, torch.Tensor]]):
        """ultralytics `model.state_dict()` keys (model.<i>....); `num_batches_tracked` is ignored."""
        want = spec.detector_param_shapes(self.cfg)
        L = native.lib()
        with torch.cuda.device(self.device):
            for key, shape in want.items():
                if key not in state_dict:
                    raise KeyError(f"missing parameter {key}")
                a = state_dict[key]
                a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
                a = np.ascontiguousarray(a, dtype=np.float32)
                assert tuple(a.shape) == tuple(shape), f"{key}: shape {tuple(a.shape)} != {tuple(shape)}"
                native.check(L.mtgv_detector_set_param(self._h, key.encode(), a.ctypes.data_as(native.c_vp), a.size))
            native.check(L.mtgv_detector_finalize(self._h))
        return self

    # ---- batched device API (what the pipeline uses) ---------------------------
    def forward(self, frames_u8: torch.Tensor, flip_rgb: bool = True, mask_rows: int = 0):
        """frames (n, 640, 640, 3) uint8 on the GPU -> dict of padded device tensors.

        n_det (n,) int32; boxes (n, max_det, 4); conf (n, max_det); cls, keep_idx (n, max_det) int32;
        mask_logits (n, mask_rows, 160, 160) if mask_rows > 0."""
        S, md = self.cfg.imgsz, self.cfg.max_det
        assert frames_u8.dtype == torch.uint8 and frames_u8.is_cuda and tuple(frames_u8.shape[1:]) == (S, S, 3), f"{tuple(frames_u8.shape)}"
        n = frames_u8.shape[0]
        assert 0 < n <= self.max_batch, f"batch {n} outside [1, {self.max_batch}]"
        dev = self.device
        # every element is written by the library (slots / mask rows beyond n_det as zeros): no fill kernels here
        out = {
            "n_det": torch.empty((n,), dtype=torch.int32, device=dev),
            "boxes": torch.empty((n, md, 4), dtype=torch.float32, device=dev),
            "conf": torch.empty((n, md), dtype=torch.float32, device=dev),
            "cls": torch.empty((n, md), dtype=torch.int32, device=dev),
            "keep_idx": torch.empty((n, md), dtype=torch.int32, device=dev),
            "mask_logits": torch.empty((n, mask_rows, S // 4, S // 4), dtype=torch.float32, device=dev) if mask_rows > 0 else None,
        }
        with torch.cuda.device(dev):
            native.check(
                native.lib().mtgv_detector_forward(
                    self._h, native.ptr(frames_u8.contiguous()), n, 1 if flip_rgb else 0, native.ptr(out["n_det"]), native.ptr(out["boxes"]),
                    native.ptr(out["conf"]), native.ptr(out["cls"]), native.ptr(out["keep_idx"]), native.ptr(out["mask_logits"]), int(mask_rows), native.stream(),
                )
            )
        return out

    def set_fork(self, mode: int):
        """the forward's internal fork-join: 1 on, 0 off, -1 the default (on unless MTGV_DET_FORK=0); mtgv_detector_set_fork"""
        native.check(native.lib().mtgv_detector_set_fork(self._h, int(mode)))

    def raw_outputs(self, n: int):
        """pred (n, 4+nc+32, 8400) and protos (n, 32, 160, 160) of the last forward (parity tests)."""
        S = self.cfg.imgsz
        pred = torch.empty((n, self.cfg.no, self.cfg.num_anchors), dtype=torch.float32, device=self.device)
        protos = torch.empty((n, self.cfg.nm, S // 4, S // 4), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            native.check(native.lib().mtgv_detector_raw(self._h, n, native.ptr(pred), native.ptr(protos), native.stream()))
        return pred, protos

    # ---- single-frame API -------------------------------------------------------
    def detect(self, frame: np.ndarray, flip_rgb: bool = True, masks: bool = True) -> Detections:
        """One HWC uint8 frame (any size) -> Detections in letterboxed 640x640 coordinates."""
        assert frame.ndim == 3 and frame.shape[-1] == 3 and frame.dtype == np.uint8, f"{frame.shape} {frame.dtype}"
        # the raw frame goes to the GPU as it is; scale-to-fit + pad there (the host `letterbox` only supplies the geometry
        # to callers that map coordinates back, e.g. CardSegmenter)
        x, _, _ = letterbox_device(torch.from_numpy(np.ascontiguousarray(frame)).to(self.device), self.cfg.imgsz)
        out = self.forward(x, flip_rgb, self.cfg.max_det if masks else 0)
        n = int(out["n_det"][0].item())
        return Detections(
            out["boxes"][0, :n], out["conf"][0, :n], out["cls"][0, :n].long(), out["keep_idx"][0, :n].long(),
            out["mask_logits"][0, :n] if masks else None,
        )

    __call__ = detect

    def flops_per_frame(self) -> float:
        f = C.c_double(0)
        native.check(native.lib().mtgv_detector_flops(self._h, C.byref(f)))
        return f.value

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                native.lib().mtgv_detector_destroy(self._h)
                self._h = native.c_vp(0)
        except Exception:
            pass


def nms(pred: torch.Tensor, nc: int, conf: float = 0.25, iou: float = 0.7, max_det: int = 300, max_wh: float = 7680.0):
    """Stand-alone NMS kernel on decoded predictions (n, 4+nc+nm, A) -> padded tensors like Detector.forward."""
    native.require_gpu()
    assert pred.is_cuda and pred.dtype == torch.float32 and pred.ndim == 3
    pred = pred.contiguous()
    n, no, na = pred.shape
    nm = no - 4 - nc
    dev = pred.device
    L = native.lib()
    ws = torch.empty((int(L.mtgv_nms_workspace_bytes(n, na)) + 3) // 4, dtype=torch.int32, device=dev)
    out = {
        "n_det": torch.zeros((n,), dtype=torch.int32, device=dev),
        "boxes": torch.zeros((n, max_det, 4), dtype=torch.float32, device=dev),
        "conf": torch.zeros((n, max_det), dtype=torch.float32, device=dev),
        "cls": torch.zeros((n, max_det), dtype=torch.int32, device=dev),
        "keep_idx": torch.zeros((n, max_det), dtype=torch.int32, device=dev),
    }
    with torch.cuda.device(dev):
        native.check(
            L.mtgv_nms(native.ptr(pred), n, nc, nm, na, conf, iou, max_det, max_wh, native.ptr(out["n_det"]), native.ptr(out["boxes"]),
                       native.ptr(out["conf"]), native.ptr(out["cls"]), native.ptr(out["keep_idx"]), native.ptr(ws), ws.numel() * 4, native.stream())
        )
    return out


def binarize_masks(mask_logits: torch.Tensor, scale: int = 4) -> torch.Tensor:
    """(n, 160, 160) cropped logits -> (n, 640, 640) uint8 {0,1}: bilinear x4 (align_corners=False), > 0."""
    native.require_gpu()
    assert mask_logits.is_cuda and mask_logits.dtype == torch.float32 and mask_logits.ndim == 3
    n, mh, mw = mask_logits.shape
    out = torch.empty((n, mh * scale, mw * scale), dtype=torch.uint8, device=mask_logits.device)
    with torch.cuda.device(mask_logits.device):
        native.check(native.lib().mtgv_mask_binarize(native.ptr(mask_logits.contiguous()), n, mh, mw, scale, native.ptr(out), native.stream()))
    return out
