"""Host side of the crop stage: batched perspective de-warp of card quads on the GPU.

Reference: `InstanceSeg.extract_dewarped(frame, out_size_hw=(192,128), expand_ratio=0.05)`,
mtgvision/od_export.py:95-111.
"""

from __future__ import annotations

import numpy as np
import torch

from . import native


def warp_workspace(nq: int, device) -> torch.Tensor:
    """device scratch for `warp_quads` of up to nq quads (the 3 x 3 coefficients in float64): callers on a hot path allocate
    it once (Pipeline does) and pass it in"""
    return torch.empty((int(native.lib().mtgv_warp_workspace_bytes(nq)) + 7) // 8, dtype=torch.float64, device=device)


def warp_quads(frames: torch.Tensor, quads: torch.Tensor, frame_idx: torch.Tensor, out_size_hw=(192, 128), expand_ratio: float = 0.05,
               workspace: torch.Tensor = None) -> torch.Tensor:
    """frames (nf, H, W, 3) uint8; quads (nq, 4, 2) float32 corner pixels (tl, tr, br, bl order of the
    card); frame_idx (nq,) int32 -> crops (nq, out_h, out_w, 3) uint8, all on the GPU.  `workspace`: see warp_workspace
    (allocated per call when absent or too small)."""
    native.require_gpu()
    assert frames.is_cuda and frames.dtype == torch.uint8 and frames.ndim == 4 and frames.shape[-1] == 3
    nq = quads.shape[0]
    oh, ow = out_size_hw
    out = torch.empty((nq, oh, ow, 3), dtype=torch.uint8, device=frames.device)
    if nq == 0:
        return out
    quads = quads.to(frames.device, torch.float32).contiguous()
    frame_idx = frame_idx.to(frames.device, torch.int32).contiguous()
    assert tuple(quads.shape) == (nq, 4, 2) and tuple(frame_idx.shape) == (nq,)
    L = native.lib()
    ws = workspace
    if ws is None or ws.device != frames.device or ws.numel() * 8 < int(L.mtgv_warp_workspace_bytes(nq)):
        ws = warp_workspace(nq, frames.device)
    with torch.cuda.device(frames.device):
        native.check(
            L.mtgv_warp_quads(native.ptr(frames.contiguous()), frames.shape[0], frames.shape[1], frames.shape[2], native.ptr(quads), native.ptr(frame_idx),
                              nq, oh, ow, float(expand_ratio), native.ptr(out), native.ptr(ws), ws.numel() * 8, native.stream())
        )
    return out


def select_cards(n_det: torch.Tensor, boxes: torch.Tensor, pad_boxes: torch.Tensor, k: int, want_quads: bool = True):
    """The K cards per frame that go on to the crop stage, in one library kernel: the K best detections of the padded
    detector outputs (n_det (F,), boxes (F, max_det, 4), score-descending) or pad_boxes[k] where a frame has fewer.
    -> (sel_boxes (F*K, 4), quads (F*K, 4, 2) or None, frame_idx (F*K,) int32)."""
    native.require_gpu()
    F, md = boxes.shape[0], boxes.shape[1]
    dev = boxes.device
    assert n_det.dtype == torch.int32 and boxes.dtype == torch.float32 and tuple(pad_boxes.shape) == (k, 4), f"{tuple(pad_boxes.shape)}"
    sel = torch.empty((F * k, 4), dtype=torch.float32, device=dev)
    quads = torch.empty((F * k, 4, 2), dtype=torch.float32, device=dev) if want_quads else None
    fidx = torch.empty((F * k,), dtype=torch.int32, device=dev)
    if F == 0:
        return sel, quads, fidx
    with torch.cuda.device(dev):
        native.check(native.lib().mtgv_select_cards(native.ptr(n_det), native.ptr(boxes), native.ptr(pad_boxes), F, md, k, native.ptr(sel),
                                                    native.ptr(quads), native.ptr(fidx), native.stream()))
    return sel, quads, fidx


def boxes_to_quads(boxes_xyxy: torch.Tensor) -> torch.Tensor:
    """(n, 4) xyxy -> (n, 4, 2) corners in the order extract_dewarped matches to [[0,0],[w,0],[w,h],[0,h]]."""
    x1, y1, x2, y2 = boxes_xyxy.unbind(-1)
    return torch.stack([torch.stack([x1, y1], -1), torch.stack([x2, y1], -1), torch.stack([x2, y2], -1), torch.stack([x1, y2], -1)], -2)


def mask_quads(masks_u8: torch.Tensor, boxes_xyxy: torch.Tensor = None, extents: bool = False):
    """masks (n, H, W) uint8 on the GPU (non-zero = card) -> (quads (n, 4, 2) float32, ok (n,) int32)
    [, extents (n, H, 2) int32: leftmost / rightmost foreground column of every row, -1 where the row is empty].

    The quad is the 4-vertex polygon cv2.approxPolyN would fit to the mask's hull, rolled so that corner 0 is the
    card's top-left, corners truncated to integers - the GPU form of `InstanceSeg._orient`
    (mtgvision/od_export.py:52-93).  Rows with an empty mask get `boxes_xyxy` (or zeros) and
    ok = 0."""
    native.require_gpu()
    assert masks_u8.is_cuda and masks_u8.dtype == torch.uint8 and masks_u8.ndim == 3, f"{tuple(masks_u8.shape)} {masks_u8.dtype}"
    n, h, w = masks_u8.shape
    quads = torch.empty((n, 4, 2), dtype=torch.float32, device=masks_u8.device)  # the kernel writes every row (an empty
    ok = torch.empty((n,), dtype=torch.int32, device=masks_u8.device)             # mask: its box or zeros, ok = 0)
    ext = torch.empty((n, h, 2), dtype=torch.int32, device=masks_u8.device) if extents else None
    if n == 0:
        return (quads, ok, ext) if extents else (quads, ok)
    if boxes_xyxy is not None:
        boxes_xyxy = boxes_xyxy.to(masks_u8.device, torch.float32).contiguous()
        assert tuple(boxes_xyxy.shape) == (n, 4), f"{tuple(boxes_xyxy.shape)}"
    with torch.cuda.device(masks_u8.device):
        native.check(native.lib().mtgv_mask_quads(native.ptr(masks_u8.contiguous()), n, h, w, native.ptr(boxes_xyxy), native.ptr(quads),
                                                  native.ptr(ok), native.ptr(ext), native.stream()))
    return (quads, ok, ext) if extents else (quads, ok)


def mask_quads_from_logits(mask_logits: torch.Tensor, boxes_xyxy: torch.Tensor = None, scale: int = 4, extents: bool = False):
    """(n, mh, mw) cropped mask logits -> (quads, ok [, extents]) of the (mh*scale, mw*scale) masks `binarize_masks` would
    produce, without materialising them (one kernel: interpolate, threshold, row extents, hull, approxPolyN, orientation)."""
    native.require_gpu()
    assert mask_logits.is_cuda and mask_logits.dtype == torch.float32 and mask_logits.ndim == 3
    n, mh, mw = mask_logits.shape
    quads = torch.empty((n, 4, 2), dtype=torch.float32, device=mask_logits.device)  # the kernel writes every row
    ok = torch.empty((n,), dtype=torch.int32, device=mask_logits.device)
    ext = torch.empty((n, mh * scale, 2), dtype=torch.int32, device=mask_logits.device) if extents else None
    if n == 0:
        return (quads, ok, ext) if extents else (quads, ok)
    if boxes_xyxy is not None:
        boxes_xyxy = boxes_xyxy.to(mask_logits.device, torch.float32).contiguous()
        assert tuple(boxes_xyxy.shape) == (n, 4), f"{tuple(boxes_xyxy.shape)}"
    with torch.cuda.device(mask_logits.device):
        native.check(native.lib().mtgv_mask_quads_logits(native.ptr(mask_logits.contiguous()), n, mh, mw, scale, native.ptr(boxes_xyxy),
                                                         native.ptr(quads), native.ptr(ok), native.ptr(ext), native.stream()))
    return (quads, ok, ext) if extents else (quads, ok)
