"""Host side of the crop stage: batched perspective de-warp of card quads on the GPU.

Reference: `InstanceSeg.extract_dewarped(frame, out_size_hw=(192,128), expand_ratio=0.05)`,
mtgvision/od_export.py:95-111.
"""

from __future__ import annotations

import numpy as np
import torch

from . import native


def warp_quads(frames: torch.Tensor, quads: torch.Tensor, frame_idx: torch.Tensor, out_size_hw=(192, 128), expand_ratio: float = 0.05) -> torch.Tensor:
    """frames (nf, H, W, 3) uint8; quads (nq, 4, 2) float32 corner pixels (tl, tr, br, bl order of the
    card); frame_idx (nq,) int32 -> crops (nq, out_h, out_w, 3) uint8, all on the GPU."""
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
    ws = torch.empty((int(L.mtgv_warp_workspace_bytes(nq)) + 7) // 8, dtype=torch.float64, device=frames.device)
    with torch.cuda.device(frames.device):
        native.check(
            L.mtgv_warp_quads(native.ptr(frames.contiguous()), frames.shape[0], frames.shape[1], frames.shape[2], native.ptr(quads), native.ptr(frame_idx),
                              nq, oh, ow, float(expand_ratio), native.ptr(out), native.ptr(ws), ws.numel() * 8, native.stream())
        )
    return out


def boxes_to_quads(boxes_xyxy: torch.Tensor) -> torch.Tensor:
    """(n, 4) xyxy -> (n, 4, 2) corners in the order extract_dewarped matches to [[0,0],[w,0],[w,h],[0,h]]."""
    x1, y1, x2, y2 = boxes_xyxy.unbind(-1)
    return torch.stack([torch.stack([x1, y1], -1), torch.stack([x2, y1], -1), torch.stack([x2, y2], -1), torch.stack([x1, y2], -1)], -2)
