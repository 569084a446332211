"""ORACLE (test infrastructure, never shipped or measured as the product).

CPU restatement of the whole recognition path on a batch of frames, composed from the stage
oracles: detector_ref (parity unpinned, third-party ultralytics), warp_ref (parity unpinned,
third-party cv2), encoder_ref (pinned to golden vectors of the reference's own modules) and
match_ref (parity unpinned, third-party Qdrant).  Mirrors mtgvision/server.py:133-207 with the
fixed K-cards-per-frame convention of SURVEY.md section 8d (config 4).
"""

from __future__ import annotations

import numpy as np

from . import detector_ref, encoder_ref, match_ref, warp_ref

PAD_BOXES = np.asarray(
    [[40.0, 60.0, 168.0, 252.0], [200.0, 60.0, 328.0, 252.0], [360.0, 60.0, 488.0, 252.0], [500.0, 60.0, 628.0, 252.0],
     [40.0, 330.0, 168.0, 522.0], [200.0, 330.0, 328.0, 522.0], [360.0, 330.0, 488.0, 522.0], [500.0, 330.0, 628.0, 522.0]],
    np.float32,
)


def boxes_to_quads(b):
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    return np.stack([np.stack([x1, y1], -1), np.stack([x2, y1], -1), np.stack([x2, y2], -1), np.stack([x1, y2], -1)], 1)


def select_boxes(dets, K):
    out = []
    reps = (K + len(PAD_BOXES) - 1) // len(PAD_BOXES)
    pad = np.tile(PAD_BOXES, (reps, 1))[:K]
    for d in dets:
        b = pad.copy()
        n = min(K, len(d["keep_idx"]))
        b[:n] = d["boxes"][:n]
        out.append(b)
    return np.stack(out)


def mask_quads_of(dets, boxes, K, imgsz):
    """the reference's dataflow (od_export.py:52-111): the oriented quadrilateral of every selected detection's mask
    (quad_ref: hull + approxPolyN + orientation); slots without a detection, and empty masks, keep their box"""
    from . import quad_ref

    quads = boxes_to_quads(boxes.reshape(-1, 4)).reshape(len(dets), K, 4, 2).copy()
    for f, d in enumerate(dets):
        n = min(K, len(d["keep_idx"]))
        if n == 0:
            continue
        masks = detector_ref.masks_binary(d["mask_logits"][:n], imgsz)
        q, ok = quad_ref.mask_quads(masks, boxes[f, :n])
        quads[f, :n][ok > 0] = q[ok > 0]
    return quads.reshape(len(dets) * K, 4, 2)


def run(det_params, det_cfg, enc_params, enc_cfg, bank, frames_u8, K=8, top_k=1, flip_rgb=True, boxes=None, quads=None,
        quad_source="box"):
    """-> dict(ids (F,K,top_k), scores, boxes (F,K,4), crops (F*K,h,w,3) u8, z (F*K, z))

    `boxes` (and `quads` (F*K,4,2), e.g. from quad_ref.mask_quads) may be supplied to check the later stages on
    identical inputs.  quad_source "mask" (when the detector runs here): crops are the mask quadrilaterals, as in
    bench.py's default GPU dataflow; "box": the detection boxes."""
    F = frames_u8.shape[0]
    dets = None
    if boxes is None:
        dets, _, _ = detector_ref.detect(det_params, det_cfg, frames_u8, flip_rgb)
        boxes = select_boxes(dets, K)
    if quads is None and quad_source == "mask" and dets is not None:
        quads = mask_quads_of(dets, boxes, K, det_cfg.imgsz)
    if quads is None:
        quads = boxes_to_quads(boxes.reshape(F * K, 4))
    h, w = enc_cfg.image_hw
    crops = np.stack([warp_ref.warp_quad(frames_u8[i // K], quads[i], (h, w), 0.05) for i in range(F * K)])
    x = encoder_ref.img_float32(crops).transpose(0, 3, 1, 2)
    z = encoder_ref.encoder_forward(enc_params, enc_cfg, x).numpy()
    ids, scores = match_ref.cosine_topk(z, bank, top_k, dtype=np.float32)
    return {"ids": ids.reshape(F, K, top_k), "scores": scores.reshape(F, K, top_k), "boxes": boxes, "crops": crops, "z": z, "dets": dets}
