"""detect -> crop -> embed -> match on one GPU without host round trips.

The reference runs this per frame and per card with batch 1 (mtgvision/server.py:133-207:
segmenter(frame) -> extract_dewarped -> encoder.predict -> vecs.query_nearby(k=3)).  Here a whole
batch of frames goes through each stage once; every stage's arithmetic is in libmtgv.so.
"""

from __future__ import annotations

import os
from typing import Callable, Optional

import torch

from . import native
from .crop import mask_quads_from_logits, select_cards, warp_quads, warp_workspace
from .detector import Detector
from .encoder import Encoder
from .matcher import Matcher

# fixed quads (pixels of the 640x640 frame) used when a frame has fewer than K detections, so that
# cards/sec is well defined on synthetic frames
_PAD_BOXES = torch.tensor(
    [[40.0, 60.0, 168.0, 252.0], [200.0, 60.0, 328.0, 252.0], [360.0, 60.0, 488.0, 252.0], [500.0, 60.0, 628.0, 252.0],
     [40.0, 330.0, 168.0, 522.0], [200.0, 330.0, 328.0, 522.0], [360.0, 330.0, 488.0, 522.0], [500.0, 330.0, 628.0, 522.0]]
)


class HostFrames:
    """Frame batches that arrive in host memory, as the reference's do (one JPEG per websocket message decoded on the host,
    mtgvision/server.py:272-280): the batches are copied to the GPU on a third stream - hipMemcpyAsync from pinned memory,
    the copy engine, no kernel - into a small ring of device buffers while the previous batches are being processed.

        src = HostFrames(pinned_batches, device)
        outs = pipe.run_many(src.leases(n_steps))

    A lease's `tensor()` makes the consuming stream wait for its copy; `done()` (called by the pipeline after the last
    stage that reads the frames - the de-warp) lets the buffer be overwritten by the copy `depth` batches later."""

    def __init__(self, host_batches, device, depth: int = 3):
        assert depth >= 2 and len(host_batches) > 0
        self.host = [b if b.is_pinned() else b.pin_memory() for b in host_batches]
        self.device = torch.device(device)
        self.depth = depth
        self.s_copy = torch.cuda.Stream(self.device)
        self.bufs = [torch.empty(self.host[0].shape, dtype=self.host[0].dtype, device=self.device) for _ in range(depth)]
        self.copied = [torch.cuda.Event() for _ in range(depth)]
        self.released = [None] * depth
        self._issued = 0

    class _Lease:
        def __init__(self, src, slot):
            self.src, self.slot = src, slot

        def tensor(self) -> torch.Tensor:
            torch.cuda.current_stream(self.src.device).wait_event(self.src.copied[self.slot])
            return self.src.bufs[self.slot]

        def done(self) -> None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.src.device))
            self.src.released[self.slot] = ev

    def _issue(self, batch_index: int):
        j = self._issued % self.depth
        h = self.host[batch_index % len(self.host)]
        self._issued += 1
        with torch.cuda.stream(self.s_copy):
            if self.released[j] is not None:
                self.s_copy.wait_event(self.released[j])  # the batch that used this buffer has been de-warped
            self.bufs[j].copy_(h, non_blocking=True)
            self.copied[j].record(self.s_copy)
        return HostFrames._Lease(self, j)

    def leases(self, n: int):
        """n leases of host batches 0, 1, ... (cyclically); copies run one to two batches ahead of their consumer.  Copy
        i + 1 is issued when lease i is handed out (i >= 1): its buffer was last used by lease i - 2, whose done() the
        pipeline recorded one batch ago - so the event the copy has to wait for always exists by then."""
        ahead = [self._issue(b) for b in range(min(n, self.depth - 1))]
        issued = len(ahead)
        for i in range(n):
            if i > 0 and issued < n:
                ahead.append(self._issue(issued))
                issued += 1
            yield ahead.pop(0)


def _frames_of(item):
    """(tensor, lease or None) of a batch handed to run / run_many: a device tensor, or a lease of HostFrames"""
    return (item.tensor(), item) if hasattr(item, "tensor") and hasattr(item, "done") else (item, None)


class Pipeline:
    def __init__(self, detector: Detector, encoder: Encoder, matcher, cards_per_frame: int = 8, top_k: int = 1,
                 match_fn: Optional[Callable] = None, quad_source: str = "box"):
        """quad_source: "box" crops the detection boxes (the synthetic bench workload);
        "mask" crops the oriented quad fitted to each detection's mask, as the reference does
        (od_export.py:52-111: InstanceSeg._orient + extract_dewarped)."""
        assert quad_source in ("box", "mask"), quad_source
        self.quad_source = quad_source
        self.detector, self.encoder, self.matcher = detector, encoder, matcher
        self.K = int(cards_per_frame)
        self.top_k = int(top_k)
        # a caller-supplied match (bench.py: the sharded bank's local top-k + all-gathers + merge) may issue collectives, which
        # run on the process group's own normal-priority stream: run_many then keeps every stream at normal priority (below)
        self._plain_match = match_fn is None
        self.match_fn = match_fn or (lambda z, k: matcher.match(z, k))
        reps = (self.K + _PAD_BOXES.shape[0] - 1) // _PAD_BOXES.shape[0]
        self._pad = _PAD_BOXES.repeat(reps, 1)[: self.K].to(detector.device).contiguous()
        self._warp_ws = None

    def _embed_match(self, frames_u8: torch.Tensor, det, lease=None):
        return self._embed(det, *self._crop(frames_u8, det, lease))

    def _crop(self, frames_u8: torch.Tensor, det, lease=None):
        """detections -> (boxes (F, K, 4), crops (F * K, h, w, 3) uint8): the K best boxes or mask quadrilaterals, de-warped"""
        F, K = frames_u8.shape[0], self.K
        # the K highest-confidence detections per frame (NMS output is score-descending), pad boxes where a frame has
        # fewer; every step of the glue is a library kernel (no PyTorch arithmetic on the streams of the step)
        want_mask = self.quad_source == "mask"
        sel, quads, frame_idx = select_cards(det["n_det"], det["boxes"], self._pad, K, want_quads=not want_mask)
        if want_mask:
            # masks of the K best detections (logits are zero outside a detection's box and in rows beyond n_det):
            # interpolated to frame resolution, thresholded and fitted inside one kernel; an empty mask - no detection
            # in that slot, or nothing above threshold - falls back to the slot's box inside the kernel
            ml = det["mask_logits"]
            assert ml.shape[1] == K, f"mask rows {ml.shape[1]} != cards per frame {K}"
            quads, _ = mask_quads_from_logits(ml.view(F * K, *ml.shape[-2:]), sel)
        boxes = sel.view(F, K, 4)
        if self._warp_ws is None or self._warp_ws.numel() < 9 * F * K:
            self._warp_ws = warp_workspace(F * K, frames_u8.device)  # once per pipeline (the largest batch seen so far)
        crops = warp_quads(frames_u8, quads, frame_idx, self.encoder.cfg.image_hw, 0.05, self._warp_ws)
        if lease is not None:
            lease.done()  # the de-warp is the last reader of the frames
        return boxes, crops

    def _embed(self, det, boxes, crops, match_stream=None):
        F, K = boxes.shape[0], self.K
        z = self.encoder.encode(crops)
        if match_stream is None:
            ids, scores = self.match_fn(z, self.top_k)
        else:  # the match on a stream of its own: the next batch's encoder does not queue behind it
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(z.device))
            with torch.cuda.stream(match_stream):
                match_stream.wait_event(ev)
                z.record_stream(match_stream)
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

    @staticmethod
    def overlap_enabled() -> bool:
        """Two-stream overlap is opt-in (MTGV_OVERLAP=on): while the split-precision GEMMs of one stream run, kernels of
        OTHER libraries that use packed-FP32 VALU instructions on another stream of the same GPU have been seen to lose
        lanes.  This library is built without those instructions, both streams of `run_many` launch
        library kernels only (output tensors are torch.empty, the glue between the stages is mtgv_select_cards) and
        tests/test_gpu_overlap.py guards the combination; an application that runs foreign kernels beside it keeps the
        default."""
        return os.environ.get("MTGV_OVERLAP", "off") == "on"

    def run_many(self, batches, flip_rgb: bool = True):
        """Process a sequence of frame batches with the detect + crop stages of batch i+1 (one stream) overlapped with the
        embed stage of batch i (a second stream) and its match (a third); with the library's own match the latter two run at
        high priority, with a caller-supplied (collective) match all three at normal priority - the comments below and
        DESIGN.md section 5 say why (round 4: +7 % cards/s over round 3's two equal-priority streams with the crops in front
        of the encoder, independent of the order in which the application created its handles).  The detector's late layers
        have too few tiles to fill 256 CUs on their own; they and the latency-bound crop kernels fill what the encoder's GEMMs
        of the previous batch leave idle.  Results are identical to `run` on each batch (same kernels, same order per stream).

        Opt-in: without MTGV_OVERLAP=on everything stays on the current stream (see overlap_enabled).  (History: with packed-FP32 VALU instructions in
        the library, kernels sharing a CU with the split-precision GEMM of the other stream sporadically lost a
        packed result in one 16-lane group; the library is built without those instructions - build.py, DESIGN.md
        section 5 - and tests/test_gpu_overlap.py guards the combination.)"""
        if not self.overlap_enabled():
            return [self.run(frames, flip_rgb) for frames in batches]
        dev = self.detector.device
        if not hasattr(self, "_s_det"):
            # The embed + match stream runs at high priority: it is the longer chain (6 of a step's 8 ms), so its kernels get the
            # CUs as if alone and the detect + crop stream fills what they leave idle (tails, half-empty rounds, small launches)
            # instead of sharing every CU half and half.  MTGV_STREAM_PRIO=none|det: equal priorities / the other way round.
            # With a match that issues collectives (sharded bank) every stream stays at normal priority: a high-priority stream
            # that calls into the process group - whose own stream is a normal-priority one, queued with long event waits in
            # the hardware queues the detect stream uses - cost 19 % (world-size-1 RCCL run: 25.6k vs 31.7k cards/s; 32.0k replicated).
            prio = os.environ.get("MTGV_STREAM_PRIO", "enc" if self._plain_match else "none")
            self._s_det = torch.cuda.Stream(dev, priority=-1 if prio == "det" else 0)
            self._s_enc = torch.cuda.Stream(dev, priority=-1 if prio == "enc" else 0)
        crop_on_det = os.environ.get("MTGV_CROP_STAGE", "det") != "enc"
        # the match (normalise, first-pass GEMM, re-rank, merge: 0.16 ms, half of it latency-bound) on a third stream:
        # the next batch's encoder starts as soon as this batch's is done (MTGV_MATCH_STREAM=0: behind the encoder)
        s_match = None
        if os.environ.get("MTGV_MATCH_STREAM", "1") == "1":
            if not hasattr(self, "_s_match"):
                # (high priority like the embed stream: in the runtime's normal-priority pool its stream would share a hardware
                # queue with another normal stream - which one depends on the order in which the application created its
                # handles - and the match's wait for the encoder, queued long before it can be satisfied, then holds up
                # whatever sits behind it in that queue: measured 8.3 vs 9.3 ms per step, tools/debug/enqueue_time.py)
                mp = os.environ.get("MTGV_MATCH_PRIO", "-1" if self._plain_match else "0")
                self._s_match = torch.cuda.Stream(dev, priority=0 if mp == "0" else -1)
            s_match = self._s_match
            s_match.wait_stream(torch.cuda.current_stream(dev))
        cur = torch.cuda.current_stream(dev)
        self._s_det.wait_stream(cur)
        self._s_enc.wait_stream(cur)
        # The detector's own fork-join is for the one-stream schedule (its late layers alone on 256 CUs); here the embed stream
        # fills those gaps, and the branch streams would only add normal-priority streams to the few hardware queues:
        # 8.05 vs 8.3 ms per step with it off, whatever the handle creation order (MTGV_OVERLAP_DET_FORK=1: left on)
        fork_off = os.environ.get("MTGV_OVERLAP_DET_FORK", "0") != "1"
        if fork_off:
            self.detector.set_fork(0)
        try:
            return self._run_overlapped(batches, flip_rgb, crop_on_det, s_match, cur)
        finally:
            if fork_off:
                self.detector.set_fork(-1)

    def _run_overlapped(self, batches, flip_rgb, crop_on_det, s_match, cur):
        import itertools

        outs, pending = [], None
        for item in itertools.chain(batches, [None]):  # (lazily: a HostFrames lease is issued when its turn comes)
            nxt = None
            if item is not None:
                with torch.cuda.stream(self._s_det):
                    frames, lease = _frames_of(item)
                    det = self.detector.forward(frames, flip_rgb, mask_rows=self.K)
                    # the crop stage (card selection, mask -> quadrilateral, de-warp: latency-bound kernels of a few hundred
                    # workgroups) belongs to the detect stream: beside the other stream's GEMMs it costs next to nothing, in
                    # front of the encoder it would leave most of the GPU idle (MTGV_CROP_STAGE=enc: the round-3 split)
                    cropped = self._crop(frames, det, lease) if crop_on_det else None
                    ev = torch.cuda.Event()
                    ev.record(self._s_det)
                nxt = (frames, det, ev, lease, cropped)
            if pending is not None:
                pf, pdet, pev, please, pcrop = pending
                with torch.cuda.stream(self._s_enc):
                    self._s_enc.wait_event(pev)
                    for t in list(pdet.values()) + list(pcrop or ()):
                        if t is not None:
                            t.record_stream(self._s_enc)
                    outs.append(self._embed(pdet, *pcrop, match_stream=s_match) if pcrop is not None
                                else self._embed_match(pf, pdet, please))
            pending = nxt
        cur.wait_stream(self._s_det)
        cur.wait_stream(self._s_enc)
        if s_match is not None:
            cur.wait_stream(s_match)
        for o in outs:
            for t in (o["ids"], o["scores"], o["z"], o["crops"], o["boxes"]):
                t.record_stream(cur)
        return outs

    def run(self, frames_u8: torch.Tensor, flip_rgb: bool = True):
        """frames (F, 640, 640, 3) uint8 on the GPU -> dict with ids (F, K, top_k) int64, scores, n_det (F,)
        and the intermediate crops / embeddings (device tensors)."""
        frames_u8, lease = _frames_of(frames_u8)
        det = self.detector.forward(frames_u8, flip_rgb, mask_rows=self.K)
        return self._embed_match(frames_u8, det, lease)
