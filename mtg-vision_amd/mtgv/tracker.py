"""Server-side temporal logic of the recognition loop.

Mirror of `TrackerCtx` / `TrackedData` in mtgvision/server.py:41-205: segment the frame, associate the detections with
tracks, de-warp every tracked card, re-embed a track at most every `update_wait_sec` seconds, keep an exponentially
weighted mean of its embeddings (`avg_z = w*z + (1-w)*avg_z`, un-normalised, server.py:183-185), query the three nearest
cards and join their payloads.  Differences, all on the host side:

* all tracks due for an update are embedded in ONE encoder batch and matched in ONE bank pass per frame (the
  reference issues one CoreML call and one Qdrant round trip per track, server.py:182-190); the results are the same.
* association: the reference uses `norfair.Tracker(mean_euclidean, distance_threshold=300, hit_counter_max=5,
  initialization_delay=2)` (server.py:100-106).  norfair is third-party and absent, so `MeanEuclideanTracker` states the
  same policy without the Kalman filter (a track's position is its last matched detection): PARITY UNPINNED.
* JPEG thumbnails (`encode_rgb_im`, server.py:223-226) use Pillow instead of cv2.
"""

from __future__ import annotations

import base64
import dataclasses
import hashlib
import io
import time
from typing import Any, Callable, Optional

import numpy as np


def get_color(seed) -> str:
    """stable '#rrggbb' per track id (server.py:215-221)"""
    h = int(hashlib.sha256(str(seed).encode()).hexdigest(), 16)
    return f"#{(h >> 16) & 0xFF:02x}{(h >> 8) & 0xFF:02x}{h & 0xFF:02x}"


def encode_rgb_im(rgb_im: np.ndarray) -> Optional[str]:
    """base64 JPEG (quality 50) of an RGB uint8 image; None when Pillow is unavailable"""
    try:
        from PIL import Image
    except Exception:
        return None
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(rgb_im)).save(buf, format="JPEG", quality=50)
    return base64.b64encode(buf.getvalue()).decode("utf-8")


@dataclasses.dataclass
class TrackedData:
    id: int
    color: str
    last_update_time: float = dataclasses.field(default_factory=lambda: time.time())
    last_instance: Any = None
    last_rgb_im: np.ndarray = None
    last_rgb_im_encoded: str = None
    avg_z: np.ndarray = None
    ave_nearby_points: list = dataclasses.field(default_factory=list)
    ave_nearby_cards: list = dataclasses.field(default_factory=list)

    def to_dict(self) -> dict:
        """wire format consumed by www/src/types.ts (server.py:57-82)"""
        matches = []
        for point, card in zip(self.ave_nearby_points, self.ave_nearby_cards):
            matches.append({
                "id": str(point.id),
                "score": point.score,
                "name": getattr(card, "name", None),
                "set_name": getattr(card, "set_name", None),
                "set_code": getattr(card, "set_code", None),
                "img_uri": getattr(card, "img_uri", None),
                "all_data": point.payload,
            })
        return {
            "id": str(self.id),
            "points": np.asarray(self.last_instance.xyxyxyxy).tolist(),
            "polygon": np.asarray(self.last_instance.points).tolist(),
            "polygon_closed": np.asarray(self.last_instance.points_closed).tolist(),
            "color": self.color,
            "img": self.last_rgb_im_encoded,
            "score": self.last_instance.conf,
            "matches": matches,
        }


@dataclasses.dataclass
class _Track:
    points: np.ndarray
    hits: int = 1          # detections matched so far (until the track is initialised)
    counter: int = 1       # norfair's hit counter: +1 on a match (capped), -1 on a miss, dropped below 0
    id: Optional[int] = None
    last_detection: int = -1


class MeanEuclideanTracker:
    """Greedy nearest association on the mean corner distance (the policy the reference configures norfair with)."""

    def __init__(self, distance_threshold: float = 300.0, hit_counter_max: int = 5, initialization_delay: int = 2):
        self.distance_threshold = float(distance_threshold)
        self.hit_counter_max = int(hit_counter_max)
        self.initialization_delay = int(initialization_delay)
        self.tracks: list[_Track] = []
        self._next_id = 1

    def update(self, detections: list[np.ndarray]) -> list[tuple[int, int]]:
        """detections: list of (P, 2) point sets -> [(track id, detection index)] for the initialised tracks that were
        matched in this frame, in track-id order."""
        for t in self.tracks:
            t.last_detection = -1
        pairs = []
        for ti, t in enumerate(self.tracks):
            for di, d in enumerate(detections):
                dist = float(np.linalg.norm(np.asarray(d, np.float64) - t.points, axis=1).mean())
                if dist < self.distance_threshold:
                    pairs.append((dist, ti, di))
        pairs.sort()
        used_t, used_d = set(), set()
        for dist, ti, di in pairs:  # closest pairs first, every track and detection at most once
            if ti in used_t or di in used_d:
                continue
            used_t.add(ti)
            used_d.add(di)
            t = self.tracks[ti]
            t.points = np.asarray(detections[di], np.float64)
            t.last_detection = di
            t.hits += 1
            t.counter = min(t.counter + 1, self.hit_counter_max)
            if t.id is None and t.hits > self.initialization_delay:
                t.id = self._next_id
                self._next_id += 1
        for ti, t in enumerate(self.tracks):
            if ti not in used_t:
                t.counter -= 1
        self.tracks = [t for t in self.tracks if t.counter >= 0]
        for di, d in enumerate(detections):
            if di not in used_d:
                self.tracks.append(_Track(points=np.asarray(d, np.float64)))
        out = [(t.id, t.last_detection) for t in self.tracks if t.id is not None and t.last_detection >= 0]
        return sorted(out)


class TrackerCtx:
    def __init__(self, update_wait_sec: float = 0.5, ewma_weight: float = 0.1, *, segmenter, encoder, vecs, data=None,
                 clock: Callable[[], float] = time.time, thumbnails: bool = True):
        """segmenter(frame) -> list[InstanceSeg]; encoder: `mtgv.Encoder` (batched `.encode`) or anything with
        `.predict(rgb_im)`; vecs: `VectorStoreQdrant`; data: optional card index with `.get_card_by_id(id)`
        (the reference's SyntheticBgFgMtgImages, server.py:190-193)."""
        self.update_wait_sec = update_wait_sec
        self.ewma_weight = ewma_weight
        self.segmenter, self.encoder, self.vecs, self.data = segmenter, encoder, vecs, data
        self.clock = clock
        self.thumbnails = thumbnails
        self.tracker = MeanEuclideanTracker(distance_threshold=300, hit_counter_max=5, initialization_delay=2)
        self.tracked_data: dict[int, TrackedData] = {}

    def _embed(self, crops: list[np.ndarray]) -> np.ndarray:
        if hasattr(self.encoder, "encode"):
            return self.encoder.encode(np.stack(crops)).cpu().numpy()
        return np.stack([self.encoder.predict(c) for c in crops])

    def update(self, rgb_frame: np.ndarray) -> list[TrackedData]:
        segments = self.segmenter(rgb_frame)
        matched = self.tracker.update([np.asarray(seg.xyxyxyxy) for seg in segments])
        now = self.clock()
        objs, due = [], []
        for tid, di in matched:
            seg = segments[di]
            trk = self.tracked_data.get(tid)
            if trk is None:
                trk = TrackedData(id=tid, color=get_color(tid), last_update_time=now, last_instance=seg)
                self.tracked_data[tid] = trk
            trk.last_instance = seg
            trk.last_rgb_im = seg.extract_dewarped(rgb_frame)
            trk.last_rgb_im_encoded = encode_rgb_im(trk.last_rgb_im) if self.thumbnails else None
            if now - trk.last_update_time > self.update_wait_sec or trk.avg_z is None:
                due.append(trk)
            objs.append(trk)
        if due:
            zs = self._embed([t.last_rgb_im for t in due])
            for t, z in zip(due, zs):
                if t.avg_z is None:
                    t.avg_z = z
                t.avg_z = self.ewma_weight * z + (1 - self.ewma_weight) * t.avg_z
            if hasattr(self.vecs, "query_nearby_batch"):
                hits = self.vecs.query_nearby_batch([t.avg_z for t in due], k=3, with_payload=True, with_vectors=False)
            else:
                hits = [self.vecs.query_nearby(t.avg_z, k=3, with_payload=True, with_vectors=False) for t in due]
            for t, h in zip(due, hits):
                t.ave_nearby_points = h
                t.ave_nearby_cards = [self.data.get_card_by_id(p.id) if self.data is not None else None for p in h]
                t.last_update_time = now
        return objs
