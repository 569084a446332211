"""Server-side temporal logic of the recognition loop.

Mirror of `TrackerCtx` / `TrackedData` in mtgvision/server.py:41-205: segment the frame, associate the detections with
tracks, de-warp every tracked card, re-embed a track at most every `update_wait_sec` seconds, keep an exponentially
weighted mean of its embeddings (`avg_z = w*z + (1-w)*avg_z`, un-normalised, server.py:183-185), query the three nearest
cards and join their payloads.  Differences, all on the host side:

* all tracks due for an update are embedded in ONE encoder batch and matched in ONE bank pass per frame (the
  reference issues one CoreML call and one Qdrant round trip per track, server.py:182-190); the results are the same.
* association: the reference uses `norfair.Tracker(mean_euclidean, distance_threshold=300, hit_counter_max=5,
  initialization_delay=2)` (server.py:100-106).  norfair is third-party and absent; `KalmanPointTracker` restates its
  published algorithm - per-coordinate constant-velocity Kalman filter with norfair's default parameters, detections
  matched greedily against the predicted positions, hit counters (-1 per frame, +2 per hit, capped), ids handed out after
  the initialisation delay: PARITY UNPINNED.  `MeanEuclideanTracker` is the same policy without the filter.
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


class _KalmanTrack:
    """One tracked object: norfair's `TrackedObject` with its default `OptimizedKalmanFilter` (R 4, Q 0.1, position
    variance 10, velocity variance 1), i.e. an independent [position, velocity] filter per coordinate with F = [[1, 1],
    [0, 1]], Q = q I, H = [1, 0]; the covariance is propagated inside `update`, `predict` only moves the state."""

    R, Q, POS_VAR, VEL_VAR = 4.0, 0.1, 10.0, 1.0

    def __init__(self, points: np.ndarray, period: int, initialization_delay: int):
        z = np.asarray(points, np.float64).reshape(-1)
        self.shape = np.asarray(points).shape
        self.pos, self.vel = z.copy(), np.zeros_like(z)
        self.pp = np.full_like(z, self.POS_VAR)
        self.pv = np.zeros_like(z)
        self.vv = np.full_like(z, self.VEL_VAR)
        self.hit_counter = period
        self.is_initializing = self.hit_counter <= initialization_delay
        self.id: Optional[int] = None
        self.last_detection = -1

    @property
    def estimate(self) -> np.ndarray:
        return self.pos.reshape(self.shape)

    def step(self):
        self.hit_counter -= 1
        self.pos = self.pos + self.vel  # filter.predict()

    def kalman_update(self, points: np.ndarray):
        z = np.asarray(points, np.float64).reshape(-1)
        e = z - self.pos
        s = self.pp + 2.0 * self.pv + self.vv + self.Q + self.R  # innovation variance of the predicted state
        k0, k1 = 1.0 - self.R / s, (self.pv + self.vv) / s
        self.pos = self.pos + k0 * e
        self.vel = self.vel + k1 * e
        self.pp, self.pv, self.vv = k0 * self.R, k1 * self.R, self.vv + self.Q - k1 * k1 * s


class KalmanPointTracker:
    """`norfair.Tracker(distance_function=mean_euclidean, distance_threshold, hit_counter_max, initialization_delay)`
    restated (server.py:100-106): every frame each object loses a hit and moves by its velocity; detections are matched
    to the predicted corner positions, closest pair first, first against the initialised objects and then against the
    initialising ones; a hit adds 2 (capped at `hit_counter_max`) and feeds the filter; an object gets its id once its
    counter exceeds the initialisation delay and is dropped when the counter falls below zero."""

    def __init__(self, distance_threshold: float = 300.0, hit_counter_max: int = 5, initialization_delay: int = 2, period: int = 1):
        self.distance_threshold = float(distance_threshold)
        self.hit_counter_max = int(hit_counter_max)
        self.initialization_delay = int(initialization_delay)
        self.period = int(period)
        self.tracks: list[_KalmanTrack] = []
        self._next_id = 1

    def _match(self, objs: list[_KalmanTrack], dets: list[np.ndarray], free: list[int]) -> list[int]:
        pairs = []
        for oi, o in enumerate(objs):
            for di in free:
                dist = float(np.linalg.norm(np.asarray(dets[di], np.float64) - o.estimate, axis=1).mean())
                if dist < self.distance_threshold:
                    pairs.append((dist, oi, di))
        pairs.sort()
        used_o, used_d = set(), set()
        for _, oi, di in pairs:
            if oi in used_o or di in used_d:
                continue
            used_o.add(oi)
            used_d.add(di)
            o = objs[oi]
            o.last_detection = di
            o.hit_counter = min(o.hit_counter + 2 * self.period, self.hit_counter_max)
            if o.is_initializing and o.hit_counter > self.initialization_delay:
                o.is_initializing = False
                o.id = self._next_id
                self._next_id += 1
            o.kalman_update(dets[di])
        return [di for di in free if di not in used_d]

    def update(self, detections: list[np.ndarray]) -> list[tuple[int, int]]:
        """detections: list of (P, 2) point sets -> [(track id, detection index)] for the initialised tracks that were
        matched in this frame, in track-id order (the reference skips tracks running on predictions, server.py:155-156)."""
        self.tracks = [t for t in self.tracks if t.hit_counter >= 0]
        for t in self.tracks:
            t.last_detection = -1
            t.step()
        free = list(range(len(detections)))
        free = self._match([t for t in self.tracks if not t.is_initializing], detections, free)
        free = self._match([t for t in self.tracks if t.is_initializing], detections, free)
        for di in free:
            t = _KalmanTrack(detections[di], self.period, self.initialization_delay)
            if not t.is_initializing:
                t.id = self._next_id
                self._next_id += 1
            t.last_detection = di
            self.tracks.append(t)
        out = [(t.id, t.last_detection) for t in self.tracks if t.id is not None and t.last_detection >= 0 and t.hit_counter >= 0]
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
        self.tracker = KalmanPointTracker(distance_threshold=300, hit_counter_max=5, initialization_delay=2)
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
