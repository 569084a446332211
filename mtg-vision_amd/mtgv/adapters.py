"""Drop-in classes with the reference's names, constructor defaults and methods, so that
`server.py` / `od_cam.py` / `qdrant_populate.py` / `encoder_validate.py` keep working when their three
imports point here:

    from mtgvision.encoder_export import CoreMlEncoder        -> mtgv.adapters.CoreMlEncoder
    from mtgvision.od_export import CardSegmenter             -> mtgv.adapters.CardSegmenter
    from mtgvision.qdrant import VectorStoreQdrant, QdrantPoint -> mtgv.adapters.VectorStoreQdrant, QdrantPoint

Everything numeric goes to the GPU library; these classes only translate types.
"""

from __future__ import annotations

from copy import deepcopy
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Iterable, Optional

import numpy as np
import torch

from . import spec
from .crop import mask_quads_from_logits, warp_quads
from .detector import Detector, binarize_masks, letterbox
from .encoder import Encoder
from .matcher import Matcher


# ---------------------------------------------------------------------------
# encoder: mtgvision/encoder_export.py:85-110
# ---------------------------------------------------------------------------
class CoreMlEncoder:
    """`CoreMlEncoder(model_path).predict(rgb_im) -> (768,)`.

    model_path: a Lightning `.ckpt` / a torch-saved state_dict (keys `model.encoder.*` or encoder
    keys), loaded with `weights_only=True`.  The reference's default (an absolute macOS path to a
    `.mlpackage`, encoder_export.py:23-27) has no meaning here; pass `encoder=` to wrap a ready
    `mtgv.Encoder`, or `state_dict=`."""

    def __init__(self, model_path: Optional[Path] = None, *, model_name: str = "cnvnxt2ae_nano", head_type: str = "conv+linear",
                 x_size_hw=(192, 128), state_dict=None, encoder: Optional[Encoder] = None, max_batch: int = 64):
        if encoder is not None:
            self.model = encoder
        elif state_dict is not None:
            self.model = Encoder(spec.encoder_config(model_name, x_size_hw, head_type), state_dict, max_batch=max_batch)
        else:
            if model_path is None:
                raise FileNotFoundError("CoreMlEncoder: no model_path given (the reference's hard-coded default path does not exist here)")
            self.model = Encoder.from_checkpoint(model_path, model_name, x_size_hw, head_type, max_batch=max_batch)

    def predict(self, rgb_im: np.ndarray):
        return self.model.predict(rgb_im)  # asserts ndim == 3, last dim 3; returns z[0] of a (1, z) result

    @property
    def input_hwc(self) -> tuple[int, int, int]:
        return self.model.input_hwc

    def ran_forward(self):
        return self.predict(np.random.rand(*self.input_hwc))


# ---------------------------------------------------------------------------
# detector: mtgvision/od_export.py:18-160
# ---------------------------------------------------------------------------
def _trace_blob(m: np.ndarray) -> np.ndarray:
    """Moore boundary trace of the single 8-connected blob in the padded boolean image `m`: (P, 2) array of (y, x)"""
    ys, xs = np.nonzero(m)
    start = (int(ys[0]), int(xs[ys == ys[0]].min()))
    nbrs = [(0, -1), (-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1)]  # clockwise from west
    pts = [start]
    cur, back = start, 0
    for _ in range(4 * m.size):
        found = False
        for i in range(8):
            d = (back + i) % 8
            ny, nx = cur[0] + nbrs[d][0], cur[1] + nbrs[d][1]
            if m[ny, nx]:
                back = (d + 5) % 8  # resume the scan just after the pixel we came from
                cur = (ny, nx)
                found = True
                break
        if not found or cur == start:
            break
        pts.append(cur)
    return np.asarray(pts, np.int64)


def _chain_approx_simple(pts: np.ndarray) -> np.ndarray:
    """cv2.CHAIN_APPROX_SIMPLE: drop the interior points of horizontal, vertical and diagonal runs of a closed chain"""
    n = len(pts)
    if n <= 2:
        return pts
    d_in = pts - np.roll(pts, 1, axis=0)
    d_out = np.roll(pts, -1, axis=0) - pts
    keep = np.any(d_in != d_out, axis=1)
    return pts[keep] if keep.any() else pts[:1]


def _mask_segments(mask: np.ndarray, strategy: str = "all") -> np.ndarray:
    """ultralytics `masks.xy` for one binary mask (`ops.masks2segments`, called behind od_export.py:152-153): the outer
    boundaries of its 8-connected blobs as cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) reports them - run end
    points only - and, with strategy "all" (ultralytics' default), every blob's points concatenated (blobs in raster
    order of their first pixel); "largest" keeps the contour with the most points.  (x, y) float32.
    cv2 / ultralytics are absent: parity unpinned; recent ultralytics additionally re-orders the concatenation so that
    the joined outline does not cross itself, which changes neither the point set nor its hull."""
    m = np.pad(np.asarray(mask).astype(bool), 1)
    if not m.any():
        return np.zeros((0, 2), np.float32)
    from scipy import ndimage

    lab, n = ndimage.label(m, structure=np.ones((3, 3), int))
    segs = [_chain_approx_simple(_trace_blob(lab == i)) for i in range(1, n + 1)]
    if strategy == "largest":
        segs = [segs[int(np.argmax([len(x) for x in segs]))]]
    elif strategy != "all":
        raise ValueError(f"unknown strategy {strategy!r}")
    p = np.concatenate(segs).astype(np.float32)
    return p[:, ::-1] - 1.0  # (x, y), undo the padding


def _outline_from_extents(ext: np.ndarray) -> np.ndarray:
    """The "outline" form of `InstanceSeg.points`: `ext` (rows, 2) holds the leftmost and rightmost mask pixel of every
    image row (-1 where the row is empty, as quads.hip reports them); returns the polygon left edge top to bottom, then
    right edge bottom to top, (x, y).  Notches that open to the left or right survive, notches that open up or down
    are filled (every row keeps only its extremes)."""
    rows = np.nonzero(ext[:, 0] >= 0)[0]
    left_edge = np.stack([ext[rows, 0], rows], 1)
    right_edge = np.stack([ext[rows[::-1], 1], rows[::-1]], 1)
    return np.concatenate([left_edge, right_edge])


def _largest_contour(mask: np.ndarray) -> np.ndarray:
    """Every boundary pixel (x, y) of the largest 8-connected blob of a binary mask, in trace order (test helper and
    the dense form of `_mask_segments(mask, "largest")`)."""
    m = np.pad(np.asarray(mask).astype(bool), 1)
    if not m.any():
        return np.zeros((0, 2), np.float32)
    from scipy import ndimage

    lab, n = ndimage.label(m, structure=np.ones((3, 3), int))
    if n > 1:
        sizes = ndimage.sum(m, lab, index=np.arange(1, n + 1))
        m = lab == (1 + int(np.argmax(sizes)))
    return _trace_blob(m).astype(np.float32)[:, ::-1] - 1.0


def _convex_hull(points: np.ndarray) -> np.ndarray:
    """Convex hull (monotone chain) of (P, 2) points, clockwise on the screen (y down), float64."""
    pts = sorted(set((float(x), float(y)) for x, y in np.asarray(points, np.float64)))
    if len(pts) <= 2:
        return np.asarray(pts, np.float64).reshape(-1, 2)

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower, upper = [], []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    for p in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    hull = np.asarray(lower[:-1] + upper[:-1], np.float64)
    x, y = hull[:, 0], hull[:, 1]
    if (x * np.roll(y, -1) - np.roll(x, -1) * y).sum() < 0:
        hull = hull[::-1]
    return hull


def _approx_poly_n(hull: np.ndarray, nsides: int = 4):
    """cv2.approxPolyN(points, nsides) as OpenCV documents it (od_export.py:74): on the convex hull, contract two
    vertices into one - prolong the two neighbours of an edge to their intersection - wherever that adds the least
    area, until `nsides` vertices remain.  Every vertex lies on the contour or outside it.  None when the hull has
    fewer vertices or nothing can be contracted."""
    v = [tuple(p) for p in np.asarray(hull, np.float64)]
    if len(v) < nsides:
        return None

    def added(a, b, c, d):
        rx, ry, qx, qy, ex, ey = b[0] - a[0], b[1] - a[1], c[0] - d[0], c[1] - d[1], c[0] - b[0], c[1] - b[1]
        den = rx * qy - ry * qx
        if den == 0.0:
            return np.inf, None
        t, u = (ex * qy - ey * qx) / den, (ex * ry - ey * rx) / den
        if not (t > 0.0 and u > 0.0):
            return np.inf, None
        p = (b[0] + t * rx, b[1] + t * ry)
        return 0.5 * abs((b[0] - p[0]) * (c[1] - p[1]) - (b[1] - p[1]) * (c[0] - p[0])), p

    while len(v) > nsides:
        n = len(v)
        cand = [added(v[i - 1], v[i], v[(i + 1) % n], v[(i + 2) % n]) for i in range(n)]
        i = int(np.argmin([c[0] for c in cand]))
        if not np.isfinite(cand[i][0]):
            return None
        v[i] = cand[i][1]
        del v[(i + 1) % n]
    return np.asarray(v, np.float64)


def _poly_centroid(p: np.ndarray) -> np.ndarray:
    """area centroid of a simple polygon (shoelace); falls back to the vertex mean for degenerate input"""
    p = np.asarray(p, np.float64)
    x, y = p[:, 0], p[:, 1]
    xn, yn = np.roll(x, -1), np.roll(y, -1)
    cr = x * yn - xn * y
    a = cr.sum() / 2.0
    if abs(a) < 1e-9:
        return p.mean(0)
    return np.asarray([((x + xn) * cr).sum(), ((y + yn) * cr).sum()]) / (6.0 * a)


@dataclass
class InstanceSeg:
    points: np.ndarray
    label: int
    conf: float

    # private
    _xyxyxyxy: np.ndarray = None
    _points_closed: np.ndarray = None
    _dir_vec: np.ndarray = None

    @property
    def scores(self) -> np.ndarray:
        return np.full_like(self.points[:, 0], self.conf)

    @property
    def center(self) -> np.ndarray:
        return np.mean(self.xyxyxyxy, axis=0)

    @property
    def points_closed(self) -> np.ndarray:
        self._orient()
        return self._points_closed

    @property
    def xyxyxyxy(self) -> np.ndarray:
        self._orient()
        return self._xyxyxyxy

    @property
    def dir_vec(self) -> np.ndarray:
        self._orient()
        return self._dir_vec

    def _orient(self, mode="u_shape") -> None:
        """od_export.py:52-93: the U-shaped mask is closed (shapely buffer(+d).buffer(-d)); v = centroid(orig) -
        centroid(closed) points at the card's top; cv2.approxPolyN(points, 4) gives a general 4-vertex polygon; the
        ray from its centroid along v picks the edge that becomes edge (0, 1); corners are truncated to int.
        shapely / cv2 are absent: the closed shape is the convex hull (the reference's own fallback, :63-64 - only the
        direction of v is used) and approxPolyN is restated from OpenCV's documentation (_approx_poly_n).  The GPU
        path (mtgv.crop.mask_quads / quads.hip) runs the same steps on the detection masks."""
        if self._xyxyxyxy is not None:
            return
        assert mode == "u_shape", "Only u_shape dataset mode is supported"
        pts = np.asarray(self.points, np.float64)
        hull = _convex_hull(pts)
        box = _approx_poly_n(hull, 4) if len(hull) >= 4 else None
        if box is None:
            x1, y1 = pts.min(0)
            x2, y2 = pts.max(0)
            box = np.asarray([[x1, y1], [x2, y1], [x2, y2], [x1, y2]], np.float64)
        # orig centroid - closed centroid (od_export.py:69-71): the card's bottom is missing from the mask,
        # so this vector points at the card's top edge
        v = _poly_centroid(pts) - (_poly_centroid(hull) if len(hull) >= 3 else pts.mean(0))
        nv = np.linalg.norm(v)
        v = v / nv if nv > 0 else np.asarray([0.0, -1.0])
        # od_export.py:76-88: first edge i in 1..3 that the ray centroid -> centroid + 1e7 v touches; else edge 0
        c = _poly_centroid(box)
        e = c + v * 10000000.0

        def touch(a, b, p, q):
            d1 = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
            d2 = (b[0] - a[0]) * (q[1] - a[1]) - (b[1] - a[1]) * (q[0] - a[0])
            d3 = (q[0] - p[0]) * (a[1] - p[1]) - (q[1] - p[1]) * (a[0] - p[0])
            d4 = (q[0] - p[0]) * (b[1] - p[1]) - (q[1] - p[1]) * (b[0] - p[0])
            return d1 * d2 <= 0.0 and d3 * d4 <= 0.0

        idx = 0
        for i in range(1, 4):
            if touch(c, e, box[i], box[(i + 1) % 4]):
                idx = i
                break
        box = np.roll(box, -idx, axis=0)
        self._xyxyxyxy = box.astype(int)
        self._points_closed = hull.astype(int)
        self._dir_vec = v

    def extract_dewarped(self, frame: np.ndarray, out_size_hw: tuple[int, int] = (192, 128), expand_ratio: float = 0.05) -> np.ndarray:
        """od_export.py:95-111 - perspective crop on the GPU (warp.hip)."""
        f = torch.from_numpy(np.ascontiguousarray(frame))[None].cuda()
        q = torch.from_numpy(np.asarray(self.xyxyxyxy).astype(np.float32))[None]
        out = warp_quads(f, q, torch.zeros(1, dtype=torch.int32), out_size_hw, expand_ratio)
        return out[0].cpu().numpy()

    def debug_draw_on(self, frame: np.ndarray, color=(128, 128, 128), id: str = None):
        raise NotImplementedError("debug drawing needs cv2; out of scope for the recognition path (SURVEY.md section 2 row 6)")


class CardSegmenter:
    """`CardSegmenter(model_path)(rgb_im) -> list[InstanceSeg]` (od_export.py:141-160).

    model_path: a torch-saved ultralytics state_dict (`model.<i>...` keys), loaded with
    `weights_only=True`; or pass `detector=` / `state_dict=`.  `.pt` pickles of whole ultralytics
    models are not loadable without the package (and are never unpickled here)."""

    def __init__(self, model_path: str | Path = None, *, state_dict=None, detector: Optional[Detector] = None, max_batch: int = 1,
                 contours="trace"):
        """contours: what `InstanceSeg.points` holds (the reference: ultralytics `masks.xy`, od_export.py:152-153).
        "trace" (default; True is accepted for it): the reference-shaped points - every 640 x 640 mask is copied to the
        host and its blobs are traced there like cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) (`_mask_segments`;
        scipy), notches of the reference's U-shaped card masks (od_export.py:57-60) included; the quad and the closed
        polygon are derived lazily on the host as `InstanceSeg._orient` does.
        "outline": the opt-in fast path - the mask's outline as the GPU's row extents (left edge top to bottom, right edge
        bottom to top, at most 2 x 640 points) with the oriented quad fitted on the GPU (quads.hip); one device-to-host
        copy of those integers per call, no mask leaves the device.  A row-extent outline FILLS any notch that opens
        upwards or downwards, so `points` / `points_closed` differ from the reference for upright and upside-down cards
        (tests/test_adapters_cpu.py pins what they hold); quads, direction vectors and crops do not.
        False: only the four GPU-fitted corners per card (they double as `points`) - for tracking loops."""
        if contours is True:
            contours = "trace"
        assert contours in ("outline", "trace", False), contours
        self.contours = contours
        if detector is not None:
            self.yolo = detector
        else:
            if state_dict is None:
                if model_path is None:
                    raise FileNotFoundError("CardSegmenter: no model_path given (the reference's hard-coded default path does not exist here)")
                state_dict = torch.load(model_path, map_location="cpu", weights_only=True)
                state_dict = state_dict.get("state_dict", state_dict) if isinstance(state_dict, dict) else state_dict
            self.yolo = Detector(spec.detector_config_for_state(state_dict), state_dict, max_batch=max_batch)  # v8n-seg or 11n-seg

    def __call__(self, rgb_im: np.ndarray) -> list[InstanceSeg]:
        img, ratio, (left, top) = letterbox(rgb_im, self.yolo.cfg.imgsz)
        det = self.yolo.detect(rgb_im)
        detections = []
        if det.mask_logits is None or det.conf.numel() == 0:
            return detections
        fh, fw = rgb_im.shape[0], rgb_im.shape[1]

        def to_frame(p):  # scale_coords + clip_coords back to the caller's frame
            p = (np.asarray(p, np.float64) - np.asarray([left, top], np.float64)) / float(ratio)
            p[..., 0] = np.clip(p[..., 0], 0, fw)
            p[..., 1] = np.clip(p[..., 1], 0, fh)
            return p

        if self.contours == "trace":
            masks = binarize_masks(det.mask_logits).cpu().numpy()
            for m, conf in zip(masks, det.conf.cpu().numpy()):
                pts = _mask_segments(m)  # masks.xy: all blobs' outlines, run end points
                if len(pts) == 0:
                    continue
                detections.append(InstanceSeg(points=to_frame(pts).astype(np.float32), label=0, conf=np.asarray(conf).tolist()))
            return detections
        # quads (and row extents) straight from the mask logits on the GPU; one small copy to the host
        want_ext = self.contours == "outline"
        res = mask_quads_from_logits(det.mask_logits, det.boxes_xyxy, extents=want_ext)
        quads, ok = to_frame(res[0].cpu().numpy()), res[1].cpu().numpy()
        ext = res[2].cpu().numpy() if want_ext else None
        for i, (q, good, conf) in enumerate(zip(quads, ok, det.conf.cpu().numpy())):
            if not good:
                continue
            if want_ext:
                pts = to_frame(_outline_from_extents(ext[i])).astype(np.float32)
            else:
                pts = q.astype(np.float32)
            seg = InstanceSeg(points=pts, label=0, conf=np.asarray(conf).tolist())
            top_mid, centre = (q[0] + q[1]) / 2, q.mean(0)
            v = top_mid - centre
            seg._xyxyxyxy = q.astype(int)
            seg._points_closed = pts.astype(int) if want_ext else q.astype(int)
            seg._dir_vec = v / (np.linalg.norm(v) or 1.0)
            detections.append(seg)
        return detections


# ---------------------------------------------------------------------------
# match: mtgvision/qdrant.py:10-111
# ---------------------------------------------------------------------------
@dataclass
class QdrantPoint:
    id: str  # UUID
    vector: list[float] | None = None
    payload: dict[str, Any] | None = None


@dataclass
class ScoredPoint:
    """the fields callers read from qdrant's ScoredPoint (server.py:61-70, :192; encoder_validate.py:92)"""

    id: str
    version: int
    score: float
    payload: dict[str, Any] | None = None
    vector: list[float] | None = None


class VectorStoreQdrant:
    _COLLECTION = "mtg"
    _VECTOR_SIZE: int = 768

    def __init__(self, location: str = "localhost:6333", *, capacity: int = 131072):
        # `location` is accepted for signature compatibility; the bank lives in this GPU's HBM
        self.location = location
        self._capacity = capacity
        self._bank = Matcher(self._VECTOR_SIZE, capacity=capacity)
        self._ids: list[str] = []
        self._row: dict[str, int] = {}
        self._payload: dict[str, dict | None] = {}

    def drop_collection(self):
        self._bank.clear()
        self._ids, self._row, self._payload = [], {}, {}

    def retrieve(self, ids: Iterable[str], *, with_payload: bool = True, with_vectors: bool = False) -> list[QdrantPoint]:
        out = []
        for i in ids:
            i = str(i)
            if i not in self._row:
                continue  # Qdrant returns only the points that exist
            vec = self._bank.rows(self._row[i], 1)[0].tolist() if with_vectors else None
            out.append(QdrantPoint(id=i, vector=vec, payload=deepcopy(self._payload.get(i)) if with_payload else None))
        return out

    def save_points(self, iter_points: Iterable[QdrantPoint]):
        new_vecs, new_ids = [], []
        for p in iter_points:
            v = np.asarray(p.vector, np.float32)
            assert v.shape == (self._VECTOR_SIZE,), f"{v.shape}"
            pid = str(p.id)
            if pid in self._row:  # upsert
                self._bank.set_row(self._row[pid], v)
            else:
                new_vecs.append(v)
                new_ids.append(pid)
            self._payload[pid] = deepcopy(p.payload)
            if len(new_vecs) == 64:  # qdrant.py:73 uploads in batches of 64
                self._flush(new_vecs, new_ids)
                new_vecs, new_ids = [], []
        self._flush(new_vecs, new_ids)

    def _flush(self, vecs, ids):
        if not vecs:
            return
        rows = self._bank.add(np.stack(vecs))
        for r, pid in zip(rows, ids):
            self._row[pid] = r
            self._ids.append(pid)

    def query_nearby(self, vector: list[float], k: int, *, with_payload: bool = True, with_vectors: bool = False,
                     score_threshold: float = None) -> list[ScoredPoint]:
        return self.query_nearby_batch([vector], k, with_payload=with_payload, with_vectors=with_vectors, score_threshold=score_threshold)[0]

    def query_nearby_batch(self, vectors, k: int, *, with_payload: bool = True, with_vectors: bool = False,
                           score_threshold: float = None) -> list[list[ScoredPoint]]:
        """`query_nearby` for several query vectors in one GPU pass (one list of hits per query)."""
        vectors = np.asarray(vectors, np.float32).reshape(-1, self._VECTOR_SIZE)
        if len(self._bank) == 0 or k <= 0 or len(vectors) == 0:
            return [[] for _ in range(len(vectors))]
        kk = min(int(k), len(self._bank))
        ids, scores = self._bank.match(vectors, kk, threshold=score_threshold)  # score_threshold applied on the device
        ids, scores = ids.cpu().numpy(), scores.cpu().numpy()
        res = []
        for row_ids, row_scores in zip(ids, scores):
            out = []
            for i, s in zip(row_ids, row_scores):
                if i < 0:
                    continue
                pid = self._ids[int(i)]
                out.append(
                    ScoredPoint(
                        id=pid, version=0, score=float(s),
                        payload=deepcopy(self._payload.get(pid)) if with_payload else None,
                        vector=self._bank.rows(int(i), 1)[0].tolist() if with_vectors else None,
                    )
                )
            res.append(out)
        return res

    def update_payload(self, id_: str, payload: dict[str, Any]) -> QdrantPoint:
        self._payload[str(id_)] = deepcopy(payload)
        return QdrantPoint(id=id_, vector=None, payload=deepcopy(payload))
