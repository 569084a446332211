"""Host logic of the temporal loop (mtgv/tracker.py) with stand-in stages: association, initialisation delay,
EWMA of the embeddings, re-embed gating, batched queries, wire format."""
import json

import numpy as np

from mtgv.tracker import MeanEuclideanTracker, TrackedData, TrackerCtx, encode_rgb_im, get_color


class _Seg:
    def __init__(self, quad, conf=0.9):
        self.xyxyxyxy = np.asarray(quad, int)
        self.points = self.xyxyxyxy.astype(np.float32)
        self.points_closed = self.xyxyxyxy
        self.conf = conf

    def extract_dewarped(self, frame, out_size_hw=(192, 128), expand_ratio=0.05):
        return np.full((192, 128, 3), int(self.xyxyxyxy[0, 0]) % 256, np.uint8)


class _Enc:
    calls = 0

    def predict(self, im):
        _Enc.calls += 1
        z = np.zeros(8, np.float32)
        z[int(im[0, 0, 0]) % 8] = 1.0
        return z


class _Pt:
    def __init__(self, i, s):
        self.id, self.score, self.payload = f"card-{i}", s, {"k": i}


class _Vecs:
    def __init__(self):
        self.batches = []

    def query_nearby_batch(self, vectors, k, *, with_payload=True, with_vectors=False, score_threshold=None):
        self.batches.append(len(vectors))
        return [[_Pt(int(np.argmax(v)), float(np.max(v))) for _ in range(k)] for v in vectors]


def _quad(x, y, w=50, h=70):
    return [[x, y], [x + w, y], [x + w, y + h], [x, y + h]]


def test_association_and_initialisation_delay():
    t = MeanEuclideanTracker(distance_threshold=300, hit_counter_max=5, initialization_delay=2)
    a, b = np.asarray(_quad(10, 10), float), np.asarray(_quad(400, 300), float)
    assert t.update([a, b]) == []                      # first sighting: not initialised
    assert t.update([a + 2, b + 3]) == []              # second hit
    assert t.update([b + 5, a + 4]) == [(1, 1), (2, 0)]  # third hit: ids in creation order, detection order swapped
    assert t.update([a + 6]) == [(1, 0)]               # b missed: still alive
    for _ in range(6):                                 # hit counter of b runs out
        t.update([a + 6])
    assert len(t.tracks) == 1
    assert t.update([a + 6, b]) == [(1, 0)] and len(t.tracks) == 2  # b returns as a new, uninitialised track
    far = np.asarray(_quad(10, 10), float) + 1000
    assert t.update([far]) == []                       # beyond the distance threshold: a new track


def test_ewma_gating_batched_queries_and_wire_format():
    now = [100.0]
    frames = [[_Seg(_quad(10, 10)), _Seg(_quad(300, 200))]] * 12
    it = iter(frames)
    vecs = _Vecs()
    _Enc.calls = 0
    ctx = TrackerCtx(0.5, 0.1, segmenter=lambda f: next(it), encoder=_Enc(), vecs=vecs, clock=lambda: now[0])
    frame = np.zeros((480, 640, 3), np.uint8)
    assert ctx.update(frame) == [] and ctx.update(frame) == []
    objs = ctx.update(frame)                           # tracks initialised: first embedding of both, one batch
    assert [o.id for o in objs] == [1, 2] and vecs.batches == [2] and _Enc.calls == 2
    z0 = objs[0].avg_z.copy()
    assert np.allclose(z0, 0.1 * z0 / z0.max() + 0.9 * z0 / z0.max())  # first update: avg_z = z
    now[0] += 0.2
    ctx.update(frame)
    assert _Enc.calls == 2                             # inside update_wait_sec: no re-embed
    now[0] += 0.4
    objs = ctx.update(frame)
    assert _Enc.calls == 4 and vecs.batches == [2, 2]  # both due again
    assert np.allclose(objs[0].avg_z, 0.1 * z0 + 0.9 * z0)
    d = objs[0].to_dict()
    json.dumps(d)
    assert d["id"] == "1" and d["color"] == get_color(1) and len(d["matches"]) == 3 and d["matches"][0]["id"].startswith("card-")
    assert d["points"] == _quad(10, 10) and (d["img"] is None or isinstance(d["img"], str))
    assert isinstance(objs[0], TrackedData)


def test_color_and_thumbnail():
    assert get_color(1) == get_color("1") and get_color(1) != get_color(2) and len(get_color(7)) == 7
    s = encode_rgb_im(np.zeros((8, 8, 3), np.uint8))
    assert s is None or isinstance(s, str)


def test_kalman_track_equals_matrix_filter():
    """the per-coordinate closed form is the textbook filter with F = [[1,1],[0,1]], Q = q I, H = [1,0], R = r"""
    from mtgv.tracker import _KalmanTrack

    rng = np.random.default_rng(0)
    z0 = rng.uniform(0, 500, (4, 2))
    t = _KalmanTrack(z0, 1, 2)
    F = np.asarray([[1.0, 1.0], [0.0, 1.0]])
    x = np.stack([z0.reshape(-1), np.zeros(8)])          # (2, 8): one filter per coordinate
    P = np.tile(np.asarray([[10.0, 0.0], [0.0, 1.0]])[:, :, None], (1, 1, 8))
    for step in range(12):
        z = z0 + 7.0 * (step + 1) + rng.normal(0, 2, (4, 2))
        t.step()
        t.kalman_update(z)
        x = F @ x
        for c in range(8):
            Pp = F @ P[:, :, c] @ F.T + 0.1 * np.eye(2)
            S = Pp[0, 0] + 4.0
            K = Pp[:, 0] / S
            x[:, c] = x[:, c] + K * (z.reshape(-1)[c] - x[0, c])
            P[:, :, c] = Pp - np.outer(K, Pp[0, :])
        assert np.allclose(t.pos, x[0], atol=1e-9) and np.allclose(t.vel, x[1], atol=1e-9)
        assert np.allclose(t.pp, P[0, 0], atol=1e-9) and np.allclose(t.pv, P[0, 1], atol=1e-9) and np.allclose(t.vv, P[1, 1], atol=1e-9)


def test_kalman_tracker_counters_prediction_and_ids():
    from mtgv.tracker import KalmanPointTracker

    sq = np.asarray([[0, 0], [100, 0], [100, 140], [0, 140]], np.float64)
    trk = KalmanPointTracker(distance_threshold=300, hit_counter_max=5, initialization_delay=2)
    # ids after the third consecutive detection (counter 1 -> 2 -> 3 > delay), as norfair with these settings
    assert trk.update([sq]) == [] and trk.update([sq + 10]) == []
    assert trk.update([sq + 20]) == [(1, 0)]
    # a second card far away gets its own id; association follows the predicted (moving) positions, not the last ones
    far = sq + 1000
    for k in range(3, 9):
        out = trk.update([far, sq + 10 * k])
    assert sorted(out) == [(1, 1), (2, 0)]
    est = trk.tracks[0].estimate
    assert np.abs(est - (sq + 80)).max() < 8.0 and np.all(trk.tracks[0].vel > 5.0)  # velocity learnt (10 px / frame)
    # misses: the counter (5 at most) drops by one per frame; the track survives exactly hit_counter_max + 1 empty frames
    for k in range(5):
        assert trk.update([]) == []
        assert any(t.id == 1 for t in trk.tracks)
    trk.update([])
    trk.update([])
    assert not any(t.id == 1 for t in trk.tracks)
    # a card crossing during the gap re-enters as a new object with a new id
    for _ in range(3):
        out = trk.update([sq])
    assert out == [(3, 0)]
