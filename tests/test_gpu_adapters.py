"""Drop-in classes (reference names / signatures) on the GPU."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_vector_store_contract():
    """mtgvision/qdrant.py:17-111 semantics: cosine score, top-k sorted desc, threshold, upsert, payloads."""
    from mtgv.adapters import QdrantPoint, VectorStoreQdrant
    from oracle import match_ref as M

    rng = np.random.default_rng(5)
    vecs = rng.standard_normal((300, 768)).astype(np.float32)
    ids = [f"00000000-0000-0000-0000-{i:012d}" for i in range(300)]
    db = VectorStoreQdrant(capacity=512)
    assert db._COLLECTION == "mtg" and db._VECTOR_SIZE == 768
    db.save_points(QdrantPoint(id=i, vector=v.tolist(), payload={"n": j}) for j, (i, v) in enumerate(zip(ids, vecs)))
    q = vecs[17] + 0.05 * rng.standard_normal(768).astype(np.float32)
    res = db.query_nearby(q.tolist(), k=3, with_payload=True)
    rid, rsc = M.cosine_topk(q[None], vecs, 3)
    assert [p.id for p in res] == [ids[i] for i in rid[0]]
    assert res[0].id == ids[17] and res[0].payload == {"n": 17} and res[0].vector is None
    np.testing.assert_allclose([p.score for p in res], rsc[0], atol=2e-6)
    thr = float(rsc[0][1]) - 1e-4
    assert len(db.query_nearby(q, k=3, score_threshold=thr)) == 2
    big = db.query_nearby(q, k=3000, with_payload=False, score_threshold=0.1)  # qdrant.py:136-142 debug call
    s64 = M.scores(q[None], vecs)[0]
    assert len(big) == int((s64 >= 0.1).sum()) and all(p.payload is None for p in big)
    # retrieve / update_payload / upsert
    [pt] = db.retrieve([ids[5]], with_payload=True, with_vectors=True)
    np.testing.assert_allclose(pt.vector, vecs[5] / np.linalg.norm(vecs[5]), atol=1e-6)
    assert db.retrieve(["missing"]) == []
    assert db.update_payload(ids[5], {"name": "x"}).payload == {"name": "x"}
    assert db.retrieve([ids[5]])[0].payload == {"name": "x"}
    db.save_points([QdrantPoint(id=ids[3], vector=(-vecs[17]).tolist(), payload=None)])
    assert db.query_nearby(-q, k=1)[0].id == ids[3]
    db.drop_collection()
    assert db.query_nearby(q, k=3) == []


def test_coreml_encoder_contract():
    from mtgv import spec
    from mtgv.adapters import CoreMlEncoder
    from oracle import encoder_ref as R

    cfg = spec.encoder_config("cnvnxt2ae_nano", (192, 128), "conv+linear")
    sd = spec.random_encoder_state(cfg, 1)
    enc = CoreMlEncoder(state_dict=sd, max_batch=2)
    assert enc.input_hwc == (192, 128, 3)
    im = np.random.default_rng(3).random((192, 128, 3))  # float64 in [0,1], like ran_forward()
    z = enc.predict(im)
    assert z.shape == (768,) and z.dtype == np.float32
    assert np.abs(z - R.predict_hwc(sd, cfg, im)).max() < 1e-4
    assert enc.ran_forward().shape == (768,)
    with pytest.raises(FileNotFoundError):
        CoreMlEncoder()


def test_card_segmenter_and_mask_binarize():
    from mtgv import spec
    from mtgv.adapters import CardSegmenter, InstanceSeg
    from mtgv.detector import Detector, binarize_masks
    from oracle import detector_ref as D

    cfg = spec.DetectorConfig()
    sd = spec.random_detector_state(cfg, 3)
    det = Detector(cfg, sd, max_batch=1)
    frame = np.random.default_rng(8).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    d = det.detect(frame)
    mb = binarize_masks(d.mask_logits[:16]).cpu().numpy()
    ref = D.masks_binary(d.mask_logits[:16].cpu().numpy())
    assert mb.shape == (16, 640, 640)
    assert (mb.astype(bool) != ref).mean() < 1e-5  # identical up to bilinear rounding exactly at 0
    segs = CardSegmenter(detector=det)(frame)
    assert isinstance(segs, list) and len(segs) > 0 and all(isinstance(s, InstanceSeg) for s in segs)
    s = segs[0]
    assert s.label == 0 and 0.25 < s.conf <= 1.0 and s.points.ndim == 2 and s.points.shape[1] == 2
    assert s.xyxyxyxy.shape == (4, 2) and s.xyxyxyxy.dtype.kind == "i"
    crop = s.extract_dewarped(frame)
    assert crop.shape == (192, 128, 3) and crop.dtype == np.uint8




@pytest.fixture
def _precision_guard():
    from mtgv import native

    before = native.get_gemm_precision()
    yield native
    native.set_gemm_precision(before)


@pytest.mark.parametrize("mode", ["f32", "f16x3"])
def test_pipeline_overlap_equals_sequential(mode, _precision_guard, monkeypatch):
    """run_many (two HIP streams, detect of batch i+1 beside embed of batch i) == run on each batch, bit for bit,
    with either GEMM operand precision"""
    from mtgv import spec

    _precision_guard.set_gemm_precision(mode)
    from mtgv.detector import Detector
    from mtgv.encoder import Encoder
    from mtgv.matcher import Matcher
    from mtgv.pipeline import Pipeline

    assert not Pipeline.overlap_enabled() or "MTGV_OVERLAP" in os.environ  # opt-in (foreign packed-FP32 kernels)
    monkeypatch.setenv("MTGV_OVERLAP", "on")
    assert Pipeline.overlap_enabled()
    det_cfg = spec.DetectorConfig()
    enc_cfg = spec.encoder_config("cnvnxt2ae_nano")
    F, K = 2, 4
    m = Matcher(768, capacity=3000)
    m.add(np.random.default_rng(2).standard_normal((3000, 768)).astype(np.float32))
    pipe = Pipeline(Detector(det_cfg, spec.random_detector_state(det_cfg, 3), max_batch=F),
                    Encoder(enc_cfg, spec.random_encoder_state(enc_cfg, 1), max_batch=F * K), m, K, 3)
    g = torch.Generator(device="cuda").manual_seed(11)
    batches = [torch.randint(0, 256, (F, 640, 640, 3), generator=g, device="cuda", dtype=torch.uint8) for _ in range(3)]
    seq = [pipe.run(b) for b in batches]
    ovl = pipe.run_many(batches)
    torch.cuda.synchronize()
    assert len(ovl) == 3
    for a, b in zip(seq, ovl):
        for k in ("ids", "scores", "z", "crops", "boxes", "n_det"):
            assert torch.equal(a[k], b[k]), k


def test_tracker_ctx_end_to_end():
    """server.py's loop on the GPU stages: one batched embed + one batched query per frame == per-track calls"""
    from mtgv import spec
    from mtgv.adapters import CardSegmenter, CoreMlEncoder, QdrantPoint, VectorStoreQdrant
    from mtgv.detector import Detector
    from mtgv.tracker import TrackerCtx

    cfg = spec.DetectorConfig()
    seg = CardSegmenter(detector=Detector(cfg, spec.random_detector_state(cfg, 3), max_batch=1))
    enc_cfg = spec.encoder_config("cnvnxt2ae_nano", (192, 128), "conv+linear")
    enc = CoreMlEncoder(state_dict=spec.random_encoder_state(enc_cfg, 1), max_batch=16)
    rng = np.random.default_rng(5)
    db = VectorStoreQdrant(capacity=256)
    db.save_points(QdrantPoint(id=f"id-{i}", vector=v.tolist(), payload={"n": i}) for i, v in enumerate(rng.standard_normal((200, 768)).astype(np.float32)))
    frame = np.random.default_rng(8).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    now = [10.0]
    ctx = TrackerCtx(0.5, 0.1, segmenter=seg, encoder=enc.model, vecs=db, clock=lambda: now[0])
    assert ctx.update(frame) == [] and ctx.update(frame) == []
    objs = ctx.update(frame)
    assert len(objs) > 0
    for o in objs[:4]:
        z = enc.predict(o.last_rgb_im)
        assert np.abs(o.avg_z - z).max() < 1e-5
        want = db.query_nearby(o.avg_z, k=3)
        assert [p.id for p in o.ave_nearby_points] == [p.id for p in want]
        d = o.to_dict()
        assert len(d["matches"]) == 3 and d["matches"][0]["all_data"] == want[0].payload and len(d["points"]) == 4


def test_card_segmenter_gpu_quads_fast_path():
    """contours=False: quads fitted on the GPU, same cards and (up to a pixel or two) the same corners as the host path"""
    from mtgv import spec
    from mtgv.adapters import CardSegmenter
    from mtgv.detector import Detector

    cfg = spec.DetectorConfig()
    det = Detector(cfg, spec.random_detector_state(cfg, 3), max_batch=1)
    frame = np.random.default_rng(8).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    slow = CardSegmenter(detector=det, contours="trace")(frame)
    fast = CardSegmenter(detector=det, contours=False)(frame)
    assert len(fast) > 0 and len(fast) <= len(slow) + 0
    assert all(s.points.shape == (4, 2) and s.xyxyxyxy.shape == (4, 2) and s.xyxyxyxy.dtype.kind == "i" for s in fast)
    assert fast[0].extract_dewarped(frame).shape == (192, 128, 3)
    assert abs(float(np.linalg.norm(fast[0].dir_vec)) - 1.0) < 1e-6
    # random-weight masks are speckle, so only compare where the host path saw one blob: same corner set within 2 px
    by_conf = {round(s.conf, 6): s for s in slow}
    close = 0
    for s in fast:
        h = by_conf.get(round(s.conf, 6))
        if h is None:
            continue
        a = np.sort(np.asarray(s.xyxyxyxy, float), axis=0)
        b = np.sort(np.asarray(h.xyxyxyxy, float), axis=0)
        close += int(np.abs(a - b).max() <= 2.0)
    print(f"{len(fast)} fast / {len(slow)} host segments, {close} with the same corners")


def _card_frame_mask(quad, size=640):
    """binary mask of a convex quadrilateral with a bite out of its bottom edge (the reference's U-shaped card masks)"""
    yy, xx = np.mgrid[0:size, 0:size]
    m = np.ones((size, size), bool)
    q = np.asarray(quad, np.float64)
    for i in range(4):
        a, b = q[i], q[(i + 1) % 4]
        m &= (b[0] - a[0]) * (yy - a[1]) - (b[1] - a[1]) * (xx - a[0]) >= 0
    c = (q[2] + q[3]) / 2
    m &= ~(((xx - c[0]) ** 2 + (yy - c[1]) ** 2) < 30**2)
    return m


def test_card_segmenter_outline_from_the_device(monkeypatch):
    """opt-in fast path (contours="outline"): points = the mask's row-extent outline computed on the GPU, quad fitted on the
    GPU; against the host path that copies the masks and traces them (contours="trace") on card-shaped masks: same cards,
    corners within 1.5 px, and the outline is a polygon of at most 2 x 640 points that covers the mask."""
    import mtgv.adapters as A
    from mtgv import spec
    from mtgv.detector import Detections, Detector

    cfg = spec.DetectorConfig()
    det = Detector(cfg, spec.random_detector_state(cfg, 3), max_batch=1)
    frame = np.zeros((640, 640, 3), np.uint8)
    quads = [[(120, 80), (330, 110), (300, 420), (90, 380)], [(400, 300), (560, 290), (590, 520), (420, 540)]]
    masks = np.stack([_card_frame_mask(q) for q in quads])
    # mask logits on the 160 x 160 grid whose x4 bilinear interpolation crosses zero on the shapes' boundaries
    big = torch.from_numpy(np.where(masks, 4.0, -4.0).astype(np.float32))[:, None]
    logits = torch.nn.functional.avg_pool2d(big, 4)[:, 0].cuda().contiguous()
    boxes = torch.tensor([[90.0, 80.0, 330.0, 420.0], [400.0, 290.0, 590.0, 540.0]], device="cuda")
    fake = Detections(boxes, torch.tensor([0.9, 0.8], device="cuda"), torch.zeros(2, dtype=torch.int64, device="cuda"),
                      torch.zeros(2, dtype=torch.int64, device="cuda"), logits)
    monkeypatch.setattr(det, "detect", lambda *a, **k: fake)
    seg_dev = A.CardSegmenter(detector=det, contours="outline")(frame)
    seg_host = A.CardSegmenter(detector=det)(frame)  # the default: reference-shaped traced points
    assert A.CardSegmenter(detector=det).contours == "trace"
    assert len(seg_dev) == len(seg_host) == 2
    for d, h, q in zip(seg_dev, seg_host, quads):
        assert d.points.ndim == 2 and d.points.shape[1] == 2 and 8 < d.points.shape[0] <= 2 * 640
        a, b = np.asarray(d.xyxyxyxy, float), np.asarray(h.xyxyxyxy, float)
        assert np.abs(a - b).max() <= 1.5, (a, b)                      # same corners, same order (corner 0 = top-left of the card)
        assert np.abs(a - np.asarray(q, float)).max() <= 6.0, (a, q)    # and they are the card's corners (mask grid: 4 px)
        # the device outline and the traced outline describe the same region: equal hulls up to a pixel
        area = lambda p: 0.5 * abs(np.sum(p[:, 0] * np.roll(p[:, 1], -1) - np.roll(p[:, 0], -1) * p[:, 1]))  # noqa: E731
        hd, hh = A._convex_hull(np.asarray(d.points, np.float64)), A._convex_hull(np.asarray(h.points, np.float64))
        assert abs(area(hd) - area(hh)) <= 0.01 * area(hh)
        assert d.extract_dewarped(frame).shape == (192, 128, 3)
