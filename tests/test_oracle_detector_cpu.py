"""CPU checks of the detector / crop oracles and of the host-side letterbox (known answers, no GPU).
Upstream parity for these stages is unpinned (third-party ultralytics / cv2 absent); these tests pin the
oracles' own definitions so that the GPU parity tests mean something."""
import numpy as np

from mtgv import spec
from oracle import detector_ref as D
from oracle import warp_ref as W


def _pred(boxes_xywh, scores, nc=3, nm=2, na=16):
    p = np.zeros((4 + nc + nm, na), np.float32)
    for i, (b, s) in enumerate(zip(boxes_xywh, scores)):
        p[:4, i] = b
        p[4 : 4 + nc, i] = s
    return p


def test_nms_known_answers():
    # anchors 0,1 overlap heavily (same class) -> lower score suppressed; anchor 2 same box but other class -> kept
    # (class offset); anchor 3 below conf; anchor 4 far away
    boxes = [(100, 100, 50, 50), (102, 101, 50, 50), (100, 100, 50, 50), (300, 300, 40, 40), (400, 100, 30, 60)]
    scores = [(0.9, 0.1, 0.0), (0.8, 0.0, 0.0), (0.1, 0.7, 0.0), (0.2, 0.1, 0.0), (0.0, 0.0, 0.6)]
    d = D.nms_single(_pred(boxes, scores), 3)
    assert d["keep_idx"].tolist() == [0, 2, 4]
    assert d["cls"].tolist() == [0, 1, 2]
    np.testing.assert_allclose(d["conf"], [0.9, 0.7, 0.6])
    np.testing.assert_allclose(d["boxes"][0], [75, 75, 125, 125])
    # exact ties keep anchor order; max_det truncates
    boxes = [(50 + 100 * i, 50, 20, 20) for i in range(5)]
    d = D.nms_single(_pred(boxes, [(0.5, 0, 0)] * 5), 3, max_det=3)
    assert d["keep_idx"].tolist() == [0, 1, 2]
    # IoU exactly at the threshold is NOT suppressed (strict >): boxes [0,10]x[0,10] and [0,10]x[3,13] -> 7/13 = 0.538
    d = D.nms_single(_pred([(5, 5, 10, 10), (5, 8, 10, 10)], [(0.9, 0, 0), (0.8, 0, 0)]), 3, iou_thres=7.0 / 13.0)
    assert len(d["keep_idx"]) == 2
    d = D.nms_single(_pred([(5, 5, 10, 10), (5, 8, 10, 10)], [(0.9, 0, 0), (0.8, 0, 0)]), 3, iou_thres=0.53)
    assert len(d["keep_idx"]) == 1
    assert len(D.nms_single(_pred([], []), 3)["keep_idx"]) == 0


def test_anchors_and_shapes():
    cfg = spec.DetectorConfig()
    a, s = D.make_anchors(cfg)
    assert a.shape == (2, 8400) and s.shape == (1, 8400)
    assert a[:, 0].tolist() == [0.5, 0.5] and a[:, 1].tolist() == [1.5, 0.5] and a[:, 80].tolist() == [0.5, 1.5]
    assert s[0, 0] == 8 and s[0, 6400] == 16 and s[0, 8000] == 32
    shapes = spec.detector_param_shapes(cfg)
    assert shapes["model.22.proto.upsample.weight"] == (64, 64, 2, 2) and shapes["model.22.cv3.0.2.weight"] == (3, 64, 1, 1)
    assert shapes["model.0.conv.weight"] == (16, 3, 3, 3) and shapes["model.9.cv2.conv.weight"] == (256, 512, 1, 1)
    assert sum(int(np.prod(v)) for v in shapes.values()) == 3275305


def test_mask_crop_and_binarize():
    nc = 3
    pred = np.zeros((4 + nc + 32, 8), np.float32)
    pred[4 + nc, :] = 1.0  # coefficient 0 = 1 -> mask = proto 0
    protos = np.zeros((32, 160, 160), np.float32)
    protos[0] = 1.0
    det = {"keep_idx": np.array([0], np.int32), "boxes": np.array([[40.0, 80.0, 120.0, 240.0]], np.float32)}
    m = D.mask_logits(pred, protos, det, nc)
    assert m.shape == (1, 160, 160) and m[0, 20:60, 10:30].min() == 1.0  # y in [20,60), x in [10,30) at 1/4 scale
    assert m[0, :20].max() == 0 and m[0, 60:].max() == 0 and m[0, :, :10].max() == 0 and m[0, :, 30:].max() == 0
    b = D.masks_binary(m)
    assert b.shape == (1, 640, 640) and b[0, 100:220, 60:100].all() and not b[0, :70].any()


def test_warp_identity_and_border():
    fr = np.random.default_rng(0).integers(0, 256, (60, 80, 3), dtype=np.uint8)
    q = np.array([[10, 5], [42, 5], [42, 53], [10, 53]], np.float32)
    out = W.warp_quad(fr, q, (48, 32), 0.0)
    assert (out == fr[5:53, 10:42]).all()
    out = W.warp_quad(fr, q - 100, (48, 32), 0.0)  # entirely outside: constant border 0
    assert (out == 0).all()
    # a degenerate quad gives a singular system: all-zero homography, every pixel samples frame[0, 0]
    assert (W.warp_quad(fr, np.zeros((4, 2), np.float32), (48, 32)) == fr[0, 0]).all()


def test_letterbox_host():
    from mtgv.detector import letterbox

    img, r, (left, top) = letterbox(np.full((480, 640, 3), 7, np.uint8))
    assert img.shape == (640, 640, 3) and r == 1.0 and (left, top) == (0, 80)
    assert (img[80:560] == 7).all() and (img[:80] == 114).all()
    img, r, (left, top) = letterbox(np.full((1280, 640, 3), 9, np.uint8))
    assert r == 0.5 and (left, top) == (160, 0) and (img[:, 160:480] == 9).all() and (img[:, :160] == 114).all()


def test_resize_oracle_against_brute_force():
    """oracle/resize_ref.make_cropped == direct evaluation of the area integral (small case, pure Python loops)"""
    from math import ceil

    from oracle import resize_ref as R

    im = np.random.default_rng(1).integers(0, 256, (50, 37, 3), dtype=np.uint8)
    got = R.make_cropped(im, (12, 8))
    H, W = im.shape[:2]
    bw = ceil(max(0.02 * H, 0.02 * W))
    c = im[bw : H - bw, bw : W - bw].astype(float)
    sy, sx = c.shape[0] / 12, c.shape[1] / 8
    ref = np.zeros((12, 8, 3))
    for oy in range(12):
        for ox in range(8):
            for y in range(c.shape[0]):
                wy = max(0.0, min(y + 1, (oy + 1) * sy) - max(y, oy * sy))
                for x in range(c.shape[1]):
                    wx = max(0.0, min(x + 1, (ox + 1) * sx) - max(x, ox * sx))
                    ref[oy, ox] += wy * wx * c[y, x]
    ref /= sx * sy * 255
    assert np.abs(got - ref).max() < 1e-6
    # integer scale = plain box mean; identity size = copy
    im = np.random.default_rng(2).integers(0, 256, (2 * 192 + 16, 2 * 128 + 16, 3), dtype=np.uint8)
    got = R.make_cropped(im, (192, 128))
    box = im[8:-8, 8:-8].astype(np.float64).reshape(192, 2, 128, 2, 3).mean((1, 3)) / 255
    assert np.abs(got - box).max() < 1e-6


def test_yolo11_keys_shapes_and_oracle_forward():
    """YOLO11n-seg (od_train.py:20, :55-56): key table, parameter count of the published order, finite oracle outputs"""
    import torch

    from mtgv import spec
    from oracle import detector_ref as D

    cfg = spec.yolo11_config()
    assert cfg.arch == "11" and cfg.head_index == 23 and cfg.rep(2) == 1
    shapes = spec.detector_param_shapes(cfg)
    nparam = sum(int(np.prod(s)) for s in shapes.values())
    assert 2.6e6 < nparam < 3.1e6  # published yolo11n-seg: 2.9 M parameters (80 classes)
    for k in ("model.2.m.0.cv1.conv.weight", "model.6.m.0.m.1.cv2.conv.weight", "model.10.m.0.attn.qkv.conv.weight",
              "model.10.m.0.attn.pe.conv.weight", "model.10.m.0.ffn.1.conv.weight", "model.23.cv3.0.0.0.conv.weight",
              "model.23.cv3.2.1.1.conv.weight", "model.23.proto.upsample.weight"):
        assert k in shapes, k
    assert shapes["model.10.m.0.attn.qkv.conv.weight"] == (256, 128, 1, 1) and shapes["model.10.m.0.attn.pe.conv.weight"] == (128, 1, 3, 3)
    assert shapes["model.2.m.0.cv1.conv.weight"] == (8, 16, 3, 3) and shapes["model.23.cv3.1.0.0.conv.weight"] == (128, 1, 3, 3)
    # the YOLOv8 table is unchanged by the generalisation
    assert len(spec.detector_param_shapes(spec.DetectorConfig())) == 351
    sd = spec.random_detector_state(cfg, 3)
    fr = np.random.default_rng(4).integers(0, 256, (1, 64, 64, 3), dtype=np.uint8)
    small = spec.yolo11_config(imgsz=64)
    pred, protos = D.forward(sd, small, fr)
    assert tuple(pred.shape) == (1, 4 + cfg.nc + cfg.nm, 8 * 8 + 4 * 4 + 2 * 2) and tuple(protos.shape) == (1, 32, 16, 16)
    assert torch.isfinite(pred).all() and torch.isfinite(protos).all()


def test_detector_config_for_state_picks_family_and_nc():
    from mtgv import spec

    for cfg in (spec.DetectorConfig(nc=3), spec.yolo11_config(nc=5)):
        sd = spec.random_detector_state(cfg, 0)
        got = spec.detector_config_for_state(sd)
        assert got.arch == cfg.arch and got.nc == cfg.nc and got.depth == cfg.depth


def test_byte_over_255_without_a_division_is_the_ieee_quotient():
    """conv0_u8_kernel (csrc/detector.hip) turns a frame byte u into u / 255 (ultralytics preprocess, img / 255) as
    q = u * fl(1/255); q' = fma(fma(-q, 255, u), fl(1/255), q).  Checked here for all 256 bytes against the correctly
    rounded quotient, with the two fused operations evaluated in exact rational arithmetic."""
    from fractions import Fraction

    def rnd32(fr):
        f = np.float32(float(fr))
        cands = [np.nextafter(f, np.float32(-np.inf)), f, np.nextafter(f, np.float32(np.inf))]
        return np.float32(min(cands, key=lambda c: (abs(Fraction(float(c)) - fr), int(np.float32(c).view(np.uint32)) & 1)))

    r = np.float32(1.0) / np.float32(255.0)
    for u in range(256):
        ref = np.float32(u) / np.float32(255.0)
        q = np.float32(u) * r
        e = rnd32(Fraction(u) - Fraction(float(q)) * 255)
        q2 = rnd32(Fraction(float(e)) * Fraction(float(r)) + Fraction(float(q)))
        assert q2 == ref, u


def test_letterbox_oracle_against_torch_interpolate():
    """oracle/resize_ref.letterbox (the restatement the GPU kernel is checked against bit for bit) vs the host path it
    replaces in `Detector.detect` (torch.nn.functional.interpolate, bilinear, align_corners=False): the same image up to one
    grey level where the two summation orders round differently; pure padding for frames that already fit"""
    import numpy as np

    from mtgv.detector import letterbox
    from oracle import resize_ref

    for h, w in [(480, 640), (720, 1280), (300, 200), (1080, 810), (64, 64)]:
        frame = np.random.default_rng(h + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = resize_ref.letterbox(frame, 640)
        host, _, _ = letterbox(frame, 640)
        d = np.abs(ref.astype(np.int32) - host.astype(np.int32))
        assert d.max() <= 1, (h, w, d.max())
        assert (d > 0).mean() < 0.02
    frame = np.random.default_rng(0).integers(0, 256, (480, 640, 3), dtype=np.uint8)
    np.testing.assert_array_equal(resize_ref.letterbox(frame, 640)[80:560], frame)
