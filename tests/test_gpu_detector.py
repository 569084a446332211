"""GPU parity of the detector stage against oracle/detector_ref.py (parity unpinned upstream: see the oracle header).

NMS indices / classes / boxes are compared bit-exactly on identical inputs; the conv stack, decode and mask
logits within the 1e-4 tolerance BASELINE.json states (boxes are in pixels up to 640, so 1e-4 relative)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _synthetic_pred(rng, n, nc, nm, na, n_hot, ties=False, cluster=True):
    """decoded predictions with n_hot anchors above threshold, clustered so that IoU suppression happens"""
    pred = np.zeros((n, 4 + nc + nm, na), np.float32)
    for i in range(n):
        centers = rng.uniform(50, 590, (12, 2))
        which = rng.integers(0, 12, na)
        jit = rng.normal(0, 6.0 if cluster else 200.0, (na, 2))
        pred[i, 0] = centers[which, 0] + jit[:, 0]
        pred[i, 1] = centers[which, 1] + jit[:, 1]
        pred[i, 2] = rng.uniform(40, 120, na)
        pred[i, 3] = rng.uniform(60, 160, na)
        sc = rng.uniform(0.0, 0.2, (nc, na))
        hot = rng.choice(na, n_hot, replace=False)
        sc[rng.integers(0, nc, n_hot), hot] = rng.uniform(0.25, 0.99, n_hot)
        if ties and n_hot >= 8:
            sc[:, hot[:8]] = 0.0
            sc[0, hot[:4]] = 0.5  # four exact ties in class 0
            sc[nc - 1, hot[4:8]] = 0.75
        pred[i, 4 : 4 + nc] = sc
        pred[i, 4 + nc :] = rng.normal(0, 1, (nm, na))
    return pred.astype(np.float32)


@pytest.mark.parametrize(
    "n,nc,nm,na,n_hot,ties,max_det",
    [
        (3, 3, 32, 8400, 400, False, 300),
        (2, 3, 32, 8400, 900, True, 300),
        (2, 1, 0, 8400, 5000, False, 300),  # > 4096 candidates: 8192-wide sort; max_det cap
        (1, 3, 4, 8400, 8400, False, 50),  # every anchor is a candidate: 16384-wide sort
        (2, 4, 8, 1000, 0, False, 300),  # no candidates
        (2, 2, 8, 777, 3, False, 300),
        (1, 3, 32, 8400, 600, False, 1000),
    ],
)
def test_nms_bit_exact(n, nc, nm, na, n_hot, ties, max_det):
    from mtgv.detector import nms
    from oracle import detector_ref as D

    rng = np.random.default_rng(n * 100 + n_hot)
    pred = _synthetic_pred(rng, n, nc, nm, na, n_hot, ties)
    out = nms(torch.from_numpy(pred).cuda(), nc, 0.25, 0.7, max_det)
    out = {k: v.cpu().numpy() for k, v in out.items()}
    for i in range(n):
        ref = D.nms_single(pred[i], nc, 0.25, 0.7, max_det)
        k = len(ref["keep_idx"])
        assert out["n_det"][i] == k
        np.testing.assert_array_equal(out["keep_idx"][i, :k], ref["keep_idx"])
        np.testing.assert_array_equal(out["cls"][i, :k], ref["cls"])
        np.testing.assert_array_equal(out["conf"][i, :k], ref["conf"])
        np.testing.assert_array_equal(out["boxes"][i, :k], ref["boxes"])
    if n_hot:
        assert (out["n_det"] > 0).all()


@pytest.fixture(scope="module")
def det_setup():
    from mtgv import spec
    from mtgv.detector import Detector
    from oracle import detector_ref as D

    cfg = spec.DetectorConfig()
    sd = spec.random_detector_state(cfg, 3)
    frames = np.random.default_rng(4).integers(0, 256, (3, 640, 640, 3), dtype=np.uint8)
    det = Detector(cfg, sd, max_batch=4)
    ref_dets, ref_pred, ref_protos = D.detect(sd, cfg, frames)
    return cfg, sd, frames, det, ref_dets, ref_pred, ref_protos


def test_forward_pred_and_protos(det_setup):
    cfg, sd, frames, det, ref_dets, ref_pred, ref_protos = det_setup
    det.forward(torch.from_numpy(frames).cuda(), True, 0)
    pred, protos = det.raw_outputs(3)
    pred, protos = pred.cpu().numpy(), protos.cpu().numpy()
    nc = cfg.nc
    box_err = np.abs(pred[:, :4] - ref_pred[:, :4]).max()
    cls_err = np.abs(pred[:, 4 : 4 + nc] - ref_pred[:, 4 : 4 + nc]).max()
    coef_err = np.abs(pred[:, 4 + nc :] - ref_pred[:, 4 + nc :]).max()
    proto_err = np.abs(protos - ref_protos).max()
    print(f"box {box_err:.2e}px cls {cls_err:.2e} coef {coef_err:.2e} protos {proto_err:.2e}")
    assert box_err < 640 * 1e-4 and cls_err < 1e-4 and coef_err < 1e-4 and proto_err < 1e-4
    assert abs(det.flops_per_frame() / 1e9 - 12.0) < 3.0  # same order as the published 12.6 GFLOP (80 classes)


def test_detect_matches_oracle(det_setup):
    from oracle import detector_ref as D

    cfg, sd, frames, det, ref_dets, ref_pred, ref_protos = det_setup
    out = det.forward(torch.from_numpy(frames).cuda(), True, cfg.max_det)
    pred, protos = det.raw_outputs(3)
    pred, protos = pred.cpu().numpy(), protos.cpu().numpy()
    o = {k: (v.cpu().numpy() if v is not None else None) for k, v in out.items()}
    for i in range(3):
        k = int(o["n_det"][i])
        # (1) the NMS kernel is bit-exact on the predictions it was given
        same_in = D.nms_single(pred[i], cfg.nc, cfg.conf, cfg.iou, cfg.max_det)
        np.testing.assert_array_equal(o["keep_idx"][i, :k], same_in["keep_idx"])
        np.testing.assert_array_equal(o["cls"][i, :k], same_in["cls"])
        np.testing.assert_array_equal(o["boxes"][i, :k], same_in["boxes"])
        # (2) end to end (HIP forward + HIP NMS vs CPU forward + CPU NMS): the two forwards differ by
        # fp32 summation order (~1e-6), so a keep/suppress decision that sits within rounding of the
        # conf or IoU threshold may flip; everything else must be identical, in the same order.
        ref = ref_dets[i]
        got_idx, ref_idx = o["keep_idx"][i, :k], ref["keep_idx"]
        common = np.intersect1d(got_idx, ref_idx)
        flips = max(k, len(ref_idx)) - len(common)
        print(f"frame {i}: kept {k} (oracle {len(ref_idx)}), threshold flips {flips}")
        # observed on MI355X with these seeds: 0 flips on every frame; one keep/suppress decision at a threshold is the
        # most fp32 summation order can explain on a frame, so that is what the test allows
        assert k > 10 and flips <= 1
        gi = {a: j for j, a in enumerate(got_idx)}
        ri = {a: j for j, a in enumerate(ref_idx)}
        gsel = np.asarray([gi[a] for a in common]); rsel = np.asarray([ri[a] for a in common])
        # order: both score-descending; anchors whose scores differ by less than the forward noise may swap
        assert (np.diff(o["conf"][i, :k]) <= 0).all()
        moved = np.abs(gsel - rsel)
        assert moved.max() <= 1 + flips  # observed: one adjacent swap of two near-equal scores
        np.testing.assert_array_equal(o["cls"][i, :k][gsel], ref["cls"][rsel])
        assert np.abs(o["boxes"][i, :k][gsel] - ref["boxes"][rsel]).max() < 640 * 1e-4
        assert np.abs(o["conf"][i, :k][gsel] - ref["conf"][rsel]).max() < 1e-4
        # (3) mask logits: coeff @ protos cropped to the box
        ml = o["mask_logits"][i, :k]
        assert np.abs(ml[gsel] - ref["mask_logits"][rsel]).max() < 1e-4
        same = D.mask_logits(pred[i], protos[i], same_in, cfg.nc)
        assert np.abs(ml - same).max() < 2e-5
        assert (ml != 0).any()


def test_single_frame_api_and_letterbox(det_setup):
    from mtgv.detector import letterbox
    from oracle import detector_ref as D

    cfg, sd, frames, det, *_ = det_setup
    frame = np.random.default_rng(8).integers(0, 256, (480, 640, 3), dtype=np.uint8)  # webcam size: pad only
    img, r, (left, top) = letterbox(frame)
    assert img.shape == (640, 640, 3) and r == 1.0 and (left, top) == (0, 80)
    assert (img[:80] == 114).all() and (img[560:] == 114).all() and (img[80:560] == frame).all()
    d = det.detect(frame)
    ref, _, _ = D.detect(sd, cfg, img[None])
    np.testing.assert_array_equal(d.keep_idx.cpu().numpy(), ref[0]["keep_idx"])
    assert d.mask_logits.shape == (len(ref[0]["keep_idx"]), 160, 160)
    with pytest.raises(AssertionError):
        det.detect(frame.astype(np.float32))


@pytest.mark.parametrize("h,w", [(480, 640), (640, 640), (720, 1280), (300, 200), (33, 1000), (1080, 810)])
def test_letterbox_kernel_bit_exact(h, w):
    """scale-to-fit + centre + pad on the GPU (resize.hip letterbox_u8_kernel) against oracle/resize_ref.letterbox, every byte;
    frames that already fit are copied exactly"""
    from mtgv.detector import letterbox_device, letterbox_geometry
    from oracle import resize_ref

    frame = np.random.default_rng(h * 7 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    got, r, (left, top) = letterbox_device(torch.from_numpy(frame).cuda(), 640)
    ref = resize_ref.letterbox(frame, 640)
    np.testing.assert_array_equal(got[0].cpu().numpy(), ref)
    r2, nh, nw, top2, left2 = letterbox_geometry(h, w, 640)
    assert (r, top, left) == (r2, top2, left2) and max(nh, nw) == 640
    if (nh, nw) == (h, w):
        np.testing.assert_array_equal(ref[top : top + nh, left : left + nw], frame)
    assert (ref[:top] == 114).all() and (ref[:, :left] == 114).all()


def test_detector_errors():
    from mtgv import spec
    from mtgv.detector import Detector

    cfg = spec.DetectorConfig()
    sd = spec.random_detector_state(cfg, 3)
    bad = dict(sd)
    bad.pop("model.22.proto.upsample.bias")
    with pytest.raises(KeyError):
        Detector(cfg, bad, max_batch=1)
    d = Detector(cfg, None, max_batch=1)
    with pytest.raises(RuntimeError):
        d.forward(torch.zeros((1, 640, 640, 3), dtype=torch.uint8, device="cuda"))


def test_no_detections_and_other_nc():
    """frames on which nothing passes conf: n_det = 0 everywhere, the pipeline falls back to its fixed quads;
    and a 1-class model (nc=1) runs through the same kernels"""
    from mtgv import spec
    from mtgv.detector import Detector
    from mtgv.encoder import Encoder
    from mtgv.matcher import Matcher
    from mtgv.pipeline import Pipeline
    from oracle import detector_ref as D

    cfg = spec.DetectorConfig()
    sd = spec.random_detector_state(cfg, 3, cls_bias=-12.0)  # class logits far below logit(0.25)
    det = Detector(cfg, sd, max_batch=2)
    frames = torch.from_numpy(np.random.default_rng(4).integers(0, 256, (2, 640, 640, 3), dtype=np.uint8)).cuda()
    out = det.forward(frames, True, 4)
    assert (out["n_det"] == 0).all()
    one = det.detect(frames[0].cpu().numpy())
    assert one.conf.numel() == 0 and one.mask_logits.shape == (0, 160, 160)
    ecfg = spec.EncoderConfig("ae", (192, 128), 3, 48, (1, 1, 1, 1), (8, 16, 32, 64), "pool+linear", True)
    m = Matcher(48, capacity=16)
    m.add(np.random.default_rng(1).standard_normal((16, 48)).astype(np.float32))
    o = Pipeline(det, Encoder(ecfg, spec.random_encoder_state(ecfg, 1), max_batch=8), m, 4, 1).run(frames)
    assert o["ids"].shape == (2, 4, 1) and (o["ids"] >= 0).all()
    assert torch.equal(o["boxes"][0], o["boxes"][1])  # the fixed pad quads

    cfg1 = spec.DetectorConfig(nc=1)
    sd1 = spec.random_detector_state(cfg1, 5)
    det1 = Detector(cfg1, sd1, max_batch=1)
    f1 = np.random.default_rng(6).integers(0, 256, (1, 640, 640, 3), dtype=np.uint8)
    det1.forward(torch.from_numpy(f1).cuda(), True, 0)
    pred, protos = det1.raw_outputs(1)
    rp, rq = D.forward(sd1, cfg1, f1)
    assert pred.shape == (1, 4 + 1 + 32, 8400)
    assert np.abs(pred.cpu().numpy()[:, 4:] - rp.numpy()[:, 4:]).max() < 1e-4
    assert np.abs(pred.cpu().numpy()[:, :4] - rp.numpy()[:, :4]).max() < 640 * 1e-4
    assert np.abs(protos.cpu().numpy() - rq.numpy()).max() < 1e-4


_WINDOW_SCRIPT = r"""
import os, sys, numpy as np, torch
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "mtg-vision_amd")]
from mtgv import spec
from mtgv.detector import Detector
arch = sys.argv[2]
cfg = spec.yolo11_config() if arch == "11" else spec.DetectorConfig()
det = Detector(cfg, spec.random_detector_state(cfg, 3), max_batch=3)
g = torch.Generator(device="cuda").manual_seed(9)
frames = torch.randint(0, 256, (3, 640, 640, 3), generator=g, device="cuda", dtype=torch.uint8)
det.forward(frames, False, 0)
pred, protos = det.raw_outputs(3)
np.savez(sys.argv[1], pred=pred.cpu().numpy(), protos=protos.cpu().numpy())
"""


@pytest.mark.parametrize("arch", ["v8", "11"])
def test_window_conv_agrees_with_tap_gather(tmp_path, arch):
    """3x3 / stride-1 convs run out of the LDS-staged input window (gemm_sp_kernel A mode 5); MTGV_SP_WINDOW=0 keeps the
    nine-tap gather.  Same network, same inputs, batch 3 (ragged last tiles on the 20x20 and 40x40 maps): the raw head
    outputs agree to rounding (the two forms accumulate K in a different order).  Fresh processes: the switch is read once."""
    import os
    import subprocess
    import sys

    script = tmp_path / "run.py"
    script.write_text(_WINDOW_SCRIPT)
    outs = []
    for flag in ("1", "0"):
        out = tmp_path / f"w{flag}.npz"
        env = dict(os.environ, MTGV_SP_WINDOW=flag)
        r = subprocess.run([sys.executable, str(script), str(out), arch], env=env, capture_output=True, text=True, timeout=600,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    for k in ("pred", "protos"):
        a, b = outs[0][k].astype(np.float64), outs[1][k].astype(np.float64)
        assert np.isfinite(a).all() and a.shape == b.shape
        assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(b).max()), k
