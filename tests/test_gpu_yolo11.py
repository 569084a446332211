"""YOLO11n-seg (what the reference trains by default, od_train.py:20, :55-56) on the GPU vs
oracle/detector_ref.py - C3k2, C3k, C2PSA attention, depthwise class branch.  Parity unpinned upstream (ultralytics is
absent); same bars as the YOLOv8 tests: conv stack / decode / prototypes within 1e-4, NMS bit-exact on its inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def det11():
    from mtgv import spec
    from mtgv.detector import Detector
    from oracle import detector_ref as D

    cfg = spec.yolo11_config()
    sd = spec.random_detector_state(cfg, 3, cls_bias=-0.9)
    frames = np.random.default_rng(4).integers(0, 256, (3, 640, 640, 3), dtype=np.uint8)
    det = Detector(cfg, sd, max_batch=4)
    ref_dets, ref_pred, ref_protos = D.detect(sd, cfg, frames)
    return cfg, sd, frames, det, ref_dets, ref_pred, ref_protos


@pytest.mark.parametrize("mode", ["f16x3", "f32"])
def test_yolo11_forward_pred_and_protos(det11, mode):
    from mtgv import native

    cfg, sd, frames, det, ref_dets, ref_pred, ref_protos = det11
    before = native.get_gemm_precision()
    native.set_gemm_precision(mode)
    try:
        det.forward(torch.from_numpy(frames).cuda(), True, 0)
        pred, protos = det.raw_outputs(3)
    finally:
        native.set_gemm_precision(before)
    pred, protos = pred.cpu().numpy(), protos.cpu().numpy()
    nc = cfg.nc
    box_err = np.abs(pred[:, :4] - ref_pred[:, :4]).max()
    cls_err = np.abs(pred[:, 4 : 4 + nc] - ref_pred[:, 4 : 4 + nc]).max()
    coef_err = np.abs(pred[:, 4 + nc :] - ref_pred[:, 4 + nc :]).max()
    proto_err = np.abs(protos - ref_protos).max()
    print(f"{mode}: box {box_err:.2e}px cls {cls_err:.2e} coef {coef_err:.2e} protos {proto_err:.2e}")
    assert box_err < 640 * 1e-4 and cls_err < 1e-4 and coef_err < 1e-4 and proto_err < 1e-4
    gf = det.flops_per_frame() / 1e9
    print(f"yolo11n-seg: {gf:.2f} GFLOP / frame (nc = {nc})")
    assert 7.0 < gf < 13.0  # published: 10.4 GFLOP (80 classes)


def test_yolo11_detect_matches_oracle(det11):
    from oracle import detector_ref as D

    cfg, sd, frames, det, ref_dets, ref_pred, ref_protos = det11
    out = det.forward(torch.from_numpy(frames).cuda(), True, cfg.max_det)
    pred, protos = det.raw_outputs(3)
    pred, protos = pred.cpu().numpy(), protos.cpu().numpy()
    o = {k: (v.cpu().numpy() if v is not None else None) for k, v in out.items()}
    total = 0
    for i in range(3):
        k = int(o["n_det"][i])
        total += k
        same_in = D.nms_single(pred[i], cfg.nc, cfg.conf, cfg.iou, cfg.max_det)
        np.testing.assert_array_equal(o["keep_idx"][i, :k], same_in["keep_idx"])
        np.testing.assert_array_equal(o["cls"][i, :k], same_in["cls"])
        np.testing.assert_array_equal(o["boxes"][i, :k], same_in["boxes"])
        ml = D.mask_logits(pred[i], protos[i], same_in, cfg.nc, cfg.imgsz)
        assert np.abs(o["mask_logits"][i, :k] - ml).max() < 1e-4
        # end to end: the kept set equals the CPU oracle's up to decisions within rounding of a threshold
        ref = ref_dets[i]
        assert abs(k - len(ref["keep_idx"])) <= 2
    assert total > 10


def test_yolo11_batch_equals_single_frames(det11):
    cfg, sd, frames, det, *_ = det11
    fr = torch.from_numpy(frames).cuda()
    det.forward(fr, True, 0)
    pred_b, protos_b = (t.clone() for t in det.raw_outputs(3))
    for i in range(3):
        det.forward(fr[i : i + 1], True, 0)
        p1, q1 = det.raw_outputs(1)
        assert torch.equal(p1[0], pred_b[i]) and torch.equal(q1[0], protos_b[i])


def test_yolo11_batch32_forward_and_nms():
    """BASELINE config 3's batch (32 frames) on the trained detector family: raw head + prototypes within 1e-4 of the CPU
    oracle, NMS bit-exact on the predictions it was given, for every frame of the batch"""
    from mtgv import spec
    from mtgv.detector import Detector
    from oracle import detector_ref as D

    cfg = spec.yolo11_config()
    sd = spec.random_detector_state(cfg, 3, cls_bias=-0.9)
    frames = np.random.default_rng(14).integers(0, 256, (32, 640, 640, 3), dtype=np.uint8)
    det = Detector(cfg, sd, max_batch=32)
    out = det.forward(torch.from_numpy(frames).cuda(), True, 8)
    pred, protos = det.raw_outputs(32)
    pred, protos = pred.cpu().numpy(), protos.cpu().numpy()
    ref_pred, ref_protos = D.forward(sd, cfg, frames)
    ref_pred, ref_protos = np.asarray(ref_pred), np.asarray(ref_protos)
    nc = cfg.nc
    assert np.abs(pred[:, :4] - ref_pred[:, :4]).max() < 640 * 1e-4
    assert np.abs(pred[:, 4:] - ref_pred[:, 4:]).max() < 1e-4
    assert np.abs(protos - ref_protos).max() < 1e-4
    o = {k: (v.cpu().numpy() if v is not None else None) for k, v in out.items()}
    total = 0
    for i in range(32):
        k = int(o["n_det"][i])
        total += k
        same_in = D.nms_single(pred[i], nc, cfg.conf, cfg.iou, cfg.max_det)
        np.testing.assert_array_equal(o["keep_idx"][i, :k], same_in["keep_idx"])
        np.testing.assert_array_equal(o["cls"][i, :k], same_in["cls"])
        np.testing.assert_array_equal(o["boxes"][i, :k], same_in["boxes"])
        assert (o["boxes"][i, k:] == 0).all() and (o["keep_idx"][i, k:] == 0).all()  # padding written by the kernel
        kk = min(k, 8)
        ml = D.mask_logits(pred[i], protos[i], {key: v[:kk] for key, v in same_in.items()}, nc, cfg.imgsz)
        assert np.abs(o["mask_logits"][i, :kk] - ml).max() < 1e-4
        assert (o["mask_logits"][i, kk:] == 0).all()
    assert total > 100
