"""GPU parity of the mask -> oriented quad kernel (quads.hip, C ABI mtgv_mask_quads) with oracle/quad_ref.py: bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _gpu(masks, boxes=None):
    from mtgv.crop import mask_quads

    q, ok = mask_quads(torch.from_numpy(np.ascontiguousarray(masks.astype(np.uint8))).cuda(), None if boxes is None else torch.from_numpy(boxes).cuda())
    return q.cpu().numpy(), ok.cpu().numpy()


def _shapes(rng, h, w, n):
    from test_oracle_quads_cpu import _rot_rect_mask

    out = []
    for i in range(n):
        kind = i % 4
        if kind == 0:  # rotated card with a notch
            m, _ = _rot_rect_mask(h, w, rng.uniform(0.3, 0.7) * w, rng.uniform(0.3, 0.7) * h, rng.uniform(0.1, 0.4) * w, rng.uniform(0.15, 0.5) * h,
                                  rng.uniform(0, 360), notch=rng.uniform(0.1, 0.4))
        elif kind == 1:  # several discs
            yy, xx = np.mgrid[0:h, 0:w]
            m = np.zeros((h, w), bool)
            for _ in range(rng.integers(1, 5)):
                m |= (yy - rng.integers(0, h)) ** 2 + (xx - rng.integers(0, w)) ** 2 <= rng.integers(2, h // 4) ** 2
        elif kind == 2:  # salt noise: hundreds of blobs, hull with many vertices
            m = rng.random((h, w)) > 0.995
        else:  # thresholded smooth field
            f = rng.standard_normal((h // 8 + 1, w // 8 + 1))
            m = np.kron(f, np.ones((8, 8)))[:h, :w] > 1.0
        out.append(m)
    return np.stack(out)


@pytest.mark.parametrize("h,w,n", [(64, 96, 24), (160, 160, 16), (640, 640, 8)])
def test_quads_bit_exact(h, w, n):
    from oracle import quad_ref as Q

    rng = np.random.default_rng(h + n)
    masks = _shapes(rng, h, w, n)
    boxes = rng.uniform(0, h, (n, 4)).astype(np.float32)
    q, ok = _gpu(masks, boxes)
    rq, rok = Q.mask_quads(masks, boxes)
    np.testing.assert_array_equal(ok, rok)
    np.testing.assert_array_equal(q, rq)
    assert ok.sum() >= n // 2


def test_degenerate_masks_and_errors():
    from mtgv import native
    from oracle import quad_ref as Q

    h, w = 32, 48
    masks = np.zeros((7, h, w), np.uint8)
    masks[1, 5, 7] = 1
    masks[2, 5, 7:20] = 200  # any non-zero byte is foreground
    masks[3, np.arange(24), np.arange(24)] = 1
    masks[4] = 1
    masks[5, :, 10] = 1
    masks[6, 3:9, 40:48] = 1
    boxes = np.arange(28, dtype=np.float32).reshape(7, 4)
    q, ok = _gpu(masks, boxes)
    rq, rok = Q.mask_quads(masks, boxes)
    np.testing.assert_array_equal(ok, rok)
    np.testing.assert_array_equal(q, rq)
    assert ok.tolist() == [0, 1, 1, 1, 1, 1, 1] and q[0].tolist() == [[0, 1], [2, 1], [2, 3], [0, 3]]
    q2, ok2 = _gpu(masks)  # no boxes: zeros for the empty mask
    assert (q2[0] == 0).all() and ok2[0] == 0
    assert _gpu(np.zeros((0, h, w), np.uint8))[0].shape == (0, 4, 2)
    with pytest.raises(AssertionError):
        native.check(native.lib().mtgv_mask_quads(None, 1, 8, 8, None, None, None, None))
    with pytest.raises(AssertionError):
        native.check(native.lib().mtgv_mask_quads(None, 1, 4096, 8, None, None, None, None))  # taller than the LDS tables


def test_pipeline_mask_quads_match_oracle():
    """detector masks -> binarize -> quads -> crops on the GPU == the oracle composition on the same masks"""
    from mtgv import spec
    from mtgv.detector import Detector, binarize_masks
    from mtgv.encoder import Encoder
    from mtgv.matcher import Matcher
    from mtgv.pipeline import Pipeline
    from oracle import quad_ref as Q
    from oracle import warp_ref

    det_cfg = spec.DetectorConfig()
    enc_cfg = spec.encoder_config("cnvnxt2ae_nano")
    F, K = 2, 4
    m = Matcher(768, capacity=500)
    m.add(np.random.default_rng(2).standard_normal((500, 768)).astype(np.float32))
    det = Detector(det_cfg, spec.random_detector_state(det_cfg, 3), max_batch=F)
    pipe = Pipeline(det, Encoder(enc_cfg, spec.random_encoder_state(enc_cfg, 1), max_batch=F * K), m, K, 1, quad_source="mask")
    g = torch.Generator(device="cuda").manual_seed(5)
    frames = torch.randint(0, 256, (F, 640, 640, 3), generator=g, device="cuda", dtype=torch.uint8)
    out = pipe.run(frames)
    masks = binarize_masks(out["det"]["mask_logits"][:, :K].reshape(F * K, 160, 160)).cpu().numpy()
    boxes = out["boxes"].cpu().numpy().reshape(F * K, 4)
    rq, rok = Q.mask_quads(masks, boxes)
    have = (np.arange(K)[None, :] < out["n_det"].cpu().numpy()[:, None]).reshape(-1)
    quads = np.where(((rok > 0) & have)[:, None, None], rq, np.stack([boxes[:, [0, 1]], boxes[:, [2, 1]], boxes[:, [2, 3]], boxes[:, [0, 3]]], 1))
    fr = frames.cpu().numpy()
    crops = np.stack([warp_ref.warp_quad(fr[i // K], quads[i], enc_cfg.image_hw, 0.05) for i in range(F * K)])
    np.testing.assert_array_equal(out["crops"].cpu().numpy(), crops)
    assert (rok > 0).any()
