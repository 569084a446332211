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


def test_perspective_card_on_gpu():
    """the GPU quad of a card seen at an angle is the generating quadrilateral (<= 1.5 px), not a rectangle"""
    from test_oracle_quads_cpu import _poly_mask

    quads = np.asarray([[[100, 80], [300, 110], [280, 420], [60, 380]], [[330, 70], [380, 300], [120, 360], [40, 150]],
                        [[250, 400], [60, 330], [110, 90], [300, 60]]], np.float64)
    masks = []
    for quad in quads:
        m = _poly_mask(480, 420, quad)
        bc = (quad[2] + quad[3]) / 2
        inward = quad[:2].mean(0) - bc
        inward /= np.linalg.norm(inward)
        along = (quad[2] - quad[3]) / np.linalg.norm(quad[2] - quad[3])
        yy, xx = np.mgrid[0:480, 0:420].astype(np.float64)
        du = (xx - bc[0]) * along[0] + (yy - bc[1]) * along[1]
        dv = (xx - bc[0]) * inward[0] + (yy - bc[1]) * inward[1]
        masks.append(m & ~((np.abs(du) < 0.3 * np.linalg.norm(quad[2] - quad[3])) & (dv < 60)))
    q, ok = _gpu(np.stack(masks))
    assert (ok == 1).all()
    assert np.abs(q - quads).max() <= 1.5, q


def test_quads_from_logits_equal_binarize_then_quads():
    """the fused kernel (interpolate + threshold + fit, no full-resolution mask) == mtgv_mask_binarize -> mtgv_mask_quads"""
    from mtgv.crop import mask_quads, mask_quads_from_logits
    from mtgv.detector import binarize_masks

    rng = np.random.default_rng(12)
    n, mh, mw = 24, 160, 160
    lg = np.zeros((n, mh, mw), np.float32)
    yy, xx = np.mgrid[0:mh, 0:mw]
    for i in range(n):
        if i % 6 == 5:
            continue  # empty mask
        f = rng.standard_normal((mh // 8 + 1, mw // 8 + 1)).astype(np.float32)
        field = np.kron(f, np.ones((8, 8), np.float32))[:mh, :mw] + 0.3 * rng.standard_normal((mh, mw)).astype(np.float32)
        y1, x1 = rng.integers(0, mh // 2), rng.integers(0, mw // 2)
        y2, x2 = y1 + rng.integers(8, mh // 2), x1 + rng.integers(8, mw // 2)
        box = (yy >= y1) & (yy < y2) & (xx >= x1) & (xx < x2)
        lg[i] = np.where(box, field + 0.5, 0.0)  # zero outside the box, as crop_mask leaves the logits
    L = torch.from_numpy(lg).cuda()
    boxes = torch.from_numpy(rng.uniform(0, 600, (n, 4)).astype(np.float32)).cuda()
    q1, ok1 = mask_quads(binarize_masks(L), boxes)
    q2, ok2 = mask_quads_from_logits(L, boxes)
    assert torch.equal(ok1, ok2) and torch.equal(q1, q2)
    assert int(ok1.sum()) >= n // 2 and int((ok1 == 0).sum()) >= 3


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
        native.check(native.lib().mtgv_mask_quads(None, 1, 8, 8, None, None, None, None, None))
    with pytest.raises(AssertionError):
        native.check(native.lib().mtgv_mask_quads(None, 1, 4096, 8, None, None, None, None, None))  # taller than the LDS tables


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
