"""Properties at BASELINE.json's full sizes (the oracle is too slow there): batch independence, determinism,
self-consistency.  Sized for one GPU; a few seconds each."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_encoder_batch256_consistency_and_oracle_sample():
    from mtgv import spec
    from mtgv.encoder import Encoder
    from oracle import encoder_ref as R

    cfg = spec.encoder_config("cnvnxt2ae_tiny", (192, 128), "conv+linear")
    sd = spec.random_encoder_state(cfg, 1)
    enc = Encoder(cfg, sd, max_batch=256)
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randint(0, 256, (256, 192, 128, 3), generator=g, device="cuda", dtype=torch.uint8)
    z = enc.encode(x)
    assert z.shape == (256, 768) and torch.isfinite(z).all()
    # determinism: same input, same bits
    assert torch.equal(z, enc.encode(x))
    # batch independence: an image's embedding does not depend on its batch neighbours (GRN sums are per image;
    # only their tile partition, hence rounding order, can change)
    z_small = torch.cat([enc.encode(x[i : i + 32]) for i in range(0, 256, 32)])
    assert (z - z_small).abs().max().item() < 1e-5
    perm = torch.randperm(256, generator=torch.Generator().manual_seed(1)).cuda()
    assert (enc.encode(x[perm]) - z[perm]).abs().max().item() < 1e-5
    # a sample against the CPU oracle
    idx = [0, 37, 128, 255]
    f = (x[idx].cpu().numpy().astype(np.float32) / 255.0).transpose(0, 3, 1, 2)
    ref = R.encoder_forward(sd, cfg, f).numpy()
    assert np.abs(z[idx].cpu().numpy() - ref).max() < 1e-4


def test_plain_tiny_batch256_at_224_config2():
    """BASELINE.json configs[1] as written: plain ConvNeXt-V2 tiny, batch 256 at 224x224 - determinism, batch
    independence and a 4-image sample within 1e-4 of the CPU oracle (the oracle takes seconds per image at this size)."""
    from mtgv import spec
    from mtgv.encoder import Encoder
    from oracle import encoder_ref as R

    cfg = spec.encoder_config("convnextv2_tiny", (224, 224))
    sd = spec.random_encoder_state(cfg, 5)
    enc = Encoder(cfg, sd, max_batch=256)
    g = torch.Generator(device="cuda").manual_seed(6)
    x = torch.randint(0, 256, (256, 224, 224, 3), generator=g, device="cuda", dtype=torch.uint8)
    z = enc.encode(x)
    assert z.shape == (256, cfg.z_size) and torch.isfinite(z).all()
    assert torch.equal(z, enc.encode(x))
    z_small = torch.cat([enc.encode(x[i : i + 64]) for i in range(0, 256, 64)])
    assert (z - z_small).abs().max().item() < 1e-5
    idx = [0, 85, 170, 255]
    f = (x[idx].cpu().numpy().astype(np.float32) / 255.0).transpose(0, 3, 1, 2)
    ref = R.encoder_forward(sd, cfg, f).numpy()
    assert np.abs(z[idx].cpu().numpy() - ref).max() < 1e-4


def test_detector_batch32_equals_single_frames():
    """conv outputs have no cross-row reduction: a frame's detections are bit-identical alone or in a batch of 32"""
    from mtgv import spec
    from mtgv.detector import Detector

    cfg = spec.DetectorConfig()
    det = Detector(cfg, spec.random_detector_state(cfg, 3), max_batch=32)
    g = torch.Generator(device="cuda").manual_seed(4)
    frames = torch.randint(0, 256, (32, 640, 640, 3), generator=g, device="cuda", dtype=torch.uint8)
    out = {k: (v.clone() if v is not None else None) for k, v in det.forward(frames, True, 8).items()}
    assert (out["n_det"] > 0).all() and (out["n_det"] <= cfg.max_det).all()
    for i in (0, 13, 31):
        one = det.forward(frames[i : i + 1], True, 8)
        n = int(one["n_det"][0])
        assert n == int(out["n_det"][i])
        assert torch.equal(one["keep_idx"][0, :n], out["keep_idx"][i, :n])
        assert torch.equal(one["boxes"][0, :n], out["boxes"][i, :n])
        assert torch.equal(one["mask_logits"][0, : min(n, 8)], out["mask_logits"][i, : min(n, 8)])
    # score-descending, boxes inside a sane range, classes valid
    for i in range(32):
        n = int(out["n_det"][i])
        c = out["conf"][i, :n]
        assert (c[:-1] >= c[1:]).all() and (c > cfg.conf).all()
        assert ((out["cls"][i, :n] >= 0) & (out["cls"][i, :n] < cfg.nc)).all()


@pytest.mark.parametrize("arch", ["yolov8n-seg", "yolo11n-seg"])
def test_detector_fork_join_and_single_launch_upsample_are_bit_identical(arch):
    """the forward's internal fork-join (prototype branch and P3 / P4 heads on library-owned streams beside the neck,
    detector.hip), the ConvTranspose as ONE scattered launch and SPPF's three max pools as ONE launch give the same bits as
    the serial schedule with their separate launches -
    raw predictions, prototypes, kept indices, boxes and mask logits - at batch 32, repeatedly (a race between the
    branches would show as run-to-run differences)"""
    import os

    from mtgv import spec
    from mtgv.detector import Detector

    cfg = spec.DetectorConfig() if arch == "yolov8n-seg" else spec.yolo11_config()
    det = Detector(cfg, spec.random_detector_state(cfg, 3), max_batch=32)
    g = torch.Generator(device="cuda").manual_seed(5)
    frames = torch.randint(0, 256, (32, 640, 640, 3), generator=g, device="cuda", dtype=torch.uint8)

    def run(fork, up1):
        os.environ["MTGV_DET_FORK"], os.environ["MTGV_PROTO_UP1"] = fork, up1
        os.environ["MTGV_SPPF_POOLS1"] = up1  # SPPF's three max pools as one launch, switched together with the upsample form
        os.environ["MTGV_DET_CHAIN"] = up1    # ... and the chained 1x1 convs (C2f cv1, Proto cv3, the heads' final convs)
        try:
            out = {k: (v.clone() if v is not None else None) for k, v in det.forward(frames, True, 8).items()}
            pred, protos = det.raw_outputs(32)
            out["pred"], out["protos"] = pred.clone(), protos.clone()
            torch.cuda.synchronize()
            return out
        finally:
            os.environ.pop("MTGV_DET_FORK", None), os.environ.pop("MTGV_PROTO_UP1", None), os.environ.pop("MTGV_SPPF_POOLS1", None), os.environ.pop("MTGV_DET_CHAIN", None)

    base = run("0", "0")
    assert (base["n_det"] > 0).any()
    for rep in range(3):
        for fork, up1 in (("1", "1"), ("1", "0"), ("0", "1")):
            got = run(fork, up1)
            for k in ("pred", "protos", "n_det", "keep_idx", "boxes", "conf", "cls", "mask_logits"):
                assert torch.equal(got[k], base[k]), (k, fork, up1, rep)


@pytest.mark.parametrize("quad_source", ["mask", "box"])
def test_pipeline_full_step_properties(quad_source):
    """one bench-sized step (both crop dataflows; "mask" is the bench default): every card gets an id in range, crops are
    uint8 192x128, scores are cosines; mask quads and their crops equal the oracle's on the GPU's own masks"""
    from mtgv import spec
    from mtgv.detector import Detector
    from mtgv.encoder import Encoder
    from mtgv.matcher import Matcher
    from mtgv.pipeline import Pipeline

    det_cfg, enc_cfg = spec.DetectorConfig(), spec.encoder_config("cnvnxt2ae_tiny")
    m = Matcher(768, capacity=100_000)
    g = torch.Generator(device="cuda").manual_seed(2)
    m.add(torch.randn((100_000, 768), generator=g, device="cuda"))
    pipe = Pipeline(Detector(det_cfg, spec.random_detector_state(det_cfg, 3), max_batch=32),
                    Encoder(enc_cfg, spec.random_encoder_state(enc_cfg, 1), max_batch=256), m, 8, 3, quad_source=quad_source)
    frames = torch.randint(0, 256, (32, 640, 640, 3), generator=g, device="cuda", dtype=torch.uint8)
    o = pipe.run(frames)
    if quad_source == "mask":
        from mtgv.detector import binarize_masks
        from oracle import pipeline_ref, quad_ref, warp_ref

        sub = slice(0, 64)  # the first 8 frames' cards against the CPU oracle (python loops: bounded)
        boxes = o["boxes"].reshape(256, 4).cpu().numpy()
        masks = binarize_masks(o["det"]["mask_logits"][:, :8].reshape(256, 160, 160)[sub]).cpu().numpy()
        rq, rok = quad_ref.mask_quads(masks, boxes[sub])
        quads = np.where((rok > 0)[:, None, None], rq, pipeline_ref.boxes_to_quads(boxes[sub]))
        fr = frames[:8].cpu().numpy()
        crops = np.stack([warp_ref.warp_quad(fr[i // 8], quads[i], enc_cfg.image_hw, 0.05) for i in range(64)])
        np.testing.assert_array_equal(o["crops"][sub].cpu().numpy(), crops)
        assert (rok > 0).sum() > 16
    assert o["ids"].shape == (32, 8, 3) and o["crops"].shape == (256, 192, 128, 3) and o["crops"].dtype == torch.uint8
    assert ((o["ids"] >= 0) & (o["ids"] < 100_000)).all()
    s = o["scores"]
    assert (s <= 1.0 + 1e-5).all() and (s >= -1.0 - 1e-5).all() and (s[..., :-1] >= s[..., 1:]).all()
    # the reported score is the cosine of the embedding with the reported bank row
    zn = torch.nn.functional.normalize(o["z"], dim=1)
    rows = torch.from_numpy(m.rows(0, 100_000)).cuda()
    re = (zn[:, None, :] * rows[o["ids"].view(256, 3)]).sum(-1)
    assert (re - s.view(256, 3)).abs().max().item() < 1e-5
    # and nothing in the bank beats the reported top-1 by more than rounding
    best = (zn @ rows.T).max(1).values
    assert (best - s.view(256, 3)[:, 0]).abs().max().item() < 1e-5
