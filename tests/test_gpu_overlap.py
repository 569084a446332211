"""Kernels of two HIP streams sharing the GPU must not disturb each other.

Regression guard for a hardware-level interaction measured on MI355X: with packed-FP32 VALU instructions in the
library, a ConvNeXt block running beside the split-precision (f16x3) GEMMs of the detector on a second stream came out
wrong in 5-50 % of the runs (one 16-lane group of the fused dwconv7+LayerNorm kernel lost a packed result).  The
library is compiled without those instructions (mtg-vision_amd/build.py: every translation unit since round 4); this
test repeats the reproducer.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def _env_guard():
    from mtgv import native

    before, tile = native.get_gemm_precision(), os.environ.get("MTGV_GEMM_TILE")
    yield native
    native.set_gemm_precision(before)
    if tile is None:
        os.environ.pop("MTGV_GEMM_TILE", None)
    else:
        os.environ["MTGV_GEMM_TILE"] = tile


@pytest.mark.parametrize("mode,tile", [("f16x3", None), ("f16x3", "1,3,16"), ("f16x3", "1,4,16"), ("f32", None)])
def test_block_beside_detector_is_bit_stable(mode, tile, _env_guard):
    from mtgv import spec
    from mtgv.detector import Detector

    nv = _env_guard
    L = nv.lib()
    nv.set_gemm_precision(mode)
    if tile:
        os.environ["MTGV_GEMM_TILE"] = tile  # read at every launch; the widest reproducer was the 128x96 tile
    else:
        os.environ.pop("MTGV_GEMM_TILE", None)
    cfg = spec.DetectorConfig()
    det = Detector(cfg, spec.random_detector_state(cfg, 3), max_batch=2)
    g = torch.Generator(device="cuda").manual_seed(11)
    frames = torch.randint(0, 256, (2, 640, 640, 3), generator=g, device="cuda", dtype=torch.uint8)
    n, h, w, c = 8, 48, 32, 80
    r = lambda *s: torch.randn(*s, generator=g, device="cuda")  # noqa: E731
    X = r(n, h, w, c)
    P = [r(49, c) / 7, 0.1 * r(c), 1 + 0.1 * r(c), 0.1 * r(c), r(4 * c, c) / c**0.5, 0.1 * r(4 * c), 0.3 * r(4 * c), 0.1 * r(4 * c),
         r(c, 4 * c) / (4 * c) ** 0.5, 0.1 * r(c)]
    nws = int(L.mtgv_op_block_workspace_floats(n, h, w, c))

    def block():
        ws = torch.zeros(nws, device="cuda")
        out = torch.empty((n, h, w, c), device="cuda")
        nv.check(L.mtgv_op_block(nv.ptr(X), nv.ptr(out), n, h, w, c, 2, *[nv.ptr(p) for p in P], nv.ptr(ws), nv.stream()))
        return out

    ref = block()
    torch.cuda.synchronize()
    s_det, s_blk = torch.cuda.Stream(), torch.cuda.Stream()
    bad = 0
    for _ in range(20):
        with torch.cuda.stream(s_det):
            det.forward(frames, True, mask_rows=4)
        with torch.cuda.stream(s_blk):
            out = block()
        torch.cuda.synchronize()
        bad += int(not torch.equal(out, ref))
    assert bad == 0, f"{bad}/20 runs of the block differ when the detector shares the GPU ({mode}, tile {tile})"


def test_pipeline_schedules_give_identical_outputs(monkeypatch):
    """the same batches through every schedule the pipeline has - one stream with the detector's branches in sequence
    (MTGV_DET_FORK=0), one stream with the detector's internal fork-join, the overlapped schedule (MTGV_OVERLAP=on) under each of its switches, and both
    of those with the frames arriving from host memory (HostFrames: a copy stream, a ring of three device buffers that the
    de-warp releases) - bit-identical ids, scores, embeddings, crops and boxes; five batches, so the ring wraps"""
    from mtgv import spec
    from mtgv.detector import Detector
    from mtgv.encoder import Encoder
    from mtgv.matcher import Matcher
    from mtgv.pipeline import HostFrames, Pipeline

    F, K = 8, 4
    det_cfg, enc_cfg = spec.DetectorConfig(), spec.encoder_config("cnvnxt2ae_nano")
    g = torch.Generator(device="cuda").manual_seed(12)
    m = Matcher(768, capacity=20_000)
    m.add(torch.randn((20_000, 768), generator=g, device="cuda"))
    pipe = Pipeline(Detector(det_cfg, spec.random_detector_state(det_cfg, 3), max_batch=F),
                    Encoder(enc_cfg, spec.random_encoder_state(enc_cfg, 1), max_batch=F * K), m, K, 1, quad_source="mask")
    batches = [torch.randint(0, 256, (F, 640, 640, 3), generator=g, device="cuda", dtype=torch.uint8) for _ in range(5)]
    host = [b.cpu() for b in batches]
    keys = ("ids", "scores", "z", "crops", "boxes", "n_det")

    def snap(outs):
        torch.cuda.synchronize()
        return [{k: o[k].clone() for k in keys} for o in outs]

    monkeypatch.setenv("MTGV_DET_FORK", "0")
    monkeypatch.delenv("MTGV_OVERLAP", raising=False)
    ref = snap([pipe.run(b) for b in batches])
    monkeypatch.delenv("MTGV_DET_FORK")
    src = HostFrames(host, "cuda")
    assert all(b.is_pinned() for b in src.host) and len(src.bufs) == 3
    got = {"one stream, fork-join": snap([pipe.run(b) for b in batches]),
           "one stream, host frames": snap([pipe.run(ls) for ls in src.leases(5)])}
    monkeypatch.setenv("MTGV_OVERLAP", "on")
    assert pipe.overlap_enabled()
    got["overlapped (detect + crop / embed at high priority / match)"] = snap(pipe.run_many(batches))
    got["overlapped, host frames"] = snap(pipe.run_many(src.leases(5)))
    got["overlapped, host frames again (ring reused)"] = snap(pipe.run_many(src.leases(5)))
    # the schedule switches of run_many (round 3's form: crops in front of the encoder, the match behind it, equal priorities)
    monkeypatch.setenv("MTGV_CROP_STAGE", "enc")
    got["overlapped, crop stage on the embed stream"] = snap(pipe.run_many(batches))
    monkeypatch.setenv("MTGV_MATCH_STREAM", "0")
    got["overlapped, crop and match on the embed stream"] = snap(pipe.run_many(batches))
    monkeypatch.delenv("MTGV_CROP_STAGE")
    got["overlapped, match on the embed stream"] = snap(pipe.run_many(src.leases(5)))
    monkeypatch.delenv("MTGV_MATCH_STREAM")
    monkeypatch.setenv("MTGV_OVERLAP_DET_FORK", "1")  # the detector's own fork-join left on inside run_many
    got["overlapped, detector fork-join on"] = snap(pipe.run_many(batches))
    monkeypatch.delenv("MTGV_OVERLAP_DET_FORK")
    monkeypatch.setenv("MTGV_MATCH_PRIO", "0")
    del pipe._s_match
    got["overlapped, match stream at normal priority"] = snap(pipe.run_many(batches))
    monkeypatch.delenv("MTGV_MATCH_PRIO")
    del pipe._s_match  # (recreated at its default priority by the next run_many)
    for prio in ("none", "det"):  # (read when the streams are created)
        monkeypatch.setenv("MTGV_STREAM_PRIO", prio)
        del pipe._s_det, pipe._s_enc
        got[f"overlapped, stream priority {prio}"] = snap(pipe.run_many(batches))
    # a caller-supplied match (the sharded bank's path): run_many keeps every stream at normal priority for it
    pipe2 = Pipeline(pipe.detector, pipe.encoder, m, K, 1, match_fn=lambda z, k: m.match(z, k), quad_source="mask")
    assert not pipe2._plain_match
    got["overlapped, caller-supplied match (normal priorities)"] = snap(pipe2.run_many(batches))
    assert pipe2._s_enc.priority == 0 and pipe2._s_match.priority == 0 and pipe._s_match.priority == -1
    for name, outs in got.items():
        assert len(outs) == 5
        for i, (a, b) in enumerate(zip(ref, outs)):
            for k in keys:
                assert torch.equal(a[k], b[k]), (name, i, k)
