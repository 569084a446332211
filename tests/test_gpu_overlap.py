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
