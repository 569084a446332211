"""Randomised shape coverage of the GEMM / conv / top-k / NMS kernels (seeded, ~100 cases, each under both GEMM
operand modes): ragged edges, tiny and odd sizes, every epilogue combination.  References are PyTorch CPU fp64 / the oracles."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ACTS = {0: lambda x: x, 1: F.gelu, 2: F.mish, 3: F.silu, 4: torch.sigmoid}


@pytest.fixture(autouse=True, params=["f16x3", "f32"])
def _both_gemm_precisions(request):
    """every case of this module runs with both GEMM operand modes"""
    from mtgv import native

    before = native.get_gemm_precision()
    native.set_gemm_precision(request.param)
    yield
    native.set_gemm_precision(before)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_linear_random_shapes():
    from mtgv import native as nv

    rng = np.random.default_rng(123)
    for case in range(48):
        m = int(rng.choice([1, 2, 31, 33, 127, 128, 129, 200, 511, 1000, 4099]))
        n = int(rng.choice([1, 3, 5, 16, 31, 32, 33, 64, 80, 96, 97, 160, 161, 200, 384]))
        k = 4 * int(rng.integers(1, 80))
        act = int(rng.integers(0, 5))
        res = bool(rng.integers(0, 2))
        a = rng.standard_normal((m, k)).astype(np.float32)
        w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
        b = rng.standard_normal(n).astype(np.float32)
        r = rng.standard_normal((m, n)).astype(np.float32) if res else None
        out = torch.full((m, n), float("nan"), device="cuda")
        A, W, B = _dev(a), _dev(w), _dev(b)
        R = _dev(r) if res else None
        nv.check(nv.lib().mtgv_op_linear(nv.ptr(A), nv.ptr(W), nv.ptr(B), nv.ptr(R), nv.ptr(out), m, n, k, act, nv.stream()))
        ref = ACTS[act](F.linear(torch.from_numpy(a).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double()))
        if res:
            ref = ref + torch.from_numpy(r).double()
        got = out.cpu().double()
        assert torch.isfinite(got).all(), (case, m, n, k)
        assert (got - ref).abs().max().item() < 3e-5, (case, m, n, k, act, res)


def test_linear_ex_grn_paths_random():
    """GRN partial sums + scale prologue with images that straddle / do not fill tiles"""
    from mtgv import native as nv

    rng = np.random.default_rng(5)
    L = nv.lib()
    for case in range(16):
        hw = int(rng.choice([1, 3, 24, 49, 96, 130, 384]))
        nimg = int(rng.integers(1, 9))
        m = hw * nimg
        n = int(rng.choice([8, 32, 96, 100, 320]))
        k = 4 * int(rng.integers(2, 60))
        a = rng.standard_normal((m, k)).astype(np.float32)
        w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
        b = rng.standard_normal(n).astype(np.float32)
        sc = (rng.random((nimg, k)) + 0.5).astype(np.float32)
        sh = rng.standard_normal(k).astype(np.float32)
        out = torch.full((m, n), float("nan"), device="cuda")
        nparts = int(L.mtgv_op_linear_ex_part_floats(m, n, k, 2, hw))
        part = torch.zeros(nparts + 4, device="cuda")
        A, W, B, SC, SH = _dev(a), _dev(w), _dev(b), _dev(sc), _dev(sh)
        # (1) scale + shift prologue
        nv.check(L.mtgv_op_linear_ex(nv.ptr(A), nv.ptr(W), nv.ptr(B), None, nv.ptr(out), m, n, k, 0, hw, nv.ptr(SC), nv.ptr(SH), None, nv.stream()))
        a2 = torch.from_numpy(a).double() * torch.from_numpy(sc).double().repeat_interleave(hw, 0) + torch.from_numpy(sh).double()
        ref = F.linear(a2, torch.from_numpy(w).double(), torch.from_numpy(b).double())
        assert (out.cpu().double() - ref).abs().max().item() < 5e-5, (case, hw, nimg, n, k)
        # (2) mish + per-image sum of squares partials: summing the partials of an image gives sum(out^2)
        nv.check(L.mtgv_op_linear_ex(nv.ptr(A), nv.ptr(W), nv.ptr(B), None, nv.ptr(out), m, n, k, 2, hw, None, None, nv.ptr(part), nv.stream()))
        o = out.cpu().double()
        want = (o * o).view(nimg, hw, n).sum(1)
        import ctypes as C

        ur, sm = C.c_int32(0), C.c_int32(0)
        nv.check(L.mtgv_op_last_grn_layout(C.byref(ur), C.byref(sm)))
        bm, segmax = ur.value, sm.value  # rows per partial unit depend on the kernel / tile the launch used
        assert segmax == (bm - 1) // hw + 2
        tiles_m = -(-m // bm)
        assert tiles_m * segmax * n <= nparts
        p = part[: tiles_m * segmax * n].cpu().double().view(tiles_m, segmax, n)
        got = torch.zeros(nimg, n, dtype=torch.float64)
        for t in range(tiles_m):
            first = (t * bm) // hw
            last = (min((t + 1) * bm, m) - 1) // hw
            for s in range(last - first + 1):
                got[first + s] += p[t, s]
        assert ((got - want).abs() / (want.abs() + 1e-3)).max().item() < 1e-5, (case, hw, nimg, n, k)


@pytest.mark.parametrize("hw,nimg,n,k", [(49, 37, 96, 392), (196, 9, 192, 776), (784, 3, 96, 384), (49, 130, 768, 3080), (24, 50, 64, 72)])
def test_linear_ex_scaled_a_with_residual_pwconv2_shapes(hw, nimg, n, k):
    """pwconv2-shaped launches through the f32-by-DMA A path: per-image multipliers as an LDS image, images that straddle
    tiles (49 and 24 rows per image: up to 4 / 7 images per 128-row tile), K tails (k % 32 != 0), ragged last tile,
    f32 residual"""
    from mtgv import native as nv

    rng = np.random.default_rng(hw * 1000 + k)
    L = nv.lib()
    m = hw * nimg
    a = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    r = rng.standard_normal((m, n)).astype(np.float32)
    sc = (rng.random((nimg, k)) + 0.5).astype(np.float32)
    out = torch.full((m, n), float("nan"), device="cuda")
    A, W, B, R, SC = _dev(a), _dev(w), _dev(b), _dev(r), _dev(sc)
    nv.check(L.mtgv_op_linear_ex(nv.ptr(A), nv.ptr(W), nv.ptr(B), nv.ptr(R), nv.ptr(out), m, n, k, 0, hw, nv.ptr(SC), None, None, nv.stream()))
    a2 = torch.from_numpy(a).double() * torch.from_numpy(sc).double().repeat_interleave(hw, 0)
    ref = F.linear(a2, torch.from_numpy(w).double(), torch.from_numpy(b).double()) + torch.from_numpy(r).double()
    assert torch.isfinite(out).all()
    assert (out.cpu().double() - ref).abs().max().item() < 5e-5
    # and without multipliers (plain f32 rows by DMA)
    out.fill_(float("nan"))
    nv.check(L.mtgv_op_linear_ex(nv.ptr(A), nv.ptr(W), nv.ptr(B), nv.ptr(R), nv.ptr(out), m, n, k, 0, hw, None, None, None, nv.stream()))
    ref = F.linear(torch.from_numpy(a).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double()) + torch.from_numpy(r).double()
    assert (out.cpu().double() - ref).abs().max().item() < 5e-5


def test_conv_random_geometry():
    from mtgv import native as nv

    rng = np.random.default_rng(77)
    for case in range(28):
        n = int(rng.integers(1, 4))
        h, w = int(rng.integers(3, 24)), int(rng.integers(3, 24))
        cin = 4 * int(rng.integers(1, 9))
        cout = int(rng.choice([1, 3, 8, 16, 33, 64, 100]))
        kh = int(rng.choice([1, 2, 3, 4]))
        stride = int(rng.choice([1, 2, kh]))
        pad = int(rng.choice([0, kh // 2]))
        if h + 2 * pad < kh or w + 2 * pad < kh:
            continue
        act = int(rng.choice([0, 3]))
        x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
        wt = (rng.standard_normal((cout, cin, kh, kh)) / np.sqrt(cin * kh * kh)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        ref = ACTS[act](
            F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), stride=stride, padding=pad)
        ).permute(0, 2, 3, 1)
        oh, ow = ref.shape[1], ref.shape[2]
        out = torch.full((n, oh, ow, cout), float("nan"), device="cuda")
        X, W, B = _dev(x), _dev(wt.transpose(0, 2, 3, 1)), _dev(b)
        nv.check(nv.lib().mtgv_op_conv2d(nv.ptr(X), nv.ptr(W), nv.ptr(B), nv.ptr(out), n, h, w, cin, cout, kh, kh, stride, pad, act, nv.stream()))
        got = out.cpu().double()
        assert torch.isfinite(got).all(), (case, n, h, w, cin, cout, kh, stride, pad)
        assert (got - ref).abs().max().item() < 3e-5, (case, n, h, w, cin, cout, kh, stride, pad)


def test_topk_random_sizes():
    from mtgv.matcher import Matcher
    from oracle import match_ref as M

    rng = np.random.default_rng(9)
    for case in range(12):
        d = 4 * int(rng.integers(1, 40))
        n = int(rng.choice([1, 2, 63, 64, 65, 127, 129, 1000, 3001]))
        b = int(rng.choice([1, 2, 127, 128, 129, 300]))
        k = int(rng.choice([1, 2, 7, 64, 65, 200]))
        bank = rng.standard_normal((n, d)).astype(np.float32)
        q = rng.standard_normal((b, d)).astype(np.float32)
        m = Matcher(d, capacity=n)
        m.add(bank)
        ids, sc = m.match(q, k)
        ids, sc = ids.cpu().numpy(), sc.cpu().numpy()
        s64 = M.scores(q, bank)
        rid, rsc = M.topk_from_scores(s64, k)
        kk = min(k, n)
        assert (ids[:, kk:] == -1).all() and np.isinf(sc[:, kk:]).all()
        got = np.take_along_axis(s64, np.maximum(ids[:, :kk], 0), axis=1)
        assert np.abs(got - rsc[:, :kk]).max() < 3e-6, (case, n, b, k, d)  # same score profile
        assert (np.sort(ids[:, :kk], axis=1) == np.sort(rid[:, :kk], axis=1)).mean() > 0.99, (case, n, b, k, d)
        for r in range(b):
            assert len(set(ids[r, :kk].tolist())) == kk  # no duplicates
