"""GPU parity of the single-op C-ABI entry points against PyTorch CPU fp32 (and fp64 for error budgets)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from mtgv import native

    return native


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


ACTS = {0: lambda x: x, 1: F.gelu, 2: F.mish, 3: F.silu, 4: torch.sigmoid}


@pytest.mark.parametrize(
    "m,n,k,act,res",
    [
        (1, 8, 4, 0, False),  # smallest legal
        (7, 3, 64, 4, False),  # N < one tile, sigmoid (cls head)
        (130, 96, 48, 0, False),  # stem-like: K=48 (BK 16), ragged M
        (257, 384, 96, 2, False),  # pwconv1 tiny stage 1, mish
        (300, 80, 320, 0, True),  # nano pwconv2 + residual, N=80 ragged
        (64, 160, 640, 1, False),  # tn=5 path, gelu
        (1000, 200, 36, 3, False),  # K=36 (padded to 48), silu
        (513, 3072, 768, 1, False),  # tiny stage 4 pwconv1
    ],
)
def test_linear(m, n, k, act, res):
    nv = _lib()
    rng = np.random.default_rng(m * 7 + n)
    a = rng.standard_normal((m, k)).astype(np.float32)
    w = (rng.standard_normal((n, k)) / np.sqrt(k)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    r = rng.standard_normal((m, n)).astype(np.float32) if res else None
    out = torch.full((m, n), float("nan"), device="cuda")
    A, W, B = _dev(a), _dev(w), _dev(b)
    R = _dev(r) if res else None
    nv.check(nv.lib().mtgv_op_linear(nv.ptr(A), nv.ptr(W), nv.ptr(B), nv.ptr(R), nv.ptr(out), m, n, k, act, nv.stream()))
    ref64 = ACTS[act](F.linear(torch.from_numpy(a).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double()))
    if res:
        ref64 = ref64 + torch.from_numpy(r).double()
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    assert (got - ref64).abs().max().item() < 2e-5


@pytest.mark.parametrize(
    "n,h,w,cin,cout,kh,kw,stride,pad,act",
    [
        (2, 16, 16, 4, 16, 3, 3, 2, 1, 3),  # yolo layer 0 style (RGBX)
        (2, 20, 12, 16, 32, 3, 3, 1, 1, 3),
        (3, 8, 8, 32, 64, 2, 2, 2, 0, 0),  # downsample k2 s2
        (1, 5, 7, 8, 8, 1, 1, 1, 0, 0),
        (2, 10, 10, 64, 3, 1, 1, 1, 0, 0),  # N=3
        (1, 9, 11, 12, 20, 3, 3, 1, 1, 3),  # odd sizes
    ],
)
def test_conv2d(n, h, w, cin, cout, kh, kw, stride, pad, act):
    nv = _lib()
    rng = np.random.default_rng(n + h * 3 + cout)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    wt = (rng.standard_normal((cout, cin, kh, kw)) / np.sqrt(cin * kh * kw)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = ACTS[act](
        F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), stride=stride, padding=pad)
    ).permute(0, 2, 3, 1)
    oh, ow = ref.shape[1], ref.shape[2]
    out = torch.full((n, oh, ow, cout), float("nan"), device="cuda")
    X, W, B = _dev(x), _dev(wt.transpose(0, 2, 3, 1)), _dev(b)
    nv.check(nv.lib().mtgv_op_conv2d(nv.ptr(X), nv.ptr(W), nv.ptr(B), nv.ptr(out), n, h, w, cin, cout, kh, kw, stride, pad, act, nv.stream()))
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("rows,c", [(1, 4), (33, 8), (100, 32), (77, 96), (50, 80), (19, 768), (5, 10), (9, 1536), (3, 2816)])
def test_layernorm(rows, c):
    nv = _lib()
    rng = np.random.default_rng(rows + c)
    x = (rng.standard_normal((rows, c)) * 2 + 0.5).astype(np.float32)
    w = (1 + 0.1 * rng.standard_normal(c)).astype(np.float32)
    b = (0.1 * rng.standard_normal(c)).astype(np.float32)
    out = torch.full((rows, c), float("nan"), device="cuda")
    X, W, B = _dev(x), _dev(w), _dev(b)
    nv.check(nv.lib().mtgv_op_layernorm(nv.ptr(X), nv.ptr(W), nv.ptr(B), nv.ptr(out), rows, c, 1e-6, nv.stream()))
    ref = F.layer_norm(torch.from_numpy(x).double(), (c,), torch.from_numpy(w).double(), torch.from_numpy(b).double(), 1e-6)
    assert (out.cpu().double() - ref).abs().max().item() < 5e-6


@pytest.mark.parametrize("n,h,w,c", [(1, 1, 1, 4), (2, 6, 4, 8), (2, 12, 8, 96), (1, 7, 7, 80), (2, 24, 16, 20), (1, 48, 32, 96), (1, 3, 2, 12)])
def test_dwconv7(n, h, w, c):
    nv = _lib()
    rng = np.random.default_rng(h * 5 + c)
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    wt = (rng.standard_normal((c, 1, 7, 7)) / 7).astype(np.float32)
    b = rng.standard_normal(c).astype(np.float32)
    out = torch.full((n, h, w, c), float("nan"), device="cuda")
    X, W, B = _dev(x), _dev(wt.reshape(c, 49).T), _dev(b)
    nv.check(nv.lib().mtgv_op_dwconv7(nv.ptr(X), nv.ptr(W), nv.ptr(B), nv.ptr(out), n, h, w, c, nv.stream()))
    ref = F.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2).double(), torch.from_numpy(wt).double(), torch.from_numpy(b).double(), padding=3, groups=c).permute(0, 2, 3, 1)
    assert (out.cpu().double() - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("n,h,w,c,act", [(2, 9, 6, 12, "gelu"), (2, 9, 6, 12, "mish"), (3, 6, 4, 64, "mish"), (5, 7, 7, 96, "gelu"), (7, 3, 2, 160, "mish"), (2, 24, 16, 40, "mish")])
def test_block_vs_oracle(n, h, w, c, act):
    """One full ConvNeXt-V2 block incl. the segmented GRN reduction (images straddling GEMM tiles)."""
    from oracle import encoder_ref as R

    nv = _lib()
    rng = np.random.default_rng(c + h)
    p = {
        "b.dwconv.weight": rng.standard_normal((c, 1, 7, 7)) / 7,
        "b.dwconv.bias": 0.1 * rng.standard_normal(c),
        "b.norm.weight": 1 + 0.1 * rng.standard_normal(c),
        "b.norm.bias": 0.1 * rng.standard_normal(c),
        "b.pwconv1.weight": rng.standard_normal((4 * c, c)) / np.sqrt(c),
        "b.pwconv1.bias": 0.1 * rng.standard_normal(4 * c),
        "b.grn.gamma": 0.3 * rng.standard_normal((1, 1, 1, 4 * c)),
        "b.grn.beta": 0.1 * rng.standard_normal((1, 1, 1, 4 * c)),
        "b.pwconv2.weight": rng.standard_normal((c, 4 * c)) / np.sqrt(4 * c),
        "b.pwconv2.bias": 0.1 * rng.standard_normal(c),
    }
    p = {k: v.astype(np.float32) for k, v in p.items()}
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    ref = R.block(torch.from_numpy(x).double(), {k: torch.from_numpy(v).double() for k, v in p.items()}, "b", act).permute(0, 2, 3, 1)
    X = _dev(x.transpose(0, 2, 3, 1))
    out = torch.full((n, h, w, c), float("nan"), device="cuda")
    ws = torch.empty(int(nv.lib().mtgv_op_block_workspace_floats(n, h, w, c)), device="cuda")
    d = {k: _dev(v) for k, v in p.items()}
    d["b.dwconv.weight"] = _dev(p["b.dwconv.weight"].reshape(c, 49).T)
    nv.check(
        nv.lib().mtgv_op_block(
            nv.ptr(X), nv.ptr(out), n, h, w, c, 1 if act == "gelu" else 2,
            nv.ptr(d["b.dwconv.weight"]), nv.ptr(d["b.dwconv.bias"]), nv.ptr(d["b.norm.weight"]), nv.ptr(d["b.norm.bias"]),
            nv.ptr(d["b.pwconv1.weight"]), nv.ptr(d["b.pwconv1.bias"]), nv.ptr(d["b.grn.gamma"]), nv.ptr(d["b.grn.beta"]),
            nv.ptr(d["b.pwconv2.weight"]), nv.ptr(d["b.pwconv2.bias"]), nv.ptr(ws), nv.stream(),
        )
    )
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("n,h,w,c", [(3, 48, 32, 96), (2, 12, 8, 384), (2, 28, 28, 192), (2, 7, 7, 768), (1, 14, 14, 384), (1, 5, 9, 80), (2, 6, 4, 768),
                                     (130, 48, 32, 96), (129, 23, 16, 192), (128, 5, 32, 96)])
def test_dwconv_ln_forms_are_bit_identical(n, h, w, c):
    """the single-row form of dwconv7_ln (MTGV_DW_ROWS=0) and the default forms - row groups (three output rows per
    thread, ragged last rows and strips included) and, from 128 images at 32 x 96 / 16 x 192 rows, the row-streaming
    kernel (LDS-DMA row ring; ragged heights included) - give the same bits (whole block)"""
    import os

    nv = _lib()
    rng = np.random.default_rng(c)
    f = lambda *sh: _dev(rng.standard_normal(sh).astype(np.float32) * 0.3)  # noqa: E731
    X = f(n, h, w, c)
    args = [f(49, c), f(c), 1 + f(c), f(c), f(4 * c, c), f(4 * c), f(1, 1, 1, 4 * c), f(1, 1, 1, 4 * c), f(c, 4 * c), f(c)]
    ws = torch.empty(int(nv.lib().mtgv_op_block_workspace_floats(n, h, w, c)), device="cuda")
    before = os.environ.get("MTGV_DW_ROWS")
    outs = []
    try:
        for rows in ("0", "1"):
            os.environ["MTGV_DW_ROWS"] = rows  # read per launch (dwconv7_ln_kernel.h)
            out = torch.full((n, h, w, c), float("nan"), device="cuda")
            nv.check(nv.lib().mtgv_op_block(nv.ptr(X), nv.ptr(out), n, h, w, c, 2, *[nv.ptr(a) for a in args], nv.ptr(ws), nv.stream()))
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        if before is None:
            os.environ.pop("MTGV_DW_ROWS", None)
        else:
            os.environ["MTGV_DW_ROWS"] = before
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


def test_error_convention():
    nv = _lib()
    X = torch.zeros(16, device="cuda")
    with pytest.raises(AssertionError):
        # K not a multiple of 4 -> status 1 -> AssertionError, like the reference's shape asserts
        nv.check(nv.lib().mtgv_op_linear(nv.ptr(X), nv.ptr(X), None, None, nv.ptr(X), 2, 2, 3, 0, nv.stream()))
    assert b"multiples of 4" in nv.lib().mtgv_last_error()
