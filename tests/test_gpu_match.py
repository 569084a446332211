"""GPU parity of the bank top-k against the brute-force oracle (ids exact; near-ties reported, none tolerated silently)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _check(ids, sc, q, bank, k, id_base=0):
    from oracle import match_ref as M

    s64 = M.scores(q, bank)
    rid, rsc = M.topk_from_scores(s64, k)
    ids = ids.cpu().numpy() - id_base
    sc = sc.cpu().numpy()
    margin = M.top_margin(s64, min(k, s64.shape[1] - 1))
    safe = margin > 1e-6  # fp32 accumulation-order noise is ~1e-7; a smaller gap is a genuine tie
    assert safe.mean() > 0.95
    assert (ids[safe] == rid[safe]).all(), "top-k ids differ from the oracle outside near-ties"
    # for near-ties the id set must still be explainable by the scores
    got_scores = np.take_along_axis(s64, np.maximum(ids, 0), axis=1)
    assert np.abs(got_scores - rsc)[ids >= 0].max() < 2e-6
    assert np.abs(sc - rsc)[ids >= 0].max() < 2e-6
    assert (np.diff(sc, axis=1) <= 0).all()


@pytest.mark.parametrize("n,b,k,d", [(1000, 4, 1, 768), (1000, 4, 5, 768), (130, 3, 3, 64), (5000, 129, 3, 768), (257, 1, 16, 32)])
def test_topk_matches_oracle(n, b, k, d):
    from mtgv.matcher import Matcher

    rng = np.random.default_rng(2)
    bank = rng.standard_normal((n, d)).astype(np.float32)
    q = (rng.standard_normal((b, d)) * 3).astype(np.float32)
    m = Matcher(d, capacity=n + 8)
    m.add(bank[: n // 2])
    m.add(torch.from_numpy(bank[n // 2 :]).cuda())
    assert len(m) == n
    ids, sc = m.match(q, k)
    _check(ids, sc, q, bank, k)
    # stored rows are unit length
    rows = m.rows(0, min(n, 16))
    np.testing.assert_allclose(np.linalg.norm(rows, axis=1), 1.0, atol=1e-6)


def test_exact_ties_break_by_id_and_padding():
    from mtgv.matcher import Matcher

    d = 8
    e0 = np.eye(d, dtype=np.float32)[0]
    bank = np.stack([e0 * 2, np.eye(d, dtype=np.float32)[1], e0 * 5, -e0, e0 * 0.5])
    m = Matcher(d, capacity=8, id_base=100)
    m.add(bank)
    ids, sc = m.match(e0, 8)
    assert ids[0].tolist() == [100, 102, 104, 101, 103, -1, -1, -1]
    assert sc[0, :3].tolist() == [1.0, 1.0, 1.0]
    assert torch.isinf(sc[0, 5:]).all()
    m.set_row(1, e0 * 7)
    ids, _ = m.match(e0, 2)
    assert ids[0].tolist() == [100, 101]


def test_full_size_properties():
    """BASELINE size (100k x 768, 256 queries): self-retrieval and shard-merge equivalence."""
    from mtgv.matcher import Matcher, merge_topk

    g = torch.Generator(device="cuda").manual_seed(2)
    bank = torch.randn((100_000, 768), generator=g, device="cuda")
    m = Matcher(768, capacity=100_000)
    m.add(bank)
    pick = torch.arange(0, 100_000, 391, device="cuda")[:256]
    q = bank[pick] * 3.0 + 0.01 * torch.randn((256, 768), generator=g, device="cuda")
    ids, sc = m.match(q, 3)
    assert (ids[:, 0] == pick).all()  # a perturbed bank row retrieves itself
    assert (sc[:, 0] > 0.99).all() and (sc[:, 1] < 0.5).all()
    # sharded (8 row shards + merge) == unsharded, bit for bit
    cs, ci = [], []
    for r in range(8):
        sh = Matcher(768, capacity=12_500, id_base=r * 12_500)
        sh.add(bank[r * 12_500 : (r + 1) * 12_500])
        i, s = sh.match(q, 3)
        cs.append(s)
        ci.append(i)
    mi, ms = merge_topk(torch.cat(cs, 1), torch.cat(ci, 1), 3)
    assert (mi == ids).all()
    assert (ms == sc).all()


def test_full_size_vs_oracle_fp64():
    """BASELINE size through the fused top-k epilogue of the LDS-DMA kernel (256 queries >= 128, 100k x 768): ids against
    the fp64 brute-force oracle (blockwise over the bank, a few seconds on the host), exact outside near-ties."""
    from mtgv.matcher import Matcher
    from oracle import match_ref as M

    rng = np.random.default_rng(12)
    bank = rng.standard_normal((100_000, 768), dtype=np.float32)
    q = rng.standard_normal((256, 768), dtype=np.float32) * 2
    # a planted near-neighbour for half of the queries, so the winners are not all at the noise floor
    pick = rng.integers(0, 100_000, 128)
    q[:128] = bank[pick] * 1.5 + 0.3 * rng.standard_normal((128, 768), dtype=np.float32)
    m = Matcher(768, capacity=100_000)
    m.add(torch.from_numpy(bank).cuda())
    k = 3
    ids, sc = m.match(torch.from_numpy(q).cuda(), k)
    ids, sc = ids.cpu().numpy(), sc.cpu().numpy()
    qn = M.l2_normalize(q.astype(np.float64))
    best_s = np.full((256, k + 1), -np.inf)
    best_i = np.full((256, k + 1), -1, np.int64)
    for b0 in range(0, 100_000, 12_500):  # blockwise fp64 scores, running top k+1 (score desc, id asc)
        s = qn @ M.l2_normalize(bank[b0 : b0 + 12_500].astype(np.float64)).T
        cs = np.concatenate([best_s, s], 1)
        ci = np.concatenate([best_i, np.broadcast_to(np.arange(b0, b0 + s.shape[1]), s.shape)], 1)
        order = np.lexsort((ci, -cs), axis=1)[:, : k + 1]
        best_s, best_i = np.take_along_axis(cs, order, 1), np.take_along_axis(ci, order, 1)
    margin = (best_s[:, :-1] - best_s[:, 1:]).min(1)
    safe = margin > 1e-6
    assert safe.mean() > 0.95
    assert (ids[safe] == best_i[safe, :k]).all(), "top-k ids differ from the fp64 oracle outside near-ties"
    assert (ids[:128, 0] == pick).all()
    assert np.abs(sc - best_s[:, :k]).max() < 2e-6


def test_exact_ties_in_the_fused_epilogue():
    """>= 128 queries go through the fused per-wave top-k epilogue (EPI 16): duplicated bank rows that sit in different
    wave column ranges (96 columns) and different 192-column tiles must come back in ascending id order."""
    from mtgv.matcher import Matcher

    rng = np.random.default_rng(5)
    d, n, b = 64, 1000, 160
    bank = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((b, d), dtype=np.float32)
    dup = [3, 97, 200, 385, 700]  # wave ranges 0 / 1 of tile 0, tiles 1, 2, 3
    for j, r in enumerate(dup):
        bank[r] = q[7] * (1.0 + j)  # same direction: identical cosine after normalisation
    m = Matcher(d, capacity=n)
    m.add(bank)
    ids, sc = m.match(torch.from_numpy(q).cuda(), 6)
    assert ids[7, :5].tolist() == dup, ids[7].tolist()
    assert (sc[7, :5] == sc[7, 0]).all() and sc[7, 5] < sc[7, 0]
    # a threshold above every other score keeps exactly the duplicates (applied on the device)
    ids_t, sc_t = m.match(torch.from_numpy(q[7:8]).cuda().repeat(130, 1), 6, threshold=float(sc[7, 0]) - 1e-3)
    assert ids_t[0].tolist() == dup + [-1] and torch.isinf(sc_t[0, 5])
    assert (ids_t == ids_t[0]).all()


def _both_paths(m, q, k, **kw):
    """(two-pass result, one-pass result) of the same query batch"""
    import os

    old = os.environ.get("MTGV_MATCH_PREPASS")
    try:
        os.environ["MTGV_MATCH_PREPASS"] = "1"
        a = m.match(q, k, **kw)
        os.environ["MTGV_MATCH_PREPASS"] = "0"
        b = m.match(q, k, **kw)
    finally:
        if old is None:
            os.environ.pop("MTGV_MATCH_PREPASS", None)
        else:
            os.environ["MTGV_MATCH_PREPASS"] = old
    torch.cuda.synchronize()
    return a, b


@pytest.mark.parametrize("nq,k", [(256, 1), (1024, 3), (131, 4)])
def test_two_pass_match_equals_one_pass(nq, k):
    """fp16 first pass + exact re-rank (match.hip) against the one-pass exact kernel on the BASELINE bank: identical ids,
    scores equal to fp32 rounding (float64 re-rank vs f32 MFMA accumulation)"""
    from mtgv.matcher import Matcher

    g = torch.Generator(device="cuda").manual_seed(3)
    bank = torch.randn((100_000, 768), generator=g, device="cuda")
    m = Matcher(768, capacity=100_000)
    m.add(bank)
    q = torch.randn((nq, 768), generator=g, device="cuda")
    pick = torch.randint(0, 100_000, (nq // 2,), generator=g, device="cuda")
    q[: nq // 2] = bank[pick] + 0.5 * torch.randn((nq // 2, 768), generator=g, device="cuda")
    (ia, sa), (ib, sb) = _both_paths(m, q, k)
    os.environ["MTGV_MATCH_PREPASS"] = "0"
    try:
        _, s1 = m.match(q, k + 1)  # near-ties: a gap below fp32 accumulation noise anywhere among the best k + 1
    finally:
        os.environ.pop("MTGV_MATCH_PREPASS", None)
    safe = (s1[:, :-1] - s1[:, 1:]).min(1).values > 1e-6
    assert safe.float().mean() > 0.95
    assert (ia[safe] == ib[safe]).all()
    assert (ia[: nq // 2, 0] == pick).all()
    assert (sa - sb).abs().max().item() < 2e-6
    # thresholds act on the exact scores in both paths
    thr = float(sb[:, 0].median())
    (it, st), (iu, su) = _both_paths(m, q, k, threshold=thr)
    clear = (sb[:, 0] - thr).abs() > 1e-5
    assert ((it[:, 0] >= 0) == (iu[:, 0] >= 0))[clear].all()
    assert (it[:, 0][clear] == iu[:, 0][clear]).all()


def test_two_pass_match_falls_back_on_clustered_banks():
    """more near-duplicates of the best match than the first pass re-ranks: the bound cannot prove the answer, the block
    scans the bank exactly - ids still equal the one-pass path's (ties by ascending id)"""
    from mtgv.matcher import Matcher

    g = torch.Generator(device="cuda").manual_seed(4)
    bank = torch.randn((8192, 768), generator=g, device="cuda")
    q = torch.randn((130, 768), generator=g, device="cuda")
    dup = torch.arange(100, 100 + 24 * 200, 200, device="cuda")  # 24 rows spread over the column tiles
    bank[dup] = q[5] * 2.0 + 1e-4 * torch.randn((24, 768), generator=g, device="cuda")  # within the fp16 noise of each other
    bank[7000] = q[9]
    m = Matcher(768, capacity=8192)
    m.add(bank)
    (ia, sa), (ib, sb) = _both_paths(m, q, 3)
    assert set(ia[5].tolist()) <= set(dup.tolist()) and ia[9, 0].item() == 7000
    # the duplicates' exact scores are separated by ~1e-8: compare as sets where the one-pass scores tie within rounding
    assert (sa - sb).abs().max().item() < 2e-6
    rows = [i for i in range(130) if i != 5]
    assert (ia[rows] == ib[rows]).all()


def test_two_pass_exact_duplicates_inside_one_range_tie_by_id():
    """exact duplicates of the best row inside ONE 96-column wave range, more of them than the range reports (PRE_KP = 2):
    the unreported duplicate with the lowest id has an exact score EQUAL to the k-th best, so only the inclusive forms of
    the proof / re-scoring comparisons (match.hip rerank_kernel) return it - as the one-pass path does (score desc, id asc)"""
    from mtgv.matcher import Matcher

    g = torch.Generator(device="cuda").manual_seed(6)
    bank = torch.randn((8192, 768), generator=g, device="cuda")
    q = torch.randn((128, 768), generator=g, device="cuda")
    for qi, rows in ((3, [960 + 5, 960 + 17, 960 + 40, 960 + 95]), (77, [4800 + 90, 4800 + 2, 4800 + 50])):  # 960 = 10 * 96, 4800 = 50 * 96
        bank[rows] = q[qi]
    m = Matcher(768, capacity=8192)
    m.add(bank)
    for k in (1, 2, 3):
        (ia, sa), (ib, sb) = _both_paths(m, q, k)
        assert ia[3].tolist() == [965, 977, 1000][:k] and ia[77].tolist() == [4802, 4850, 4890][:k]
        assert (ia == ib).all()
        assert (sa - sb).abs().max().item() < 2e-6


def test_small_bank_single_query():
    """one query against a 300-row, 768-d bank with k = 3 (tools/debug/launch_fail_probe.py, the round-3 reproducer of a
    failing launch) - the convert-on-load top-k path with a single 128-row tile of which one row is valid"""
    from mtgv.matcher import Matcher

    rng = np.random.default_rng(5)
    bank = rng.standard_normal((300, 768)).astype(np.float32)
    m = Matcher(768, capacity=512)
    m.add(bank)
    q = np.ones((1, 768), np.float32)
    ids, sc = m.match(q[0], 3)
    _check(ids, sc, q, bank, 3)


def test_score_threshold_small_batch():
    from mtgv.matcher import Matcher, merge_topk

    d = 8
    e0 = np.eye(d, dtype=np.float32)[0]
    e1 = np.eye(d, dtype=np.float32)[1]
    bank = np.stack([e0, (e0 + e1) / np.sqrt(2), e1, -e0]).astype(np.float32)
    m = Matcher(d, capacity=4)
    m.add(bank)
    ids, sc = m.match(e0, 4, threshold=0.5)
    assert ids[0].tolist() == [0, 1, -1, -1] and torch.isinf(sc[0, 2:]).all()
    ids, sc = m.match(e0, 4, threshold=None)
    assert ids[0].tolist() == [0, 1, 2, 3]
    cs = torch.tensor([[0.9, 0.2, 0.7]], device="cuda")
    ci = torch.tensor([[5, 6, 7]], device="cuda")
    mi, ms = merge_topk(cs, ci, 3, threshold=0.6)
    assert mi[0].tolist() == [5, 7, -1]
    with pytest.raises(AssertionError):
        m.match(e0, 1, threshold=float("nan"))


def test_errors():
    from mtgv.matcher import Matcher

    with pytest.raises(AssertionError):
        Matcher(770 + 1)  # dim must be a multiple of 4
    m = Matcher(8, capacity=4)
    with pytest.raises(RuntimeError):
        m.match(np.ones(8, np.float32), 1)  # empty bank
    m.add(np.ones((4, 8), np.float32))
    with pytest.raises(AssertionError):
        m.add(np.ones((1, 8), np.float32))  # over capacity
    with pytest.raises(AssertionError):
        m.match(np.ones(8, np.float32), 0)
