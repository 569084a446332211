"""GPU parity of the bank top-k against the brute-force oracle (ids exact; near-ties reported, none tolerated silently)."""
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
