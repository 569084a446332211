"""CPU checks of the match oracle (no reference fixture exists for this stage: parity unpinned)."""
import numpy as np

from oracle import match_ref as M


def test_topk_order_and_ties():
    bank = np.array([[1, 0], [0, 1], [1, 0], [-1, 0], [1, 1]], np.float32)
    q = np.array([[2, 0]], np.float32)
    ids, sc = M.cosine_topk(q, bank, 3)
    assert ids.tolist() == [[0, 2, 4]]  # tie 0/2 broken by id
    np.testing.assert_allclose(sc[0], [1, 1, 2**-0.5], atol=1e-12)
    ids, sc = M.cosine_topk(q, bank, 5, score_threshold=0.5)
    assert ids.tolist() == [[0, 2, 4, -1, -1]]


def test_matches_bruteforce_random():
    rng = np.random.default_rng(2)
    bank = rng.standard_normal((1000, 768)).astype(np.float32)
    q = rng.standard_normal((4, 768)).astype(np.float32)
    ids, sc = M.cosine_topk(q, bank, 5)
    s = M.scores(q, bank)
    for i in range(4):
        assert ids[i, 0] == int(np.argmax(s[i]))
        assert np.all(np.diff(sc[i]) <= 0)
    ids32, _ = M.cosine_topk(q, bank, 5, dtype=np.float32)
    assert (ids32 == ids).all()
