"""Bank build row: make_cropped kernel vs oracle, batched build vs per-card oracle, persistence."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cards(rng, n):
    sizes = [(680, 488), (1040, 745), (936, 672), (204, 146), (192 + 8, 128 + 8)]
    return [(f"00000000-0000-0000-0000-{i:012d}", rng.integers(0, 256, (*sizes[i % len(sizes)], 3), dtype=np.uint8)) for i in range(n)]


def test_make_cropped_matches_oracle():
    from mtgv.bank import make_cropped
    from oracle import resize_ref as R

    rng = np.random.default_rng(3)
    cards = _cards(rng, 7)
    out = make_cropped([im for _, im in cards], (192, 128)).cpu().numpy()
    assert out.shape == (7, 192, 128, 3) and out.dtype == np.float32
    for i, (_, im) in enumerate(cards):
        ref = R.make_cropped(im, (192, 128))
        assert np.abs(out[i] - ref).max() < 1e-6, i
    assert (out >= 0).all() and (out <= 1).all()
    # a 200x136 card loses a 4-pixel border and is copied 1:1
    im = cards[4][1]
    np.testing.assert_allclose(out[4], im[4:-4, 4:-4].astype(np.float32) / 255.0, atol=1e-7)
    assert make_cropped([], (192, 128)).shape == (0, 192, 128, 3)


def test_build_bank_and_persistence(tmp_path):
    from mtgv import spec
    from mtgv.adapters import VectorStoreQdrant
    from mtgv.bank import build_bank, load_store, save_store
    from mtgv.encoder import Encoder
    from oracle import encoder_ref as E
    from oracle import resize_ref as R

    cfg = spec.encoder_config("cnvnxt2ae_nano")
    sd = spec.random_encoder_state(cfg, 1)
    enc = Encoder(cfg, sd, max_batch=8)
    rng = np.random.default_rng(4)
    cards = _cards(rng, 11)
    store = VectorStoreQdrant(capacity=64)
    assert build_bank(cards, enc, store, batch_size=4) == 11
    assert build_bank(cards, enc, store, batch_size=4) == 0  # already present: nothing re-embedded
    # every stored vector is the normalised oracle embedding of the oracle crop
    x = np.stack([R.make_cropped(im, (192, 128)) for _, im in cards[:4]]).transpose(0, 3, 1, 2)
    z = E.encoder_forward(sd, cfg, x).numpy()
    zn = z / np.linalg.norm(z, axis=1, keepdims=True)
    got = np.asarray([store.retrieve([cid], with_vectors=True)[0].vector for cid, _ in cards[:4]])
    assert np.abs(got - zn).max() < 1e-5
    # a card retrieves itself
    q = enc.encode(torch.from_numpy(x)).cpu().numpy()[2]
    assert store.query_nearby(q, k=1)[0].id == cards[2][0]
    # persistence round trip
    store.update_payload(cards[5][0], {"name": "Black Lotus"})
    save_store(store, str(tmp_path / "bank"))
    again = load_store(str(tmp_path / "bank"), capacity=64)
    assert [p.id for p in again.query_nearby(q, k=3)] == [p.id for p in store.query_nearby(q, k=3)]
    assert again.retrieve([cards[5][0]])[0].payload == {"name": "Black Lotus"}
