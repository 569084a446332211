"""ORACLE (test infrastructure, never shipped or measured as the product).

Brute-force cosine top-k over a vector bank: the exact answer that the
reference's `VectorStoreQdrant.query_nearby` (mtgvision/qdrant.py:76-95, Qdrant
collection with Distance.COSINE, qdrant.py:29-32) approximates with its HNSW
index.  Qdrant normalises stored and query vectors for COSINE and returns
points sorted by score descending; ties are broken here by ascending id.

PARITY UNPINNED: `qdrant_client` and the Qdrant server are third-party and absent
from /root/reference, and the reference holds no test or golden vector for this
stage.  The restated algorithm is the published definition of
cosine similarity search.
"""

from __future__ import annotations

import numpy as np


def l2_normalize(x: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    x = np.asarray(x)
    n = np.sqrt((x.astype(np.float64) ** 2).sum(-1, keepdims=True))
    return (x / np.maximum(n, eps)).astype(x.dtype)


def scores(q: np.ndarray, bank: np.ndarray, dtype=np.float64) -> np.ndarray:
    """cosine similarity matrix (B, N), inputs raw (normalised here)."""
    qn = l2_normalize(np.asarray(q, dtype)).astype(dtype)
    bn = l2_normalize(np.asarray(bank, dtype)).astype(dtype)
    return qn @ bn.T


def topk_from_scores(s: np.ndarray, k: int, score_threshold=None):
    """(ids (B,k) int64, scores (B,k)): score desc, id asc; -1 / -inf pad."""
    b, n = s.shape
    ids = np.full((b, k), -1, np.int64)
    out = np.full((b, k), -np.inf, s.dtype)
    idx = np.arange(n)
    for i in range(b):
        order = np.lexsort((idx, -s[i]))[:k]  # primary: -score asc; secondary: id asc
        if score_threshold is not None:
            order = order[s[i][order] >= score_threshold]
        ids[i, : len(order)] = order
        out[i, : len(order)] = s[i][order]
    return ids, out


def cosine_topk(q, bank, k: int, dtype=np.float64, score_threshold=None):
    return topk_from_scores(scores(q, bank, dtype), k, score_threshold)


def top_margin(s: np.ndarray, k: int) -> np.ndarray:
    """per query: smallest gap between consecutive scores among the top k+1 (tie detector)."""
    part = -np.sort(-s, axis=1)[:, : k + 1]
    return (part[:, :-1] - part[:, 1:]).min(axis=1)
