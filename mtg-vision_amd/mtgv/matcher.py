"""Host side of the match stage: exact cosine top-k over the whole card bank.

`Matcher.match(embedding, k)` is the north-star name.  The reference goes through
`VectorStoreQdrant.query_nearby` (mtgvision/qdrant.py:76-95; collection of 768-d
vectors with Distance.COSINE) - see `mtgv.adapters.VectorStoreQdrant` for that
signature.  All arithmetic is in libmtgv.so (match.hip / gemm_f32.hip).
"""

from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple, Union

import numpy as np
import torch

from . import native


class Matcher:
    """Device-resident bank of L2-normalised vectors (rows = local ids 0..size-1).

    `id_base` is added to every returned id: with the bank sharded by rows over
    ranks, rank r passes the global index of its first row (mtgv.dist).
    """

    def __init__(self, dim: int = 768, capacity: int = 131072, id_base: int = 0, device=None):
        native.require_gpu()
        self.dim = int(dim)
        self.capacity = int(capacity)
        self.id_base = int(id_base)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._h = native.c_vp(0)
        with torch.cuda.device(self.device):
            native.check(native.lib().mtgv_bank_create(self.dim, self.capacity, C.byref(self._h)))

    def __len__(self) -> int:
        return int(native.lib().mtgv_bank_size(self._h))

    def add(self, vectors: Union[np.ndarray, torch.Tensor]) -> range:
        """Append rows; returns the local row range they occupy.  Vectors are normalised on the GPU."""
        start = len(self)
        if isinstance(vectors, np.ndarray):
            vectors = torch.from_numpy(np.ascontiguousarray(vectors, dtype=np.float32))
        vectors = vectors.to(torch.float32)
        if vectors.ndim == 1:
            vectors = vectors[None]
        assert vectors.ndim == 2 and vectors.shape[1] == self.dim, f"{tuple(vectors.shape)}"
        vectors = vectors.contiguous()
        with torch.cuda.device(self.device):
            native.check(
                native.lib().mtgv_bank_append(
                    self._h, native.ptr(vectors), vectors.shape[0], 1 if vectors.is_cuda else 0, native.stream()
                )
            )
            if vectors.is_cuda:
                torch.cuda.current_stream().synchronize()
        return range(start, start + vectors.shape[0])

    def set_row(self, row: int, vector):
        v = np.ascontiguousarray(np.asarray(vector, dtype=np.float32).reshape(-1))
        assert v.size == self.dim
        with torch.cuda.device(self.device):
            native.check(native.lib().mtgv_bank_set_row(self._h, int(row), v.ctypes.data_as(native.c_vp), native.stream()))

    def rows(self, start: int, n: int) -> np.ndarray:
        out = np.empty((n, self.dim), np.float32)
        with torch.cuda.device(self.device):
            native.check(native.lib().mtgv_bank_get_rows(self._h, int(start), int(n), out.ctypes.data_as(native.c_vp)))
        return out

    def clear(self):
        native.check(native.lib().mtgv_bank_clear(self._h))

    def match(self, embedding, k: int = 1, threshold: Optional[float] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """(D,) or (B,D) raw embeddings -> (ids int64 (B,k), scores float32 (B,k)) on the GPU.

        Sorted by cosine score descending, ties by ascending id; id -1 / score -inf pads
        when the bank holds fewer than k rows.  `threshold` is `score_threshold` of
        `query_nearby` (mtgvision/qdrant.py:83,93): hits scoring below it are dropped on the
        device (they become pads)."""
        if isinstance(embedding, np.ndarray) or isinstance(embedding, (list, tuple)):
            embedding = torch.from_numpy(np.ascontiguousarray(np.asarray(embedding, dtype=np.float32)))
        q = embedding.to(self.device, torch.float32)
        if q.ndim == 1:
            q = q[None]
        assert q.ndim == 2 and q.shape[1] == self.dim, f"{tuple(q.shape)}"
        q = q.contiguous()
        b = q.shape[0]
        ids = torch.empty((b, k), dtype=torch.int64, device=self.device)
        scores = torch.empty((b, k), dtype=torch.float32, device=self.device)
        if b == 0:
            return ids, scores
        with torch.cuda.device(self.device):
            native.check(
                native.lib().mtgv_bank_topk(self._h, native.ptr(q), b, int(k), self.id_base, _thr(threshold), native.ptr(ids), native.ptr(scores),
                                            native.stream())
            )
        return ids, scores

    def match_packed(self, q: torch.Tensor, k: int = 1) -> torch.Tensor:
        """The local top-k in the sharded match's exchange format (mtgv.dist, include/mtgv.h): q (B, D) float32 on the
        device, contiguous -> (B, k, 2) int64 = (global id or -1, float32 bit pattern of the score).  One library call,
        no other kernel: what one all-gather then carries to every rank."""
        assert q.is_cuda and q.dtype == torch.float32 and q.ndim == 2 and q.shape[1] == self.dim and q.is_contiguous(), f"{tuple(q.shape)} {q.dtype}"
        b = q.shape[0]
        packed = torch.empty((b, k, 2), dtype=torch.int64, device=self.device)
        if b:
            with torch.cuda.device(self.device):
                native.check(native.lib().mtgv_bank_topk_packed(self._h, native.ptr(q), b, int(k), self.id_base, native.ptr(packed), native.stream()))
        return packed

    def prepass_fallbacks(self) -> int:
        """Queries of two-pass matches (>= 128 queries per call) that had to be scanned exactly so far (include/mtgv.h)."""
        v = native.c_i64(0)
        native.check(native.lib().mtgv_bank_prepass_fallbacks(self._h, C.byref(v)))
        return int(v.value)

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                native.lib().mtgv_bank_destroy(self._h)
                self._h = native.c_vp(0)
        except Exception:
            pass


def _thr(threshold: Optional[float]) -> float:
    return float("-inf") if threshold is None else float(threshold)


def merge_topk(cand_scores: torch.Tensor, cand_ids: torch.Tensor, k: int, threshold: Optional[float] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge (B, ncand) candidate (score, id) pairs - e.g. the all-gathered per-shard top-k -
    into the global top-k with the same (score desc, id asc) order.  Consumes cand_scores."""
    assert cand_scores.is_cuda and cand_ids.is_cuda and cand_scores.shape == cand_ids.shape and cand_scores.ndim == 2
    cs = cand_scores.to(torch.float32).contiguous().clone()
    ci = cand_ids.to(torch.int64).contiguous()
    b, n = cs.shape
    ids = torch.empty((b, k), dtype=torch.int64, device=cs.device)
    scores = torch.empty((b, k), dtype=torch.float32, device=cs.device)
    with torch.cuda.device(cs.device):
        native.check(native.lib().mtgv_topk_merge(native.ptr(cs), native.ptr(ci), b, n, int(k), _thr(threshold), native.ptr(ids), native.ptr(scores), native.stream()))
    return ids, scores


def merge_gathered(gathered: torch.Tensor, row0: int, b: int, k: int, threshold: Optional[float] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge the all-gathered per-shard candidates (R, B_total, k, 2) int64 in the exchange format of
    `Matcher.match_packed` for the queries [row0, row0 + b): global (ids (b, k) int64, scores (b, k) float32), score desc
    then id asc.  One library kernel reads the collective's output buffer in place."""
    assert gathered.is_cuda and gathered.dtype == torch.int64 and gathered.ndim == 4 and gathered.shape[2] == k and gathered.shape[3] == 2
    assert gathered.is_contiguous()
    R, b_total = gathered.shape[0], gathered.shape[1]
    ids = torch.empty((b, k), dtype=torch.int64, device=gathered.device)
    scores = torch.empty((b, k), dtype=torch.float32, device=gathered.device)
    if b:
        with torch.cuda.device(gathered.device):
            native.check(native.lib().mtgv_topk_merge_gathered(native.ptr(gathered), R, b_total, int(k), int(row0), int(b), _thr(threshold),
                                                               native.ptr(ids), native.ptr(scores), native.stream()))
    return ids, scores
