"""Multi-GPU layout: one process per GPU, frames data-parallel, bank sharded by rows.

The only exchange step of the path is the match: every query must see every bank row.
With the bank split by rows over R ranks (rank r holds rows [r*N/R, (r+1)*N/R), global id =
local row + offset), top-k of the union = merge of per-shard top-k:

    1. all_gather the local query embeddings        (B_local x D fp32 each)
    2. every rank scores ALL queries against its shard, local top-k (ids already global)
    3. all_gather the (id, score) candidates, packed into ONE int64 buffer (B_total x k x 2 per rank)
    4. every rank merges the R*k candidates of its OWN queries (score desc, id asc)

Both messages are KB-scale: latency-bound, xGMI bandwidth is irrelevant, so the step has exactly two
collectives (RCCL `all_gather_into_tensor`, backend "nccl" on ROCm).  With `local_topk_packed` / `merge_gathered`
(`Matcher.match_packed`, `matcher.merge_gathered`) step 2 writes the exchange format itself and step 4 reads the
gathered buffer in place: the exchange then launches NOTHING but two library kernels and RCCL's two all-gathers - no
PyTorch arithmetic - which is what lets it run on one of the two streams of `Pipeline.run_many` (include/mtgv.h,
concurrency contract).  MTGV_FORCE_COLLECTIVE=1 runs them at
world size 1 too (a one-GPU box can then exercise the RCCL path: tests/test_gpu_dist.py).  The local top-k and the merge
are pluggable so the collective logic is exercised on CPU with gloo (tests/test_dist_cpu.py).
"""

from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_rows(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous row range [start, stop) of `rank`; remainders go to the first ranks"""
    base, rem = divmod(n_rows, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_frames(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    return shard_rows(n_frames, rank, world)


def _all_gather_cat(x: torch.Tensor, group=None) -> torch.Tensor:
    world = dist.get_world_size(group)
    out = torch.empty((world, *x.shape), dtype=x.dtype, device=x.device)
    try:
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    except (RuntimeError, NotImplementedError):  # backends without the fused form
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x.contiguous(), group=group)
        out = torch.stack(parts)
    return out


def sharded_topk(q_local: torch.Tensor, k: int, local_topk: Callable, merge: Callable, group=None, *,
                 local_topk_packed: Optional[Callable] = None, merge_gathered: Optional[Callable] = None):
    """q_local (B_local, D) -> global (ids (B_local, k) int64, scores (B_local, k)) for this rank's queries.

    local_topk(q (B,D), k) -> (ids int64 (B,k) GLOBAL ids, scores (B,k)) over this rank's bank shard.
    merge(cand_scores (B, R*k), cand_ids (B, R*k), k) -> (ids, scores).
    Lean form (both given): local_topk_packed(q (B,D), k) -> (B, k, 2) int64 (id, float32 score bits);
    merge_gathered(gathered (R, B_total, k, 2), row0, b, k) -> (ids, scores) - no tensor arithmetic in between.
    Every rank must pass the same B_local."""
    force = os.environ.get("MTGV_FORCE_COLLECTIVE") == "1"
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return local_topk(q_local, k)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    b_local = q_local.shape[0]
    if local_topk_packed is not None and merge_gathered is not None:
        q_all = _all_gather_cat(q_local, group).view(world * b_local, -1)  # a view of the collective's output
        packed = local_topk_packed(q_all, k)
        return merge_gathered(_all_gather_cat(packed, group), rank * b_local, b_local, k)
    q_all = _all_gather_cat(q_local, group).reshape(world * b_local, -1)
    ids, scores = local_topk(q_all, k)
    # one message: [..., 0] = id, [..., 1] = the score's float32 bit pattern
    packed = torch.stack((ids.to(torch.int64), scores.to(torch.float32).contiguous().view(torch.int32).to(torch.int64)), dim=-1)
    allp = _all_gather_cat(packed.contiguous(), group)  # (R, B_total, k, 2)
    mine = allp[:, rank * b_local : (rank + 1) * b_local].permute(1, 0, 2, 3).reshape(b_local, world * k, 2)
    cand_i = mine[..., 0].contiguous()
    cand_s = mine[..., 1].to(torch.int32).contiguous().view(torch.float32)
    return merge(cand_s, cand_i, k)
