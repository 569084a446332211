"""world_size-2 gloo test of the sharded match exchange (mtgv.dist) on CPU.

The per-shard top-k and the merge are the oracle's (the HIP ones need a GPU); what is under test is
the partitioning, the two all-gathers and the candidate bookkeeping."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_bank, b_local, k, ret):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "mtg-vision_amd")]
    from mtgv import dist as mdist
    from oracle import match_ref as M

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(2)
        bank = rng.standard_normal((n_bank, 64)).astype(np.float32)
        q_all = rng.standard_normal((world * b_local, 64)).astype(np.float32)
        lo, hi = mdist.shard_rows(n_bank, rank, world)
        shard = bank[lo:hi]

        def local_topk(q, kk):
            ids, sc = M.cosine_topk(q.numpy(), shard, kk, dtype=np.float32)
            ids = np.where(ids >= 0, ids + lo, -1)
            return torch.from_numpy(ids), torch.from_numpy(sc.astype(np.float32))

        def merge(cs, ci, kk):
            cs, ci = cs.numpy(), ci.numpy()
            out_i = np.full((cs.shape[0], kk), -1, np.int64)
            out_s = np.full((cs.shape[0], kk), -np.inf, np.float32)
            for r in range(cs.shape[0]):
                valid = ci[r] >= 0
                order = np.lexsort((ci[r][valid], -cs[r][valid]))[:kk]
                out_i[r, : len(order)] = ci[r][valid][order]
                out_s[r, : len(order)] = cs[r][valid][order]
            return torch.from_numpy(out_i), torch.from_numpy(out_s)

        # the lean form of the exchange (include/mtgv.h: mtgv_bank_topk_packed / mtgv_topk_merge_gathered), restated in numpy:
        # the local top-k writes (id, float32 score bits) pairs, the merge reads the gathered (R, B_total, k, 2) buffer
        def local_topk_packed(q, kk):
            ids, sc = local_topk(q, kk)
            bits = sc.numpy().astype(np.float32).view(np.uint32).astype(np.int64)
            return torch.from_numpy(np.stack([ids.numpy().astype(np.int64), bits], -1))

        def merge_gathered(g, row0, b, kk):
            g = g.numpy()
            assert g.shape[1:] == (world * b_local, kk, 2)
            mine = g[:, row0 : row0 + b].transpose(1, 0, 2, 3).reshape(b, -1, 2)
            cs = mine[..., 1].astype(np.uint32).view(np.float32)
            return merge(torch.from_numpy(cs.copy()), torch.from_numpy(mine[..., 0].copy()), kk)

        q_local = torch.from_numpy(q_all[rank * b_local : (rank + 1) * b_local])
        ref_i, ref_s = M.cosine_topk(q_all[rank * b_local : (rank + 1) * b_local], bank, k, dtype=np.float32)
        ok = True
        for lean in (False, True):
            kw = dict(local_topk_packed=local_topk_packed, merge_gathered=merge_gathered) if lean else {}
            ids, sc = mdist.sharded_topk(q_local, k, local_topk, merge, **kw)
            ok = ok and bool((ids.numpy() == ref_i).all() and np.allclose(sc.numpy(), ref_s, atol=1e-6))
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_bank,b_local,k", [(1001, 5, 3), (64, 2, 1)])
def test_sharded_topk_gloo(n_bank, b_local, k):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_bank, b_local, k, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world))


def test_shard_rows_partition():
    from mtgv.dist import shard_rows

    for n, w in [(100000, 8), (10, 3), (7, 8)]:
        spans = [shard_rows(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert shard_rows(100000, 3, 8) == (37500, 50000)
