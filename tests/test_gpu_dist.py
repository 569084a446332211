"""Two-rank rehearsal of bench.py's sharded path on ONE GPU (gloo collectives on host copies, both ranks on
device 0): row-sharded bank + all-gather of per-shard top-1 must give the same ids as the replicated bank."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_sharded_bench_runs():
    env = dict(os.environ, MTGV_SHARE_GPU="1", MTGV_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
        "--frames", "4", "--bank", "20000", "--encoder", "cnvnxt2ae_nano",
    ]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["value"] > 0 and res["scaling"] == "weak"
    assert "row-sharded 2-way" in res["config"]["bank_layout"]
    assert res["roofline"]["achieved"] > 0  # the roofline leg contains a collective: it must run on every rank


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` from a plain shell (no torch.distributed.run in front, RANK unset): the parent starts
    the ranks as a child process and relays rank 0's line"""
    env = dict(os.environ, MTGV_SHARE_GPU="1", MTGV_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "4",
           "--bank", "20000", "--encoder", "cnvnxt2ae_nano", "--no-f32-roofline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["value"] > 0
    assert res["config"]["rccl_world"] == 2 and res["config"]["dist_backend"] == "gloo"


def test_two_ranks_sharded_with_overlap_equal_the_replicated_ids():
    """the N > 1 default as the driver launches it - bank row-sharded, the overlapped schedule ON, the exchange in its lean form
    (Matcher.match_packed -> all-gather -> merge_gathered: no PyTorch arithmetic on the streams) - rehearsed with two
    ranks on one GPU (gloo): rank 0's top-1 ids over the timed steps equal those of the one-rank replicated run and of
    the one-stream run."""
    base = ["--steps", "3", "--warmup", "1", "--frames", "4", "--bank", "20000", "--encoder", "cnvnxt2ae_nano", "--settle-steps", "0",
            "--no-cpu-baseline", "--no-roofline"]
    env = dict(os.environ, MTGV_SHARE_GPU="1", MTGV_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MTGV_OVERLAP"):
        env.pop(k, None)

    def run(extra):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra, *base], env=env, capture_output=True, text=True, timeout=400)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])

    two = run(["--gpus", "2"])
    assert two["n_gpus"] == 2 and "row-sharded 2-way" in two["config"]["bank_layout"] and two["config"]["streams"].startswith("3")
    one = run(["--gpus", "1"])
    assert one["config"]["bank_layout"] == "replicated" and one["config"]["streams"].startswith("3")
    serial = run(["--gpus", "1", "--no-overlap"])
    assert serial["config"]["streams"] == "1"
    assert two["config"]["ids_crc32_rank0"] == one["config"]["ids_crc32_rank0"] == serial["config"]["ids_crc32_rank0"]


def test_exchange_format_roundtrip():
    """Matcher.match_packed + merge_gathered (the exchange without PyTorch arithmetic) against match + merge_topk on three
    row shards: identical ids and scores, threshold included"""
    import torch
    from mtgv.dist import shard_rows
    from mtgv.matcher import Matcher, merge_gathered

    g = torch.Generator(device="cuda").manual_seed(8)
    bank = torch.randn((13_001, 768), generator=g, device="cuda")  # shards of >= 4096 rows: full bank and shards both take the two-pass
    q = torch.randn((140, 768), generator=g, device="cuda")        # kernel at >= 128 queries (exact float64 re-rank: identical score bits)
    full = Matcher(768, capacity=13_001)
    full.add(bank)
    for k in (1, 3):
        ids, sc = full.match(q, k)
        parts = []
        for r in range(3):
            lo, hi = shard_rows(13_001, r, 3)
            m = Matcher(768, capacity=hi - lo, id_base=lo)
            m.add(bank[lo:hi])
            p = m.match_packed(q, k)
            i2, s2 = m.match(q, k)
            assert (p[..., 0] == i2).all() and (p[..., 1].to(torch.int32).view(torch.float32) == s2).all()
            parts.append(p)
        gathered = torch.stack(parts).contiguous()
        mi, ms = merge_gathered(gathered, 0, 140, k)
        assert (mi == ids).all() and (ms == sc).all()
        mi, ms = merge_gathered(gathered, 100, 40, k)  # a rank's own slice of the queries
        assert (mi == ids[100:]).all() and (ms == sc[100:]).all()
        thr = float(sc[:, 0].median())
        mi, ms = merge_gathered(gathered, 0, 140, k, threshold=thr)
        ti, ts = full.match(q, k, threshold=thr)
        assert (mi == ti).all() and (ms == ts).all()
    small = full.match_packed(q[:5].contiguous(), 2)  # < 128 queries: the one-pass kernel
    i5, s5 = full.match(q[:5], 2)
    assert (small[..., 0] == i5).all() and (small[..., 1].to(torch.int32).view(torch.float32) == s5).all()


def test_sharded_equals_replicated_ids():
    """same queries against a replicated bank and against 3 row shards merged: identical ids and scores"""
    import torch
    from mtgv.dist import shard_rows
    from mtgv.matcher import Matcher, merge_topk

    g = torch.Generator(device="cuda").manual_seed(7)
    bank = torch.randn((10_007, 768), generator=g, device="cuda")
    q = torch.randn((33, 768), generator=g, device="cuda")
    full = Matcher(768, capacity=10_007)
    full.add(bank)
    ids, sc = full.match(q, 3)
    cs, ci = [], []
    for r in range(3):
        lo, hi = shard_rows(10_007, r, 3)
        m = Matcher(768, capacity=hi - lo, id_base=lo)
        m.add(bank[lo:hi])
        i, s = m.match(q, 3)
        ci.append(i)
        cs.append(s)
    mi, ms = merge_topk(torch.cat(cs, 1), torch.cat(ci, 1), 3)
    assert (mi == ids).all() and (ms == sc).all()
