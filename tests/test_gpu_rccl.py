"""The RCCL code path on a one-GPU box: bench.py under torch.distributed.run with ONE rank, backend nccl and
MTGV_FORCE_COLLECTIVE=1 runs the sharded match (all-gather of the queries, all-gather of the packed (id, score)
candidates, merge) through RCCL on the device.  Its ids must equal the non-distributed path's."""
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


_CHILD = r"""
import os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "mtg-vision_amd")]
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from mtgv import dist as mdist
from mtgv.matcher import Matcher, merge_topk
g = torch.Generator(device="cuda").manual_seed(11)
bank = torch.randn((5003, 768), generator=g, device="cuda")
q = torch.randn((37, 768), generator=g, device="cuda")
m = Matcher(768, capacity=5003)
m.add(bank)
ref_i, ref_s = m.match(q, 3)
calls = []
orig = mdist._all_gather_cat
def spy(x, group=None):
    calls.append(tuple(x.shape))
    assert x.is_cuda
    return orig(x, group)
mdist._all_gather_cat = spy
ids, sc = mdist.sharded_topk(q, 3, m.match, merge_topk)
torch.cuda.synchronize()
assert len(calls) == 2, calls            # queries + one packed (id, score) message
assert (ids == ref_i).all() and (sc == ref_s).all()
from mtgv.matcher import merge_gathered
del calls[:]
ids, sc = mdist.sharded_topk(q, 3, m.match, merge_topk, local_topk_packed=m.match_packed, merge_gathered=merge_gathered)
torch.cuda.synchronize()
assert calls == [(37, 768), (37, 3, 2)], calls   # the lean form: the library writes and reads the exchange format itself
assert (ids == ref_i).all() and (sc == ref_s).all()
dist.barrier()
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK", calls)
"""


def test_sharded_topk_through_rccl_one_rank(tmp_path):
    script = tmp_path / "child.py"
    script.write_text(_CHILD % {"root": ROOT})
    env = dict(os.environ, MTGV_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "RCCL_ONE_RANK_OK" in r.stdout


def test_bench_one_rank_nccl_forced_collective():
    env = dict(os.environ, MTGV_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--frames", "4", "--bank", "20000", "--encoder", "cnvnxt2ae_nano", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["n_gpus"] == 1 and res["value"] > 0
    assert "row-sharded 1-way" in res["config"]["bank_layout"]
