#!/usr/bin/env python3
"""Where do the two match paths disagree?  python tools/debug/prepass_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mtg-vision_amd")]
import torch
from mtgv.matcher import Matcher
nq, k = 1024, 3
g = torch.Generator(device="cuda").manual_seed(3)
bank = torch.randn((100_000, 768), generator=g, device="cuda")
m = Matcher(768, capacity=100_000); m.add(bank)
q = torch.randn((nq, 768), generator=g, device="cuda")
pick = torch.randint(0, 100_000, (nq // 2,), generator=g, device="cuda")
q[: nq // 2] = bank[pick] + 0.5 * torch.randn((nq // 2, 768), generator=g, device="cuda")
os.environ["MTGV_MATCH_PREPASS"] = "1"; ia, sa = m.match(q, k)
os.environ["MTGV_MATCH_PREPASS"] = "0"; ib, sb = m.match(q, k)
torch.cuda.synchronize()
bad = (ia != ib).any(1).nonzero().flatten()
print("rows differing:", bad.numel())
bn = torch.nn.functional.normalize(bank, dim=1); qn = torch.nn.functional.normalize(q, dim=1)
for r in bad[:6].tolist():
    s = (qn[r].double() @ bn.double().T)
    top = torch.topk(s, 10)
    print(r, "two-pass", ia[r].tolist(), sa[r].tolist(), "| one-pass", ib[r].tolist(), sb[r].tolist())
    print("   fp64 top10", top.indices.tolist(), [round(x, 6) for x in top.values.tolist()])
