#!/usr/bin/env python3
"""Which hardware queue did each kind of kernel run on?  python tools/debug/queue_map.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict, Counter
rows = list(csv.DictReader(open(sys.argv[1])))
rows = rows[len(rows) // 2:]   # the second half of the trace: steady state
def kind(n):
    for k, t in (("mlp_fused", "embed"), ("dwconv7", "embed"), ("grn_finalize", "embed"), ("conv0_u8", "detect"), ("nms_kernel", "detect-tail"),
                 ("decode_kernel", "detect-tail"), ("mask_quads", "crop"), ("warp_kernel", "crop"), ("select_cards", "crop"), ("rerank", "match"),
                 ("topk_merge", "match"), ("l2norm", "match"), ("sppf", "detect"), ("upsample2x", "detect"), ("mask_logits", "detect-tail")):
        if k in n: return t
    return "gemm/other"
q = defaultdict(Counter)
for r in rows: q[(r.get("Queue_Id"), r.get("Stream_Id"))][kind(r["Kernel_Name"])] += 1
for k, c in sorted(q.items(), key=lambda kv: -sum(kv[1].values())):
    print("queue", k[0], "stream", k[1], dict(c))
