#!/usr/bin/env python3
"""W4A16 decode of M = 2...16 rows: one list launch over 40 x 4096x11008 (96 x 4096x4096) layers against one launch per layer.
   BIE_LUTM_LIST_PF=0: no next-unit prefetch in the list kernel (two workgroups per CU instead of one).
   usage: python tools/lutm_list_ab.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd")); sys.path.insert(0, ROOT)
import torch
from bench import Bench
B = Bench(torch.device("cuda", 0))
out = {"env": {k: os.environ.get(k) for k in ("BIE_LUTM_LIST_PF",) if os.environ.get(k)}}
for (k, n, nl) in ((4096, 11008, 40), (4096, 4096, 96)):
    for M in (2, 8):
        r = B.gemv_list(k, n, nl, nl, 10, 100 + M, M=M)
        row = {"list_us_per_layer": r["us_per_layer"], "list_frac": r["roofline"]["frac"]}
        if M > 1:
            p = B.gemv(k, n, nl, 10, 200 + M, M=M)
            row.update({"per_layer_launch_us": p["us_per_launch"], "per_layer_frac": p["roofline"]["frac"]})
        out[f"{k}x{n}_M{M}"] = row
print(json.dumps(out))
