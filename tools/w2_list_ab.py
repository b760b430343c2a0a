#!/usr/bin/env python3
"""W2A16 one-row layer lists (96 x 4096x4096, 40 x 4096x11008, 40 x 11008x4096; bf16 and fp16) under the process environment (BIE_LIST_W2_NW = 8 | 4): us per layer, HBM fraction."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import Bench, BF16  # noqa: E402

B = Bench(torch.device("cuda", 0))
out = {"BIE_LIST_W2_NW": os.environ.get("BIE_LIST_W2_NW", "8")}
for dn, dt in (("bf16", BF16), ("f16", torch.float16)):
    for (k, n, nl) in ((4096, 4096, 96), (4096, 11008, 40), (11008, 4096, 40)):
        r = B.gemv_list(k, n, nl, nl, 10, 44, w_bit=2, dt=dt)
        out[f"{dn}_{k}x{n}"] = [r["us_per_layer"], r["roofline"]["frac"]]
print(json.dumps(out))
