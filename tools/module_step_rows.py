#!/usr/bin/env python3
"""The Llama-7B decode step through the unchanged module tree at 1 / 16 / 17 / 24 / 32 rows (a batch of that many sequences), fp16 and bf16: us per block with the sibling grouping
   on and off, under the process environment (BIE_LUT_RB2=0: the round-5 routing beyond 16 rows -- every call a fused GEMM + finalize)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import Bench, BF16  # noqa: E402

B = Bench(torch.device("cuda", 0))
out = {"BIE_LUT_RB2": os.environ.get("BIE_LUT_RB2", "1")}
for dn, dt in (("f16", torch.float16), ("bf16", BF16)):
    for rows in (1, 16, 17, 24, 32):
        for ag in (True, False):
            r = B.decode_step_modules(8, 6, 3, auto_group=ag, rows=rows, dt=dt)
            out[f"{dn}_M{rows}_{'grouped' if ag else 'alone'}"] = [r["us_per_layer"], r["launches_per_layer"]]
print(json.dumps(out))
