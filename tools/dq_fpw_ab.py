#!/usr/bin/env python3
"""M = 4096 GEMM (both launches timed) under the process environment's BIE_DQ_FPW (fragments per wave of the dequantise pass): us per call.
   usage: BIE_DQ_FPW=2 python tools/dq_fpw_ab.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import Bench  # noqa: E402

B = Bench(torch.device("cuda", 0))
out = {"BIE_DQ_FPW": os.environ.get("BIE_DQ_FPW", "8")}
for (k, n, nl) in ((4096, 4096, 24), (4096, 11008, 12), (11008, 4096, 12)):
    B.gemm(4096, k, n, nl, 5, 7)
    r = B.gemm(4096, k, n, nl, 30, 7)
    out[f"{k}x{n}"] = r["us_per_launch"]
print(json.dumps(out))
