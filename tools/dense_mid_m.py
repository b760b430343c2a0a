#!/usr/bin/env python3
"""Fused MFMA kernel (default below 1024 rows) against the dense two-launch form forced (BIE_GEMM_DENSE=2: dequantise pass + dense GEMM, no K split on this path) at 128..1023 rows."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
os.environ["BIE_TUNING"] = "1"
import torch  # noqa: E402
from bench import Bench  # noqa: E402

B = Bench(torch.device("cuda", 0))
for (k, n) in ((4096, 11008), (4096, 4096), (11008, 4096)):
    for M in (128, 256, 384, 512, 768, 1023):
        os.environ.pop("BIE_GEMM_DENSE", None)
        a = B.gemm(M, k, n, 8, 8, 5)["us_per_launch"]
        os.environ["BIE_GEMM_DENSE"] = "2"
        b = B.gemm(M, k, n, 8, 8, 5)["us_per_launch"]
        os.environ.pop("BIE_GEMM_DENSE", None)
        print(json.dumps({"K": k, "N": n, "M": M, "fused_us": a, "dense_us": b}), flush=True)
