#!/usr/bin/env python3
"""Fused MFMA kernel against the dense two-launch form over a grid of layer shapes and row counts (W4 g128 bf16): the data behind mpq_dense_ok's rule, round 6.
   One JSON line per (K, N, M): fused us, dense us, what the rule picks."""
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
dims = (2048, 3072, 4096, 5120, 6144, 8192, 11008, 14336)
for k in dims:
    for n in dims + (28672,):
        if k * n > 8192 * 28672:
            continue
        for M in (897, 1024, 1536, 2048, 3072, 4096):
            nl = max(2, min(6, int(6e8 // (k * n))))
            r = {"K": k, "N": n, "M": M}
            for name, v in (("fused", "0"), ("dense", "2"), ("rule", "1")):
                os.environ["BIE_GEMM_DENSE"] = v
                try:
                    r[name] = B.gemm(M, k, n, nl, 4, 5)["us_per_launch"]
                except Exception as e:
                    r[name] = None
            os.environ.pop("BIE_GEMM_DENSE", None)
            print(json.dumps(r), flush=True)
