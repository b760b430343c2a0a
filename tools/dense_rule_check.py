#!/usr/bin/env python3
"""The dense form's dispatch rule on shapes outside the round-3 sweep (Llama-3 8B / 70B projections): fused (BIE_GEMM_DENSE=0) against dense forced (=2) and the rule's own choice (=1)."""
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
for (k, n) in ((4096, 14336), (14336, 4096), (4096, 6144), (8192, 10240), (5120, 13824), (4096, 4096), (4096, 11008)):
    for M in (1024, 2048, 4096):
        r = {"K": k, "N": n, "M": M}
        for name, v in (("fused", "0"), ("dense", "2"), ("rule", "1")):
            os.environ["BIE_GEMM_DENSE"] = v
            r[name] = B.gemm(M, k, n, 6, 6, 5)["us_per_launch"]
        os.environ.pop("BIE_GEMM_DENSE", None)
        print(json.dumps(r), flush=True)
