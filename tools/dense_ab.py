#!/usr/bin/env python3
"""Dense form (mpq_dense.hip, forced: BIE_GEMM_DENSE=2) against the fused kernel (=0), bie_mpq_forward per call, bf16 W4 g128, one box.
   usage: python tools/dense_ab.py"""
import json, os, sys
os.environ["BIE_TUNING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd")); sys.path.insert(0, ROOT)
import sweep
from bitorch_engine import _hip
for (K, N) in ((4096, 4096), (4096, 11008), (11008, 4096), (8192, 28672), (4096, 14336), (5120, 5120), (8192, 8192)):
    for M in (512, 1024, 2048, 4096, 8192):
        if 2.0 * M * K * N > 4.5e12:
            continue
        row = {}
        for name, v in (("fused", "0"), ("dense", "2")):
            os.environ["BIE_GEMM_DENSE"] = v
            row[name + "_us"] = sweep.time_case(M, K, N, _hip.BF16, layers=3, reps=4)["us"]
        row["dense/fused"] = round(row["dense_us"] / row["fused_us"], 3)
        t256 = -(-M // 256) * -(-N // 256)
        row["grid"] = t256 if t256 >= 192 else -(-M // 128) * -(-N // 128)
        row["tile"] = 256 if t256 >= 192 else 128
        print(f"K{K} N{N} M{M}", json.dumps(row), flush=True)
