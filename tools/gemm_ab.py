#!/usr/bin/env python3
"""MFMA GEMM A/B on one MI355X: table-lookup dequant (BIE_GEMM_LUT=1) against the arithmetic dequant (=0); one process per setting.
   usage: python tools/gemm_ab.py [f16]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import sweep
from bitorch_engine import _hip
dt = _hip.F16 if (len(sys.argv) > 1 and sys.argv[1] == "f16") else _hip.BF16
out = {"BIE_GEMM_LUT": os.environ.get("BIE_GEMM_LUT", "1"), "dtype": "f16" if dt == _hip.F16 else "bf16"}
for (M, K, N) in ((4096, 4096, 4096), (4096, 4096, 11008), (4096, 11008, 4096), (4096, 8192, 28672), (1024, 4096, 4096), (512, 4096, 11008)):
    r = sweep.time_case(M, K, N, dt, layers=4 if K * N < 1e8 else 2, reps=5, graph=True)
    out[f"{M}x{K}x{N}"] = {"us": r["us"], "tflops": round(2.0 * M * K * N / r["us"] / 1e6, 1), "frac": round(2.0 * M * K * N / r["us"] / 1e6 / 2500.0, 4)}
print(json.dumps(out))
