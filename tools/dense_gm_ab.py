"""Dense-form W4A16 GEMM: tile rows per XCD run (BIE_GEMM_DENSE_GM) A/B, microseconds per call (dequantise + GEMM), bf16.
usage (GPU): python tools/dense_gm_ab.py"""
import json, os, sys
os.environ["BIE_TUNING"] = "1"; os.environ["BIE_GEMM_DENSE"] = "2"
sys.path.insert(0, "tools"); sys.path.insert(0, "bitorch-engine_amd"); sys.path.insert(0, ".")
import sweep
from bitorch_engine import _hip
for (K, N) in ((4096, 4096), (4096, 11008), (11008, 4096), (4096, 14336), (8192, 8192)):
    for M in (1024, 2048, 4096, 8192):
        row = {}
        for g in ("1", "2", "4", "8", "16"):
            os.environ["BIE_GEMM_DENSE_GM"] = g
            row["gm" + g] = sweep.time_case(M, K, N, _hip.BF16, layers=3, reps=4)["us"]
        print(K, N, M, json.dumps(row), flush=True)
