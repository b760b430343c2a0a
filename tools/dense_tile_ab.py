import json, os, sys
os.environ["BIE_TUNING"] = "1"; os.environ["BIE_GEMM_DENSE"] = "2"
sys.path.insert(0, "tools"); sys.path.insert(0, "bitorch-engine_amd"); sys.path.insert(0, ".")
import sweep
from bitorch_engine import _hip
for (K, N) in ((4096, 4096), (4096, 11008), (11008, 4096), (4096, 14336)):
    for M in (2048, 4096, 8192):
        row = {}
        for t in ("128", "256"):
            os.environ["BIE_GEMM_DENSE_TILE"] = t
            row["t" + t] = sweep.time_case(M, K, N, _hip.BF16, layers=3, reps=4)["us"]
        os.environ["BIE_GEMM_DENSE"] = "0"
        row["fused"] = sweep.time_case(M, K, N, _hip.BF16, layers=3, reps=4)["us"]
        os.environ["BIE_GEMM_DENSE"] = "2"
        print(K, N, M, json.dumps(row), flush=True)
