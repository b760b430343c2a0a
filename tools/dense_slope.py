import sys, os, json
sys.path.insert(0, "tools"); sys.path.insert(0, "bitorch-engine_amd"); sys.path.insert(0, ".")
import sweep
from bitorch_engine import _hip
for M in (256, 4096):
    row = {}
    for K in (1024, 4096, 16384):
        r = sweep.time_case(M, K, 4096, _hip.BF16, layers=4, reps=5)
        row[f"K{K}"] = r["us"]
    print("M", M, json.dumps(row), flush=True)
