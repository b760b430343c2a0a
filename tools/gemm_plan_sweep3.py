"""In-process sweep of the fused GEMM's plan knobs (BIE_GEMM_BM / BIE_GEMM_S, read per call under BIE_TUNING) at mid M; BIE_GEMM_DENSE=0."""
import os, sys, json
os.environ["BIE_TUNING"] = "1"; os.environ["BIE_GEMM_DENSE"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd")); sys.path.insert(0, ROOT)
import sweep
from bitorch_engine import _hip
for (K, N) in ((11008, 4096), (4096, 4096), (4096, 11008), (8192, 8192)):
    for M in (64, 128, 256, 512, 1024, 2048):
        res = []
        for bm in (32, 64, 128, 256):
            if bm > 32 and bm >= 2 * M:
                continue
            for S in (1, 2, 3, 4, 6, 8, 12, 16):
                tiles = -(-M // bm) * -(-N // 256)
                if (tiles * S > 1400 and S > 1) or tiles > 3000 or (K // 64) // S < 2:
                    continue
                os.environ["BIE_GEMM_BM"], os.environ["BIE_GEMM_S"] = str(bm), str(S)
                us = sweep.time_case(M, K, N, _hip.BF16, layers=3, reps=3)["us"]
                res.append((us, bm, S, tiles * S))
        os.environ["BIE_GEMM_BM"], os.environ["BIE_GEMM_S"] = "0", "0"
        cur = sweep.time_case(M, K, N, _hip.BF16, layers=3, reps=3)["us"]
        res.sort()
        print(f"K={K} N={N} M={M}: current {cur} us; best {res[:4]}", flush=True)
