"""In-process sweep of the GEMM plan knobs (BIE_GEMM_BM / BIE_GEMM_S are read per call)."""
import os
os.environ.setdefault("BIE_TUNING", "1"), sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import sweep
from bitorch_engine import _hip
for (K, N) in ((4096, 11008), (4096, 4096)):
    for M in (32, 64, 128, 256, 512, 1024, 2048, 4096):
        res = []
        for bm in (32, 64, 128, 256):
            if bm > 32 and bm >= 2 * M:
                continue
            for S in (1, 2, 4, 8, 16):
                tiles = -(-M // bm) * -(-N // 256)
                if tiles * S > 1400 and S > 1:
                    continue
                if tiles > 3000:
                    continue
                os.environ["BIE_GEMM_BM"], os.environ["BIE_GEMM_S"] = str(bm), str(S)
                us = sweep.time_case(M, K, N, _hip.BF16, layers=4, reps=3)["us"]
                res.append((us, bm, S, tiles * S))
                print(f"K={K} N={N} M={M} BM={bm} S={S} blocks={tiles*S}: {us} us", flush=True)
        os.environ["BIE_GEMM_BM"], os.environ["BIE_GEMM_S"] = "0", "0"
        cur = sweep.time_case(M, K, N, _hip.BF16, layers=4, reps=3)["us"]
        print(f"## K={K} N={N} M={M}: best {min(res)}  current plan {cur} us", flush=True)
