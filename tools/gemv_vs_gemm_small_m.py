"""M = 2..8: dot-product GEMV (BIE_GEMV_MAX_M=8) against the MFMA GEMM (default dispatch from M = 3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sweep
from bitorch_engine import _hip
for dt in (_hip.BF16, _hip.F16):
    for (K, N) in ((4096, 11008), (4096, 4096)):
        row = " ".join("M=%d:%.2f" % (M, sweep.time_case(M, K, N, dt, layers=8, reps=10)["us"]) for M in (2, 3, 4, 6, 8))
        print(row, "dt", dt, "K", K, "N", N, flush=True)
