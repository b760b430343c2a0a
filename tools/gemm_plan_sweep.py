"""Sweeps the GEMM tile height / split-K plan through the BIE_GEMM_BM / BIE_GEMM_S knobs (one subprocess per point)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sweep
    from bitorch_engine import _hip
    M, K, N = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    print(json.dumps(sweep.time_case(M, K, N, _hip.BF16, layers=4, reps=5)))
    sys.exit(0)
for (K, N) in ((4096, 11008), (4096, 4096)):
    for M in (8, 16, 32, 64, 128, 256, 512, 1024, 2048):
        best = None
        for bm in (32, 64, 128, 256):
            if bm > 32 and bm >= 2 * M:
                continue
            for S in (1, 2, 3, 4, 6, 8, 16):
                tiles = -(-M // bm) * -(-N // 256)
                if tiles * S > 2100 or (S > 1 and tiles >= 512):
                    continue
                env = dict(os.environ, BIE_GEMM_BM=str(bm), BIE_GEMM_S=str(S))
                r = subprocess.run([sys.executable, __file__, "one", str(M), str(K), str(N)], env=env, capture_output=True, text=True)
                try:
                    us = json.loads(r.stdout.strip().splitlines()[-1])["us"]
                except Exception:
                    continue
                print(f"K={K} N={N} M={M} BM={bm} S={S} tiles={tiles} blocks={tiles*S}: {us} us", flush=True)
                if best is None or us < best[0]:
                    best = (us, bm, S)
        r = subprocess.run([sys.executable, __file__, "one", str(M), str(K), str(N)], capture_output=True, text=True)
        cur = json.loads(r.stdout.strip().splitlines()[-1])["us"]
        print(f"## K={K} N={N} M={M}: best {best}  current plan {cur} us", flush=True)
