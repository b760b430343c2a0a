#!/usr/bin/env python3
"""Fused MFMA GEMM (mpq_gemm_kernel, 17 <= M < 1024) under every forced (BM, S) plan against the cost model's own choice: us per call, W4 g128 bf16.
   One line per (shape, M): the model's plan time, the best forced plan and its time.  usage: python tools/gemm_plan_sweep_r06.py"""
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
shapes = tuple(tuple(int(v) for v in x.split("x")) for x in os.environ.get("SWEEP_SHAPES", "4096x11008,4096x4096,11008x4096").split(","))
Ms = tuple(int(m) for m in os.environ.get("SWEEP_MS", "17,32,48,64,96,128,192,256,384,512,768,1023").split(","))


def t(M, k, n):
    try:
        return B.gemm(M, k, n, 8, 8, 5)["us_per_launch"]
    except Exception as e:  # a plan the library refuses
        return None


for (k, n) in shapes:
    for M in Ms:
        os.environ.pop("BIE_GEMM_BM", None)
        os.environ.pop("BIE_GEMM_S", None)
        base = t(M, k, n)
        res = {}
        for bm in (32, 64, 128, 256):
            if bm >= 2 * M and bm > 32:
                continue
            for s in (1, 2, 3, 4, 6, 8, 12, 16):
                os.environ["BIE_GEMM_BM"], os.environ["BIE_GEMM_S"] = str(bm), str(s)
                v = t(M, k, n)
                if v is not None:
                    res[f"{bm}x{s}"] = v
        os.environ.pop("BIE_GEMM_BM", None)
        os.environ.pop("BIE_GEMM_S", None)
        best = min(res, key=res.get)
        top = sorted(res.items(), key=lambda kv: kv[1])[:4]
        print(json.dumps({"K": k, "N": n, "M": M, "model_us": base, "best": best, "best_us": res[best], "gain": round(base / res[best], 3), "top4": top, "all": res}), flush=True)
