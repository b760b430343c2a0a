#!/usr/bin/env python3
"""Lone calls of 17..32 rows: the fused MFMA GEMM (default) against the matrix-pipe lookup / arithmetic kernel with two row blocks (BIE_LUT_MFMA_MAX_M=32 BIE_LUT_MAX_M=32), fp16 and bf16."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import Bench, BF16  # noqa: E402

B = Bench(torch.device("cuda", 0))
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("BIE_LUT")}}  # arms: BIE_LUT_RB2=0 (the fused GEMM beyond 16 rows) against the default (measured table)
for dn, dt in (("f16", torch.float16), ("bf16", BF16)):
    shapes = ((2048, 2048), (2048, 8192), (4096, 2048), (4096, 4096), (4096, 6144), (4096, 8192), (4096, 11008), (4096, 14336), (4096, 28672), (5120, 5120), (5120, 13824), (8192, 4096), (8192, 8192), (8192, 28672), (11008, 4096), (14336, 4096))
    if os.environ.get("LONE_SHAPES"):  # e.g. "4096x1024,3584x18944"
        shapes = tuple(tuple(int(v) for v in x.split("x")) for x in os.environ["LONE_SHAPES"].split(","))
    for (k, n) in shapes:
        for M in (17, 20, 24, 28, 32):
            print(f"{dn}_{k}x{n}_M{M}", file=sys.stderr, flush=True)
            out[f"{dn}_{k}x{n}_M{M}"] = B.gemv(k, n, max(6, min(40, int(9e8 // (k * n // 2)))), 8, 7, M=M, dt=dt)["us_per_launch"]
print(json.dumps(out))
