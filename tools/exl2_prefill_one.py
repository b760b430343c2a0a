#!/usr/bin/env python3
"""One exl2 prefill shape, a few eager calls (for rocprofv3 --kernel-trace --stats): python tools/exl2_prefill_one.py M K N"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd")); sys.path.insert(0, ROOT)
import torch
from bitorch_engine.extensions import q_linear_cuda
import exl2_prefill_ab as ab  # noqa: F401  (its module-level sweep is skipped below)
M, K, N = [int(v) for v in sys.argv[1:4]]
gen = torch.Generator().manual_seed(1)
L = ab.layer(K, N, gen)
x = torch.randn((M, K), generator=gen).half().to(ab.dev)
for _ in range(6):
    y = q_linear_cuda.mbwq_exl2_forward(x, *L, False)
torch.cuda.synchronize()
print("ok", float(y.float().abs().max()))
