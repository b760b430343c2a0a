#!/usr/bin/env python3
"""One W4 g128 layer list of L distinct K x N layers at M rows in fp16 / bf16, launched three times without a graph (for rocprofv3 passes).
   usage: python tools/xs_run.py K N L M f16|bf16"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import Bench, BF16, make_layer  # noqa: E402

K, N, L, M = (int(a) for a in sys.argv[1:5])
dt = torch.float16 if sys.argv[5] == "f16" else BF16
dev = torch.device("cuda", 0)
B = Bench(dev)
gen = torch.Generator(device=dev).manual_seed(5)
layers = [make_layer(dev, gen, K, N, 4, dt) for _ in range(L)]
plan = B.make_list(layers, K, N, gen, M=M)
print("form", plan.form)
for _ in range(3):
    plan()
torch.cuda.synchronize()
