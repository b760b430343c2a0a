#!/usr/bin/env python3
"""Host-side cost of ONE eager layer call (what an unchanged caller without HIP graphs pays per linear): MPQLinearCuda 4096x4096 W4 g128 at M = 1,
   wall clock per call over a long asynchronous loop (the GPU is never the limit: a launch takes 6 us), next to torch.nn.Linear in fp16, plus the
   cProfile top of the MPQ loop.  usage: python tools/host_overhead.py [calls=3000]"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bitorch_engine.layers.qlinear.nbit.cuda import MPQLinearCuda  # noqa: E402

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
K = N = 4096
g = torch.Generator().manual_seed(0)
layer = MPQLinearCuda(K, N, w_bit=4, dtype=torch.half, group_size=128, dq_group_size=32, use_gba_quant=True, asym=False)
layer.qweight.data = torch.randint(-2 ** 31, 2 ** 31 - 1, layer.qweight.shape, generator=g, dtype=torch.int64).to(torch.int32)
layer.prepare_params()
layer.scales = (torch.rand(layer.scales.shape, generator=g) * 0.01 + 0.005).half()
layer.zeros = (layer.scales.float() * torch.rand(layer.scales.shape, generator=g) * 15).half()
layer.eval().to(dev)
lin = torch.nn.Linear(K, N, bias=False).half().to(dev).eval()
x = torch.randn((1, 1, K), generator=g).half().to(dev)


def loop(f, calls):
    with torch.no_grad():
        for _ in range(200):
            f(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            f(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    return (t1 - t0) / calls * 1e6, (t2 - t0) / calls * 1e6


for name, f in (("MPQLinearCuda", layer), ("nn.Linear fp16", lin), ("MPQLinearCuda", layer), ("nn.Linear fp16", lin)):
    h, w = loop(f, n)
    print(f"{name:16s} host {h:7.2f} us per call (issue loop), {w:7.2f} us per call incl. the final drain")
pr = cProfile.Profile()
with torch.no_grad():
    pr.enable()
    for _ in range(n):
        layer(x)
    pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print("\n".join(l[:160] for l in s.getvalue().splitlines()[:34]))
