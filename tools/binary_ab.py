#!/usr/bin/env python3
"""Binary (XNOR-popcount) linear 4096x4096 over M on one MI355X: packed-x launch (bie_binary_linear_forward) and the one-launch layer
   forward (bie_binary_linear_fused, bf16 x), graph-timed over 16 distinct weight sets.  BIE_BINARY_MID_MAX=0 disables xnor_mid_kernel
   (the previous dispatch); BIE_BINARY_MID=<rt><g> pins a tile form.
   usage: python tools/binary_ab.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd")); sys.path.insert(0, ROOT)
import torch
from bench import capture, time_graph, BF16
from bitorch_engine import _hip
dev = torch.device("cuda", 0)
L = _hip.lib()
K = N = 4096
wsets = [torch.randint(0, 256, (N, K // 8), dtype=torch.int32, device=dev).to(torch.uint8) for _ in range(16)]
bias_a = torch.randn(K, device=dev).to(BF16)
sa, sw = torch.tensor(0.7, device=dev).to(BF16), torch.tensor(0.01, device=dev).to(BF16)
out = {"env": {k: os.environ.get(k) for k in ("BIE_BINARY_MID_MAX", "BIE_BINARY_MID") if os.environ.get(k)}}
for M in (8, 16, 32, 64, 128, 256, 512, 1024):
    xp = torch.randint(0, 256, (M, K // 8), dtype=torch.int32, device=dev).to(torch.uint8)
    y = torch.empty((M, N), dtype=torch.float32, device=dev)
    def run(st):
        for w in wsets:
            assert L.bie_binary_linear_forward(xp.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, 0, 1.0, st) == 0
    us = time_graph(capture(run), 10) / len(wsets)
    row = {"packed_us": round(us, 2), "packed_TOPs": round(2.0 * M * K * N / us / 1e6, 1)}
    if L.bie_binary_linear_fused_ok(M, N, K):
        x = torch.randn((M, K), device=dev).to(BF16)
        yb = torch.empty((M, N), dtype=BF16, device=dev)
        def runf(st):
            for w in wsets:
                assert L.bie_binary_linear_fused(x.data_ptr(), bias_a.data_ptr(), w.data_ptr(), sa.data_ptr(), sw.data_ptr(), yb.data_ptr(), M, N, K, 1, 0, st) == 0
        us = time_graph(capture(runf), 10) / len(wsets)
        row.update({"layer_us": round(us, 2), "layer_TOPs": round(2.0 * M * K * N / us / 1e6, 1)})
    out[f"M{M}"] = row
print(json.dumps(out))
