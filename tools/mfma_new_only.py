#!/usr/bin/env python3
"""Workload for PMC / kernel-stat passes over the two ordered-asm MFMA kernels of this round, no graph:
   xnor_fp4_gemm_kernel (4096^3, fp32 out) and mpq_dense_gemm_kernel (+ its dequantise pass; M = 4096, 4096 -> 4096, bf16 W4 g128)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import sweep
from bitorch_engine import _hip
L = _hip.lib()
dev = torch.device("cuda", 0)
M = N = K = 4096
st = torch.cuda.current_stream().cuda_stream
xp = torch.randint(0, 256, (M, K // 8), dtype=torch.int32, device=dev).to(torch.uint8)
wp = torch.randint(0, 256, (N, K // 8), dtype=torch.int32, device=dev).to(torch.uint8)
ximg = torch.empty(L.bie_binary_fp4_image_bytes(M, K), dtype=torch.uint8, device=dev)
wimg = torch.empty(L.bie_binary_fp4_image_bytes(N, K), dtype=torch.uint8, device=dev)
y = torch.empty((M, N), dtype=torch.float32, device=dev)
assert L.bie_binary_fp4_image(xp.data_ptr(), ximg.data_ptr(), M, K, st) == 0 and L.bie_binary_fp4_image(wp.data_ptr(), wimg.data_ptr(), N, K, st) == 0
for _ in range(6):
    assert L.bie_binary_linear_forward_fp4(ximg.data_ptr(), wimg.data_ptr(), y.data_ptr(), M, N, K, 1.0, st) == 0
torch.cuda.synchronize()
print(sweep.time_case(4096, 4096, 4096, _hip.BF16, layers=2, reps=3, graph=False))
