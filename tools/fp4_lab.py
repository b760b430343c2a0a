#!/usr/bin/env python3
"""Wall-clock ablations of xnor_fp4_gemm_kernel<4,4> (lab build: -DBIE_FP4_LAB, BIE_HIP_LIB=bitorch-engine_amd/ab/libbie_hip_fp4lab.so):
   BIE_FP4_VAR 0 product / 1 burst / 2 no LDS-DMA in the loop / 3 no fragment reads / 4 neither / 5 neither, no barrier
   at M = 256 (16 workgroups: the lone-workgroup pipeline) and M = 4096 (one workgroup per CU), N = K = 4096.
   usage: BIE_HIP_LIB=... python tools/fp4_lab.py [vars] [tiles]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd")); sys.path.insert(0, ROOT)
import torch
from bench import capture, time_graph
from bitorch_engine import _hip
dev = torch.device("cuda", 0)
L = _hip.lib()
N = 4096
K = int(os.environ.get("FP4_LAB_K", "4096"))
VARS = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1", "2", "3", "4", "5"]
TILES = sys.argv[2].split(",") if len(sys.argv) > 2 else ["256"]
st0 = torch.cuda.current_stream().cuda_stream
wp = torch.randint(0, 256, (N, K // 8), dtype=torch.int32, device=dev).to(torch.uint8)
wimg = torch.empty(L.bie_binary_fp4_image_bytes(N, K), dtype=torch.uint8, device=dev)
assert L.bie_binary_fp4_image(wp.data_ptr(), wimg.data_ptr(), N, K, st0) == 0
for M in (256, 4096):
    xp = torch.randint(0, 256, (M, K // 8), dtype=torch.int32, device=dev).to(torch.uint8)
    ximg = torch.empty(L.bie_binary_fp4_image_bytes(M, K), dtype=torch.uint8, device=dev)
    assert L.bie_binary_fp4_image(xp.data_ptr(), ximg.data_ptr(), M, K, st0) == 0
    y = torch.empty((M, N), dtype=torch.float32, device=dev)
    row = {}
    for tile in TILES:
        for var in VARS:
            os.environ["BIE_FP4_TILE"], os.environ["BIE_FP4_VAR"] = tile, var
            def run(st):
                for _ in range(8):
                    assert L.bie_binary_linear_forward_fp4(ximg.data_ptr(), wimg.data_ptr(), y.data_ptr(), M, N, K, 1.0, st) == 0
            us = time_graph(capture(run), 10) / 8
            row[f"t{tile}_v{var}"] = round(us, 2)
    print(f"K{K} M{M}", json.dumps(row), flush=True)
