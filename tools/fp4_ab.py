#!/usr/bin/env python3
"""Binary GEMM 4096x4096 over M on one MI355X: the XNOR-popcount dispatch (bie_binary_linear_forward) against the matrix-pipe form
   (FP4 images + v_mfma_scale_f32_32x32x64_f8f6f4): GEMM alone per tile (BIE_FP4_TILE) and schedule (BIE_FP4_VAR), the x-image
   pass from bits and from bf16 values, and the end-to-end call (image pass + GEMM).  Graph-timed over 8 distinct weight sets.
   usage: python tools/fp4_ab.py [N K]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd")); sys.path.insert(0, ROOT)
import torch
from bench import capture, time_graph, BF16
from bitorch_engine import _hip
dev = torch.device("cuda", 0)
L = _hip.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
NS = 8
wsets = [torch.randint(0, 256, (N, K // 8), dtype=torch.int32, device=dev).to(torch.uint8) for _ in range(NS)]
st0 = torch.cuda.current_stream().cuda_stream
wimgs = []
for w in wsets:
    img = torch.empty(L.bie_binary_fp4_image_bytes(N, K), dtype=torch.uint8, device=dev)
    assert L.bie_binary_fp4_image(w.data_ptr(), img.data_ptr(), N, K, st0) == 0
    wimgs.append(img)
out = {"N": N, "K": K}
for M in (128, 192, 256, 384, 512, 1024, 2048, 4096, 8192):
    xp = torch.randint(0, 256, (M, K // 8), dtype=torch.int32, device=dev).to(torch.uint8)
    xv = torch.randn((M, K), device=dev).to(BF16)
    y = torch.empty((M, N), dtype=torch.float32, device=dev)
    ximg = torch.empty(L.bie_binary_fp4_image_bytes(M, K), dtype=torch.uint8, device=dev)
    ops = 2.0 * M * K * N
    row = {}
    def run_x(st):
        for w in wsets:
            assert L.bie_binary_linear_forward(xp.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, 0, 1.0, st) == 0
    us = time_graph(capture(run_x), 10) / NS
    row["xnor_us"] = round(us, 2); row["xnor_TOPs"] = round(ops / us / 1e6, 1)
    yx = y.clone()
    for tile in ("64", "128", "256"):
        for var in ("0",):
            os.environ["BIE_FP4_TILE"], os.environ["BIE_FP4_VAR"] = tile, var
            def run_g(st):
                for wi in wimgs:
                    assert L.bie_binary_linear_forward_fp4(ximg.data_ptr(), wi.data_ptr(), y.data_ptr(), M, N, K, 1.0, st) == 0
            assert L.bie_binary_fp4_image(xp.data_ptr(), ximg.data_ptr(), M, K, st0) == 0
            us = time_graph(capture(run_g), 10) / NS
            row[f"fp4_t{tile}_v{var}_us"] = round(us, 2); row[f"fp4_t{tile}_v{var}_TOPs"] = round(ops / us / 1e6, 1)
            torch.cuda.synchronize()
            row[f"fp4_t{tile}_v{var}_exact"] = bool(torch.equal(y, yx))
    del os.environ["BIE_FP4_TILE"], os.environ["BIE_FP4_VAR"]
    def run_i(st):
        for _ in range(NS):
            assert L.bie_binary_fp4_image(xp.data_ptr(), ximg.data_ptr(), M, K, st) == 0
    row["image_bits_us"] = round(time_graph(capture(run_i), 10) / NS, 2)
    def run_v(st):
        for _ in range(NS):
            assert L.bie_binary_fp4_image_from_values(xv.data_ptr(), None, ximg.data_ptr(), M, K, 1, st) == 0
    row["image_values_us"] = round(time_graph(capture(run_v), 10) / NS, 2)
    def run_e(st):
        for wi in wimgs:
            assert L.bie_binary_fp4_image(xp.data_ptr(), ximg.data_ptr(), M, K, st) == 0
            assert L.bie_binary_linear_forward_fp4(ximg.data_ptr(), wi.data_ptr(), y.data_ptr(), M, N, K, 1.0, st) == 0
    us = time_graph(capture(run_e), 10) / NS
    row["fp4_end_to_end_us"] = round(us, 2); row["fp4_end_to_end_TOPs"] = round(ops / us / 1e6, 1)
    yb = torch.empty((M, N), dtype=BF16, device=dev)
    bias_a = torch.randn(K, device=dev).to(BF16)
    sa, sw = torch.tensor(0.7, device=dev).to(BF16), torch.tensor(0.01, device=dev).to(BF16)
    def run_l(st):
        for wi in wimgs:
            assert L.bie_binary_fp4_image_from_values(xv.data_ptr(), bias_a.data_ptr(), ximg.data_ptr(), M, K, 1, st) == 0
            assert L.bie_binary_linear_layer_fp4(ximg.data_ptr(), wi.data_ptr(), sa.data_ptr(), sw.data_ptr(), yb.data_ptr(), M, N, K, 1, st) == 0
    us = time_graph(capture(run_l), 10) / NS
    row["layer_fp4_us"] = round(us, 2); row["layer_fp4_TOPs"] = round(ops / us / 1e6, 1)
    def run_lg(st):
        for wi in wimgs:
            assert L.bie_binary_linear_layer_fp4(ximg.data_ptr(), wi.data_ptr(), sa.data_ptr(), sw.data_ptr(), yb.data_ptr(), M, N, K, 1, st) == 0
    row["layer_fp4_gemm_only_us"] = round(time_graph(capture(run_lg), 10) / NS, 2)
    if L.bie_binary_linear_fused_ok(M, N, K):
        def run_f(st):
            for w in wsets:
                assert L.bie_binary_linear_fused(xv.data_ptr(), bias_a.data_ptr(), w.data_ptr(), sa.data_ptr(), sw.data_ptr(), yb.data_ptr(), M, N, K, 1, 0, st) == 0
        row["layer_xnor_one_launch_us"] = round(time_graph(capture(run_f), 10) / NS, 2)
    out[f"M{M}"] = row
    print(f"M{M}", json.dumps(row), flush=True)
print(json.dumps(out))
