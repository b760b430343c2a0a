"""Timing of the hot-path kernels outside the headline config (BASELINE configs[2], [3]): exl2 mixed 3/2-bit decode, MBWQ uniform
q4/q2 decode, binary XNOR linear / conv.  Events around graph-free launch loops with rotating weight sets."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bitorch_engine.extensions import q_linear_cuda, binary_linear_cutlass, binary_conv_cpp
from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
dev = "cuda:0"


def timeit(fn, sets, reps=20):
    for i in range(len(sets)):
        fn(sets[i])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for s_ in sets:
            fn(s_)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        for s_ in sets:
            fn(s_)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(sets))


gen = torch.Generator().manual_seed(0)
# ---- binary linear 4096x4096 (row-packed)
K = N = 4096
wsets = [torch.randint(0, 256, (N, K // 8), generator=gen, dtype=torch.int32).to(torch.uint8).to(dev) for _ in range(16)]
for LAYOUT, M in ((0, 1), (0, 4), (0, 64), (0, 512), (0, 4096), (1, 1), (1, 64)):
    xp = torch.randint(0, 256, (M, K // 8), generator=gen, dtype=torch.int32).to(torch.uint8).to(dev)
    from bitorch_engine import _hip
    y = torch.empty((M, N), dtype=torch.float32, device=dev)
    L = _hip.lib()
    def f(w):
        rc = L.bie_binary_linear_forward(xp.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, LAYOUT, 1.0, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    us = timeit(f, wsets, reps=10 if M < 4096 else 3)
    byts = K * N // 8 + M * K // 8 + 4 * M * N
    print(json.dumps({"op": f"binary linear layout {LAYOUT}", "M": M, "K": K, "N": N, "us": round(us, 2), "GB/s": round(byts / us / 1e3, 1), "TOP/s": round(2.0 * M * K * N / us / 1e6, 1)}), flush=True)
