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
# ---- exl2 3/2-bit mixed, g32 rows
for (K, N) in ((4096, 11008), (4096, 4096), (11008, 4096)):
    qg, row = [], 0
    for b in (3, 2):
        for _ in range(K // 2 // 32):
            qg += [b, row]
            row += 32 * b // 32
    groups = len(qg) // 2
    q_groups = torch.tensor(qg, dtype=torch.short)
    gmap = make_group_map(q_groups, row).to(dev)
    perm = torch.randperm(K, generator=gen).to(torch.short).to(dev)
    sets = []
    nset = max(2, min(32, int(600e6 // (row * N * 4))))
    for _ in range(nset):
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (row, N), generator=gen, dtype=torch.int64).to(torch.int32).to(dev)
        sc = (torch.rand((groups, N), generator=gen) * 0.02 + 0.001).half().to(dev)
        ze = (torch.randn((groups, N), generator=gen) * 0.05).half().to(dev)
        sets.append((qw, sc, ze))
    rows = [q_linear_cuda.mbwq_trans_qweight(s_[0], q_groups, True, K, groups, 4)[1] for s_ in sets][0]  # the load-time step, every tensor
    for M in (1, 4):
        x = torch.randn((M, K), generator=gen).half().to(dev)
        us = timeit(lambda s_: q_linear_cuda.mbwq_exl2_forward(x, s_[0], s_[1], s_[2], perm, gmap, rows, False), sets)
        byts = row * N * 4 + 4 * groups * N + 6 * K + 2 * M * K + 2 * M * N
        print(json.dumps({"op": "exl2 w3/w2 g32", "M": M, "K": K, "N": N, "us": round(us, 2), "GB/s": round(byts / us / 1e3, 1), "bytes": byts}), flush=True)
# ---- MBWQ uniform q4 / q2 decode
for bits in (4, 2):
    K, N, gs = 4096, 11008, 128
    sets = []
    for _ in range(26):
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K * bits // 32, N), generator=gen, dtype=torch.int64).to(torch.int32).to(dev)
        sc = (torch.rand((K // gs, N), generator=gen) * 0.02 + 0.001).half().to(dev)
        ze = (torch.randn((K // gs, N), generator=gen) * 0.05).half().to(dev)
        sets.append((qw, sc, ze))
    perm = torch.zeros(K, dtype=torch.short, device=dev)
    x = torch.randn((1, K), generator=gen).half().to(dev)
    us = timeit(lambda s_: q_linear_cuda.mbwq_q4_forward(x, s_[0], s_[1], s_[2], gs, perm, bits), sets)
    byts = K * N * bits // 8 + 4 * (K // gs) * N + 2 * K + 2 * N
    print(json.dumps({"op": f"mbwq q{bits} g128", "M": 1, "K": K, "N": N, "us": round(us, 2), "GB/s": round(byts / us / 1e3, 1)}), flush=True)
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
