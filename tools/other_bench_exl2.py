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
