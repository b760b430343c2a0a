#!/usr/bin/env python3
"""exl2 decode rows of bench_exl2 only (M = 1: per-layer launches and the 32-layer list), for A/B runs of library variants (BIE_HIP_LIB)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
import bench
from bench import capture, time_graph
from bitorch_engine.extensions import q_linear_cuda
from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
from bitorch_engine.layers.qlinear.nbit.cuda import MBWQExl2ForwardList
dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(5)
res = []
for (K, N) in ((4096, 4096), (4096, 11008), (11008, 4096)):
    qg, row = [], 0
    MIX = os.environ.get("EXL2_MIX", "3:1,2:1")  # bits:share, in band order
    mix = [(int(a.split(":")[0]), int(a.split(":")[1])) for a in MIX.split(",")]
    tot = sum(w for _, w in mix)
    left = K // 32
    for i, (b, w) in enumerate(mix):
        cnt = left if i == len(mix) - 1 else K // 32 * w // tot  # the last band takes the remainder: every chunk of K belongs to a group
        left -= cnt
        for _ in range(cnt):
            qg += [b, row]; row += b
    groups = len(qg) // 2
    q_groups = torch.tensor(qg, dtype=torch.short)
    gmap = make_group_map(q_groups, row).to(dev)
    perm = torch.randperm(K, generator=gen).to(torch.short).to(dev)
    if os.environ.get("EXL2_PERM") == "identity":
        perm = torch.arange(K).to(torch.short).to(dev)
    nset = max(2, min(32, int(500e6 // (row * N * 4))))
    sets = [(torch.randint(-2 ** 31, 2 ** 31 - 1, (row, N), dtype=torch.int32, device=dev),
             (torch.rand((groups, N), device=dev) * 0.02 + 0.001).half(), (torch.randn((groups, N), device=dev) * 0.05).half()) for _ in range(nset)]
    rows = [q_linear_cuda.mbwq_trans_qweight(s_[0], q_groups, True, K, groups, 4)[1] for s_ in sets][0]  # the load-time step, every tensor
    byts = row * N * 4 + 4 * groups * N + 6 * K + 2 * K + 2 * N
    x = torch.randn((1, K), device=dev).half()
    g = capture(lambda st: [q_linear_cuda.mbwq_exl2_forward(x, s_[0], s_[1], s_[2], perm, gmap, rows, False) for s_ in sets])
    us1 = min(time_graph(g, 10) for _ in range(3)) / nset
    ents = [{"x": torch.randn((1, K), device=dev).half(), "qweight": s_[0], "scales": s_[1], "zeros": s_[2], "q_perm": perm, "q_group_map": gmap,
             "rows": rows, "y": torch.empty((1, N), dtype=torch.float16, device=dev)} for s_ in sets]
    plan = MBWQExl2ForwardList(ents)
    g2 = capture(lambda st: [plan.forward(st) for _ in range(4)])
    usl = min(time_graph(g2, 10) for _ in range(3)) / (nset * 4)
    res.append(f"{K}x{N}: per-layer {us1:.2f} us ({byts / us1 / 8e6:.3f})  list {usl:.2f} us/layer ({byts / usl / 8e6:.3f})")
print(os.environ.get("EXL2_MIX", ""), os.environ.get("BIE_HIP_LIB", "./default/x").split("/")[-2], "direct=" + os.environ.get("BIE_EXL2_DIRECT", "1"), " | ".join(res))
