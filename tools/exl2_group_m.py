#!/usr/bin/env python3
"""exl2 q/k/v-sized sibling group at M = 1 ... 16 rows: three lone calls against one grouped call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bench import capture, time_graph
from bitorch_engine.extensions import q_linear_cuda
from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(5)
for (K, N, nm) in ((4096, 4096, 3), (4096, 11008, 2)):
    qg, row = [], 0
    for b in (3, 2):
        for _ in range(K // 2 // 32):
            qg += [b, row]; row += b
    groups = len(qg) // 2
    q_groups = torch.tensor(qg, dtype=torch.short)
    gmap = make_group_map(q_groups, row).to(dev)
    perm = torch.randperm(K, generator=gen).to(torch.short).to(dev)
    nset = 12
    sets = [(torch.randint(-2 ** 31, 2 ** 31 - 1, (row, N), dtype=torch.int32, device=dev),
             (torch.rand((groups, N), device=dev) * 0.02 + 0.001).half(), (torch.randn((groups, N), device=dev) * 0.05).half()) for _ in range(nset)]
    rows = [q_linear_cuda.mbwq_trans_qweight(s_[0], q_groups, True, K, groups, 4)[1] for s_ in sets][0]
    grps = [[(s_[0], s_[1], s_[2], perm, gmap, rows) for s_ in sets[i:i + nm]] for i in range(0, nset, nm)]
    for M in (1, 2, 4, 8, 16, 32):
        x = torch.randn((M, K), device=dev).half()
        g = capture(lambda st: [q_linear_cuda.mbwq_exl2_forward(x, s_[0], s_[1], s_[2], perm, gmap, rows, False) for s_ in sets])
        us1 = min(time_graph(g, 10) for _ in range(3)) / len(grps)
        g3 = capture(lambda st: [q_linear_cuda.mbwq_exl2_forward_grouped(x, m) for m in grps])
        usg = min(time_graph(g3, 10) for _ in range(3)) / len(grps)
        print(f"{nm} x {K}x{N} M={M}: {nm} lone calls {us1:.2f} us   grouped call {usg:.2f} us")
