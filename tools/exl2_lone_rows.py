#!/usr/bin/env python3
"""exl2 lone call with 3 ... 16 rows: the fused matrix-pipe kernel against the group-of-one path (BIE_EXL2_LONE_ROWS_LO / _HI select it)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bench import capture, time_graph
from bitorch_engine.extensions import q_linear_cuda
from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(5)
for (K, N) in ((4096, 4096), (4096, 11008), (11008, 4096)):
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
    out = []
    for M in (3, 4, 5, 8, 12, 16):
        x = torch.randn((M, K), device=dev).half()
        g = capture(lambda st: [q_linear_cuda.mbwq_exl2_forward(x, s_[0], s_[1], s_[2], perm, gmap, rows, False) for s_ in sets])
        out.append(f"M={M}: {min(time_graph(g, 10) for _ in range(3)) / nset:.2f}")
    print(f"{K}x{N} lone call us: " + "  ".join(out))
