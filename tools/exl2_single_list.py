#!/usr/bin/env python3
"""exl2 decode of ONE layer per call: the per-layer launch against a one-entry list (permute kernel + list kernel), rotating weight sets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bench import capture, time_graph
from bitorch_engine.extensions import q_linear_cuda
from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
from bitorch_engine.layers.qlinear.nbit.cuda import MBWQExl2ForwardList
dev = torch.device("cuda", 0)
gen = torch.Generator().manual_seed(5)
for (K, N) in ((4096, 4096), (4096, 11008), (11008, 4096), (4096, 12288), (8192, 8192), (14336, 4096), (28672, 8192)):
    qg, row = [], 0
    for b in (3, 2):
        for _ in range(K // 2 // 32):
            qg += [b, row]; row += b
    groups = len(qg) // 2
    q_groups = torch.tensor(qg, dtype=torch.short)
    gmap = make_group_map(q_groups, row).to(dev)
    perm = torch.randperm(K, generator=gen).to(torch.short).to(dev)
    nset = max(2, min(32, int(500e6 // (row * N * 4))))
    sets = [(torch.randint(-2 ** 31, 2 ** 31 - 1, (row, N), dtype=torch.int32, device=dev),
             (torch.rand((groups, N), device=dev) * 0.02 + 0.001).half(), (torch.randn((groups, N), device=dev) * 0.05).half()) for _ in range(nset)]
    rows = [q_linear_cuda.mbwq_trans_qweight(s_[0], q_groups, True, K, groups, 4)[1] for s_ in sets][0]
    byts = row * N * 4 + 4 * groups * N + 6 * K + 2 * K + 2 * N
    x = torch.randn((1, K), device=dev).half()
    g = capture(lambda st: [q_linear_cuda.mbwq_exl2_forward(x, s_[0], s_[1], s_[2], perm, gmap, rows, False) for s_ in sets])
    us1 = min(time_graph(g, 10) for _ in range(3)) / nset
    NE = int(os.environ.get("NENT", "1"))
    groups_ = [sets[i:i + NE] for i in range(0, nset - NE + 1, NE)]
    plans = [MBWQExl2ForwardList([{"x": x, "qweight": s_[0], "scales": s_[1], "zeros": s_[2], "q_perm": perm, "q_group_map": gmap, "rows": rows,
                                   "y": torch.empty((1, N), dtype=torch.float16, device=dev)} for s_ in grp]) for grp in groups_]
    g2 = capture(lambda st: [p.forward(st) for p in plans])
    usl = min(time_graph(g2, 10) for _ in range(3)) / len(plans)
    g3 = capture(lambda st: [q_linear_cuda.mbwq_exl2_forward_grouped(x, [(s_[0], s_[1], s_[2], perm, gmap, rows)]) for s_ in sets])
    usg = min(time_graph(g3, 10) for _ in range(3)) / nset
    print(f"{K}x{N}: group call with ONE member {usg:.2f} us")
    print(f"{K}x{N}: per-layer launch {us1:.2f} us ({byts / us1 / 8e6:.3f})   {NE}-entry list (2 launches) {usl:.2f} us per list = {usl / NE:.2f} per layer ({NE * byts / usl / 8e6:.3f})")
