import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bitorch_engine.extensions import q_linear_cuda
from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
dev = torch.device("cuda:0")
for spec in ([(4, 32)] * 6 + [(2, 32)] * 2, [(3, 32)] * 4, [(2, 32)] * 4, [(4, 32)] * 4):
    qg, row, K = [], 0, 0
    for bits, k in spec:
        qg += [bits, row]; row += k * bits // 32; K += k
    groups = len(spec); N = 64
    q_groups = torch.tensor(qg, dtype=torch.short)
    gen = torch.Generator().manual_seed(1)
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (row, N), generator=gen, dtype=torch.int64).to(torch.int32).to(dev)
    gmap = make_group_map(q_groups, row).to(dev)
    _, rows = q_linear_cuda.mbwq_trans_qweight(qw, q_groups, True, K, groups, 4)
    perm = torch.arange(K).to(torch.short).to(dev)
    for name, sc, ze, x in (("s=1 z=0 x=1", 1.0, 0.0, torch.ones((1, K))), ("s=0 z=1 x=1", 0.0, 1.0, torch.ones((1, K))), ("s=1 z=0 x=e0", 1.0, 0.0, torch.eye(K)[:1]),
                            ("s=1 z=0 x=randn", 1.0, 0.0, torch.randn((1, K), generator=gen))):
        s_ = torch.full((groups, N), sc).half().to(dev); z_ = torch.full((groups, N), ze).half().to(dev)
        xd = x.half().to(dev)
        W = q_linear_cuda.mbwq_exl2fp_weight(qw, s_, z_, perm, gmap, rows).float()
        ref = xd.float() @ W
        y = q_linear_cuda.mbwq_exl2_forward(xd, qw, s_, z_, perm, gmap, rows, False).float()
        print(spec[0], spec[-1], name, "ref", ref[0, :4].tolist(), "y", y[0, :4].tolist())
