import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bitorch_engine.extensions import q_linear_cuda
from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
dev = torch.device("cuda:0")
for (G, gk, N) in ((224, 128, 10240), (224, 128, 4096), (32, 128, 10240), (56, 128, 10240), (224, 32, 10240)):
    spec = [(4, gk)] * G
    qg, row, K = [], 0, 0
    for bits, k in spec:
        qg += [bits, row]; row += k * bits // 32; K += k
    groups = len(spec)
    q_groups = torch.tensor(qg, dtype=torch.short)
    gen = torch.Generator().manual_seed(1)
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (row, N), generator=gen, dtype=torch.int64).to(torch.int32).to(dev)
    gmap = make_group_map(q_groups, row).to(dev)
    _, rows = q_linear_cuda.mbwq_trans_qweight(qw, q_groups, True, K, groups, 4)
    perm = torch.arange(K).to(torch.short).to(dev)
    for name, sc, ze, x in (("s=2^-6 z=0 x=1", 2.0 ** -6, 0.0, torch.ones((1, K))), ("s=0 z=2^-4 x=1", 0.0, 2.0 ** -4, torch.ones((1, K))),
                            ("s=2^-6 z=0 x=randn", 2.0 ** -6, 0.0, torch.randn((1, K), generator=gen))):
        s_ = torch.full((groups, N), sc).half().to(dev); z_ = torch.full((groups, N), ze).half().to(dev)
        xd = x.half().to(dev)
        W = q_linear_cuda.mbwq_exl2fp_weight(qw, s_, z_, perm, gmap, rows).float()
        ref = xd.float() @ W
        y = q_linear_cuda.mbwq_exl2_forward(xd, qw, s_, z_, perm, gmap, rows, False).float()
        print(G, gk, N, name, "max|y-ref|", float((y - ref).abs().max()), "max|ref|", float(ref.abs().max()), "ref", ref[0, :3].tolist(), "y", y[0, :3].tolist())
