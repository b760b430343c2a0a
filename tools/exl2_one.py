"""exl2 (3/2-bit g32, random q_perm) decode at 4096x11008, M = 1: a few plain launches (PMC workload)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bitorch_engine.extensions import q_linear_cuda
from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
dev = torch.device("cuda:0")
K, N = 4096, 11008
gen = torch.Generator().manual_seed(5)
qg, row = [], 0
for b in (3, 2):
    for _ in range(K // 2 // 32):
        qg += [b, row]; row += b
groups = len(qg) // 2
q_groups = torch.tensor(qg, dtype=torch.short)
gmap = make_group_map(q_groups, row).to(dev)
perm = torch.randperm(K, generator=gen).to(torch.short).to(dev)
sets = []
for _ in range(8):
    sets.append((torch.randint(-2 ** 31, 2 ** 31 - 1, (row, N), dtype=torch.int32, device=dev),
                 (torch.rand((groups, N), device=dev) * 0.02 + 0.001).half(), (torch.randn((groups, N), device=dev) * 0.05).half()))
_, rows = q_linear_cuda.mbwq_trans_qweight(sets[0][0], q_groups, True, K, groups, 4)
x = torch.randn((1, K), device=dev).half()
for _ in range(3):
    for s_ in sets:
        q_linear_cuda.mbwq_exl2_forward(x, s_[0], s_[1], s_[2], perm, gmap, rows, False)
torch.cuda.synchronize()
