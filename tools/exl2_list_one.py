"""exl2 (3/2-bit g32, random q_perm) decode as a LIST of 32 layers of 4096x4096 (or argv K N), M = 1: a few plain launches (PMC workload)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bitorch_engine.extensions import q_linear_cuda
from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
from bitorch_engine.layers.qlinear.nbit.cuda import MBWQExl2ForwardList
dev = torch.device("cuda:0")
K, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4096)
gen = torch.Generator().manual_seed(5)
qg, row = [], 0
for b in (3, 2):
    for _ in range(K // 2 // 32):
        qg += [b, row]; row += b
groups = len(qg) // 2
q_groups = torch.tensor(qg, dtype=torch.short)
gmap = make_group_map(q_groups, row).to(dev)
perm = torch.randperm(K, generator=gen).to(torch.short).to(dev)
nset = max(2, min(32, int(500e6 // (row * N * 4))))
sets = []
for _ in range(nset):
    sets.append((torch.randint(-2 ** 31, 2 ** 31 - 1, (row, N), dtype=torch.int32, device=dev),
                 (torch.rand((groups, N), device=dev) * 0.02 + 0.001).half(), (torch.randn((groups, N), device=dev) * 0.05).half()))
rows = [q_linear_cuda.mbwq_trans_qweight(s_[0], q_groups, True, K, groups, 4)[1] for s_ in sets][0]  # the load-time step, every tensor
ents = [{"x": torch.randn((1, K), device=dev).half(), "qweight": s_[0], "scales": s_[1], "zeros": s_[2], "q_perm": perm, "q_group_map": gmap,
         "rows": rows, "y": torch.empty((1, N), dtype=torch.float16, device=dev)} for s_ in sets]
plan = MBWQExl2ForwardList(ents)
st = torch.cuda.current_stream()
for _ in range(6):
    plan.forward(st)
torch.cuda.synchronize()
