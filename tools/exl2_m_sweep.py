"""exl2 (3/2-bit mix, g32 rows, random q_perm) forward time against M at 4096x11008: which path serves which batch size."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch, bench
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
for _ in range(6):
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (row, N), dtype=torch.int32, device=dev)
    sc = (torch.rand((groups, N), device=dev) * 0.02 + 0.001).half()
    ze = (torch.randn((groups, N), device=dev) * 0.05).half()
    sets.append((qw, sc, ze))
rows = [q_linear_cuda.mbwq_trans_qweight(s_[0], q_groups, True, K, groups, 4)[1] for s_ in sets][0]  # the load-time step, every tensor
res = {}
for M in (1, 2, 3, 4, 8, 16, 32, 33, 64):
    x = torch.randn((M, K), device=dev).half()
    g = bench.capture(lambda st: [q_linear_cuda.mbwq_exl2_forward(x, s_[0], s_[1], s_[2], perm, gmap, rows, False) for s_ in sets])
    res[f"M{M}"] = round(bench.time_graph(g, 5) / len(sets), 1)
print(json.dumps(res))
