"""XP list form against the per-layer forward on one layer, fresh process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bitorch_engine.extensions import q_linear_cuda
from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
from bitorch_engine.layers.qlinear.nbit.cuda import MBWQExl2ForwardList
dev = torch.device("cuda:0")
spec = [(4, 64)] * 3 + [(4, 32)] + [(3, 128)] * 2 + [(3, 96)] + [(2, 64)] * 4
if len(sys.argv) > 1: spec = [(3, 32)] * 8 + [(2, 32)] * 8
qg, row, K = [], 0, 0
for bits, k in spec:
    qg += [bits, row]; row += k * bits // 32; K += k
groups = len(spec); N = 328
q_groups = torch.tensor(qg, dtype=torch.short)
gen = torch.Generator().manual_seed(1)
qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (row, N), generator=gen, dtype=torch.int64).to(torch.int32).to(dev)
sc = (torch.rand((groups, N), generator=gen) * 0.02 + 0.001).half().to(dev)
ze = (torch.randn((groups, N), generator=gen) * 0.1).half().to(dev)
perm = torch.randperm(K, generator=gen).to(torch.short).to(dev)
gmap = make_group_map(q_groups, row).to(dev)
_, rows = q_linear_cuda.mbwq_trans_qweight(qw, q_groups, True, K, groups, 4)
print("rows", rows)
x = torch.randn((1, K), generator=gen).half().to(dev)
y0 = q_linear_cuda.mbwq_exl2_forward(x, qw, sc, ze, perm, gmap, rows, False)
ent = [{"x": x, "qweight": qw, "scales": sc, "zeros": ze, "q_perm": perm, "q_group_map": gmap, "rows": rows, "y": torch.zeros((1, N), dtype=torch.float16, device=dev)}]
pl = MBWQExl2ForwardList(ent); pl(); torch.cuda.synchronize()
print("max diff", float((ent[0]["y"].float() - y0.float()).abs().max()), float(y0.float().abs().max()))
pl(); torch.cuda.synchronize()
print("2nd max diff", float((ent[0]["y"].float() - y0.float()).abs().max()))

y1 = q_linear_cuda.mbwq_exl2_forward(x, qw, sc, ze, torch.arange(K).to(torch.short).to(dev), gmap, rows, False)
ent2 = [{"x": x, "qweight": qw, "scales": sc, "zeros": ze, "q_perm": None, "q_group_map": gmap, "rows": rows, "y": torch.zeros((1, N), dtype=torch.float16, device=dev)}]
pl2 = MBWQExl2ForwardList(ent2); pl2(); torch.cuda.synchronize()
print("no perm: max diff", float((ent2[0]["y"].float() - y1.float()).abs().max()))
