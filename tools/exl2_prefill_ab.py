#!/usr/bin/env python3
"""exl2 (3/2-bit g32, random q_perm) prefill form against (a) the matrix-pipe streaming kernel at the rows both take (49..64) and (b) the
MPQ W4 dense form of the same shape; one box.   usage: python tools/exl2_prefill_ab.py"""
import json, os, sys
os.environ["BIE_TUNING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd")); sys.path.insert(0, ROOT)
import torch
import sweep
from bitorch_engine import _hip
from bitorch_engine.extensions import q_linear_cuda
from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
dev = torch.device("cuda:0")


def layer(K, N, gen):
    qg, row = [], 0
    for b in (3, 2):
        for _ in range(K // 2 // 32):
            qg += [b, row]
            row += b
    groups = len(qg) // 2
    q_groups = torch.tensor(qg, dtype=torch.short)
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (row, N), dtype=torch.int64, generator=gen).to(torch.int32).to(dev)
    sc = (torch.rand((groups, N), generator=gen) * 0.02 + 0.001).half().to(dev)
    ze = (torch.randn((groups, N), generator=gen) * 0.05).half().to(dev)
    perm = torch.randperm(K, generator=gen).to(torch.short).to(dev)
    gmap = make_group_map(q_groups, row).to(dev)
    _, rows = q_linear_cuda.mbwq_trans_qweight(qw, q_groups, True, K, groups, 4)
    return qw, sc, ze, perm, gmap, rows


def time_exl2(M, K, N, nl=3, reps=5):
    gen = torch.Generator().manual_seed(1)
    Ls = [layer(K, N, gen) for _ in range(nl)]
    x = torch.randn((M, K), generator=gen).half().to(dev)
    run = lambda: [q_linear_cuda.mbwq_exl2_forward(x, *l, False) for l in Ls]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        run()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * nl)


def main():
  for (K, N) in ((4096, 4096), (4096, 11008), (11008, 4096)):
      for M in (49, 64):
          row = {}
          for name, v in (("stream", "65"), ("dense", "49")):
              os.environ["BIE_EXL2_DENSE_MIN_M"] = v
              row[name + "_us"] = round(time_exl2(M, K, N), 2)
          print(f"K{K} N{N} M{M}", json.dumps(row), flush=True)
      os.environ["BIE_EXL2_DENSE_MIN_M"] = "49"
      for M in (128, 512, 2048, 4096):
          us = time_exl2(M, K, N, nl=2, reps=3)
          mpq = sweep.time_case(M, K, N, _hip.F16, layers=2, reps=3)["us"]
          print(f"K{K} N{N} M{M}", json.dumps({"exl2_dense_us": round(us, 2), "mpq_w4_us": mpq, "exl2/mpq": round(us / mpq, 3), "exl2_TFLOPs": round(2.0 * M * K * N / us / 1e6, 1)}), flush=True)


if __name__ == "__main__":
    main()
