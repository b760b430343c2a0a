#!/usr/bin/env python3
"""Producer form (mpq_prod.hip, BIE_GEMM_PROD=2 forced / 0 off) against the dense form, bie_mpq_forward per call, bf16 W4 g128, one box; plus a
correctness probe against the dense form's output (same weights, fp32 accumulation: equal up to association order).
   usage: python tools/prod_ab.py [gm ...]"""
import json, os, sys
os.environ["BIE_TUNING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd")); sys.path.insert(0, ROOT)
import torch
import sweep
from bitorch_engine import _hip
from bitorch_engine.extensions import q_linear_cuda
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
K, N, M = 4096, 4096, 4096
qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), dtype=torch.int64, generator=gen).to(torch.int32).to(dev)
sc = (torch.rand((K // 128, N), generator=gen) * 0.01 + 0.005).bfloat16().to(dev)
ze = (sc.float().cpu() * torch.rand((K // 128, N), generator=gen) * 15).bfloat16().to(dev)
x = torch.randn((M, K), generator=gen).bfloat16().to(dev)
ys = {}
for name, v in (("dense", "0"), ("prod", "2")):
    os.environ["BIE_GEMM_PROD"] = v
    ys[name] = q_linear_cuda.mpq_forward_impl(x, qw, sc, ze, None, 4, 0, 128).float()
torch.cuda.synchronize()
d = (ys["prod"] - ys["dense"]).abs()
print("prod vs dense: max|d|", float(d.max()), "max|y|", float(ys["dense"].abs().max()), "exact frac", float((d == 0).float().mean()), flush=True)
DT = _hip.F16 if os.environ.get("AB_DT") == "f16" else _hip.BF16
gms = [int(v) for v in sys.argv[1:]] or [1, 2, 4]
for (K, N) in ((4096, 4096), (4096, 11008), (11008, 4096), (8192, 8192)):
    for M in (2048, 4096, 8192):
        if 2.0 * M * K * N > 4.5e12:
            continue
        row = {}
        os.environ["BIE_GEMM_PROD"] = "0"
        row["dense_us"] = sweep.time_case(M, K, N, DT, layers=3, reps=4)["us"]
        os.environ["BIE_GEMM_PROD"] = "2"
        for gm in gms:
            os.environ["BIE_GEMM_PROD_GM"] = str(gm)
            row[f"prod_gm{gm}_us"] = sweep.time_case(M, K, N, DT, layers=3, reps=4)["us"]
        best = min(v for k, v in row.items() if k.startswith("prod"))
        row["prod/dense"] = round(best / row["dense_us"], 3)
        row["TFLOPs"] = round(2.0 * M * K * N / best / 1e6, 1)
        print(f"K{K} N{N} M{M}", json.dumps(row), flush=True)
