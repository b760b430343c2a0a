#!/usr/bin/env python3
"""Every (tile height BM, K splits S) plan of the fused MFMA GEMM (mpq_gemm_kernel, 17 <= M < 897) over a grid of layer shapes and row counts, W4 g128 bf16:
   the data behind csrc/mpq_gemm_plan_table.inc (tools/gemm_plan_table_gen.py).  One JSON line per (K, N, M): the cost model's own time and {"BMxS": us}.
   Layers, activations and workspace are built once per cell; the plan knobs are re-read per launch under BIE_TUNING."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
os.environ["BIE_TUNING"] = "1"
os.environ["BIE_GEMM_DENSE"] = "0"
import torch  # noqa: E402
import bench  # noqa: E402
from bench import Bench, BF16, capture, time_graph, make_layer  # noqa: E402

bench.PREROLL_S = 0.008
B = Bench(torch.device("cuda", 0))
KD = tuple(int(v) for v in os.environ.get("SWEEP_K", "2048,4096,5120,8192,11008,14336").split(","))
ND = tuple(int(v) for v in os.environ.get("SWEEP_N", "2048,4096,6144,8192,11008,14336,28672").split(","))
MD = tuple(int(v) for v in os.environ.get("SWEEP_M", "17,32,48,64,96,128,192,256,384,512,768").split(","))
SS = (1, 2, 3, 4, 5, 6, 8, 10, 12, 16)
NL = 6


def clear():
    os.environ.pop("BIE_GEMM_BM", None)
    os.environ.pop("BIE_GEMM_S", None)


for k in KD:
    for n in ND:
        if k * n > 8192 * 28672:
            continue
        gen = torch.Generator(device=B.dev).manual_seed(5)
        layers = [make_layer(B.dev, gen, k, n) for _ in range(NL)]
        for M in MD:
            x = torch.randn((M, k), generator=gen, device=B.dev).to(BF16)
            y = torch.empty((M, n), dtype=BF16, device=B.dev)
            plans = [(bm, s) for bm in (32, 64, 128, 256) if not (bm > 32 and bm >= 2 * M) for s in SS if s == 1 or (k // 64) // s >= 2]
            need = 16
            for (bm, s) in plans:
                os.environ["BIE_GEMM_BM"], os.environ["BIE_GEMM_S"] = str(bm), str(s)
                need = max(need, B.L.bie_mpq_workspace_bytes(M, k, n, 4))
            clear()
            need = max(need, B.L.bie_mpq_workspace_bytes(M, k, n, 4))
            ws = torch.zeros(need, dtype=torch.uint8, device=B.dev)

            def t():
                try:
                    g = capture(lambda st: [B.forward(x, l, y, ws, M, k, n, st) for l in layers])
                    return round(time_graph(g, 6) / NL, 2)
                except Exception:
                    return None

            rec = {"K": k, "N": n, "M": M, "model": t(), "plans": {}}
            for (bm, s) in plans:
                os.environ["BIE_GEMM_BM"], os.environ["BIE_GEMM_S"] = str(bm), str(s)
                v = t()
                if v is not None:
                    rec["plans"][f"{bm}x{s}"] = v
            clear()
            print(json.dumps(rec), flush=True)
        del layers
        torch.cuda.empty_cache()
