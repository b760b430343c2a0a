#!/usr/bin/env python3
"""Held-out check of the measured dispatch table (mpq_dense_table.inc): layer shapes and row counts that are NOT grid points -- fused (BIE_GEMM_DENSE=0), dense (=2),
   the table (default) and the round-3 rule alone (BIE_GEMM_DENSE_TABLE=0).  Prints one line per cell and the mean regret of both rules against the better form."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
os.environ["BIE_TUNING"] = "1"
import torch  # noqa: E402
from bench import Bench  # noqa: E402

B = Bench(torch.device("cuda", 0))
shapes = ((2560, 2560), (2560, 10240), (3584, 3584), (3584, 18944), (18944, 3584), (7168, 7168), (4096, 12288), (12288, 4096), (5120, 13824), (13824, 5120), (6656, 6656), (4096, 22016),
          (8192, 7168), (7168, 28672), (10240, 10240), (4096, 4096), (4096, 11008), (11008, 4096))
tot = {"table": 0.0, "round3": 0.0}
n = 0
for (k, n_) in shapes:
    for M in (1000, 1280, 1800, 2500, 3500, 4096):
        nl = max(2, min(6, int(6e8 // (k * n_))))
        r = {"K": k, "N": n_, "M": M}
        for name, env in (("fused", {"BIE_GEMM_DENSE": "0"}), ("dense", {"BIE_GEMM_DENSE": "2"}), ("table", {}), ("round3", {"BIE_GEMM_DENSE_TABLE": "0"})):
            for kk in ("BIE_GEMM_DENSE", "BIE_GEMM_DENSE_TABLE"):
                os.environ.pop(kk, None)
            os.environ.update(env)
            r[name] = B.gemm(M, k, n_, nl, 4, 5)["us_per_launch"]
        best = min(r["fused"], r["dense"])
        # which form a rule ran = the closer of the two timings
        for rule in ("table", "round3"):
            ran = "dense" if abs(r[rule] - r["dense"]) < abs(r[rule] - r["fused"]) else "fused"
            tot[rule] += r[ran] / best
            r[rule + "_ran"] = ran
        n += 1
        print(json.dumps(r), flush=True)
print(json.dumps({"cells": n, "mean_regret_table": round(tot["table"] / n, 4), "mean_regret_round3": round(tot["round3"] / n, 4)}))
