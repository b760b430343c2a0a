#!/usr/bin/env python3
"""Held-out check of the measured plan table of the fused MFMA GEMM (mpq_gemm_plan_table.inc): shapes and row counts that are not grid points, the table on (default) against
   BIE_GEMM_PLAN_TABLE=0 (the cost model alone), dense form off so that only the fused kernel is compared.  Also the grid cells themselves (a second box: does the gain hold?)."""
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
from bench import Bench  # noqa: E402

bench.PREROLL_S = 0.01
B = Bench(torch.device("cuda", 0))
held = ((2560, 2560), (2560, 10240), (3584, 3584), (3584, 18944), (7168, 7168), (4096, 12288), (12288, 4096), (5120, 13824), (13824, 5120), (6656, 6656), (4096, 22016), (4096, 5120))
grid = ((4096, 4096), (4096, 11008), (11008, 4096), (2048, 2048), (4096, 28672), (8192, 8192), (4096, 2048), (4096, 14336))
for tag, shapes, Ms in (("held_out_shapes", held, (20, 56, 112, 224, 448, 850)), ("grid_shapes_other_rows", grid, (20, 40, 56, 80, 112, 160, 224, 320, 448, 640, 850)), ("grid", grid, (17, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768))):
    tot_on = tot_off = 0.0
    n = 0
    worst = []
    for (k, n_) in shapes:
        for M in Ms:
            r = {}
            for name, v in (("on", "1"), ("off", "0"), ("on2", "1"), ("off2", "0")):
                os.environ["BIE_GEMM_PLAN_TABLE"] = v
                r[name] = B.gemm(M, k, n_, 6, 6, 5)["us_per_launch"]
            on, off = min(r["on"], r["on2"]), min(r["off"], r["off2"])
            tot_on += on / min(on, off)
            tot_off += off / min(on, off)
            n += 1
            worst.append((round(on / off, 3), k, n_, M))
            print(json.dumps({"set": tag, "K": k, "N": n_, "M": M, "table_us": on, "model_us": off}), flush=True)
    worst.sort()
    print(json.dumps({"set": tag, "cells": n, "mean_table_over_best": round(tot_on / n, 4), "mean_model_over_best": round(tot_off / n, 4), "best5": worst[:5], "worst5": worst[-5:]}), flush=True)
