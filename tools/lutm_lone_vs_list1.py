#!/usr/bin/env python3
"""Lone calls with 2..32 rows (bie_mpq_forward: the matrix-pipe kernel's per-layer plan) against the SAME layer presented as a list of one entry
   (bie_mpq_list_*: four-wave workgroups, K sliced by the list plan).  us per launch, 40 distinct layers in one captured graph."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import Bench  # noqa: E402

B = Bench(torch.device("cuda", 0))
out = {}
for (k, n) in ((4096, 4096), (4096, 11008), (11008, 4096)):
    for M in (2, 4, 8, 16, 32):
        lone = B.gemv(k, n, 40, 10, 300 + M, M=M)["us_per_launch"] if M <= 32 else None
        one = B.gemv_list(k, n, 40, 1, 10, 300 + M, M=M)["us_per_layer"]
        out[f"{k}x{n}_M{M}"] = {"lone_us": lone, "list_of_one_us": one}
print(json.dumps(out, indent=0))
