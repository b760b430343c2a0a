#!/usr/bin/env python3
"""Lone decode launches (bie_mpq_forward, one layer per launch, 40 distinct layers in one HIP graph): us per launch for the current process
   environment.  usage: python tools/lone_ab.py [f16|bf16]   (LONE_AB_ROWS=1,2,4  LONE_AB_SHAPES=0,1,2)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import Bench, BF16  # noqa: E402

B = Bench(torch.device("cuda", 0))
dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "f16") else BF16
shapes = ((4096, 4096, 64), (4096, 11008, 40), (11008, 4096, 40))
if os.environ.get("LONE_AB_SHAPES"):
    shapes = tuple(shapes[int(i)] for i in os.environ["LONE_AB_SHAPES"].split(","))
rows = tuple(int(m) for m in os.environ.get("LONE_AB_ROWS", "1,2,4,8,16").split(","))
out = {"dtype": str(dt)}
for (k, n, nl) in shapes:
    for M in rows:
        r = B.gemv(k, n, nl, 10, 200 + M, M=M, dt=dt)
        out[f"{k}x{n}_M{M}"] = {"us": r["us_per_launch"], "frac": r["roofline"]["frac"]}
print(json.dumps(out))
