#!/usr/bin/env python3
"""A/B of the matrix-pipe list form (2 <= M <= 32 rows, bie_mpq_list_*) on one MI355X: us per layer and HBM fraction per row count for
   the current process environment (BIE_LUTM_NW32 = 8 | 4 chooses the workgroup size of the 17..32-row instance).
   usage: [BIE_LUTM_NW32=4] python tools/lutm_ab.py [f16]"""
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
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("BIE_LUTM")}, "dtype": str(dt)}
shapes = ((4096, 11008, 40), (4096, 4096, 96), (11008, 4096, 40))
if os.environ.get("LUTM_AB_SHAPES"):  # e.g. "2" or "0,2": indices into the list above
    shapes = tuple(shapes[int(i)] for i in os.environ["LUTM_AB_SHAPES"].split(","))
rows = tuple(int(m) for m in os.environ.get("LUTM_AB_ROWS", "2,8,16,17,24,32").split(","))
for (k, n, nl) in shapes:
    for M in rows:
        r = B.gemv_list(k, n, nl, nl, 10, 100 + M, M=M, dt=dt)
        out[f"{k}x{n}_M{M}"] = {"us": r["us_per_layer"], "frac": r["roofline"]["frac"]}
print(json.dumps(out))
