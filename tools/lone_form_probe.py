#!/usr/bin/env python3
"""Lone decode launches (M = 1, W4 g128 bf16) over layer shapes beyond Llama-7B's under the process environment (BIE_DECODE_INLINE = 0 never / 2 always the inline list form;
   default: from 40 MB of packed weights): us per launch over distinct layers in one graph."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import Bench  # noqa: E402

B = Bench(torch.device("cuda", 0))
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("BIE_DECODE") or k.startswith("BIE_LUT")}}
for (k, n) in ((2048, 2048), (2048, 8192), (4096, 4096), (5120, 5120), (4096, 11008), (11008, 4096), (4096, 14336), (14336, 4096), (5120, 13824), (13824, 5120), (8192, 8192), (8192, 10240),
               (4096, 28672), (8192, 28672), (28672, 8192)):
    nl = max(6, min(96, int(9e8 // (k * n // 2))))
    out[f"{k}x{n}"] = B.gemv(k, n, nl, 10, 3)["us_per_launch"]
print(json.dumps(out))
