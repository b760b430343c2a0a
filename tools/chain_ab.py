#!/usr/bin/env python3
"""Dependent chains inside one launch against per-layer launches (4096x4096 W4 g128 bf16, M = 1)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from list_ab import list_case, B
out = {}
for (pl, ch) in ((4, 4), (8, 8), (32, 8), (96, 4), (96, 2)):
    out[f"chain{ch}_per_launch{pl}"] = list_case(4096, 4096, 96, pl, chain=ch)
out["chain4_8192x8192"] = list_case(8192, 8192, 24, 4, chain=4)
out["per_layer_launch_8192x8192"] = B.gemv(8192, 8192, 24, 10, 1)
print(json.dumps(out))
