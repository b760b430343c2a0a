"""Lone decode launches (bie_mpq_forward / bie_mpq_forward_grouped, M = 1, W4 g128 bf16) under the current BIE_* environment: us per launch,
rotating over distinct layers (> 256 MiB in all), HIP graph.  One line of JSON."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bench import Bench
dev = torch.device("cuda", 0)
B = Bench(dev)
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("BIE_") and k != "BIE_HIP_LIB"}}
out["4096x4096"] = B.gemv(4096, 4096, 64, 10, 1)["us_per_launch"]
out["4096x11008"] = B.gemv(4096, 11008, 24, 10, 2)["us_per_launch"]
out["11008x4096"] = B.gemv(11008, 4096, 24, 10, 3)["us_per_launch"]
out["qkv"] = B.grouped(4096, (4096, 4096, 4096), 20, 10, 4, "")["us_per_launch"]
out["gate_up"] = B.grouped(4096, (11008, 11008), 12, 10, 5, "")["us_per_launch"]
out["8192x28672"] = B.gemv(8192, 28672, 4, 10, 6)["us_per_launch"]
if "--step" in sys.argv:
    out["decode_step_us_per_layer"] = B.decode_step(16, 5, 77)["us_per_layer"]
print(json.dumps(out))
