import json, os, sys
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bench import Bench
B = Bench(torch.device("cuda", 0))
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("BIE_") and k != "BIE_HIP_LIB"}}
out["8192x28672"] = B.gemv(8192, 28672, 6, 10, 6)["us_per_launch"]
out["gate_up_8192x28672"] = B.grouped(8192, (28672, 28672), 3, 10, 33, "")["us_per_launch"]
print(json.dumps(out))
