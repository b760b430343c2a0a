import json, os, sys
sys.path.insert(0, "/root/repo/bitorch-engine_amd"); sys.path.insert(0, "/root/repo")
import torch
from bench import Bench
B = Bench(torch.device("cuda", 0))
out = {}
for nl in (1, 2, 4, 32, 1, 2, 32):
    r = B.decode_step(nl, 20 if nl < 8 else 5, 77)
    out.setdefault(f"layers={nl}", []).append(r["us_per_layer"])
print(json.dumps(out))
