"""FMA form against matrix-pipe form of the W4 lookup GEMV at M = 1 as the launch grows (run once per BIE_LUT_MFMA_MIN_M setting)."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch, bench
dev = torch.device("cuda:0")
B = bench.Bench(dev)
out = {"MIN_M": os.environ.get("BIE_LUT_MFMA_MIN_M", "default")}
out["gemv_4096x11008"] = B.gemv(4096, 11008, 30, 10, 11)["us_per_launch"]
out["gemv_8192x8192"] = B.gemv(8192, 8192, 16, 10, 12)["us_per_launch"]
out["gemv_8192x28672"] = B.gemv(8192, 28672, 6, 10, 13)["us_per_launch"]
out["gemv_28672x8192"] = B.gemv(28672, 8192, 6, 10, 14)["us_per_launch"]
out["grouped_qkv"] = B.grouped(4096, (4096, 4096, 4096), 24, 10, 21, "qkv")["us_per_launch"]
out["grouped_gate_up"] = B.grouped(4096, (11008, 11008), 16, 10, 22, "gate/up")["us_per_launch"]
out["grouped_gate_up_70b"] = B.grouped(8192, (28672, 28672), 3, 10, 23, "gate/up 70B")["us_per_launch"]
print(json.dumps(out))
