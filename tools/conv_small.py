"""Small-batch binary conv (ResNet-18 deep layers) through binary_conv_cpp.forward: us per call under the current BIE_* environment."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bench import capture, time_graph
from bitorch_engine.extensions import binary_conv_cpp
from bitorch_engine.extensions._binary_common import pack_rows
dev = torch.device("cuda", 0)
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("BIE_") and k != "BIE_HIP_LIB"}}
for (C, H) in ((512, 7), (256, 14), (128, 28)):
    w = torch.randn((C, C, 3, 3), device=dev)
    wp = pack_rows(w.reshape(C, -1)).contiguous()
    for B in (1, 2, 4, 8):
        x = torch.randn((B, C, H, H), device=dev)
        fn = lambda st: binary_conv_cpp.forward(x, wp, C, B * H * H, C * 9, 3, 1, 1, 1, H)
        out[f"{C}x{H}x{H}_B{B}"] = round(time_graph(capture(lambda st: [fn(st) for _ in range(16)]), 20) / 16, 2)
print(json.dumps(out))
