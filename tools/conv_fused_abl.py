"""Where the one-launch binary conv spends its time: ablations through BIE_CONV_FUSED_ABL (1 no x loads, 2 no popcount passes, 4 no LDS adds, 8 no weight loads)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bench import capture, time_graph
from bitorch_engine.extensions import _binary_common as bc
dev = torch.device("cuda", 0)
os.environ["BIE_CONV_FUSED_MAX_ROWS"] = "100000000"
for B in (1, 32, 128):
    x = torch.randn((B, 512, 7, 7), device=dev)
    w = torch.randn((512, 512, 3, 3), device=dev)
    wp = bc.pack_rows(w.reshape(512, -1)).contiguous()
    row = {"B": B}
    for abl in (0, 1, 2, 3, 4, 8, 15):
        os.environ["BIE_CONV_FUSED_ABL"] = str(abl)
        fn = lambda s_: [bc.conv2d(x, wp, 512, 3, 1, 1, 1, 1.0) for _ in range(16)]
        row[f"abl{abl}"] = round(time_graph(capture(fn), 10) / 16, 2)
    print(json.dumps(row), flush=True)
