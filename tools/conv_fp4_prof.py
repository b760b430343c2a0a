"""The matrix-pipe (FP4) form of the binary conv at B = 128 / 512, 200 calls each: run under rocprofv3 --kernel-trace --stats to see how the call splits over its launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
os.environ["BIE_CONV_FUSED_MAX_ROWS"] = os.environ.get("BIE_CONV_FUSED_MAX_ROWS", "0")
import torch
from bitorch_engine.extensions import _binary_common as bc
dev = torch.device("cuda", 0)
for B in [int(v) for v in (sys.argv[1:] or ["128"])]:
    x = torch.randn((B, 512, 7, 7), device=dev)
    w = torch.randn((512, 512, 3, 3), device=dev)
    wp = bc.pack_rows(w.reshape(512, -1)).contiguous()
    for _ in range(200):
        y = bc.conv2d(x, wp, 512, 3, 1, 1, 1, 1.0)
    torch.cuda.synchronize()
