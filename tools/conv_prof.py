import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bitorch-engine_amd"))
import torch
from bitorch_engine.extensions import binary_conv_cpp
from bitorch_engine.extensions._binary_common import pack_rows
dev = torch.device("cuda:0")
for B in (1, 32):
    x = torch.randn((B, 512, 7, 7), device=dev)
    w = torch.randn((512, 512, 3, 3), device=dev)
    wp = pack_rows(w.reshape(512, -1)).contiguous()
    for _ in range(20):
        y = binary_conv_cpp.forward(x, wp, 512, B * 49, 512 * 9, 3, 1, 1, 1, 7)
    torch.cuda.synchronize()
