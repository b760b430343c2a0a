#!/usr/bin/env python3
"""Binary conv 512->512 3x3 on 7x7 (and the 256 / 128-channel ResNet-18 stages) over batch, graph-timed, through binary_conv_cpp.forward.
   BIE_CONV_QUAD=0: the lane-per-channel tap kernel; BIE_CONV_DMA=0: no one-wave DMA form for small problems.
   usage: python tools/conv_ab.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd")); sys.path.insert(0, ROOT)
import torch
from bench import capture, time_graph
from bitorch_engine.extensions import binary_conv_cpp
from bitorch_engine.extensions._binary_common import pack_rows
dev = torch.device("cuda", 0)
out = {"env": {k: os.environ.get(k) for k in ("BIE_CONV_QUAD", "BIE_CONV_DMA") if os.environ.get(k)}}
for (C, HW) in ((512, 7), (256, 14), (128, 28)):
    w = torch.randn((C, C, 3, 3), device=dev)
    wp = pack_rows(w.reshape(C, -1)).contiguous()
    for B in (1, 4, 32, 128):
        x = torch.randn((B, C, HW, HW), device=dev)
        fn = lambda st: binary_conv_cpp.forward(x, wp, C, B * HW * HW, C * 9, 3, 1, 1, 1, HW)
        us = time_graph(capture(fn), 20)
        out[f"C{C}_{HW}x{HW}_B{B}"] = {"us": round(us, 2), "TOPs": round(2.0 * B * HW * HW * C * C * 9 / us / 1e6, 1)}
print(json.dumps(out))
