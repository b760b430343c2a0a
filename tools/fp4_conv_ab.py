#!/usr/bin/env python3
"""Binary conv2d 3x3 (ResNet-18 stage shapes) over the batch on one MI355X: the XNOR tap form against the matrix-pipe form (channel-minor
   bits -> FP4 image of the (pixel) x (tap, channel) matrix -> FP4 GEMM, NCHW epilogue), both through binary_conv_cpp.forward.
   usage: python tools/fp4_conv_ab.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd")); sys.path.insert(0, ROOT)
import torch
from bench import capture, time_graph
from bitorch_engine.extensions import binary_conv_cpp
from bitorch_engine.extensions._binary_common import pack_rows
dev = torch.device("cuda", 0)
for (C, HW) in ((512, 7), (256, 14), (128, 28), (64, 56)):
    w = torch.randn((C, C, 3, 3), device=dev)
    wp = pack_rows(w.reshape(C, -1)).contiguous()
    for B in (4, 8, 16, 32, 64, 128):
        if B * HW * HW * C * 9 > 3e9:
            continue
        x = torch.randn((B, C, HW, HW), device=dev)
        row = {}
        ys = {}
        for name, env in (("xnor_taps", "0"), ("fp4", "1")):
            os.environ["BIE_FP4_CONV_MIN_ROWS"] = env
            fn = lambda st: binary_conv_cpp.forward(x, wp, C, B * HW * HW, C * 9, 3, 1, 1, 1, HW)
            ys[name] = fn(None)
            row[name + "_us"] = round(time_graph(capture(fn), 10), 2)
        row["exact"] = bool(torch.equal(ys["xnor_taps"], ys["fp4"]))
        ops = 2.0 * B * HW * HW * C * C * 9
        row["fp4_TOPs"] = round(ops / row["fp4_us"] / 1e6, 1); row["xnor_TOPs"] = round(ops / row["xnor_taps_us"] / 1e6, 1)
        print(f"C{C} {HW}x{HW} B{B}", json.dumps(row), flush=True)
