#!/usr/bin/env python3
"""W4 M = 1 lists, bf16 and fp16: the D16 form's v_pk_fma_f32 / v_fma_mix_f32 FMAs (BIE_LIST_DOT=0) against the paired-lookup v_dot2 form (=1); bench.py's
list helper, one process (BIE_TUNING re-reads the switch), arms alternated; plus a bit-level comparison of the two forms' outputs.
   usage: python tools/list_dot_ab.py"""
import json, os, sys
os.environ["BIE_TUNING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
import bench
B = bench.Bench(torch.device("cuda:0"))
dev = torch.device("cuda:0")
# outputs of the two forms on one list
for dt in (torch.bfloat16, torch.float16):
    gen = torch.Generator(device=dev).manual_seed(5)
    layers = [bench.make_layer(dev, gen, 4096, 4096, 4, dt) for _ in range(8)]
    ys = {}
    for arm in ("0", "1"):
        os.environ["BIE_LIST_DOT"] = arm
        gen2 = torch.Generator(device=dev).manual_seed(6)
        y_all = torch.zeros((8, 4096), dtype=dt, device=dev)
        plan = B.make_list(layers, 4096, 4096, gen2, ys=[y_all[i:i + 1] for i in range(8)])
        plan.forward(); torch.cuda.synchronize()
        ys[arm] = y_all.float().clone()
    d = (ys["0"] - ys["1"]).abs()
    print(str(dt), "dot vs fma: max|d|", float(d.max()), "max|y|", float(ys["0"].abs().max()), "identical fraction", float((d == 0).float().mean()), flush=True)
for rnd in range(2):
    for (dt, name) in ((torch.bfloat16, "bf16"), (torch.float16, "f16")):
        for (k, n, nl) in ((4096, 4096, 96), (4096, 11008, 40), (11008, 4096, 40)):
            row = {}
            for arm in ("0", "1"):
                os.environ["BIE_LIST_DOT"] = arm
                r = B.gemv_list(k, n, nl, nl, 8, 5, dt=dt)
                row["dot" if arm == "1" else "fma"] = [r["us_per_layer"], r["roofline"]["frac"]]
            print(name, f"{k}x{n}", json.dumps(row), flush=True)
