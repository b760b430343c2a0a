#!/usr/bin/env python3
"""fp16 / W2 layer lists: the algebraic form (BIE_LIST_ALG=1, default) against the table form (=0), one subprocess per arm (the switch is read
once), bench.py's own list helper: 96 x 4096x4096, 40 x 4096x11008, 40 x 11008x4096, each layer its own x, ONE launch.
   usage: python tools/list_alg_ab.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "bitorch-engine_amd"))
import torch, bench
B = bench.Bench(torch.device("cuda:0"))
out = {}
for (dt, name) in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
    if name == "bf16" and os.environ.get("AB_BF16", "0") != "1":
        continue
    for w in (4, 2):
        for (k, n, nl) in ((4096, 4096, 96), (4096, 11008, 40), (11008, 4096, 40)):
            for M in (1, 2):
                if M == 2 and (k, n) != (4096, 4096):
                    continue
                r = B.gemv_list(k, n, nl, nl, 8, 5, w_bit=w, M=M, dt=dt)
                out[f"{name}_w{w}_M{M}_{k}x{n}"] = [r["us_per_layer"], r["roofline"]["frac"]]
print(json.dumps(out))
'''
res = {}
ARMS = {"d16 (default)": {}, "fp32 table": {"BIE_LIST_D16": "0"}, "algebraic": {"BIE_LIST_ALG": "1"}}
for arm, extra in ARMS.items():
    env = dict(os.environ, **extra)
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True)
    try:
        res[arm] = json.loads(p.stdout.strip().splitlines()[-1])
    except Exception:
        print("arm", arm, "failed:", p.stderr[-800:])
        res[arm] = {}
for k in res.get("d16 (default)", {}):
    print(f"{k:24s} " + "   ".join(f"{arm}: {res[arm].get(k, [0, 0])[0]:.3f} us {res[arm].get(k, [0, 0])[1]:.3f}" for arm in ARMS), flush=True)
