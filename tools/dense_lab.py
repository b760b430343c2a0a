#!/usr/bin/env python3
"""Where the time of the M = 4096 dense-form launch goes: the same timing (sweep.time_case through bie_mpq_forward, both launches) under
variant libraries built by tools/build_variant.sh from mpq_dense.hip with -DBIE_DENSE_LAB=n (1 no epilogue stores, 2 no main loop,
3 the round-4 direct-store epilogue), plus the two kernels alone (events around each).  One subprocess per library (BIE_HIP_LIB).
   usage: python tools/dense_lab.py [variant ...]      (default: every directory under bitorch-engine_amd/variants/ named dense_*)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
sys.path.insert(0, os.path.join(%(root)r, "tools")); sys.path.insert(0, os.path.join(%(root)r, "bitorch-engine_amd"))
os.environ["BIE_TUNING"] = "1"
import sweep
from bitorch_engine import _hip
out = {}
for (M, K, N) in ((4096, 4096, 4096), (4096, 4096, 11008), (4096, 11008, 4096)):
    r = [sweep.time_case(M, K, N, _hip.BF16, layers=3, reps=6)["us"] for _ in range(3)]
    out[f"{M}x{K}x{N}"] = sorted(r)[1]
print(json.dumps(out))
'''
names = sys.argv[1:] or ["product"] + sorted(d for d in os.listdir(os.path.join(ROOT, "bitorch-engine_amd", "variants")) if d.startswith("dense_"))
for name in names:
    env = dict(os.environ)
    if name != "product":
        env["BIE_HIP_LIB"] = os.path.join(ROOT, "bitorch-engine_amd", "variants", name, "libbie_hip.so")
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True)
    print(name, p.stdout.strip() or p.stderr[-400:], flush=True)
