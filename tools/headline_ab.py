#!/usr/bin/env python3
"""The headline launch (96 x 4096x4096 W4 g128 bf16, one list launch) under two or more builds of libbie_hip.so, alternating, one box:
   python tools/headline_ab.py <variant dir name under bitorch-engine_amd/variants | product> ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1:] or ["base", "product"]
res = {n: [] for n in names}
for rnd in range(3):
    for n in names:
        env = dict(os.environ)
        if n.startswith("env:"):  # "env:K=V,K=V": the product library under these environment variables
            env.update(dict(kv.split("=", 1) for kv in n[4:].split(",")))
        elif n != "product":
            env["BIE_HIP_LIB"] = os.path.join(ROOT, "bitorch-engine_amd", "variants", n, "libbie_hip.so")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(p.stdout.strip().splitlines()[-1])
            res[n].append((d["roofline"]["frac"], d["cold_start"]["roofline_frac"], d["verified"]))
        except Exception:
            res[n].append(("failed", p.stderr[-300:]))
for n in names:
    print(n, res[n], flush=True)
