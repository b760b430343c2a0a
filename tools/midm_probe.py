#!/usr/bin/env python3
"""Lone calls of 8..32 rows (bie_mpq_forward) against the same layer as a LIST OF ONE entry under several list plans (k-split default, x-sharing
   forced with K sliced 2..16 units per workgroup): us per layer over 40 distinct layers, one launch per layer in one graph.  Round 6 question: the
   x-sharing form did not exist when round 5 rejected "lone multi-row calls as a list of one"."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
os.environ["BIE_TUNING"] = "1"
import torch  # noqa: E402
from bench import Bench, BF16  # noqa: E402

B = Bench(torch.device("cuda", 0))
dts = {"bf16": BF16, "f16": torch.float16}
plans = {
    "list1_default": {},
    "list1_xs_gpw2": {"BIE_LUTM_XS_WANT_WAVES": "1", "BIE_LUTM_XS_MAX_GPW": "2", "BIE_LUTM_XS_MAX_S": "64", "BIE_LUTM_XS_MIN_M_BF16": "1", "BIE_LUTM_XS_BF16_WHOLE_PCT": "0"},
    "list1_xs_gpw4": {"BIE_LUTM_XS_WANT_WAVES": "1", "BIE_LUTM_XS_MAX_GPW": "4", "BIE_LUTM_XS_MAX_S": "64", "BIE_LUTM_XS_MIN_M_BF16": "1", "BIE_LUTM_XS_BF16_WHOLE_PCT": "0"},
    "list1_xs_gpw8": {"BIE_LUTM_XS_WANT_WAVES": "1", "BIE_LUTM_XS_MAX_GPW": "8", "BIE_LUTM_XS_MAX_S": "64", "BIE_LUTM_XS_MIN_M_BF16": "1", "BIE_LUTM_XS_BF16_WHOLE_PCT": "0"},
    "list1_xs_gpw16": {"BIE_LUTM_XS_WANT_WAVES": "1", "BIE_LUTM_XS_MAX_GPW": "16", "BIE_LUTM_XS_MAX_S": "64", "BIE_LUTM_XS_MIN_M_BF16": "1", "BIE_LUTM_XS_BF16_WHOLE_PCT": "0"},
}
knobs = sorted({k for p in plans.values() for k in p})
shapes = ((4096, 11008), (4096, 4096), (11008, 4096))
rows = tuple(int(m) for m in os.environ.get("MIDM_ROWS", "8,16,24,32").split(","))
for dn in os.environ.get("MIDM_DT", "bf16,f16").split(","):
    dt = dts[dn]
    for (k, n) in shapes:
        for M in rows:
            rec = {"dtype": dn, "K": k, "N": n, "M": M}
            rec["lone"] = B.gemv(k, n, 40, 10, 50 + M, M=M, dt=dt)["us_per_launch"]
            for name, env in plans.items():
                for kk in knobs:
                    os.environ.pop(kk, None)
                os.environ.update(env)
                try:
                    rec[name] = B.gemv_list(k, n, 40, 1, 10, 50 + M, M=M, dt=dt)["us_per_layer"]
                except Exception as e:  # a plan the library refuses
                    rec[name] = str(e)[:60]
            for kk in knobs:
                os.environ.pop(kk, None)
            print(json.dumps(rec), flush=True)
