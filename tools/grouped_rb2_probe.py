#!/usr/bin/env python3
"""Sibling sets (q/k/v: 3 x 4096x4096; gate/up: 2 x 4096x11008) at 17..32 rows: ONE grouped launch of the matrix-pipe decode kernel with two row blocks (arm: BIE_LUT_MFMA_MAX_M=32)
   against the members' lone calls (default routing), fp16 and bf16.  us per set."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import Bench, BF16, capture, time_graph, make_layer, GROUP, WBIT  # noqa: E402

B = Bench(torch.device("cuda", 0))
arm = os.environ.get("BIE_LUT_MFMA_MAX_M") == "32"
out = {"arm": arm}
for dn, dt in (("f16", torch.float16), ("bf16", BF16)):
    code = B._hip.BF16 if dt == BF16 else B._hip.F16
    for what, k, ns in (("qkv", 4096, (4096, 4096, 4096)), ("gate_up", 4096, (11008, 11008)), ("qkv8k", 8192, (8192, 1024, 1024))):
        for M in (16, 17, 24, 32):
            gen = torch.Generator(device=B.dev).manual_seed(9)
            nl = 12
            groups = [[make_layer(B.dev, gen, k, n, WBIT, dt) for n in ns] for _ in range(nl)]
            x = torch.randn((M, k), generator=gen, device=B.dev).to(dt)
            ys = [torch.empty((M, n), dtype=dt, device=B.dev) for n in ns]
            cnt = len(ns)
            if arm or M <= 16:
                Narr = (ctypes.c_int * cnt)(*ns)
                ws = torch.zeros(max(B.L.bie_mpq_grouped_workspace_bytes(cnt, Narr, M, k, WBIT), 16), dtype=torch.uint8, device=B.dev)
                arr = lambda ts: (ctypes.c_void_p * cnt)(*[t.data_ptr() for t in ts])
                args = [(arr([l[0] for l in g]), arr([l[1] for l in g]), arr([l[2] for l in g])) for g in groups]
                yarr = arr(ys)

                def run(st):
                    for (q, s, z) in args:
                        rc = B.L.bie_mpq_forward_grouped(x.data_ptr(), cnt, q, s, z, None, yarr, Narr, ws.data_ptr(), ws.numel(), M, k, WBIT, GROUP, 0, code, st)
                        if rc:
                            raise RuntimeError(B.L.bie_last_error().decode())
            else:
                wss = [B.workspace(M, k, n) for n in ns]

                def run(st):
                    for g in groups:
                        for l, y, n, w in zip(g, ys, ns, wss):
                            B.forward(x, l, y, w, M, k, n, st)
            try:
                us = time_graph(capture(run), 8) / nl
            except Exception as e:
                us = str(e)[:80]
            out[f"{dn}_{what}_M{M}"] = us if isinstance(us, str) else round(us, 2)
print(json.dumps(out))
