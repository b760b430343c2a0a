#!/usr/bin/env python3
"""bie_mpq_forward_grouped at 17 / 24 / 32 rows over sibling sets beyond the three the rule was drawn from, under the process environment: default (one two-row-block decode launch where
   bie_mpq_grouped_max_rows allows, the members' own calls otherwise) against BIE_LUT_RB2=0 (round-5 routing: every member a fused GEMM + finalize).  us per set."""
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
out = {"BIE_LUT_RB2": os.environ.get("BIE_LUT_RB2", "1")}
sets = (("qkv4k", 4096, (4096, 4096, 4096)), ("gqa4k", 4096, (4096, 1024, 1024)), ("gateup11k", 4096, (11008, 11008)), ("gateup14k", 4096, (14336, 14336)), ("qkv5k", 5120, (5120, 5120, 5120)),
        ("gateup13k", 5120, (13824, 13824)), ("gqa8k", 8192, (8192, 1024, 1024)), ("gateup28k", 8192, (28672, 28672)), ("qkv2k", 2048, (2048, 2048, 2048)), ("gateup8k_2k", 2048, (8192, 8192)))
for dn, dt in (("f16", torch.float16), ("bf16", BF16)):
    code = B._hip.BF16 if dt == BF16 else B._hip.F16
    for what, k, ns in sets:
        for M in (17, 24, 32):
            gen = torch.Generator(device=B.dev).manual_seed(9)
            nl = max(3, min(12, int(6e8 // (k * sum(ns) // 2))))
            groups = [[make_layer(B.dev, gen, k, n, WBIT, dt) for n in ns] for _ in range(nl)]
            x = torch.randn((M, k), generator=gen, device=B.dev).to(dt)
            ys = [torch.empty((M, n), dtype=dt, device=B.dev) for n in ns]
            cnt = len(ns)
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
            out[f"{dn}_{what}_M{M}"] = round(time_graph(capture(run), 6) / nl, 2)
            del groups
print(json.dumps(out))
