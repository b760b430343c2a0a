#!/usr/bin/env python3
"""Vocabulary-sized and other very wide / very deep layers through bie_mpq_forward against the CPU restatement (sampled rows of x at large M):
   lm_head shapes (4096 x 32000, 4096 x 128256, 8192 x 128256), a very deep one (28672 x 8192) and a tall-skinny one (65536 x 256).
   usage: python tests/sweeps/big_shapes_check.py   (test infrastructure: imports oracle/)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from oracle import oracle as orc  # noqa: E402

out = []
rng = np.random.default_rng(7)
for (K, N, w_bit, gs, dt, asym) in ((4096, 32000, 4, 128, orc.BF16, False), (4096, 128256, 4, 128, orc.BF16, False), (4096, 128256, 4, 128, orc.F16, True),
                                   (8192, 128256, 4, 128, orc.BF16, False), (28672, 8192, 4, 128, orc.BF16, False), (65536, 256, 4, 128, orc.F16, False),
                                   (4096, 128256, 2, 128, orc.BF16, False), (4096, 32000, 8, 128, orc.F16, True)):
    qw, scales, zeros, gen = T.rand_case(rng, K, N, w_bit, gs, dt, asym)
    scales = (scales.float() * (64.0 / K) ** 0.5).to(scales.dtype)  # keep the sums in range for the deep layers
    if not asym:
        zeros = (scales.float() * (2 ** w_bit - 1) / 2).to(scales.dtype)
    for M in (1, 2, 16, 33, 512, 2048):
        x = torch.randn((M, K), generator=gen).to(T.TDT[dt])
        tag = f"w{w_bit} g{gs} {'asym' if asym else 'sym'} {'f16' if dt == orc.F16 else 'bf16'} K={K} N={N} M={M}"
        try:
            y = T.hip_forward(x, qw, scales, zeros, None, w_bit, gs, asym, None)
            torch.cuda.synchronize()
        except RuntimeError as e:
            out.append({"case": tag, "refused": str(e)[:160]})
            continue
        rows = list(range(M)) if M <= 2 else sorted(set(int(v) for v in np.linspace(0, M - 1, 3)))
        ref = T.oracle_forward(x[rows], qw, scales, zeros, None, w_bit, gs, asym, dt, None)
        try:
            T.assert_close(y[rows], ref, dt, tag)
            out.append({"case": tag, "ok": True, "finite": bool(torch.isfinite(y.float()).all())})
        except AssertionError as e:
            out.append({"case": tag, "bad": str(e)[:300]})
print(json.dumps({"cases": len(out), "ok": sum(1 for o in out if o.get("ok")), "problems": [o for o in out if not o.get("ok") or not o.get("finite")]}, indent=1))
