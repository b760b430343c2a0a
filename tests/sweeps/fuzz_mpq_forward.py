#!/usr/bin/env python3
"""Randomised shape sweep of the W-n / A16 forward (extensions.q_linear_cuda.mpq_forward_impl -> bie_mpq_forward) against the CPU restatement:
   w_bit 1/2/4/8, group sizes 32..256, sym / asym, fp16 / bf16, ragged N (not a multiple of 64 / 8 / 4), every dispatch boundary of M
   (1, 2, 3, 8, 9, 16, 17, 32, 33, 64, 65, ... 1024+), bias on / off, act-order g_idx now and then.  A configuration may be REFUSED (a
   RuntimeError with the library's message: loud is fine); a wrong value, a NaN or a crash is a finding.
   usage: python tests/sweeps/fuzz_mpq_forward.py [cases=200] [seed=1]   (test infrastructure: imports oracle/)"""
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



def run(cases=200, seed=1):
    rng = np.random.default_rng(seed)
    MS = [1, 2, 3, 4, 7, 8, 9, 15, 16, 17, 31, 32, 33, 48, 63, 64, 65, 100, 127, 128, 129, 255, 256, 300, 511, 512, 1023, 1024, 1025, 1500]
    refused, ok, bad = {}, 0, []
    for c in range(cases):
        w_bit = int(rng.choice([1, 2, 4, 4, 4, 8]))
        gs = int(rng.choice([32, 64, 128, 128, 256]))
        dt = orc.F16 if rng.random() < 0.5 else orc.BF16
        asym = bool(rng.random() < 0.4)
        K = gs * int(rng.integers(1, 2048 // gs + 1))
        mode = rng.random()
        if mode < 0.3:
            N = 64 * int(rng.integers(1, 12))
        elif mode < 0.6:
            N = 8 * int(rng.integers(1, 90))
        elif mode < 0.8:
            N = 4 * int(rng.integers(1, 180))
        else:
            N = int(rng.integers(1, 700))
        if asym:
            N = max(32 // w_bit, N // (32 // w_bit) * (32 // w_bit))  # the packed zero points need whole words
        M = int(rng.choice(MS))
        if M * K * N > 3e8:
            M = int(rng.choice(MS[:12]))
        qw, scales, zeros, gen = T.rand_case(rng, K, N, w_bit, gs, dt, asym)
        x = torch.randn((M, K), generator=gen).to(T.TDT[dt])
        bias = torch.randn((N,), generator=gen).to(T.TDT[dt]) if rng.random() < 0.3 else None
        g_idx = None
        if rng.random() < 0.15:
            g_idx = (torch.arange(K, dtype=torch.int32) // gs)[torch.randperm(K, generator=gen)]
        tag = f"w{w_bit} g{gs} {'asym' if asym else 'sym'} {'f16' if dt == orc.F16 else 'bf16'} M={M} K={K} N={N} bias={bias is not None} gidx={g_idx is not None}"
        try:
            y = T.hip_forward(x, qw, scales, zeros, g_idx, w_bit, gs, asym, bias)
            torch.cuda.synchronize()
        except RuntimeError as e:
            key = str(e)[:90]
            refused[key] = refused.get(key, 0) + 1
            continue
        ref = T.oracle_forward(x, qw, scales, zeros, g_idx, w_bit, gs, asym, dt, bias)
        try:
            if bias is None:
                T.assert_close(y, ref, dt, tag)
            else:
                # y = dt(dt(acc) + bias): where the two nearly cancel, ONE ulp of the rounded accumulator (or of the final sum) is many ulps of the small
                # result, and the accumulator may round the other way on a 1e-7 difference in summation order -- the gate widens by one ulp of each
                # magnitude involved (seen: acc -2.21875 / -2.234375, bias 1.7265625 -> -0.4921875 / -0.5078125)
                yf, rf, bf = T.to_f32(y), T.to_f32(ref), T.to_f32(bias)[None, :]
                ulp = 2.0 ** -7 if dt == orc.BF16 else 2.0 ** -10
                tol = 1e-3 * np.abs(rf).max() + ulp * (np.abs(rf) + np.abs(bf) + np.abs(rf - bf))
                nbad = int((~(np.abs(yf - rf) <= tol)).sum())
                if nbad:
                    raise AssertionError(f"{tag}: {nbad} outside tolerance, max err {np.abs(yf - rf).max():.4g} vs max|ref| {np.abs(rf).max():.4g}")
            ok += 1
        except AssertionError as e:
            bad.append(str(e)[:400])

    return {"cases": cases, "seed": seed, "ok": ok, "refused": refused, "bad": bad}


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 1), indent=1))
