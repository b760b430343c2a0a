#!/usr/bin/env python3
"""Randomised sweep of the mixed-bit (exl2) layout: random band structures (bits 8/6/5/4/3/2 in descending order, groups of 32/64/128 k, whole
   groups), ragged N, random q_perm, every dispatch boundary of M -- load-time step (shuffle + table), dequantised weight BIT-EXACT and forward
   within the parity gate against the CPU restatement.  A refusal (RuntimeError) is fine; a wrong value, a NaN or a crash is a finding.
   usage: python tests/sweeps/fuzz_exl2_forward.py [cases=150] [seed=1]   (test infrastructure: imports oracle/)"""
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
from bitorch_engine.extensions import q_linear_cuda  # noqa: E402
from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map  # noqa: E402



def run(cases=150, seed=1):
    rng = np.random.default_rng(seed)
    MS = [1, 2, 3, 4, 5, 8, 9, 16, 17, 31, 32, 33, 47, 48, 49, 50, 64, 65, 100, 128, 129, 300, 513, 800]
    DEV = T.DEV
    refused, ok, bad = {}, 0, []
    for c in range(cases):
        gs = int(rng.choice([32, 32, 64, 128]))
        nbands = int(rng.integers(1, 5))
        bits = sorted(rng.choice([8, 6, 5, 4, 3, 2], size=nbands, replace=False).tolist(), reverse=True)
        qg, row, K = [], 0, 0
        for b in bits:
            for _ in range(int(rng.integers(1, 9))):
                qg += [b, row]
                row += b * gs // 32
                K += gs
        groups = len(qg) // 2
        mode = rng.random()
        N = 64 * int(rng.integers(1, 8)) if mode < 0.4 else (8 * int(rng.integers(1, 60)) if mode < 0.75 else int(rng.integers(1, 400)))
        M = int(rng.choice(MS))
        tag = f"bits={bits} gs={gs} K={K} N={N} M={M}"
        q_groups = torch.tensor(qg, dtype=torch.short)
        gen = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
        qw = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, (row, N), dtype=np.int64).astype(np.int32))
        scales = (torch.rand((groups, N), generator=gen) * 0.02 + 0.001).half()
        zeros = (torch.randn((groups, N), generator=gen) * 0.05).half()
        q_perm = torch.randperm(K, generator=gen).to(torch.short)
        try:
            gmap = make_group_map(q_groups, row)
            qd = qw.to(DEV).clone()
            _, rows = q_linear_cuda.mbwq_trans_qweight(qd, q_groups, True, K, groups, 4)
            d = lambda t: t.to(DEV)
            Wd = q_linear_cuda.mbwq_exl2fp_weight(qd, d(scales), d(zeros), d(q_perm), d(gmap), rows)
            x = torch.randn((M, K), generator=gen).half()
            y = q_linear_cuda.mbwq_exl2_forward(d(x), qd, d(scales), d(zeros), d(q_perm), d(gmap), rows, False)
            torch.cuda.synchronize()
        except RuntimeError as e:
            key = str(e)[:100]
            refused[key] = refused.get(key, 0) + 1
            continue
        Wo = orc.exl2_dequant(qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), q_perm.numpy(), q_groups.numpy(), K)
        if not np.array_equal(orc.torch_to_np(Wd), Wo):
            bad.append(tag + ": dequantised weight not bit-exact")
            continue
        try:
            T.assert_close(y, T.t16(orc.gemm(orc.torch_to_np(x), Wo, orc.F16), orc.F16), orc.F16, tag)
            ok += 1
        except AssertionError as e:
            bad.append(str(e)[:400])

    return {"cases": cases, "seed": seed, "ok": ok, "refused": refused, "bad": bad}


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 150, int(sys.argv[2]) if len(sys.argv) > 2 else 1), indent=1))
