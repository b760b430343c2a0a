#!/usr/bin/env python3
"""Randomised sweep of the one-launch layer lists (MPQForwardList -> bie_mpq_list_*): 1..24 entries of random (K, N) (ragged N included), w_bit 4 / 2,
   groups 32..256, sym / asym, fp16 / bf16, 1..64 rows, bias on some entries, dependent chains now and then; every entry against the CPU
   restatement, a second launch of the same plan (generation words / granule tags carry over), lists small (K sliced over workgroups) and
   large (whole K per workgroup).  A refusal (RuntimeError) is fine; a wrong value, a NaN or a crash is a finding.
   usage: python tests/sweeps/fuzz_mpq_lists.py [cases=120] [seed=1]   (test infrastructure: imports oracle/)"""
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
from bitorch_engine.layers.qlinear.nbit.cuda import MPQForwardList  # noqa: E402



XS_KNOBS = ("BIE_LUTM_XS_WANT_WAVES", "BIE_LUTM_XS_MAX_GPW", "BIE_LUTM_XS_MAX_S", "BIE_LUTM_XS_MIN_M_BF16", "BIE_LUTM_XS_BF16_WHOLE_PCT")


def run(cases=120, seed=1):
    os.environ.setdefault("BIE_TUNING", "1")
    rng = np.random.default_rng(seed)
    refused, ok, bad, forms = {}, 0, [], {}
    for c in range(cases):
        w_bit = 4 if rng.random() < 0.75 else 2
        gs = int(rng.choice([32, 64, 128, 128, 256])) if w_bit == 4 else int(rng.choice([64, 128, 256]))
        dt = orc.F16 if rng.random() < 0.5 else orc.BF16
        asym = int(rng.random() < 0.35)
        M = int(rng.choice([1, 1, 1, 2, 2, 3, 4, 5, 8, 13, 16, 17, 24, 31, 32, 33, 40, 64]))
        if w_bit == 2 and M > 2:
            M = int(rng.choice([1, 2]))
        n = int(rng.integers(1, 25))
        chain = bool(rng.random() < 0.2) and M <= 2
        specs = []
        if chain:
            d = gs * int(rng.integers(1, 9))
            d = max(d, 64) // 64 * 64
            d = (d // gs) * gs if (d // gs) * gs else gs
            specs = [(d, d, bool(rng.random() < 0.5)) for _ in range(n)]
        else:
            for _ in range(n):
                K = gs * int(rng.integers(1, 1024 // gs + 1))
                mode = rng.random()
                N = 64 * int(rng.integers(1, 10)) if mode < 0.4 else (8 * int(rng.integers(1, 80)) if mode < 0.8 else 4 * int(rng.integers(1, 150)))
                if asym:
                    N = max(32 // w_bit, N // (32 // w_bit) * (32 // w_bit))
                specs.append((K, N, bool(rng.random() < 0.3)))
        # the x-sharing matrix-pipe form (list_xs_plan) is the default for BIG lists only: force it onto half of the W4 cases in one of its three plans
        # (the knobs are re-read per plan under BIE_TUNING)
        xs_plan = str(rng.choice(["", "", "", "whole_k", "sliced", "split_groups"])) if w_bit == 4 else ""
        for k_ in XS_KNOBS:
            os.environ.pop(k_, None)
        if xs_plan:
            os.environ.update(T._XS_PLANS[xs_plan])
            os.environ.update({"BIE_LUTM_XS_MIN_M_BF16": "1", "BIE_LUTM_XS_BF16_WHOLE_PCT": "0"})
        tag = f"w{w_bit} g{gs} asym={asym} {'f16' if dt == orc.F16 else 'bf16'} M={M} n={n} chain={chain} xs={xs_plan or '-'} specs={specs[:3]}"
        try:
            entries, host = T._list_case(specs, dt, w_bit, gs, asym, M, seed=int(rng.integers(1 << 30)), chain=chain)
            plan = MPQForwardList(entries, w_bit=w_bit, group_size=gs, asym=bool(asym))
            fk = f"form{getattr(plan, 'form', 'blocks')}"
            forms[fk] = forms.get(fk, 0) + 1
            plan.forward()
            torch.cuda.synchronize()
            first = [e["y"].clone() for e in entries]
            for e in entries:
                e["y"].fill_(float("nan"))
            plan.forward()
            torch.cuda.synchronize()
        except RuntimeError as e:
            key = str(e)[:100]
            refused[key] = refused.get(key, 0) + 1
            continue
        good = True
        prev = None
        for i, (e, (x, qw, scales, zeros, bias)) in enumerate(zip(entries, host)):
            xin = x if x is not None else prev
            ref = T.oracle_forward(xin, qw, scales, zeros, None, w_bit, gs, asym, dt, bias)
            prev = e["y"].cpu()  # a chain is judged link by link: the next entry's reference starts from what this launch produced
            try:
                yf, rf = T.to_f32(e["y"]), T.to_f32(ref)
                if chain:  # a long random chain overflows fp16 sooner or later: where BOTH sides are non-finite they agree (inf - inf = NaN on both)
                    both = ~np.isfinite(yf) & ~np.isfinite(rf)
                    yf, rf = np.where(both, 0.0, yf), np.where(both, 0.0, rf)
                ulp = 2.0 ** -7 if dt == orc.BF16 else 2.0 ** -10
                tol = 1e-3 * np.abs(rf).max() + ulp * np.abs(rf)
                if bias is not None:  # y = dt(dt(acc) + bias): one ulp of each magnitude involved (tests/sweeps/fuzz_mpq_forward.py)
                    bf = T.to_f32(bias)[None, :]
                    tol = tol + ulp * (np.abs(bf) + np.abs(rf - bf))
                nbad = int((~(np.abs(yf - rf) <= tol)).sum())
                if nbad:
                    raise AssertionError(f"{tag} entry {i}: {nbad} outside tolerance, max err {np.nanmax(np.abs(yf - rf)):.4g} vs max|ref| {np.abs(rf).max():.4g}")
                if not torch.equal(torch.nan_to_num(first[i]), torch.nan_to_num(e["y"])):
                    raise AssertionError(tag + f" entry {i}: the second launch of the plan differs from the first")
            except AssertionError as err:
                bad.append(str(err)[:400])
                good = False
                break
        ok += good

    for k_ in XS_KNOBS:
        os.environ.pop(k_, None)
    return {"cases": cases, "seed": seed, "ok": ok, "forms": forms, "refused": refused, "bad": bad}


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 1), indent=1))
