#!/usr/bin/env python3
"""A/B of the one-launch-many-layers decode (bie_mpq_list_*) on one MI355X: per-layer microseconds and HBM fraction for
   list sizes / shapes / chains, next to the per-layer launches of bie_mpq_forward.  Kernel variants are chosen per PROCESS through
   BIE_LIST_VAR (0 scalar FMAs, 1 v_pk_fma_f32 pairs, 3 stream only), BIE_LIST_WANT_WAVES, BIE_LIST_MAX_GPW.
   usage: python tools/list_ab.py [quick]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import Bench, make_layer, capture, time_graph, alg_bytes, BF16, GROUP  # noqa: E402
from bitorch_engine.layers.qlinear.nbit.cuda import MPQForwardList  # noqa: E402

dev = torch.device("cuda", 0)
B = Bench(dev)


def list_case(k, n, nl, per_launch, chain=0, reps=20, seed=1):
    """nl distinct layers, `per_launch` entries per list launch (nl % per_launch == 0); chain > 0: entries form dependent chains of
    that length (needs k == n)."""
    gen = torch.Generator(device=dev).manual_seed(seed)
    wb = int(os.environ.get("LIST_AB_WBIT", "4"))                       # PMC passes of other instances: W2, fp16
    tdt = torch.float16 if os.environ.get("LIST_AB_DT") == "f16" else BF16
    layers = [make_layer(dev, gen, k, n, wb, tdt) for _ in range(nl)]
    plans = []
    for p0 in range(0, nl, per_launch):
        entries = []
        for i in range(per_launch):
            qw, sc, ze = layers[p0 + i]
            dep = i - 1 if (chain and i % chain) else -1
            x = entries[-1]["y"] if dep >= 0 else torch.randn((1, k), generator=gen, device=dev).to(tdt)
            entries.append({"x": x, "qweight": qw, "scales": sc, "zeros": ze, "y": torch.empty((1, n), dtype=tdt, device=dev), "depends_on": dep})
        plans.append(MPQForwardList(entries, w_bit=wb, group_size=GROUP))
    if os.environ.get("LIST_AB_NOGRAPH"):  # PMC passes: every launch a dispatch row of its own
        for _ in range(3):
            for p in plans:
                p.forward()
        torch.cuda.synchronize()
        return {"K": k, "N": n, "layers": nl, "per_launch": per_launch, "nograph": True}
    passes = int(os.environ.get("LIST_AB_PASSES", "1"))  # passes per captured graph
    g = capture(lambda st: [p.forward(st) for _ in range(passes) for p in plans])
    us = time_graph(g, reps) / nl / passes
    b = alg_bytes(1, k, n)
    return {"K": k, "N": n, "layers": nl, "per_launch": per_launch, "chain": chain, "us_per_layer": round(us, 3), "frac": round(b / us / 1e3 / 8000.0, 4)}


def main():
    if len(sys.argv) > 3 and sys.argv[1] == "shape":  # PMC / rocprof passes: one list launch form of one shape, e.g. `shape 4096 11008 40`
        k, n, nl = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
        print(json.dumps(list_case(k, n, nl, nl, reps=5)))
        return
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    tag = {k: os.environ.get(k) for k in ("BIE_LIST_VAR", "BIE_LIST_WANT_WAVES", "BIE_LIST_MAX_GPW", "BIE_LIST_H") if os.environ.get(k)}
    out = {"env": tag}
    out["list96x1"] = list_case(4096, 4096, 96, 96)
    if not quick:
        out["list96x4"] = list_case(4096, 4096, 96, 24)
        out["list96x12"] = list_case(4096, 4096, 96, 8)
        out["list96x24"] = list_case(4096, 4096, 96, 4)
        out["list96x96"] = list_case(4096, 4096, 96, 1)
        out["chain4"] = list_case(4096, 4096, 96, 4, chain=4)
        out["chain8_in_list32"] = list_case(4096, 4096, 96, 32, chain=8)
        out["list40_4096x11008"] = list_case(4096, 11008, 40, 40, reps=10)
        out["list40_11008x4096"] = list_case(11008, 4096, 40, 40, reps=10)
        out["list6_8192x28672"] = list_case(8192, 28672, 6, 6, reps=10)
        out["per_layer_launch_4096x4096"] = B.gemv(4096, 4096, 96, 20, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
