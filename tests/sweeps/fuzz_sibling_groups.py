#!/usr/bin/env python3
"""Random call programs against the automatic sibling grouping of the module path (layers/qlinear/nbit/cuda/mpq_layer.SiblingGroup): a parent with
   3..6 MPQLinearCuda children (prepare_bie_layers attaches the group), and a program of calls per "forward" -- each call picks a layer and an
   input: the shared tensor, a temporary computed from it (freed right after the call: the allocator recycles its address), a clone, a row
   slice, the shared tensor after an in-place update, a second tensor -- repeated for several rounds with fresh or refilled inputs, the program
   mutated now and then (a call dropped / two swapped / row count changed).  EVERY output is compared with the same layer's own launch on the
   same input (grouping off): whatever the protocol concludes, it must never hand out another tensor's result.
   usage: python tests/sweeps/fuzz_sibling_groups.py [parents=40] [seed=1]   (test infrastructure: imports the tests' helpers)"""
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

DEV = T.DEV


def run(parents=40, seed=1):
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQLinearCuda, mpq_layer
    from bitorch_engine.utils.model_helper import prepare_bie_layers
    rng = np.random.default_rng(seed)
    bad, calls, served, grouped = [], 0, 0, 0
    for pi in range(parents):
        K = int(rng.choice([256, 512, 1024]))
        tdt = torch.bfloat16 if rng.random() < 0.5 else torch.float16
        dt = orc.BF16 if tdt == torch.bfloat16 else orc.F16
        n = int(rng.integers(3, 7))
        g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))

        class Parent(torch.nn.Module):
            def __init__(self):
                super().__init__()
                for i in range(n):
                    N = 64 * int(rng.integers(1, 9))
                    layer = MPQLinearCuda(K, N, w_bit=4, dtype=tdt, group_size=128, dq_group_size=32, use_gba_quant=True, asym=False)
                    layer.qweight.data = torch.randint(-2 ** 31, 2 ** 31 - 1, layer.qweight.shape, generator=g, dtype=torch.int64).to(torch.int32)
                    setattr(self, f"l{i}", layer)

        model = Parent()
        prepare_bie_layers(model)
        for i in range(n):
            l = getattr(model, f"l{i}")
            l.scales = (torch.rand(l.scales.shape, generator=g) * 0.01 + 0.005).to(tdt)
            l.zeros = (l.scales.float() * torch.rand(l.scales.shape, generator=g) * 15).to(tdt)
        model = model.to(DEV).eval()
        layers = [getattr(model, f"l{i}") for i in range(n)]
        M = int(rng.choice([1, 1, 1, 2, 4, 16, 20, 24, 32]))
        kinds = ["shared", "shared", "shared", "temp", "temp", "clone", "slice", "bumped", "other"]
        program = [(int(rng.integers(0, n)), str(rng.choice(kinds))) for _ in range(int(rng.integers(2, 2 * n + 2)))]
        h = torch.randn((M, K), generator=g).to(tdt).to(DEV)
        other = torch.randn((M, K), generator=g).to(tdt).to(DEV)
        for rnd in range(int(rng.integers(4, 9))):
            if rng.random() < 0.5:
                h = torch.randn((M, K), generator=g).to(tdt).to(DEV)   # a fresh tensor every forward (the usual case)
            else:
                h.copy_(torch.randn((M, K), generator=g).to(tdt))      # the same buffer refilled in place (static input, graph-style)
            if rng.random() < 0.25 and len(program) > 2:               # the caller changes its mind
                u = rng.random()
                if u < 0.4:
                    program.pop(int(rng.integers(0, len(program))))
                elif u < 0.8:
                    i, j = rng.integers(0, len(program), 2)
                    program[i], program[j] = program[j], program[i]
                else:
                    M = int(rng.choice([1, 2, 4, 16, 20, 24, 32]))
                    h = torch.randn((M, K), generator=g).to(tdt).to(DEV)
                    other = torch.randn((M, K), generator=g).to(tdt).to(DEV)
            before = dict(mpq_layer.GROUP_STATS)
            with torch.no_grad():
                for step, (li, kind) in enumerate(program):
                    layer = layers[li]
                    if kind == "shared":
                        x = h
                    elif kind == "temp":
                        x = h * float(1.0 + 0.25 * (step % 3))            # a temporary nobody else holds
                    elif kind == "clone":
                        x = h.clone()
                    elif kind == "slice":
                        x = h[: max(1, M // 2)]
                    elif kind == "bumped":
                        h.add_(0.125)                                      # in place: same address, new version
                        x = h
                    else:
                        x = other
                    y = layer(x)
                    saved, layer._bie_group = layer._bie_group, None       # the layer's own launch on the same input
                    ref = layer(x)
                    layer._bie_group = saved
                    calls += 1
                    try:
                        T.assert_close(y.reshape(-1, y.shape[-1]), ref.reshape(-1, ref.shape[-1]), dt, f"parent {pi} round {rnd} step {step} layer {li} input {kind} M={x.shape[0]}")
                    except AssertionError as e:
                        bad.append(str(e)[:300] + f" program={program}")
                    del x, y, ref
            d = {k: mpq_layer.GROUP_STATS[k] - before[k] for k in before}
            served += d["served_from_group"]
            grouped += d["grouped_launches"]
        if len(bad) > 5:
            break
    return {"parents": parents, "seed": seed, "calls": calls, "grouped_launches": grouped, "served_from_group": served, "bad": bad}


if __name__ == "__main__":
    print(json.dumps(run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1), indent=1))
