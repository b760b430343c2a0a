#!/usr/bin/env python3
"""REJECTED (profiles/r05_prefetch_probe_*.txt; needs the bie_prefetch entry of profiles/r05_prefetch_kernel_rejected.patch).  Hypothesis probe: the Llama-7B decode step (bench.Bench.decode_step's launches: grouped q/k/v -> o -> grouped gate/up -> down, true
dependencies) with the NEXT launch's packed weights read on a second stream while the current launch runs, so that they sit in the 256 MB
Infinity Cache when their own launch asks for them.  The prefetch here is a crude stand-in (torch.sum over the int32 tensor on the side stream).
   usage: python tools/prefetch_probe.py"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
import bench
from bench import BF16, WBIT, GROUP, time_graph

dev = torch.device("cuda:0")
B = bench.Bench(dev)
L = B.L
hidden, inter, n_layers = 4096, 11008, 32
gen = torch.Generator(device=dev).manual_seed(77)


def unit_layer(k, n):
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (k * WBIT // 32, n), dtype=torch.int32, generator=gen, device=dev)
    s0 = 1.0 / (21.25 * k) ** 0.5
    sc = (s0 * (0.8 + 0.4 * torch.rand((k // GROUP, n), generator=gen, device=dev))).to(BF16)
    ze = (sc.float() * (7.5 + 0.5 * (torch.rand((k // GROUP, n), generator=gen, device=dev) - 0.5))).to(BF16)
    return qw, sc, ze


h = [torch.randn((1, hidden), generator=gen, device=dev).to(BF16)] + [torch.empty((1, hidden), dtype=BF16, device=dev) for _ in range(n_layers)]
steps = []
for l in range(n_layers):
    qkv = [unit_layer(hidden, hidden) for _ in range(3)]
    yq = [torch.empty((1, hidden), dtype=BF16, device=dev) for _ in range(3)]
    o = unit_layer(hidden, hidden)
    yo = torch.empty((1, hidden), dtype=BF16, device=dev)
    gu = [unit_layer(hidden, inter) for _ in range(2)]
    yg = [torch.empty((1, inter), dtype=BF16, device=dev) for _ in range(2)]
    dn = unit_layer(inter, hidden)
    steps += [(h[l], qkv, yq), (yq[0], [o], [yo]), (yo, gu, yg), (yg[0], [dn], [h[l + 1]])]
ws = torch.zeros(1 << 24, dtype=torch.uint8, device=dev)
arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
calls = []
for (x, ls, ys) in steps:
    calls.append((x, len(ls), arr([l[0] for l in ls]), arr([l[1] for l in ls]), arr([l[2] for l in ls]), arr(ys), (ctypes.c_int * len(ls))(*[l[0].shape[1] for l in ls]), x.shape[1], [l[0] for l in ls]))
keep = []
WGS = int(os.environ.get('PF_WGS', '128'))


def launch(c, st):
    (x, cnt, q, s_, z, y, Narr, k, _) = c
    rc = L.bie_mpq_forward_grouped(x.data_ptr(), cnt, q, s_, z, None, y, Narr, ws.data_ptr(), ws.numel(), 1, k, WBIT, GROUP, 0, B._hip.BF16, st)
    assert rc == 0, L.bie_last_error()


def build(prefetch, dist=1):
    main = torch.cuda.Stream()
    side = torch.cuda.Stream()
    def run():
        for i, c in enumerate(calls):
            if prefetch and i + dist < len(calls):
                ev = torch.cuda.Event()
                ev.record(main)            # the prefetch of launch i + dist starts when launch i is about to start
                side.wait_event(ev)
                ts = calls[i + dist][8]
                pa = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
                ba = (ctypes.c_size_t * len(ts))(*[t.numel() * 4 for t in ts])
                keep.append((pa, ba))
                rc = L.bie_prefetch(len(ts), pa, ba, WGS, side.cuda_stream)
                assert rc == 0, L.bie_last_error()
            launch(c, main.cuda_stream)
        if prefetch:
            main.wait_stream(side)
    with torch.cuda.stream(main):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main):
        run()
    return g


for name, pf, dist, wgs in (("no prefetch", False, 0, 0), ("prefetch next launch, 128 workgroups", True, 1, 128), ("next, 64", True, 1, 64), ("next, 256", True, 1, 256), ("next, 512", True, 1, 512),
                            ("two ahead, 128", True, 2, 128), ("two ahead, 256", True, 2, 256), ("no prefetch again", False, 0, 0)):
    WGS = wgs
    g = build(pf, dist)
    us = time_graph(g, 5) / n_layers
    print(json.dumps({"arm": name, "us_per_layer": round(us, 2)}), flush=True)
