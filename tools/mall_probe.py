#!/usr/bin/env python3
"""Does a lone decode launch run faster when its packed weights are already in the 256 MB Infinity Cache?  128 launches per graph replay, rotating
over 2 / 4 layers (cache-resident) against 64 / 24 layers (HBM every time).   usage: python tools/mall_probe.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bitorch_engine import _hip
L = _hip.lib()
dev = torch.device("cuda:0")
for (K, N) in ((4096, 4096), (4096, 11008), (11008, 4096)):
    gen = torch.Generator().manual_seed(0)
    nmax = 24 if K * N > 2e7 else 64
    Ls = []
    for _ in range(nmax):
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), dtype=torch.int64, generator=gen).to(torch.int32).to(dev)
        sc = (torch.rand((K // 128, N), generator=gen) * 0.01 + 0.005).bfloat16().to(dev)
        ze = (sc.float().cpu() * torch.rand((K // 128, N), generator=gen) * 15).bfloat16().to(dev)
        Ls.append((qw, sc, ze))
    x = torch.randn((1, K), generator=gen).bfloat16().to(dev)
    y = torch.empty((1, N), dtype=torch.bfloat16, device=dev)
    ws = torch.zeros(max(L.bie_mpq_workspace_bytes(1, K, N, 4), 16), dtype=torch.uint8, device=dev)
    for nl in (1, 2, 4, nmax):
        total = 120 // nl * nl
        def run(st):
            for i in range(total):
                qw, sc, ze = Ls[i % nl]
                rc = L.bie_mpq_forward(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), ze.data_ptr(), None, None, y.data_ptr(), ws.data_ptr(), ws.numel(), 1, K, N, 4, 128, 0, _hip.BF16, st)
                assert rc == 0
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            run(side.cuda_stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            run(torch.cuda.current_stream().cuda_stream)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        print(json.dumps({"K": K, "N": N, "layers_in_rotation": nl, "MB_in_rotation": round(nl * K * N / 2 / 1e6, 1), "us": round(e0.elapsed_time(e1) * 1e3 / (10 * total), 2)}), flush=True)
