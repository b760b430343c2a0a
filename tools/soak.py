"""Determinism soak: the same W4A16 inputs through bie_mpq_forward many times (ticketed split-K GEMV, split-K GEMM with LDS-DMA
staging, finalize kernel); every output must be bit-identical to the first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bitorch_engine import _hip
L = _hip.lib(); dev = "cuda:0"
total_bad = 0
for (M, K, N) in ((1, 4096, 11008), (2, 4096, 4096), (37, 4096, 11008), (300, 1088, 520), (4096, 4096, 4096)):
    for dt, tdt in ((_hip.BF16, torch.bfloat16), (_hip.F16, torch.float16)):
        g = torch.Generator().manual_seed(M)
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), dtype=torch.int64, generator=g).to(torch.int32).to(dev)
        sc = (torch.rand((K // 64, N), generator=g) * 0.01 + 0.005).to(tdt).to(dev)
        ze = (sc.float().cpu() * torch.rand((K // 64, N), generator=g) * 15).to(tdt).to(dev)
        x = torch.randn((M, K), generator=g).to(tdt).to(dev)
        ws = torch.zeros(max(L.bie_mpq_workspace_bytes(M, K, N, 4), 4096), dtype=torch.uint8, device=dev)
        ref, bad = None, 0
        for it in range(200 if M < 1000 else 40):
            y = torch.empty((M, N), dtype=tdt, device=dev)
            rc = L.bie_mpq_forward(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), ze.data_ptr(), None, None, y.data_ptr(), ws.data_ptr(), ws.numel(),
                                   M, K, N, 4, 64, 0, dt, None)
            assert rc == 0, L.bie_last_error()
            if ref is None:
                ref = y.clone()
            elif not torch.equal(y, ref):
                bad += 1
        total_bad += bad
        print("M", M, "K", K, "N", N, "dt", dt, "mismatching repeats:", bad, flush=True)

# ---- round 4: the list kernel's D16 form (big list: whole K per workgroup; small lists: K sliced over workgroups, granules), the same inputs
# replayed many times under a HIP graph, and the inline form of a >= 96 MB lone launch: every replay bit-identical to the first
sys.path.insert(0, ROOT)
import bench
B = bench.Bench(torch.device(dev))
gen = torch.Generator(device=dev).manual_seed(5)
for (K, N, nl, reps) in ((4096, 4096, 96, 200), (4096, 4096, 4, 300), (2048, 1024, 6, 300), (11008, 4096, 8, 200)):
    layers = [bench.make_layer(torch.device(dev), gen, K, N) for _ in range(nl)]
    y_all = torch.empty((nl, N), dtype=torch.bfloat16, device=dev)
    plan = B.make_list(layers, K, N, gen, ys=[y_all[i:i + 1] for i in range(nl)])
    g = bench.capture(lambda st: plan.forward(st))
    ref, bad = y_all.clone(), 0
    for it in range(reps):
        y_all.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        bad += 0 if torch.equal(y_all, ref) else 1
    total_bad += bad
    print("list", nl, "x", K, "x", N, "replays", reps, "mismatching:", bad, flush=True)
K, N = 8192, 28672
qw, sc, ze = bench.make_layer(torch.device(dev), gen, K, N)
x = torch.randn((1, K), generator=gen, device=dev).to(torch.bfloat16)
ws = B.workspace(1, K, N)
y = torch.empty((1, N), dtype=torch.bfloat16, device=dev)
ref, bad = None, 0
for it in range(100):
    y.fill_(float("nan"))
    B.forward(x, (qw, sc, ze), y, ws, 1, K, N, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    if ref is None:
        ref = y.clone()
    bad += 0 if torch.equal(y, ref) else 1
total_bad += bad
print("inline form 8192x28672 repeats 100 mismatching:", bad, flush=True)
sys.exit(1 if total_bad else 0)
