"""Act-order (random permuted g_idx) against trivial g_idx at 4096x11008, M = 1 and 4096: graph replay of
the Python-level call (sorted-copy + gather + fast kernel), per-call microseconds."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bitorch-engine_amd"))
import torch
from bitorch_engine.extensions import q_linear_cuda as q

dev = torch.device("cuda:0")
K, N, gs = 4096, 11008, 128
g = torch.Generator().manual_seed(1)
out = {}
for M in (1, 4096):
    layers = 8 if M == 1 else 2
    sets = []
    for i in range(layers):
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
        sc = (torch.rand((K // gs, N), generator=g) * 0.01 + 0.005).bfloat16().to(dev)
        zr = (sc.float() * 8).bfloat16()
        gi = (torch.arange(K, dtype=torch.int32) // gs)[torch.randperm(K, generator=g)].to(dev)
        sets.append((qw, sc, zr, gi))
    x = torch.randn((M, K), generator=g).bfloat16().to(dev)
    for label in ("trivial", "act_order"):
        def run():
            for (qw, sc, zr, gi) in sets:
                q.mpq_forward_impl(x, qw, sc, zr, gi if label == "act_order" else None, 4, 0, gs)
        run(); torch.cuda.synchronize()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            run()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                run()
        torch.cuda.synchronize()
        reps = 50 if M == 1 else 10
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            gr.replay()
        e1.record(); torch.cuda.synchronize()
        out[f"M{M}_{label}_us"] = round(e0.elapsed_time(e1) * 1000 / (reps * layers), 2)
    out[f"M{M}_ratio"] = round(out[f"M{M}_act_order_us"] / out[f"M{M}_trivial_us"], 3)
print(json.dumps(out))
