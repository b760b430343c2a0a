"""GPU micro-benchmarks used while tuning (not part of the product): per-launch time of bie_mpq_forward for a list of
(M, K, N, dtype) cases, each over a rotating set of distinct layers captured in a HIP graph."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bitorch_engine import _hip

L = _hip.lib()
dev = torch.device("cuda:0")


def time_case(M, K, N, dt, layers=32, reps=20, w_bit=4, gs=128, graph=True):
    gen = torch.Generator().manual_seed(0)
    tdt = torch.bfloat16 if dt == _hip.BF16 else torch.float16
    Ls = []
    nl = max(2, min(layers, int(600e6 // (K * N * w_bit // 8))))
    for _ in range(nl):
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K * w_bit // 32, N), dtype=torch.int64, generator=gen).to(torch.int32).to(dev)
        sc = (torch.rand((K // gs, N), generator=gen) * 0.01 + 0.005).to(tdt).to(dev)
        ze = (sc.float().cpu() * torch.rand((K // gs, N), generator=gen) * 15).to(tdt).to(dev)
        Ls.append((qw, sc, ze))
    x = torch.randn((M, K), generator=gen).to(tdt).to(dev)
    y = torch.empty((M, N), dtype=tdt, device=dev)
    ws = torch.zeros(max(L.bie_mpq_workspace_bytes(M, K, N, w_bit), 16), dtype=torch.uint8, device=dev)

    def run(st):
        for (qw, sc, ze) in Ls:
            rc = L.bie_mpq_forward(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), ze.data_ptr(), None, None, y.data_ptr(), ws.data_ptr(), ws.numel(), M, K, N, w_bit, gs, 0, dt, st)
            assert rc == 0, L.bie_last_error()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        run(side.cuda_stream)
    torch.cuda.synchronize()
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            run(torch.cuda.current_stream().cuda_stream)
        fn = g.replay
    else:
        fn = lambda: run(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * nl)
    G = K // gs
    byts = K * N * w_bit // 8 + 4 * G * N + 2 * M * K + 2 * M * N
    return {"M": M, "K": K, "N": N, "dt": "bf16" if dt == _hip.BF16 else "f16", "us": round(us, 2), "GB/s": round(byts / us / 1e3, 1),
            "TFLOP/s": round(2.0 * M * K * N / us / 1e6, 1), "layers": nl}


if __name__ == "__main__":  # noqa
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    out = []
    if which in ("all", "gemv"):
        for dt in (_hip.BF16, _hip.F16):
            for (K, N) in ((4096, 11008), (4096, 4096), (11008, 4096)):
                for M in (1, 4, 8):
                    out.append(time_case(M, K, N, dt))
                    print(json.dumps(out[-1]), flush=True)
    if which == "ab":
        for dt in (_hip.BF16, _hip.F16):
            out.append(time_case(1, 4096, 11008, dt)); print(json.dumps(out[-1]), flush=True)
            out.append(time_case(2, 4096, 11008, dt)); print(json.dumps(out[-1]), flush=True)
            out.append(time_case(4096, 4096, 11008, dt, layers=4, reps=5)); print(json.dumps(out[-1]), flush=True)
    if which == "midm":
        for (K, N) in ((4096, 11008), (4096, 4096), (11008, 4096)):
            for M in (4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096):
                out.append(time_case(M, K, N, _hip.BF16, layers=4, reps=5)); print(json.dumps(out[-1]), flush=True)
    if which in ("all", "gemm"):
        for dt in (_hip.BF16, _hip.F16):
            for (M, K, N) in ((4096, 4096, 11008), (4096, 4096, 4096), (512, 4096, 11008), (64, 4096, 11008), (16, 4096, 11008)):
                out.append(time_case(M, K, N, dt, layers=4, reps=5))
                print(json.dumps(out[-1]), flush=True)
