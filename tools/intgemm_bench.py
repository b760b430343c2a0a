"""Times bie_q4_gemm / bie_q8_gemm (events around 20 launches)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bitorch-engine_amd"))
import torch
from bitorch_engine.extensions import q_linear_cutlass as qc

dev = "cuda:0"
for (M, N, K) in ((4096, 4096, 4096), (8192, 8192, 8192), (4096, 11008, 4096), (256, 4096, 4096)):
    a8 = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
    w8 = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
    x = torch.randn((M, K), dtype=torch.half, device=dev)
    w4 = qc.q4_w_pack(torch.randn((N, K), dtype=torch.half, device=dev), 0.3)
    a4 = qc.q4_w_pack(x, 0.3)
    for name, fn in (("q8", lambda: qc.q8_forward(a8, w8, False, 0.01, 0.01)),
                     ("q4gemm", lambda: qc._q4_gemm(a4, w4, M, N, K, 0.3, 0.3, torch.half)),
                     ("q4fwd", lambda: qc.q4_forward(x, w4, 0.3, 0.3, False, False))):
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1000 / 20
        print(f"{name:7s} M={M} N={N} K={K}: {us:9.1f} us  {2.0*M*N*K/us/1e6:8.1f} TOP/s", flush=True)
