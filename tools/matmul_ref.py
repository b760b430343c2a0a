"""Yardstick: plain bf16/fp16 library GEMM (torch.matmul -> hipBLASLt/rocBLAS) on the bench shapes."""
import torch
dev = "cuda:0"
for dt in (torch.bfloat16, torch.float16):
    for (M, K, N) in ((4096, 4096, 11008), (4096, 4096, 4096), (512, 4096, 11008), (64, 4096, 11008), (8192, 8192, 8192)):
        a = torch.randn((M, K), device=dev, dtype=dt)
        ws = [torch.randn((K, N), device=dev, dtype=dt) * 0.02 for _ in range(4)]
        for w in ws:
            torch.matmul(a, w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20):
            torch.matmul(a, ws[i % 4])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{dt} M={M} K={K} N={N}: {us:8.1f} us {2.0*M*K*N/us/1e6:8.1f} TFLOP/s", flush=True)
# int8 yardstick
for (M, K, N) in ((4096, 4096, 4096), (8192, 8192, 8192)):
    a = torch.randint(-128, 127, (M, K), device=dev, dtype=torch.int8)
    b = torch.randint(-128, 127, (K, N), device=dev, dtype=torch.int8)
    try:
        torch._int_mm(a, b); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(10):
            torch._int_mm(a, b)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 10
        print(f"int8 _int_mm M={M} K={K} N={N}: {us:8.1f} us {2.0*M*K*N/us/1e6:8.1f} TOP/s", flush=True)
    except Exception as ex:
        print("int_mm failed", ex)
