"""Reads the phase cycle sums written by the BIE_GEMM_LAB=7 build (bitorch-engine_amd/lab7/libbie_hip.so)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import torch
from bitorch_engine import _hip
L = _hip.lib()
dev = "cuda:0"
M, K, N, w_bit, gs = 4096, 4096, 11008, 4, 128
for dt, tdt in ((_hip.BF16, torch.bfloat16), (_hip.F16, torch.float16)):
    gen = torch.Generator().manual_seed(0)
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K * w_bit // 32, N), dtype=torch.int64, generator=gen).to(torch.int32).to(dev)
    sc = (torch.rand((K // gs, N), generator=gen) * 0.01 + 0.005).to(tdt).to(dev)
    ze = (sc.float().cpu() * torch.rand((K // gs, N), generator=gen) * 15).to(tdt).to(dev)
    x = torch.randn((M, K), generator=gen).to(tdt).to(dev)
    y = torch.zeros((M, N), dtype=tdt, device=dev)
    ws = torch.zeros(max(L.bie_mpq_workspace_bytes(M, K, N, w_bit), 4096), dtype=torch.uint8, device=dev)
    for it in range(3):
        rc = L.bie_mpq_forward(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), ze.data_ptr(), None, None, y.data_ptr(), ws.data_ptr(), ws.numel(), M, K, N, w_bit, gs, 0, dt, None)
        assert rc == 0, L.bie_last_error()
    torch.cuda.synchronize()
    d = y[M - 1].view(torch.int32)[:32].cpu().tolist()
    names = ["kk0", "kk1", "kk2", "store_a(+vmcnt)", "barrier", "kk3+prime"]
    for w in range(4):
        v = d[w * 8: w * 8 + 7]
        tiles = max(v[6], 1)
        print(("bf16" if dt == _hip.BF16 else "f16"), "wave", w, "tiles", tiles, " cycles per K-tile:", {n: round(c / tiles) for n, c in zip(names, v[:6])}, "total", round(sum(v[:6]) / tiles))
