import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from bitorch_engine.extensions import q_linear_cuda as ql
from oracle import oracle as orc
dev = "cuda:0"
for (w_bit, K, N) in ((2, 11008, 4096), (2, 11008, 1024), (2, 11008, 2048), (2, 8192, 4096), (2, 4096, 4096), (4, 11008, 4096), (2, 11008, 11008)):
        gs, dt = 128, orc.F16
        rng = np.random.default_rng(K)
        gen = torch.Generator().manual_seed(K)
        qw = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, (K * w_bit // 32, N), dtype=np.int64).astype(np.int32))
        sc = (torch.rand((K // gs, N), generator=gen) * 0.01 + 0.005).half()
        ze = (sc.float() * torch.rand((K // gs, N), generator=gen) * (2 ** w_bit - 1)).half()
        x = torch.randn((1, K), generator=gen).half()
        y = ql.mpq_forward_impl(x.to(dev), qw.to(dev), sc.to(dev), ze.to(dev), None, w_bit, 0, gs).float().cpu().numpy()
        r = orc.np_to_torch(orc.mpq_forward(orc.torch_to_np(x), qw.numpy(), orc.torch_to_np(sc), orc.torch_to_np(ze), None, w_bit, gs, 0, dt), torch.half).float().numpy()
        bad = np.abs(y - r) > 1e-2
        print(f"w{w_bit} K={K} N={N}: max err {np.abs(y - r).max():.4g} / {np.abs(r).max():.4g} bad cols {bad.sum()} first bad {np.argwhere(bad)[:3].tolist()} good {np.argwhere(~bad)[:20, 1].tolist()}", flush=True)
