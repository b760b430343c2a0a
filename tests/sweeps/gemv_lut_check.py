"""GPU check of the table-lookup decode GEMV (mpq_gemv_lut.hip): parity vs the oracle on small shapes, then per-launch time
at the bench shapes (rotating distinct layers, HIP graph), for the LUT kernel and -- BIE_GEMV_LUT=0 in a second process --
the dot2 kernel."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from bitorch_engine import _hip
from bitorch_engine.extensions import q_linear_cuda as ql
from oracle import oracle as orc
sys.path.insert(0, os.path.join(ROOT, "tools"))

dev = torch.device("cuda:0")
BF = torch.bfloat16


def case(K, N, gs, asym, M, seed):
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    G = K // gs
    qw = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, (K // 8, N), dtype=np.int64).astype(np.int32))
    sc = (torch.rand((G, N), generator=gen) * 0.01 + 0.005).to(BF)
    if asym:
        ze = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, (G, N // 8), dtype=np.int64).astype(np.int32))
    else:
        ze = (sc.float() * torch.rand((G, N), generator=gen) * 15).to(BF)
    x = torch.randn((M, K), generator=gen).to(BF)
    return qw, sc, ze, x


def ref(x, qw, sc, ze, gs, asym):
    W = orc.mpq_dequant(qw.numpy(), orc.torch_to_np(sc), orc.torch_to_np(ze), None, 4, gs, asym, orc.BF16)
    return orc.np_to_torch(orc.gemm(orc.torch_to_np(x), W, orc.BF16), BF).float().numpy()


def check(y, r, what):
    y = y.float().cpu().numpy()
    tol = 1e-3 * np.abs(r).max() + 2.0 ** -7 * np.abs(r)
    bad = np.abs(y - r) > tol
    print(("OK  " if not bad.any() else "FAIL") + f" {what}: max err {np.abs(y - r).max():.4g} / max|ref| {np.abs(r).max():.4g}, bad {bad.sum()}", flush=True)
    return not bad.any()


ok = True
if os.environ.get("BIE_GEMV_LUT", "1") != "0" and "--no-parity" not in sys.argv:
    for (K, N, gs, asym, M) in ((512, 384, 128, 0, 1), (512, 384, 128, 1, 2), (1024, 200, 64, 0, 1), (768, 520, 256, 0, 2), (640, 64, 32, 1, 1),
                                (1408, 136, 128, 0, 1), (4096, 4096, 128, 0, 1)):
        qw, sc, ze, x = case(K, N, gs, asym, M, K + N + M)
        y = ql.mpq_forward_impl(x.to(dev), qw.to(dev), sc.to(dev), ze.to(dev), None, 4, asym, gs)
        ok &= check(y, ref(x, qw, sc, ze, gs, asym), f"K={K} N={N} g={gs} asym={asym} M={M}")
        y2 = ql.mpq_forward_impl(x.to(dev), qw.to(dev), sc.to(dev), ze.to(dev), None, 4, asym, gs)
        assert torch.equal(y, y2), "not deterministic / counters not reset"
    # grouped: three sets sharing x
    K, gs = 1024, 128
    sets, refs = [], []
    x = None
    for i, N in enumerate((384, 200, 72)):
        qw, sc, ze, xx = case(K, N, gs, 0, 2, 77 + i)
        x = xx if x is None else x
        bias = (torch.randn(N) * 0.1).to(BF) if i == 1 else None
        sets.append((qw.to(dev), sc.to(dev), ze.to(dev), None if bias is None else bias.to(dev)))
        r = ref(x, qw, sc, ze, gs, 0)
        if bias is not None:
            r = (torch.from_numpy(r).to(BF) + bias).float().numpy()
        refs.append(r)
    ys = ql.mpq_forward_grouped_impl(x.to(dev), sets, 4, 0, gs)
    for i, (y, r) in enumerate(zip(ys, refs)):
        ok &= check(y, r, f"grouped set {i}")
    print("PARITY", "OK" if ok else "FAILED", flush=True)

from sweep import time_case
for (M, K, N) in ((1, 4096, 11008), (1, 4096, 4096), (1, 11008, 4096), (2, 4096, 11008)):
    print(json.dumps(time_case(M, K, N, _hip.BF16)), flush=True)
