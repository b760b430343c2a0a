"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI (ctypes) and through
the layer API, against
  * the committed golden vectors produced by the reference itself (tests/golden/), and
  * the CPU oracle (oracle/bie_oracle.c, pinned to the same vectors by test_oracle_golden.py) on seeded inputs.
Bars: bit-exact for dequant / pack / binary / integer helpers; for fp16/bf16 GEMV/GEMM
max|y - y_ref| <= 1e-3 * max|y_ref| (north_star's 1e-3, norm-wise because outputs near zero are sums with
cancellation) plus one output ulp of the storage type (<= 2^-10 |y| fp16, <= 2^-7 |y| bf16) elementwise:
the fp32 sums differ in association order, so a result sitting on a rounding boundary may flip by one ulp.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MANIFEST = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))
DEV = "cuda:0"
TDT = {orc.F16: torch.float16, orc.BF16: torch.bfloat16, orc.F32: torch.float32}


def t16(a, dt):
    return orc.np_to_torch(np.ascontiguousarray(a), TDT[dt])


def to_f32(t):
    return t.detach().float().cpu().numpy()


def rel_err_report(y, ref, what):
    """Elementwise relative error distribution |y - ref| / |ref| over the elements with |ref| >= 2^-6 max|ref| (smaller references are
    sums with cancellation: their relative error is unbounded by construction).  Printed (pytest -s / the captured log), NOT a gate:
    north_star words the tolerance as "1e-3 rel", the gate is norm-wise (assert_close); this shows what the elementwise figure is."""
    y, ref = to_f32(y).ravel(), to_f32(ref).ravel()
    keep = np.abs(ref) >= np.abs(ref).max() * 2.0 ** -6
    r = np.abs(y[keep] - ref[keep]) / np.abs(ref[keep])
    pct = np.percentile(r, [50, 90, 99, 99.9, 100]) if r.size else np.zeros(5)
    line = (f"[rel-err] {what}: n={r.size} of {ref.size}; p50 {pct[0]:.2e} p90 {pct[1]:.2e} p99 {pct[2]:.2e} p99.9 {pct[3]:.2e} max {pct[4]:.2e}; "
            f"exact {float((y == ref).mean()):.4f}; norm-wise max|d|/max|ref| {np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30):.2e}")
    print(line)
    try:
        with open(os.path.join(os.path.dirname(GOLDEN), "..", "gpurun_out", "rel_err_report.txt"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
    return pct


def assert_close(y, ref, dt, what=""):
    y, ref = to_f32(y), to_f32(ref)
    ulp = 2.0 ** -7 if dt == orc.BF16 else (2.0 ** -10 if dt == orc.F16 else 2.0 ** -22)  # one ulp, worst case within a binade
    tol = 1e-3 * np.abs(ref).max() + ulp * np.abs(ref)
    bad = ~(np.abs(y - ref) <= tol)  # written so that a NaN / inf on either side is BAD (x > tol is False for NaN)
    bad &= ~((y == ref) & np.isinf(ref))  # the same infinity on both sides is agreement
    if bad.any():
        idx = np.argwhere(bad)
        rows, cols = np.unique(idx[:, 0]), np.unique(idx[:, 1])
        raise AssertionError(
            f"{what}: {bad.sum()} of {bad.size} outside tolerance; max err {np.abs(y - ref).max():.4g} vs max|ref| "
            f"{np.abs(ref).max():.4g}; first {idx[:6].tolist()}; bad rows {rows[:10].tolist()} (n={rows.size}) "
            f"bad cols {cols[:10].tolist()} (n={cols.size}); y={y[tuple(idx[0])]:.5g} ref={ref[tuple(idx[0])]:.5g}")


def rand_case(rng, K, N, w_bit, gs, dt, asym):
    G = (K + gs - 1) // gs
    qw = rng.integers(-2 ** 31, 2 ** 31 - 1, (K * w_bit // 32, N), dtype=np.int64).astype(np.int32)
    gen = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    scales = (torch.rand((G, N), generator=gen) * 0.01 + 0.005).to(TDT[dt])
    if asym:
        zeros = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, (G, N * w_bit // 32), dtype=np.int64).astype(np.int32))
    else:
        zeros = (scales.float() * torch.rand((G, N), generator=gen) * (2 ** w_bit - 1)).to(TDT[dt])
    return torch.from_numpy(qw), scales, zeros, gen


def oracle_forward(x, qw, scales, zeros, g_idx, w_bit, gs, asym, dt, bias=None):
    W = orc.mpq_dequant(qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), None if g_idx is None else g_idx.numpy(), w_bit, gs, asym, dt)
    y = orc.gemm(orc.torch_to_np(x), W, dt, None if bias is None else orc.torch_to_np(bias))
    return t16(y, dt) if dt != orc.F32 else torch.from_numpy(y)


def hip_forward(x, qw, scales, zeros, g_idx, w_bit, gs, asym, bias=None):
    from bitorch_engine.extensions import q_linear_cuda
    d = lambda t: None if t is None else t.to(DEV)
    return q_linear_cuda.mpq_forward_impl(d(x), d(qw), d(scales), d(zeros), d(g_idx), w_bit, asym, gs, d(bias))


# ------------------------------------------------------------------------------------------------ dequant / pack
@pytest.mark.parametrize("name", MANIFEST["mpq_dequant_pack"])
def test_dequant_and_pack_kernels_bit_exact_vs_reference(name):
    from bitorch_engine.extensions import q_linear_cuda
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    K, N, gs, w_bit, asym, has_gidx = [int(v) for v in d["meta"]]
    dt = orc.BF16 if "_bf16" in name else orc.F16
    qw = torch.from_numpy(d["qweight"]).to(DEV)
    scales = t16(d["scales"], dt).to(DEV)
    zeros = (torch.from_numpy(d["zeros"]) if asym else t16(d["zeros"], dt)).to(DEV)
    g_idx = torch.from_numpy(d["g_idx"]).to(DEV) if has_gidx else None
    W = q_linear_cuda.mpq_dequant(qw, scales, zeros, g_idx, w_bit, asym, gs)
    assert np.array_equal(orc.torch_to_np(W), d["W"]), "dequant kernel differs from reference unpack_qweight"
    packed = q_linear_cuda.mpq_pack(t16(d["Wp"], dt).to(DEV), scales, zeros, g_idx, w_bit, asym, gs)
    assert np.array_equal(packed.cpu().numpy(), d["packed"]), "pack kernel differs from reference pack_fp_weight"


def test_unpack_qweight_pack_fp_weight_api_roundtrip():
    from bitorch_engine.layers.qlinear.nbit import MPQWeightParameter
    from bitorch_engine.layers.qlinear.nbit.cuda.utils import unpack_qweight, pack_fp_weight
    rng = np.random.default_rng(5)
    for w_bit in (2, 4, 8):
        qw, scales, zeros, _ = rand_case(rng, 256, 128, w_bit, 64, orc.F16, False)
        p = MPQWeightParameter(qw.to(DEV), w_bit=w_bit, asym=False, group_size=64, layer_type=1)
        p.scales, p.zeros = scales.to(DEV), zeros.to(DEV)
        p.g_idx = (torch.arange(256, dtype=torch.int32) // 64).to(DEV)
        W = unpack_qweight(p)
        assert torch.equal(pack_fp_weight(W, p).cpu(), qw)


# ------------------------------------------------------------------------------------------------ forward: C ABI
CASES = []
for dt in (orc.F16, orc.BF16):
    for w_bit in (1, 2, 4, 8):
        for asym in (0, 1):
            CASES.append((dt, w_bit, asym))


@pytest.mark.parametrize("dt,w_bit,asym", CASES)
@pytest.mark.parametrize("M", [1, 2, 3, 8, 9, 17, 24, 32, 33, 70, 300])  # W4: 17 .. 32 = the lookup / matrix-pipe kernel with two row blocks
def test_mpq_forward_vs_oracle(dt, w_bit, asym, M):
    rng = np.random.default_rng(1000 * w_bit + 10 * M + asym + 7 * dt)
    K, N, gs = 512, 384, 128
    qw, scales, zeros, gen = rand_case(rng, K, N, w_bit, gs, dt, asym)
    x = torch.randn((M, K), generator=gen).to(TDT[dt])
    y = hip_forward(x, qw, scales, zeros, None, w_bit, gs, asym)
    ref = oracle_forward(x, qw, scales, zeros, None, w_bit, gs, asym, dt)
    assert_close(y, ref, dt, f"dt={dt} w{w_bit} asym={asym} M={M}")


@pytest.mark.parametrize("dt,K,N,M,asym", [(orc.F16, 4096, 4096, 17, 0), (orc.F16, 4096, 4096, 32, 1), (orc.BF16, 4096, 4096, 29, 0), (orc.F16, 8192, 4096, 24, 0),
                                            (orc.BF16, 2048, 8192, 20, 1), (orc.F16, 4096, 11008, 31, 0), (orc.BF16, 4096, 6144, 28, 0), (orc.F16, 11008, 4096, 20, 0)])
def test_lone_calls_of_17_to_32_rows_on_the_measured_shapes_take_the_decode_kernel(dt, K, N, M, asym):
    """Round 6: on the layer shapes where the matrix-pipe decode kernel with two row blocks measured ahead of the fused MFMA GEMM (csrc/mpq_lut_rb2_table.inc
    from profiles/r06_lone_rb2_sweep.txt) a lone call of 17 .. 32 rows is routed to it -- bie_mpq_rows_form says so -- and gives the oracle's numbers; one row more,
    another dtype's limit or a shape outside the table keeps the GEMM."""
    from bitorch_engine import _hip
    L = _hip.lib()
    code = _hip.F16 if dt == orc.F16 else _hip.BF16
    assert L.bie_mpq_rows_form(M, K, N, 4, 128, code) == 2
    assert L.bie_mpq_rows_form(33, K, N, 4, 128, code) != 2 and L.bie_mpq_rows_form(M, K, N + 64, 4, 128, code) != 2
    rng = np.random.default_rng(31 * M + K + N + asym)
    qw, scales, zeros, gen = rand_case(rng, K, N, 4, 128, dt, asym)
    bias = (torch.randn((N,), generator=gen) * 0.1).to(TDT[dt])
    x = torch.randn((M, K), generator=gen).to(TDT[dt])
    y = hip_forward(x, qw, scales, zeros, None, 4, 128, asym, bias)
    ref = oracle_forward(x, qw, scales, zeros, None, 4, 128, asym, dt, bias)
    assert_close(y, ref, dt, f"rows form 2: dt={dt} {K}x{N} M={M} asym={asym}")
    y2 = hip_forward(x, qw, scales, zeros, None, 4, 128, asym, bias)
    assert torch.equal(y, y2), "a second call differs"


@pytest.mark.parametrize("bm", [32, 64, 128, 256])
@pytest.mark.parametrize("S", [1, 2, 3, 5])
def test_mfma_gemm_every_tile_height_and_split(bm, S, monkeypatch):
    """The launcher picks (tile height, split-K factor) from a cost model; force every combination through the tuning knobs
    (read per call) so that no plan the model could ever choose is untested: odd tile counts per split, clamped rows,
    a ragged last column tile, both dtypes, sym and asym."""
    monkeypatch.setenv("BIE_GEMM_BM", str(bm))
    monkeypatch.setenv("BIE_GEMM_S", str(S))
    M, K, N, gs = 300, 1088, 520, 64  # 17 K tiles, 3 column tiles (the last one 8 columns wide)
    for dt in (orc.BF16, orc.F16):
        for asym in (0, 1):
            rng = np.random.default_rng(bm + 7 * S + dt + 3 * asym)
            qw, scales, zeros, gen = rand_case(rng, K, N, 4, gs, dt, asym)
            x = torch.randn((M, K), generator=gen).to(TDT[dt])
            y = hip_forward(x, qw, scales, zeros, None, 4, gs, asym)
            ref = oracle_forward(x, qw, scales, zeros, None, 4, gs, asym, dt)
            assert_close(y, ref, dt, f"BM={bm} S={S} dt={dt} asym={asym}")


@pytest.mark.parametrize("tile", ["128", "256"])
@pytest.mark.parametrize("dt,w_bit,asym", CASES)
def test_dense_form_of_the_gemm_every_flavour(dt, w_bit, asym, tile, monkeypatch):
    """mpq_dense.hip (dequantise once into MFMA fragment order + dense GEMM; opt-in with BIE_GEMM_DENSE=1, see the file header), forced
    onto small ragged shapes: every bit width, sym / asym, both dtypes, both tile sizes, clamped rows and column blocks, bias, K tiles
    1 / 2 / 34 (pipeline prologue and tail), group sizes below and above a stage."""
    monkeypatch.setenv("BIE_TUNING", "1")
    monkeypatch.setenv("BIE_GEMM_DENSE", "2")
    monkeypatch.setenv("BIE_GEMM_DENSE_TILE", tile)
    for (M, K, N, gs) in ((300, 1088, 544, 64), (33, 64, 288, 32), (130, 128, 96, 128)):
        rng = np.random.default_rng(M + K + N + 31 * w_bit + asym + dt)
        qw, scales, zeros, gen = rand_case(rng, K, N, w_bit, gs, dt, asym)
        x = torch.randn((M, K), generator=gen).to(TDT[dt])
        bias = torch.randn((N,), generator=gen).to(TDT[dt]) if M == 300 else None
        y = hip_forward(x, qw, scales, zeros, None, w_bit, gs, asym, bias)
        ref = oracle_forward(x, qw, scales, zeros, None, w_bit, gs, asym, dt, bias)
        assert_close(y, ref, dt, f"dense dt={dt} w{w_bit} asym={asym} tile={tile} M={M} K={K} N={N}")
        monkeypatch.setenv("BIE_GEMM_DENSE", "0")
        y0 = hip_forward(x, qw, scales, zeros, None, w_bit, gs, asym, bias)  # the fused kernel on the same operands: same weights, fp32 sums
        monkeypatch.setenv("BIE_GEMM_DENSE", "2")
        assert_close(y, y0, dt, "dense vs fused")


@pytest.mark.parametrize("gm", ["1", "3", "16"])
@pytest.mark.parametrize("tile", ["128", "256"])
def test_dense_form_tile_order(gm, tile, monkeypatch):
    """The workgroup -> tile mapping of the pipeline GEMMs (pipe_tile, mfma_pipe.cuh): every tile exactly once whatever the tile rows per
    XCD run (default 4) and the grid -- tile counts that are not multiples of 8, a last group of fewer rows, one tile row, one tile column."""
    monkeypatch.setenv("BIE_TUNING", "1")
    monkeypatch.setenv("BIE_GEMM_DENSE", "2")
    monkeypatch.setenv("BIE_GEMM_DENSE_TILE", tile)
    monkeypatch.setenv("BIE_GEMM_DENSE_GM", gm)
    for (M, K, N) in ((1300, 128, 1440), (200, 64, 3000), (1700, 64, 200)):
        rng = np.random.default_rng(M + K + N)
        qw, scales, zeros, gen = rand_case(rng, K, N, 4, 64, orc.BF16, 0)
        x = torch.randn((M, K), generator=gen).to(TDT[orc.BF16])
        y = hip_forward(x, qw, scales, zeros, None, 4, 64, 0)
        ref = oracle_forward(x, qw, scales, zeros, None, 4, 64, 0, orc.BF16)
        assert_close(y, ref, orc.BF16, f"dense tile order gm={gm} tile={tile} M={M} N={N}")


@pytest.mark.parametrize("K,N,gs,M", [(128, 64, 32, 1), (256, 260, 64, 4), (4096, 128, 128, 1), (1024, 2048, 1024, 2),
                                      (192, 132, 64, 16), (4096, 512, 32, 48), (2048, 1000, 128, 130), (64, 32, 64, 5)])
@pytest.mark.parametrize("dt", [orc.F16, orc.BF16])
def test_mpq_forward_ragged_shapes_w4(K, N, gs, M, dt):
    rng = np.random.default_rng(K + N + M)
    qw, scales, zeros, gen = rand_case(rng, K, N, 4, gs, dt, 0)
    x = torch.randn((M, K), generator=gen).to(TDT[dt])
    bias = torch.randn((N,), generator=gen).to(TDT[dt])
    y = hip_forward(x, qw, scales, zeros, None, 4, gs, 0, bias)
    ref = oracle_forward(x, qw, scales, zeros, None, 4, gs, 0, dt, bias)
    assert_close(y, ref, dt, f"K={K} N={N} gs={gs} M={M}")


@pytest.mark.parametrize("M", [1, 7, 40])
@pytest.mark.parametrize("dt", [orc.F16, orc.BF16, orc.F32])
def test_mpq_forward_act_order_gidx_and_fp32(M, dt):
    rng = np.random.default_rng(M + 3 * dt)
    K, N, gs, w_bit = 256, 192, 64, 4
    qw, scales, zeros, gen = rand_case(rng, K, N, w_bit, gs, orc.F16 if dt == orc.F32 else dt, 0)
    if dt == orc.F32:
        scales, zeros = scales.float(), zeros.float()
    g_idx = (torch.arange(K, dtype=torch.int32) // gs)[torch.randperm(K, generator=gen)]
    x = torch.randn((M, K), generator=gen).to(TDT[dt])
    y = hip_forward(x, qw, scales, zeros, g_idx, w_bit, gs, 0)
    ref = oracle_forward(x, qw, scales, zeros, g_idx, w_bit, gs, 0, dt)
    assert_close(y, ref, dt, f"act-order M={M} dt={dt}")


def test_mpq_forward_deterministic_and_linear():
    rng = np.random.default_rng(11)
    K, N, gs = 4096, 4096, 128
    qw, scales, zeros, gen = rand_case(rng, K, N, 4, gs, orc.F16, 0)
    x = torch.randn((1, K), generator=gen).half()
    a = hip_forward(x, qw, scales, zeros, None, 4, gs, 0)
    b = hip_forward(x, qw, scales, zeros, None, 4, gs, 0)
    assert torch.equal(a, b), "split-K reduction must be deterministic"
    # linearity in x for exactly representable scalings (size-independent property): f(2x) == 2 f(x) bit for bit
    c = hip_forward(x * 2, qw, scales, zeros, None, 4, gs, 0)
    assert torch.equal(c, a * 2)


# ------------------------------------------------------------------------------------------------ BASELINE.json sizes
@pytest.mark.parametrize("K,N", [(4096, 4096), (4096, 11008), (11008, 4096)])
@pytest.mark.parametrize("dt", [orc.F16, orc.BF16])
def test_full_size_decode_gemv(K, N, dt):
    rng = np.random.default_rng(K * 3 + N + dt)
    qw, scales, zeros, gen = rand_case(rng, K, N, 4, 128, dt, 0)
    for M in (1, 4):
        x = torch.randn((M, K), generator=gen).to(TDT[dt])
        y = hip_forward(x, qw, scales, zeros, None, 4, 128, 0)
        ref = t16(orc.mpq_forward(orc.torch_to_np(x), qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), None, 4, 128, 0, dt), dt)
        assert_close(y, ref, dt, f"full-size GEMV K={K} N={N} M={M}")
        rel_err_report(y, ref, f"decode GEMV {K}x{N} M={M} {'bf16' if dt == orc.BF16 else 'f16'}")
        if dt == orc.F16:  # north_star's 1e-3 element by element on the lone decode launches too
            assert_close_elementwise_f16(y, ref, f"full-size GEMV K={K} N={N} M={M} (lone launch, fp16)")


@pytest.mark.parametrize("dt", [orc.BF16, orc.F16])
def test_full_size_prefill_gemm_sampled_rows(dt):
    """M=4096, 4096x11008: every output column, a sample of rows checked against the oracle; all rows checked for
    agreement with the same rows computed in a separate small launch (row-independence property)."""
    rng = np.random.default_rng(99 + dt)
    K, N, M = 4096, 11008, 4096
    qw, scales, zeros, gen = rand_case(rng, K, N, 4, 128, dt, 0)
    x = torch.randn((M, K), generator=gen).to(TDT[dt])
    y = hip_forward(x, qw, scales, zeros, None, 4, 128, 0)
    rows = torch.tensor([0, 1, 31, 32, 255, 256, 1000, 2047, 2048, 3333, 4094, 4095])
    ref = t16(orc.mpq_forward(orc.torch_to_np(x[rows]), qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), None, 4, 128, 0, dt), dt)
    assert_close(y[rows.to(DEV)], ref, dt, "full-size GEMM sampled rows")
    if dt == orc.F16:
        assert_close_elementwise_f16(y[rows.to(DEV)], ref, "full-size GEMM M=4096 sampled rows (dense form, fp16)")
    # the same rows through a separate launch (different M tiling + split-K): same values up to fp32 summation order
    y2 = hip_forward(x[1024:1024 + 256], qw, scales, zeros, None, 4, 128, 0)
    assert_close(y2, y[1024:1024 + 256], dt, "rows must not depend on the M tile they land in")
    assert torch.isfinite(y.float()).all()


# ------------------------------------------------------------------------------------------------ layer API
@pytest.mark.parametrize("name", MANIFEST["mpq_layers"])
def test_mpq_layer_matches_reference_layer_outputs(name):
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQLinearCuda
    d = np.load(os.path.join(GOLDEN, f"layer_{name}.npz"))
    cfgs = {
        "gba_sym_w4_g128_dq2": dict(w_bit=4, dtype=torch.half, group_size=128, dq_group_size=32, dq_mode=2, use_gba_quant=True, asym=False),
        "gba_sym_w2_g32_dq1": dict(w_bit=2, dtype=torch.half, group_size=32, dq_group_size=1, dq_mode=1, use_gba_quant=True, asym=False),
        "gba_sym_w4_g128_bf16": dict(w_bit=4, dtype=torch.bfloat16, group_size=128, dq_group_size=32, dq_mode=2, use_gba_quant=True, asym=False),
        "gba_asym_w4_g64": dict(w_bit=4, dtype=torch.half, group_size=64, dq_group_size=32, dq_mode=2, use_gba_quant=True, asym=True),
        "gptq_w4_g64": dict(w_bit=4, dtype=torch.half, group_size=64, use_gba_quant=False, asym=True),
        "gptq_w8_g128": dict(w_bit=8, dtype=torch.half, group_size=128, use_gba_quant=False, asym=True),
        "gba_sym_w4_g256_nodq": dict(w_bit=4, dtype=torch.half, group_size=256, dq_group_size=32, dq_mode=2, use_gba_quant=True, asym=False),
    }
    kw = cfgs[name]
    dt = orc.BF16 if kw["dtype"] == torch.bfloat16 else orc.F16
    layer = MPQLinearCuda(256, 128, **kw)
    sd = {}
    for k, v in layer.state_dict().items():
        a = d["sd_" + k]
        sd[k] = t16(a, dt) if a.dtype == np.uint16 else torch.from_numpy(a)
    layer.load_state_dict(sd)
    layer.eval().to(DEV)
    layer.prepare_params()
    assert np.array_equal(orc.torch_to_np(layer.scales), d["prep_scales"])
    assert np.array_equal(orc.torch_to_np(layer.zeros), d["prep_zeros"])
    for M in (33, 64):
        y = layer(t16(d[f"x{M}"], dt).to(DEV))
        assert_close(y, t16(d[f"y{M}"], dt), dt, f"layer {name} M={M} vs reference CPU path")
    x3 = t16(d["x64"], dt).to(DEV).reshape(4, 16, 256)  # leading dims are flattened like the reference
    assert layer(x3).shape == (4, 16, 128)
    y1 = layer(t16(d["x33"], dt)[:1].to(DEV))  # decode path agrees with the prefill path on the same row
    assert_close(y1, t16(d["y33"], dt)[:1], dt, f"layer {name} M=1")


def test_layer_fails_loudly_without_gpu_tensors():
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQLinearCuda
    layer = MPQLinearCuda(64, 32, w_bit=4, dtype=torch.half, group_size=32, use_gba_quant=False)
    layer.prepare_params()
    with pytest.raises(RuntimeError):
        layer(torch.randn(1, 64).half())


# ------------------------------------------------------------------------------------------------ MBWQ
@pytest.mark.parametrize("bits", [2, 4])
@pytest.mark.parametrize("M", [1, 4, 40])
@pytest.mark.parametrize("perm", [False, True])
def test_mbwq_q4_dequant_and_forward(bits, M, perm):
    from bitorch_engine.extensions import q_linear_cuda
    rng = np.random.default_rng(bits * 10 + M + perm)
    K, N, gs = 512, 256, 64
    qw, scales, zeros, gen = rand_case(rng, K, N, bits, gs, orc.F16, 0)
    zeros = torch.randn(zeros.shape, generator=gen).half() * 0.05
    q_perm = (torch.randperm(K, generator=gen) if perm else torch.zeros(K)).to(torch.short)
    Wd = q_linear_cuda.mbwq_q42fp_weight(qw.to(DEV), scales.to(DEV), zeros.to(DEV), gs, bits, q_perm.to(DEV))
    Wo = orc.mbwq_q4_dequant(qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), q_perm.numpy() if perm else None, bits, gs)
    assert np.array_equal(orc.torch_to_np(Wd), Wo), "uniform MBWQ dequant not bit-exact"
    x = torch.randn((M, K), generator=gen).half()
    y = q_linear_cuda.mbwq_q4_forward(x.to(DEV), qw.to(DEV), scales.to(DEV), zeros.to(DEV), gs, q_perm.to(DEV), bits)
    ref = t16(orc.gemm(orc.torch_to_np(x), Wo, orc.F16), orc.F16)
    assert_close(y, ref, orc.F16, f"mbwq q{bits} M={M} perm={perm}")


@pytest.mark.parametrize("bits,M,K,N,gs,perm", [(4, 300, 608, 368, 32, True), (2, 16, 256, 200, 64, False), (4, 17, 960, 12, 64, True), (4, 33, 928, 152, 32, False),
                                                 (2, 1100, 768, 396, 128, True), (4, 5, 96, 7, 32, True), (2, 3, 160, 30, 32, False)])
def test_mbwq_uniform_forward_takes_every_shape_the_layout_allows(bits, M, K, N, gs, perm):
    """K that is not a multiple of 64, N that is not a multiple of 8 (4-bit) / 16 (2-bit) / 4: bie_mbwq_q4_forward used to REFUSE these
    (unsupported shape) although the reference's kernels bound-check K and N and take them (mbwq_linear_cuda_kernel.cu:740-830) -- found by
    tests/sweeps/fuzz_other_ops.py.  Now the one-column-per-lane kernel with the same per-weight rounding serves them; values against the oracle."""
    from bitorch_engine.extensions import q_linear_cuda
    rng = np.random.default_rng(bits * 1000 + M + K + N)
    qw, scales, zeros, gen = rand_case(rng, K, N, bits, gs, orc.F16, 0)
    zeros = torch.randn(zeros.shape, generator=gen).half() * 0.05
    q_perm = (torch.randperm(K, generator=gen) if perm else torch.zeros(K)).to(torch.short)
    Wo = orc.mbwq_q4_dequant(qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), q_perm.numpy() if perm else None, bits, gs)
    Wd = q_linear_cuda.mbwq_q42fp_weight(qw.to(DEV), scales.to(DEV), zeros.to(DEV), gs, bits, q_perm.to(DEV))
    assert np.array_equal(orc.torch_to_np(Wd), Wo)
    x = torch.randn((M, K), generator=gen).half()
    y = q_linear_cuda.mbwq_q4_forward(x.to(DEV), qw.to(DEV), scales.to(DEV), zeros.to(DEV), gs, q_perm.to(DEV), bits)
    assert_close(y, t16(orc.gemm(orc.torch_to_np(x), Wo, orc.F16), orc.F16), orc.F16, f"mbwq q{bits} M={M} K={K} N={N} perm={perm}")


@pytest.mark.parametrize("bits", [2, 4])
def test_mbwq_q4_forward_through_the_dense_form(bits, monkeypatch):
    """The uniform MBWQ layers reach the MFMA GEMM with the one-rounding fma (ZM_FUSED); forced through mpq_dense.hip (no q_perm: the
    gather path stays on the fused kernel) and compared with the oracle and with the fused kernel."""
    from bitorch_engine.extensions import q_linear_cuda
    monkeypatch.setenv("BIE_TUNING", "1")
    rng = np.random.default_rng(bits * 77)
    K, N, gs, M = 512, 256, 64, 200
    qw, scales, zeros, gen = rand_case(rng, K, N, bits, gs, orc.F16, 0)
    zeros = torch.randn(zeros.shape, generator=gen).half() * 0.05
    q_perm = torch.zeros(K).to(torch.short)
    Wo = orc.mbwq_q4_dequant(qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), None, bits, gs)
    x = torch.randn((M, K), generator=gen).half()
    ref = t16(orc.gemm(orc.torch_to_np(x), Wo, orc.F16), orc.F16)
    ys = {}
    for form in ("2", "0"):
        monkeypatch.setenv("BIE_GEMM_DENSE", form)
        ys[form] = q_linear_cuda.mbwq_q4_forward(x.to(DEV), qw.to(DEV), scales.to(DEV), zeros.to(DEV), gs, q_perm.to(DEV), bits)
        assert_close(ys[form], ref, orc.F16, f"mbwq q{bits} dense={form}")
    assert_close(ys["2"], ys["0"], orc.F16, "dense vs fused")


def exl2_load(qw_cpu, q_groups, K, groups):
    """The load-time step (q_linear_cuda.mbwq_trans_qweight: in-place re-arrangement + band table) on a DEVICE COPY; the raw
    checkpoint-order tensor stays with the caller for the oracle.  The table's first seven ints are the reference's."""
    from bitorch_engine.extensions import q_linear_cuda
    qd = qw_cpu.to(DEV).clone()
    same, rows = q_linear_cuda.mbwq_trans_qweight(qd, q_groups, True, K, groups, 4)
    ref = orc.exl2_rows(q_groups.numpy(), K)
    assert same is qd and len(rows) == q_linear_cuda.EXL2_ROWS_LEN
    assert rows[:6] == ref[:6] and (rows[6] & 0xff) == ref[6] and (rows[6] & 0x100)
    return qd, rows


@pytest.mark.parametrize("cfg", ["q_proj", "k_proj", "w3w2", "all6"])
@pytest.mark.parametrize("M", [1, 2, 3, 7, 11, 16, 17, 33, 48, 49, 64, 70, 130, 300])  # 3..48: matrix-pipe kernel (1-3 row blocks); from 49: fragment image + dense MFMA GEMM (no vendor GEMM)
def test_mbwq_exl2_dequant_and_forward(cfg, M, monkeypatch):
    from bitorch_engine.extensions import q_linear_cuda
    from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
    g = np.load(os.path.join(GOLDEN, "exl2_group_maps.npz"))
    K, groups, rows_packed = [int(v) for v in g[cfg + "_meta"]]
    q_groups = torch.from_numpy(g[cfg + "_q_groups"])
    N = 192
    rng = np.random.default_rng(K + M)
    gen = torch.Generator().manual_seed(K + M)
    qw = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, (rows_packed, N), dtype=np.int64).astype(np.int32))
    scales = (torch.rand((groups, N), generator=gen) * 0.02 + 0.001).half()
    zeros = (torch.randn((groups, N), generator=gen) * 0.1).half()
    q_perm = torch.randperm(K, generator=gen).to(torch.short)
    gmap = make_group_map(q_groups, rows_packed)
    assert np.array_equal(gmap.numpy(), g[cfg + "_group_map"]), "make_group_map differs from the reference"
    qs, rows = exl2_load(qw, q_groups, K, groups)
    Wd = q_linear_cuda.mbwq_exl2fp_weight(qs, scales.to(DEV), zeros.to(DEV), q_perm.to(DEV), gmap.to(DEV), rows)
    Wo = orc.exl2_dequant(qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), q_perm.numpy(), q_groups.numpy(), K)
    assert np.array_equal(orc.torch_to_np(Wd), Wo), f"exl2 dequant ({cfg}) not bit-exact"
    x = torch.randn((M, K), generator=gen).half()
    y = q_linear_cuda.mbwq_exl2_forward(x.to(DEV), qs, scales.to(DEV), zeros.to(DEV), q_perm.to(DEV), gmap.to(DEV), rows, False)
    ref = t16(orc.gemm(orc.torch_to_np(x), Wo, orc.F16), orc.F16)
    assert_close(y, ref, orc.F16, f"exl2 {cfg} M={M}")
    if M in (16, 64):
        # the C-ABI takes M <= 64 on the matrix-pipe kernel whatever the Python switch says (M = 64: four row blocks), and a NULL
        # q_perm (no act-order: x is read in place, no permute launch)
        from bitorch_engine import _hip
        monkeypatch.setenv("BIE_TUNING", "1")  # the switch is re-read per call
        monkeypatch.setenv("BIE_EXL2_DENSE_MIN_M", "65")
        y64 = q_linear_cuda.mbwq_exl2_forward(x.to(DEV), qs, scales.to(DEV), zeros.to(DEV), q_perm.to(DEV), gmap.to(DEV), rows, False)
        assert_close(y64, ref, orc.F16, f"exl2 {cfg} M={M} fused")
        L = _hip.lib()
        xd, qd, sd, zd, gd = x.to(DEV), qs, scales.to(DEV), zeros.to(DEV), gmap.to(DEV)
        yn = torch.empty((M, N), dtype=torch.float16, device=DEV)
        ws = _hip.workspace(L.bie_mbwq_workspace_bytes(M, K, N), DEV)
        keep, rp = q_linear_cuda._rows_arg(rows)
        rc = L.bie_mbwq_exl2_forward(_hip.ptr(xd), _hip.ptr(qd), _hip.ptr(sd), _hip.ptr(zd), None, _hip.ptr(gd), rp, _hip.ptr(yn), _hip.ptr(ws),
                                     ws.numel(), M, K, N, groups, _hip.stream())
        assert rc == 0
        Wn = orc.exl2_dequant(qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), None, q_groups.numpy(), K)
        assert_close(yn, t16(orc.gemm(orc.torch_to_np(x), Wn, orc.F16), orc.F16), orc.F16, f"exl2 {cfg} M={M} no q_perm")
        if M == 64:  # ... and the same NULL-q_perm call in the prefill form (x read in place by the dense kernel)
            monkeypatch.setenv("BIE_EXL2_DENSE_MIN_M", "49")
            yn.fill_(float("nan"))
            ws = _hip.workspace(L.bie_mbwq_workspace_bytes(M, K, N), DEV)
            rc = L.bie_mbwq_exl2_forward(_hip.ptr(xd), _hip.ptr(qd), _hip.ptr(sd), _hip.ptr(zd), None, _hip.ptr(gd), rp, _hip.ptr(yn), _hip.ptr(ws),
                                         ws.numel(), M, K, N, groups, _hip.stream())
            assert rc == 0
            assert_close(yn, t16(orc.gemm(orc.torch_to_np(x), Wn, orc.F16), orc.F16), orc.F16, f"exl2 {cfg} M={M} no q_perm, prefill form")
    if M >= 49:
        # the prefill form multiplies the reference's own weight values (exl2_dq per weight): its result is the fp32-accumulated product of
        # x with the bit-exact dequantised matrix -- the same statement as the oracle's, so the tolerance above is association order only.
        # use_cublas (the reference's switch to reconstruct + at::matmul) changes nothing here
        y2 = q_linear_cuda.mbwq_exl2_forward(x.to(DEV), qs, scales.to(DEV), zeros.to(DEV), q_perm.to(DEV), gmap.to(DEV), rows, True)
        assert torch.equal(y2, y)


@pytest.mark.parametrize("N,M,off", [(100, 800, False), (192, 900, True), (100, 70, False)])
def test_mbwq_exl2_many_rows_where_the_prefill_form_does_not_apply(N, M, off, monkeypatch):
    """M beyond the matrix-pipe kernel's 64 rows on a layer the fragment-image form cannot take (N % 8 != 0) or with that form switched off
    (BIE_EXL2_DENSE_MIN_M): the streaming kernel in slabs of eight rows.  The workspace function used to die on the host for M >= 769 here
    (SIGFPE: the one/two-row plan asked about 769 rows); values against the oracle."""
    from bitorch_engine.extensions import q_linear_cuda
    from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
    if off:
        monkeypatch.setenv("BIE_TUNING", "1")
        monkeypatch.setenv("BIE_EXL2_DENSE_MIN_M", "1000000")
    g = np.load(os.path.join(GOLDEN, "exl2_group_maps.npz"))
    cfg = "w3w2"
    K, groups, rows_packed = [int(v) for v in g[cfg + "_meta"]]
    q_groups = torch.from_numpy(g[cfg + "_q_groups"])
    rng = np.random.default_rng(K + M + N)
    gen = torch.Generator().manual_seed(K + M + N)
    qw = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, (rows_packed, N), dtype=np.int64).astype(np.int32))
    scales = (torch.rand((groups, N), generator=gen) * 0.02 + 0.001).half()
    zeros = (torch.randn((groups, N), generator=gen) * 0.1).half()
    q_perm = torch.randperm(K, generator=gen).to(torch.short)
    gmap = make_group_map(q_groups, rows_packed)
    qs, rows = exl2_load(qw, q_groups, K, groups)
    Wo = orc.exl2_dequant(qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), q_perm.numpy(), q_groups.numpy(), K)
    x = torch.randn((M, K), generator=gen).half()
    y = q_linear_cuda.mbwq_exl2_forward(x.to(DEV), qs, scales.to(DEV), zeros.to(DEV), q_perm.to(DEV), gmap.to(DEV), rows, False)
    assert_close(y, t16(orc.gemm(orc.torch_to_np(x), Wo, orc.F16), orc.F16), orc.F16, f"exl2 {cfg} N={N} M={M} streaming slabs")


def exl2_half_pair_words(q, bits):
    """The half-pair layout as DESIGN.md section 3 states it, in numpy: q [32, N] values of one chunk -> [bits, N] words.
    F = 16 // bits whole fields per 16-bit half; pair j = (q[2j], q[2j+1]) for j < bits * F sits at bit (j % F) * bits of the low /
    high half of word j // F; the remaining pairs fill, LSB first, the stream made of the R = 16 - F * bits spare top bits of the
    halves of words 0, 1, ..."""
    F = 16 // bits
    R = 16 - F * bits
    w = np.zeros((bits, q.shape[1]), dtype=np.uint64)
    for j in range(16):
        lo, hi = q[2 * j].astype(np.uint64), q[2 * j + 1].astype(np.uint64)
        if j < bits * F:
            p = (j % F) * bits
            w[j // F] |= (lo << np.uint64(p)) | (hi << np.uint64(16 + p))
        else:
            for t in range(bits):
                bit = (j - bits * F) * bits + t
                d, off = bit // R, F * bits + bit % R
                w[d] |= (((lo >> np.uint64(t)) & np.uint64(1)) << np.uint64(off)) | (((hi >> np.uint64(t)) & np.uint64(1)) << np.uint64(16 + off))
    return w.astype(np.uint32)


def test_exl2_load_time_layout_is_the_documented_one_and_the_kernels_refuse_an_unmarked_table():
    """bie_mbwq_exl2_shuffle (behind q_linear_cuda.mbwq_trans_qweight, the reference's shuffle hook mbwq_linear_cuda_kernel.cu:63-86):
    every chunk of every band (8/6/5/4/3/2 bit) re-arranged exactly as the layout is specified; a second pass on the same tensor
    and a plain bie_mbwq_rows table are refused."""
    import ctypes
    from bitorch_engine import _hip
    from bitorch_engine.extensions import q_linear_cuda
    g = np.load(os.path.join(GOLDEN, "exl2_group_maps.npz"))
    K, groups, rows_packed = [int(v) for v in g["all6_meta"]]
    q_groups = torch.from_numpy(g["all6_q_groups"])
    N = 200
    rng = np.random.default_rng(11)
    qw = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, (rows_packed, N), dtype=np.int64).astype(np.int32))
    qs, rows = exl2_load(qw, q_groups, K, groups)
    raw = qw.numpy().view(np.uint32).astype(np.uint64)
    want = np.zeros((rows_packed, N), dtype=np.uint32)
    row, kprev = 0, 0
    for b, bits in enumerate((8, 6, 5, 4, 3, 2)):
        for _ in range((rows[b] - kprev) // 32):
            stream = np.zeros(N, dtype=object)
            for i in range(bits):
                stream = stream + (raw[row + i].astype(object) << (32 * i))
            q = np.stack([np.array([(int(v) >> (bits * j)) & ((1 << bits) - 1) for v in stream], dtype=np.uint32) for j in range(32)])
            want[row:row + bits] = exl2_half_pair_words(q, bits)
            row += bits
        kprev = rows[b]
    assert row == rows_packed
    assert np.array_equal(qs.cpu().numpy().view(np.uint32), want), "the re-arranged tensor is not the documented half-pair layout"
    with pytest.raises(RuntimeError, match="already been re-arranged"):
        q_linear_cuda.mbwq_trans_qweight(qs, q_groups, True, K, groups, 4)
    # a table that did not come from the load-time step: refused by every entry point, nothing launched
    L = _hip.lib()
    plain = (ctypes.c_int * 20)()
    qg = q_groups.numpy().astype(np.int16)
    assert L.bie_mbwq_rows(qg.ctypes.data, groups, K, ctypes.cast(plain, ctypes.c_void_p)) == 0
    out = torch.empty((K, N), dtype=torch.float16, device=DEV)
    dev = lambda a: torch.from_numpy(a).to(DEV)
    sc, gm = torch.ones((groups, N), dtype=torch.float16, device=DEV), dev(g["all6_group_map"])
    rc = L.bie_mbwq_exl2_dequant(_hip.ptr(qs), _hip.ptr(sc), _hip.ptr(sc), None, _hip.ptr(gm), ctypes.cast(plain, ctypes.c_void_p), _hip.ptr(out), K, N, groups, None)
    assert rc == -1 and "SHUFFLED" in L.bie_last_error().decode()
    with pytest.raises(RuntimeError, match="20-int table"):
        q_linear_cuda.mbwq_exl2fp_weight(qs, sc, sc, torch.arange(K).short().to(DEV), gm, list(plain)[:7])


def test_exl2_state_dict_is_the_reference_format_and_a_reloaded_tensor_is_prepared_exactly_once(tmp_path):
    """prepare_params() re-arranges the mixed-bit qweight in place (the reference's shuffle hook is a no-op, so ITS
    saved tensor is the checkpoint's stream).  (1) state_dict() of a prepared layer holds the ORIGINAL words bit for bit
    (bie_mbwq_exl2_unshuffle is the exact inverse, every band width); (2) saved -> loaded into a fresh layer -> prepare_params() gives the
    same outputs; (3) loading into the PREPARED layer makes forward() fail loudly until prepare_params() ran again -- never a second
    shuffle, never unshuffled words under a shuffled table; (4) .cpu()/.to() keep a prepared layer usable."""
    g = torch.Generator().manual_seed(77)
    K, N = 512, 256
    spec = [(8, 32), (6, 64), (5, 32), (4, 128), (3, 128), (2, 128)]
    layer, qw, qg = _exl2_layer(K, N, spec, g)
    layer.eval().to(DEV)
    layer.prepare_params()
    x = torch.randn((3, K), generator=g).half().to(DEV)
    y0 = layer(x).clone()
    assert not torch.equal(layer.qweight.data.cpu(), qw)            # the private layout really differs from the stream
    sd = layer.state_dict()
    assert torch.equal(sd["qweight"].cpu(), qw), "state_dict() does not hold the checkpoint's streams"
    assert torch.equal(layer(x), y0)                                 # ... and the layer's own tensor was not touched
    path = str(tmp_path / "exl2.pt")
    torch.save(sd, path)
    fresh, _, _ = _exl2_layer(K, N, spec, torch.Generator().manual_seed(1))
    fresh.load_state_dict(torch.load(path))
    fresh.eval().to(DEV)
    with pytest.raises(RuntimeError, match="prepare_params"):
        fresh(x)                                                     # loaded, not prepared: refuses
    fresh.prepare_params()
    assert torch.equal(fresh(x), y0)
    # (3) into the prepared layer, through a parent module (the nn.Module machinery) and through the layer's own tolerant loader
    parent = torch.nn.Sequential(layer)
    parent.load_state_dict({"0." + k: v for k, v in sd.items()})
    with pytest.raises(RuntimeError, match="prepare_params"):
        layer(x)
    with pytest.raises(RuntimeError):
        layer.exl2fp_weight(layer.qweight.data, layer.scales, layer.zeros, layer.q_perm, layer.q_group_map, layer.qweight.rows)
    layer.prepare_params()
    assert torch.equal(layer(x), y0)
    with pytest.raises(RuntimeError, match="already been re-arranged"):
        layer.prepare_params()                                       # a second pass over the same contents is refused, as before
    layer.load_state_dict(sd)
    with pytest.raises(RuntimeError, match="prepare_params"):
        layer(x)
    layer.prepare_params()
    assert torch.equal(layer(x), y0)
    # (4) device moves carry the prepared state; a prepared layer moved to the CPU still saves the checkpoint's format
    layer.cpu()
    sd_cpu = layer.state_dict()
    assert sd_cpu["qweight"].device.type == "cpu" and torch.equal(sd_cpu["qweight"], qw)
    layer.to(DEV)
    assert torch.equal(layer(x), y0)
    assert torch.equal(layer.state_dict()["qweight"].cpu(), qw)


def test_exl2_second_checkpoint_with_another_bit_allocation_gets_its_own_group_map(tmp_path):
    """a checkpoint with the SAME tensor shapes but different q_groups, loaded into an already prepared layer.  The
    second prepare_params() must rebuild the group map (it used to keep the first checkpoint's: scales / zeros of the wrong groups, silently);
    a REJECTED load (wrong shape through the nn.Module machinery) must leave the prepared layer working."""
    K, N = 512, 192
    spec_a = [(4, 128), (3, 128), (3, 128), (2, 128)]   # 16 + 12 + 12 + 8 = 48 packed rows, 4 groups
    spec_b = [(4, 64), (4, 192), (2, 128), (2, 128)]    #  8 + 24 +  8 + 8 = 48 packed rows, 4 groups: same shapes, other groups
    a, qw_a, qg_a = _exl2_layer(K, N, spec_a, torch.Generator().manual_seed(5))
    b, qw_b, qg_b = _exl2_layer(K, N, spec_b, torch.Generator().manual_seed(6))
    assert a.qweight.shape == b.qweight.shape and a.scales.shape == b.scales.shape and qg_a != qg_b
    keep = ("qweight", "scales", "zeros", "q_perm", "q_groups", "channel_scale")   # what a PREPARED layer still owns (prepare_params drops the load-only buffers)
    sd_b = {k: v.clone() for k, v in b.state_dict().items() if k in keep}            # b is unprepared: its state_dict IS the checkpoint
    b.eval().to(DEV)
    b.prepare_params()
    x = torch.randn((2, K), generator=torch.Generator().manual_seed(7)).half().to(DEV)
    want_b = b(x).clone()
    ref = orc.np_to_torch(orc.gemm(orc.torch_to_np(x.cpu()), orc.exl2_dequant(qw_b.numpy(), orc.torch_to_np(b.scales.cpu()), orc.torch_to_np(b.zeros.cpu()),
                                                                               b.q_perm.cpu().numpy(), np.array(qg_b, np.int16), K), orc.F16), torch.half)
    assert_close(want_b.cpu(), ref, orc.F16, "exl2 layer b against the oracle")
    a.eval().to(DEV)
    a.prepare_params()
    map_a = a.q_group_map.clone()
    y_a = a(x).clone()
    # a load the base class rejects: the layer keeps working (it used to refuse forward() AND prepare_params())
    bad = {"0." + k: v for k, v in sd_b.items()}
    bad["0.qweight"] = torch.zeros((47, N), dtype=torch.int32)
    with pytest.raises(RuntimeError, match="size mismatch"):
        torch.nn.Sequential(a).load_state_dict(bad)
    # (the other keys of `bad` were copied: put a's own back before comparing)
    a.load_state_dict({k: v for k, v in _exl2_layer(K, N, spec_a, torch.Generator().manual_seed(5))[0].state_dict().items() if k in keep and k != "qweight"}, strict=False)
    assert a._exl2_current() and torch.equal(a(x), y_a)
    # the second checkpoint, through the nn.Module machinery
    torch.nn.Sequential(a).load_state_dict({"0." + k: v for k, v in sd_b.items()})
    assert a.q_group_map is None and a.qweight.q_group_map is None
    with pytest.raises(RuntimeError, match="prepare_params"):
        a(x)
    a.prepare_params()
    assert not torch.equal(a.q_group_map.cpu(), map_a.cpu())
    assert torch.equal(a.q_group_map.cpu(), b.q_group_map.cpu())
    assert torch.equal(a(x), want_b)


@pytest.mark.parametrize("name", ["g64_g128_ragged", "one_group_per_band", "mixed_sizes", "g16", "g96"])
def test_exl2_group_structures_direct_and_staged_decode(name):
    """The decode kernel has two forms: DIRECT (regular groups -- every band's groups hold the same power-of-two number of whole
    chunks, a shorter last group allowed: group index by shift on the band table, nothing staged) and the staged form (anything
    else, and M = 2).  Both against the oracle on group structures of either kind; the REGULAR mark of the table says which ran."""
    from bitorch_engine.extensions import q_linear_cuda
    from bitorch_engine.layers.qlinear.nbit.cuda import MBWQExl2ForwardList
    from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
    structures = {  # (bits, k per group) in band order 8 > 6 > 5 > 4 > 3 > 2
        "g64_g128_ragged": ([(4, 64)] * 3 + [(4, 32)] + [(3, 128)] * 2 + [(3, 96)] + [(2, 64)] * 4, True),
        "one_group_per_band": ([(6, 96), (4, 160), (2, 224)], True),
        "mixed_sizes": ([(4, 32), (4, 64), (4, 32)] + [(2, 64)] * 4, False),
        "g16": ([(4, 16)] * 8 + [(2, 16)] * 12, False),
        "g96": ([(4, 96)] * 3 + [(2, 32)] * 4, False),
    }
    spec, regular = structures[name]
    qg, row, K = [], 0, 0
    for bits, k in spec:
        qg += [bits, row]
        row += k * bits // 32
        K += k
    assert K % 32 == 0
    groups = len(spec)
    q_groups = torch.tensor(qg, dtype=torch.short)
    N = 328
    rng = np.random.default_rng(K + groups)
    gen = torch.Generator().manual_seed(K + groups)
    qw = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, (row, N), dtype=np.int64).astype(np.int32))
    scales = (torch.rand((groups, N), generator=gen) * 0.02 + 0.001).half()
    zeros = (torch.randn((groups, N), generator=gen) * 0.1).half()
    q_perm = torch.randperm(K, generator=gen).to(torch.short)
    gmap = make_group_map(q_groups, row)
    qs, rows = exl2_load(qw, q_groups, K, groups)
    assert bool(rows[6] & 0x200) == regular, f"{name}: REGULAR mark {rows[6]:#x}"
    d = lambda t: t.to(DEV)
    Wd = q_linear_cuda.mbwq_exl2fp_weight(qs, d(scales), d(zeros), d(q_perm), d(gmap), rows)
    Wo = orc.exl2_dequant(qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), q_perm.numpy(), q_groups.numpy(), K)
    assert np.array_equal(orc.torch_to_np(Wd), Wo), f"exl2 dequant ({name}) not bit-exact"
    for M in (1, 2, 5):
        x = torch.randn((M, K), generator=gen).half()
        ref = t16(orc.gemm(orc.torch_to_np(x), Wo, orc.F16), orc.F16)
        y = q_linear_cuda.mbwq_exl2_forward(d(x), qs, d(scales), d(zeros), d(q_perm), d(gmap), rows, False)
        assert_close(y, ref, orc.F16, f"exl2 {name} M={M}")
        if M <= 2:
            ent = [{"x": d(x), "qweight": qs, "scales": d(scales), "zeros": d(zeros), "q_perm": d(q_perm), "q_group_map": d(gmap), "rows": rows,
                    "y": torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)}]
            MBWQExl2ForwardList(ent)()
            assert_close(ent[0]["y"], ref, orc.F16, f"exl2 {name} M={M} list")
    # no q_perm: the direct form keeps its index load and ignores the value
    x = torch.randn((1, K), generator=gen).half()
    Wn = orc.exl2_dequant(qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), None, q_groups.numpy(), K)
    ent = [{"x": d(x), "qweight": qs, "scales": d(scales), "zeros": d(zeros), "q_perm": None, "q_group_map": d(gmap), "rows": rows,
            "y": torch.full((1, N), float("nan"), dtype=torch.float16, device=DEV)}]
    MBWQExl2ForwardList(ent)()
    assert_close(ent[0]["y"], t16(orc.gemm(orc.torch_to_np(x), Wn, orc.F16), orc.F16), orc.F16, f"exl2 {name} list, no q_perm")


def test_mbwq_layer_llama_shapes_w3w2_decode():
    """BASELINE configs[2]: mixed 3/2-bit, Llama-7B shape 4096x4096, M=1 through the layer API."""
    from bitorch_engine.layers.qlinear.nbit.cuda import MBWQLinearCuda
    K, N = 4096, 4096
    bits, gsz = [3, 2], 32
    # q_groups laid out like the reference's test helper: half the rows 3-bit, half 2-bit, groups of 32 rows
    qg, row = [], 0
    for b in bits:
        for _ in range(K // 2 // gsz):
            qg += [b, row]
            row += gsz * b // 32
    groups = len(qg) // 2
    gen = torch.Generator().manual_seed(3)
    layer = MBWQLinearCuda(in_channels=K, out_channels=N, w_bit=4, dtype=torch.half, group_size=32, dq_group_size=1,
                           use_gba_quant=True, asym=False, dq_mode=2, use_mbw=True, groups=groups, rows_packed=row)
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (row, N), generator=gen, dtype=torch.int64).to(torch.int32)
    layer.set_qweight_data(qw)
    scales = (torch.rand((groups, N), generator=gen) * 0.02 + 0.001).half()
    zeros = (torch.randn((groups, N), generator=gen) * 0.05).half()
    layer.set_scales(scales)
    layer.set_zeros(zeros)
    layer.q_perm = torch.arange(K).to(torch.short)
    layer.q_groups = torch.tensor(qg, dtype=torch.short)
    layer.eval().to(DEV)
    layer.prepare_params()
    x = torch.randn((1, K), generator=gen).half()
    y = layer(x.to(DEV)).reshape(-1, N)  # channel_scale is [1, 1, K]: the reference broadcasts to 3-D too
    Wo = orc.exl2_dequant(qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), None, np.array(qg, np.int16), K)
    ref = t16(orc.gemm(orc.torch_to_np(x), Wo, orc.F16), orc.F16)
    assert_close(y, ref, orc.F16, "MBWQ exl2 layer 4096x4096 w3/w2")
    y40 = layer(torch.randn((40, K), generator=gen).half().to(DEV))
    assert y40.reshape(-1, N).shape == (40, N) and torch.isfinite(y40.float()).all()


# ------------------------------------------------------------------------------------------------ binary
def test_binary_linear_vs_reference_cpp_golden():
    from bitorch_engine.extensions import binary_linear_cpp, binary_linear_cuda, binary_linear_cutlass
    d = np.load(os.path.join(GOLDEN, "binary_linear_cpp.npz"))
    for tag in ("M1N64K128", "M4N96K256", "M33N40K64"):
        x, w, yref, wp = (torch.from_numpy(d[tag + s]) for s in ("_x", "_w", "_y", "_wpacked"))
        M, K = x.shape
        N = w.shape[0]
        mine = binary_linear_cpp.w_pack(w.to(DEV), N, K)
        assert torch.equal(mine.cpu(), wp), "w_pack layout differs from the reference"
        assert torch.equal(binary_linear_cpp.forward(x.to(DEV), mine, M, N, K).cpu(), yref)
        assert torch.equal(binary_linear_cpp.forward(x.to(DEV), w.to(DEV), M, N, K).cpu(), yref)
        assert torch.equal(binary_linear_cuda.forward(x.to(DEV), w.to(DEV), 3, True).cpu(), yref)
        assert torch.equal(binary_linear_cutlass.forward(x.to(DEV), binary_linear_cutlass.w_pack(w.to(DEV), False), 0.5, False, 3).cpu(), yref * 0.5)


@pytest.mark.parametrize("M", [1, 3, 64, 4096])
def test_binary_linear_4096_vs_oracle(M):
    from bitorch_engine.extensions import binary_linear_cuda
    gen = torch.Generator().manual_seed(M)
    K = N = 4096
    x = torch.randn((M, K), generator=gen)
    w = torch.randn((N, K), generator=gen)
    y = binary_linear_cuda.forward(x.to(DEV), w.to(DEV), 3, True).cpu().numpy()
    rows = np.unique(np.linspace(0, M - 1, min(M, 16)).astype(int))
    ref = orc.binary_linear_rowpacked(orc.binary_pack_rows(x.numpy()[rows]), orc.binary_pack_rows(w.numpy()), K)
    assert np.array_equal(y[rows], ref)
    # checksum property over the full output: sum_n y[m][n] = sum_k sx[m][k] * (sum_n sw[n][k])  (exact in int64)
    sx = np.where(x.numpy() >= 0, 1, -1).astype(np.int64)
    colsum = np.where(w.numpy() >= 0, 1, -1).astype(np.int64).sum(axis=0)
    assert np.array_equal(y.astype(np.int64).sum(axis=1), sx @ colsum)


def test_binary_linear_large_ragged_shapes_take_the_128_tile_kernel():
    """xnor_gemm128_kernel (>= 256 tiles of 128 x 128, K % 128 == 0): M, N not multiples of 128, K of 3 chunks -- exact against the
    oracle on sampled rows and through the integer checksum over the whole output."""
    from bitorch_engine.extensions import binary_linear_cuda
    gen = torch.Generator().manual_seed(77)
    M, N, K = 2100, 2000, 384
    x = torch.randn((M, K), generator=gen)
    w = torch.randn((N, K), generator=gen)
    y = binary_linear_cuda.forward(x.to(DEV), w.to(DEV), 3, True).cpu().numpy()
    rows = np.array([0, 1, 127, 128, 1000, 2047, 2048, 2099])
    ref = orc.binary_linear_rowpacked(orc.binary_pack_rows(x.numpy()[rows]), orc.binary_pack_rows(w.numpy()), K)
    assert np.array_equal(y[rows], ref)
    sx = np.where(x.numpy() >= 0, 1, -1).astype(np.int64)
    colsum = np.where(w.numpy() >= 0, 1, -1).astype(np.int64).sum(axis=0)
    assert np.array_equal(y.astype(np.int64).sum(axis=1), sx @ colsum)


def test_binary_layers_api():
    from bitorch_engine.layers.qlinear.binary.cpp import BinaryLinearCPP
    from bitorch_engine.layers.qlinear.binary.cuda import BinaryLinearCuda
    from bitorch_engine.layers.qlinear.binary.cutlass import BinaryLinearCutlass
    gen = torch.Generator().manual_seed(0)
    K, N, M = 256, 64, 5
    w = torch.randn((N, K), generator=gen)
    x = torch.randn((M, K), generator=gen)
    sign = lambda t: torch.where(t >= 0, 1.0, -1.0)
    cpp = BinaryLinearCPP(K, N)
    cpp.set_weight_data(w)
    cpp.eval().to(DEV)
    assert torch.equal(cpp(x.to(DEV)).cpu(), sign(x) @ sign(w).t())
    for cls in (BinaryLinearCuda, BinaryLinearCutlass):
        layer = cls(K, N)
        layer.set_weight_data(w.clone())
        layer.eval().to(DEV)
        y = layer(x.to(DEV)).cpu()
        # exact expectation: the integers come from the oracle's XNOR-popcount on the layer's own int8 sign carriers, the
        # scaling is the layer's fp32 expression evaluated in the same order
        ints = orc.binary_linear_rowpacked(orc.binary_pack_rows(x.numpy()), orc.binary_pack_rows(layer.weight.data.float().cpu().numpy()), K)
        ints = torch.from_numpy(ints.astype(np.float32))
        sa, sw = layer.scale_a.detach().cpu(), layer.scale_w.detach().cpu()
        assert torch.allclose(sa, (2 * x.abs().mean()).to(sa.dtype), rtol=1e-5)  # GPU reduction order
        if cls is BinaryLinearCuda:
            expect = ints * sa * sw
        else:
            expect = ints * torch.tensor(sa.item() * sw.item(), dtype=torch.float32)
        assert torch.equal(y, expect), f"{cls.__name__}: layer output differs from oracle integers x scales"
        wc = w - w.mean()
        assert torch.equal(layer.weight.data.cpu() >= 0, wc >= 0)


def test_binary_conv_vs_reference_cpp_golden_and_resnet_shape():
    from bitorch_engine.extensions import binary_conv_cpp
    from bitorch_engine.layers.qconv.binary.cpp import BinaryConv2dCPP
    d = np.load(os.path.join(GOLDEN, "binary_conv_cpp.npz"))
    for tag in sorted({k.rsplit("_", 1)[0] for k in d.files}):
        st = int(tag.split("s")[1].split("p")[0])
        pad = int(tag.split("p")[1].split("d")[0])
        dil = int(tag.split("d")[1])
        x, w, yref = (torch.from_numpy(d[tag + s]) for s in ("_x", "_w", "_y"))
        OC, C, ks, _ = w.shape
        oe = yref.shape[-1]
        y = binary_conv_cpp.forward(x.to(DEV), w.reshape(OC, -1).to(DEV), OC, oe * oe, C * ks * ks, ks, st, pad, dil, oe)
        assert torch.equal(y.cpu(), yref), tag
    # BASELINE configs[3]: ResNet-18 3x3x512 on 7x7, through the layer
    gen = torch.Generator().manual_seed(1)
    layer = BinaryConv2dCPP(512, 512, 3, stride=1, padding=1)
    w = torch.randn((512, 512, 3, 3), generator=gen)
    layer.set_weight_data(w)
    layer.eval().to(DEV)
    x = torch.randn((4, 512, 7, 7), generator=gen)
    y = layer(x.to(DEV)).cpu().numpy()
    assert np.array_equal(y, orc.binary_conv2d(x.numpy(), w.numpy(), 1, 1, 1))


@pytest.mark.parametrize("B,C,H,W,OC,ks,st,pad,dil", [(2, 64, 14, 14, 64, 3, 1, 1, 1), (1, 24, 9, 11, 36, 3, 2, 1, 1), (3, 8, 7, 7, 4, 1, 1, 0, 1),
                                                     (2, 40, 12, 10, 70, 3, 1, 2, 2), (1, 128, 5, 17, 130, 5, 1, 2, 1), (2, 32, 8, 8, 64, 2, 2, 0, 1),
                                                     (1, 64, 6, 300, 8, 7, 2, 3, 1),
                                                     # small batches of the deep ResNet layers (the one-wave DMA-staged tap kernel)
                                                     (1, 512, 7, 7, 512, 3, 1, 1, 1), (4, 512, 7, 7, 512, 3, 1, 1, 1), (2, 256, 14, 14, 256, 3, 1, 1, 1),
                                                     (3, 128, 9, 11, 64, 3, 2, 1, 1), (1, 128, 28, 28, 128, 3, 1, 1, 1), (2, 256, 7, 7, 128, 1, 1, 0, 1)])
def test_binary_conv_tap_form_equals_the_im2col_form_and_the_oracle(B, C, H, W, OC, ks, st, pad, dil):
    """The implicit conv (tap-major weights x channel-minor activation bits, no im2col image) against the round-1 bit-im2col path
    through the C ABI and against the oracle: strides, dilation, channel counts that are not multiples of 32, output widths that
    are not multiples of 8, output channel counts that are not multiples of 64; the last geometry (7 rows of 300 pixels) does not
    fit the LDS slab and must take the im2col form."""
    from bitorch_engine import _hip
    from bitorch_engine.extensions._binary_common import pack_rows, conv2d
    gen = torch.Generator().manual_seed(B + C + H + W + OC)
    x = torch.randn((B, C, H, W), generator=gen)
    w = torch.randn((OC, C, ks, ks), generator=gen)
    wp = pack_rows(w.reshape(OC, -1).to(DEV)).contiguous()
    y = conv2d(x.to(DEV), wp, OC, ks, st, pad, dil, 1.0)
    L = _hip.lib()
    assert bool(L.bie_binary_conv2d_taps_ok(C, W, ks)) == (W != 300)
    OH, OW = y.shape[2], y.shape[3]
    y2 = torch.empty_like(y)
    xd = x.to(DEV)
    ws = torch.zeros(L.bie_binary_conv2d_workspace_bytes(B, C, H, W, OC, ks, st, pad, dil), dtype=torch.uint8, device=DEV)
    rc = L.bie_binary_conv2d_forward(xd.data_ptr(), wp.data_ptr(), y2.data_ptr(), ws.data_ptr(), ws.numel(), B, C, H, W, OC, ks, st, pad, dil,
                                     1.0, _hip.F32, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert torch.equal(y, y2)
    assert np.array_equal(y.cpu().numpy(), orc.binary_conv2d(x.numpy(), w.numpy(), st, pad, dil))
    for tdt in (torch.float16, torch.bfloat16):  # the sign of a value does not depend on the carrier dtype
        assert torch.equal(conv2d(x.to(tdt).to(DEV), wp, OC, ks, st, pad, dil, 0.5), y * 0.5)


@pytest.mark.parametrize("B,C,H,W,OC,ks,st,pad", [(1, 512, 7, 7, 512, 3, 1, 1), (32, 512, 7, 7, 512, 3, 1, 1), (128, 512, 7, 7, 512, 3, 1, 1), (5, 512, 7, 7, 200, 3, 1, 1),
                                                 (2, 256, 14, 14, 256, 3, 1, 1), (3, 256, 14, 14, 512, 3, 2, 1), (1, 128, 28, 28, 128, 3, 1, 1), (9, 128, 28, 28, 256, 3, 2, 1),
                                                 (2, 256, 14, 14, 512, 1, 2, 0), (3, 128, 9, 11, 64, 3, 2, 1), (2, 512, 5, 60, 72, 3, 1, 1), (1, 128, 13, 64, 130, 1, 1, 0),
                                                 (2, 256, 6, 6, 96, 3, 1, 0), (70, 128, 4, 4, 64, 3, 1, 2),
                                                 # stride 3: picture columns beyond the last tap (the fuzz sweep's finding: they were stored onto the next row's border)
                                                 (1, 128, 19, 12, 32, 3, 3, 1), (8, 512, 29, 6, 200, 3, 3, 1), (10, 256, 9, 7, 129, 1, 3, 1)])
def test_binary_conv_one_launch_form_against_the_oracle_and_the_other_forms(B, C, H, W, OC, ks, st, pad, monkeypatch):
    """bie_binary_conv2d_forward_fused (configs[3] as ONE launch, no workspace): sign-pack into an LDS bit image, register-resident
    weight quarters, XNOR-popcount against uniform LDS reads.  Bit-exact against the oracle's integers (pinned by the compiled binary_conv.cpp,
    tests/test_oracle_golden.py) and against the forms it replaces on these shapes (tap kernels / FP4 matrix-pipe GEMM, forced through
    BIE_CONV_FUSED_MAX_ROWS=0); ResNet stage shapes, stride 2, 1x1, pad 0 and 2, output channel counts that are not multiples of 64 / 128,
    one / several row chunks per image, both channel-group widths, fp32 / fp16 / bf16 inputs with zeros, negative zeros and NaNs."""
    from bitorch_engine import _hip
    from bitorch_engine.extensions import _binary_common as bc
    gen = torch.Generator().manual_seed(B * 7 + C + H + W + OC + ks)
    x = torch.randn((B, C, H, W), generator=gen)
    x.view(-1)[::97] = 0.0
    x.view(-1)[5::131] = -0.0
    x.view(-1)[11::257] = float("nan")
    w = torch.randn((OC, C, ks, ks), generator=gen)
    wp = bc.pack_rows(w.reshape(OC, -1).to(DEV)).contiguous()
    L = _hip.lib()
    assert L.bie_binary_conv2d_fused_ok(B, C, H, W, OC, ks, st, pad, 1) == 1
    OH, OW = (H + 2 * pad - ks) // st + 1, (W + 2 * pad - ks) // st + 1
    wl = bc.conv_weight_lanes(wp, OC, C, ks)
    assert wl.numel() * 4 == L.bie_binary_conv_weight_lanes_bytes(OC, C, ks)
    xd = x.to(DEV)
    y = torch.full((B, OC, OH, OW), float("nan"), dtype=torch.float32, device=DEV)
    rc = L.bie_binary_conv2d_forward_fused(xd.data_ptr(), wl.data_ptr(), y.data_ptr(), B, C, H, W, OC, ks, st, pad, 1, 0.25, _hip.F32, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, L.bie_last_error().decode()
    xo = torch.where(torch.isnan(x), torch.tensor(-1.0), x)  # the sign rule (value >= 0) makes a NaN a -1; the oracle takes the value it stands for
    want = orc.binary_conv2d(xo.numpy(), w.numpy(), st, pad, 1).astype(np.float32) * np.float32(0.25)
    assert np.array_equal(y.cpu().numpy(), want), "one-launch conv differs from the oracle"
    # the one-launch MATRIX-PIPE form (FP4 image of the input rows in LDS, fragments gathered tap by tap) through the C ABI
    mfma = L.bie_binary_conv2d_mfma_ok(B, C, H, W, OC, ks, st, pad, 1) == 1
    assert mfma or (H, W) == (5, 60)   # (three input rows of 62 pixels x 512 channels as FP4 do not fit its 64 KiB LDS image)
    wimg = bc.conv_weight_fp4_image(wp, OC, C, ks)
    y2 = torch.full_like(y, float("nan"))
    rc = L.bie_binary_conv2d_forward_mfma(xd.data_ptr(), wimg.data_ptr(), y2.data_ptr(), B, C, H, W, OC, ks, st, pad, 1, 0.25, _hip.F32, torch.cuda.current_stream().cuda_stream)
    if mfma:
        assert rc == 0, L.bie_last_error().decode()
        assert torch.equal(y2, y), "one-launch matrix-pipe conv differs from the one-launch VALU conv"
    else:
        assert rc == -2
    # the dispatch (whichever one-launch form it picks for this size) and the forms of round 5 (both one-launch forms switched off)
    assert torch.equal(bc.conv2d(xd, wp, OC, ks, st, pad, 1, 0.25), y)
    monkeypatch.setenv("BIE_CONV_FUSED_MAX_ROWS", "0")
    assert torch.equal(bc.conv2d(xd, wp, OC, ks, st, pad, 1, 0.25), y)          # forced onto the matrix-pipe one-launch form
    monkeypatch.setenv("BIE_CONV_MFMA_MAX_ROWS", "0")
    assert torch.equal(bc.conv2d(xd, wp, OC, ks, st, pad, 1, 0.25), y)          # tap kernels / three-launch FP4 GEMM agree bit for bit
    monkeypatch.delenv("BIE_CONV_FUSED_MAX_ROWS")
    monkeypatch.delenv("BIE_CONV_MFMA_MAX_ROWS")
    for tdt in (torch.float16, torch.bfloat16):
        xt = x.to(tdt).to(DEV)
        assert torch.equal(bc.conv2d(xt, wp, OC, ks, st, pad, 1, 0.25), y)
        if mfma:
            rc = L.bie_binary_conv2d_forward_mfma(xt.data_ptr(), wimg.data_ptr(), y2.data_ptr(), B, C, H, W, OC, ks, st, pad, 1, 0.25, _hip._DT[tdt], torch.cuda.current_stream().cuda_stream)
            assert rc == 0 and torch.equal(y2, y)


def test_binary_conv_one_launch_form_refuses_what_it_cannot_run():
    from bitorch_engine import _hip
    L = _hip.lib()
    ok = lambda *a: L.bie_binary_conv2d_fused_ok(*a)  # noqa: E731  (B, C, H, W, OC, k, stride, pad, dil)
    assert ok(1, 512, 7, 7, 512, 3, 1, 1, 1) == 1
    assert ok(1, 64, 56, 56, 64, 3, 1, 1, 1) == 0      # C = 64: no four K quarters of whole words
    assert ok(1, 512, 7, 7, 512, 5, 1, 2, 1) == 0      # k = 5
    assert ok(1, 512, 7, 7, 512, 3, 1, 2, 2) == 0      # dilation
    assert ok(1, 128, 8, 300, 8, 3, 1, 1, 1) == 0      # an output row longer than a wave
    x = torch.zeros((1, 64, 8, 8), device=DEV)
    y = torch.zeros((1, 8, 8, 8), device=DEV)
    rc = L.bie_binary_conv2d_forward_fused(x.data_ptr(), x.data_ptr(), y.data_ptr(), 1, 64, 8, 8, 8, 3, 1, 1, 1, 1.0, _hip.F32, None)
    assert rc == -2 and "one-launch form" in L.bie_last_error().decode()


@pytest.mark.parametrize("B,C,H,OC,ks,st,pad,dil", [(2, 64, 14, 64, 3, 1, 1, 1), (1, 512, 7, 512, 3, 1, 1, 1), (3, 32, 9, 48, 3, 2, 1, 1), (2, 128, 8, 64, 1, 1, 0, 1)])
def test_binary_conv2d_cutlass_layer_packed_equals_unpacked_and_the_a15_oracle(B, C, H, OC, ks, st, pad, dil):
    """SURVEY A16 / VERDICT r3: BinaryConv2dCutlass and binary_conv2d_cutlass.forward / w_pack.  What the reference's own test pins
    (tests/layers/test_binary_conv.py:157-170) is that the packed (eval) and the unpacked (training-mode weight) forward agree; this
    build additionally DEFINES the result as the A15 convolution (NCHW in, [B, OC, OH, OW] out, padding counted as -1,
    (C k k - 2 popc) * scale_a * scale_w) -- INTEGRATION.md states why the reference's NHWC-view / raw-popcount kernel is not reproduced
    -- so the layer is also held to the oracle's integers."""
    from bitorch_engine.extensions import binary_conv2d_cutlass
    from bitorch_engine.layers.qconv.binary.cutlass import BinaryConv2dCutlass
    gen = torch.Generator().manual_seed(B * 1000 + C + H + OC)
    x = torch.randn((B, C, H, H), generator=gen)
    w = torch.randn((OC, C, ks, ks), generator=gen)
    layer = BinaryConv2dCutlass(C, OC, ks, stride=st, padding=pad, dilation=dil)
    layer.set_weight_data(w.clone())
    layer.to(DEV)
    layer.train()
    with torch.no_grad():
        y_unpacked = layer(x.to(DEV))
    layer.generate_quantized_weight(qweight_only=True)
    layer.eval()
    with torch.no_grad():
        y_packed = layer(x.to(DEV))
    assert layer.weight is None and layer.qweight.dtype == torch.uint8 and tuple(layer.qweight.shape) == (OC, C * ks * ks // 8)
    assert torch.equal(y_unpacked, y_packed), "packed and unpacked forward differ"
    # the extension functions by themselves: w_pack + forward on the centred sign carriers == the oracle's integers x scale
    wc = w - w.mean()
    carriers = torch.where(wc >= 0, 1.0, -1.0)
    xa = x  # bias_a is zero after construction
    ints = torch.from_numpy(orc.binary_conv2d(xa.numpy(), carriers.numpy(), st, pad, dil))
    scale = layer.scale_a.item() * layer.scale_w.item()
    assert scale > 0
    y_ext = binary_conv2d_cutlass.forward(xa.to(DEV), binary_conv2d_cutlass.w_pack(carriers.to(DEV)), scale, False, ks, st, pad, dil)
    assert torch.equal(y_ext.cpu(), ints * torch.tensor(scale, dtype=torch.float32))
    assert torch.equal(y_packed.cpu(), (ints * torch.tensor(scale, dtype=torch.float32)).to(y_packed.dtype))


@pytest.mark.parametrize("B,C,H,W,OC,ks,st,pad,dil", [(2, 1024, 6, 6, 8, 3, 1, 1, 1), (1, 2048, 5, 5, 16, 1, 2, 0, 1), (2, 1024, 7, 7, 72, 3, 2, 2, 1),
                                                       (1, 1024, 8, 8, 5, 3, 1, 1, 2), (3, 1024, 5, 7, 8, 5, 1, 1, 1), (8, 1024, 12, 12, 64, 3, 1, 1, 1)])
def test_binary_conv2d_cutlass_reference_convention_opt_in(B, C, H, W, OC, ks, st, pad, dil):
    """VERDICT r4 missing #3: the reference kernel's OWN output convention as an opt-in mode (checkpoints trained against that layer):
    viewed layouts, C/8 one-bit channels, flipped filter, raw XOR popcounts, zero bits outside the image, [B, out_edge, out_edge, OC]
    float32 = int32 * scale (binary_conv2d_cutlass_kernel.cu:206-228,260,271,414-453).  The kernel needs CUDA + CUTLASS: parity UNPINNED by
    any reference output; the comparison is with the numpy restatement of those lines (oracle.binary_conv2d_cutlass_reference_convention),
    bit-exact, for training-mode (value) and eval-mode (w_pack bytes) weights; the layer switch; and what the reference's own test asserts
    (packed == unpacked, tests/layers/test_binary_conv.py:157-170).  Geometries: dilation 2 and H != W reach past the padding (the
    reference's out_edge ignores both); 8 x 12 x 12 x 64 outputs = the matrix-pipe form of the popcount GEMM."""
    from bitorch_engine.extensions import binary_conv2d_cutlass
    from bitorch_engine.layers.qconv.binary.cutlass import BinaryConv2dCutlass
    gen = torch.Generator().manual_seed(B * 977 + C + H * 13 + W + OC + ks)
    x = torch.randn((B, C, H, W), generator=gen)
    w = torch.randn((OC, C, ks, ks), generator=gen)
    ref = torch.from_numpy(orc.binary_conv2d_cutlass_reference_convention(x.numpy(), w.numpy(), 0.375, ks, st, pad, dil))
    oe = (W - ks + 2 * pad) // st + 1
    assert tuple(ref.shape) == (B, oe, oe, OC)
    y_vals = binary_conv2d_cutlass.forward_reference_convention(x.to(DEV), w.to(DEV), 0.375, True, ks, st, pad, dil)
    assert y_vals.dtype == torch.float32 and torch.equal(y_vals.cpu(), ref), "training-mode weights (values)"
    y_packed = binary_conv2d_cutlass.forward_reference_convention(x.to(DEV), binary_conv2d_cutlass.w_pack(w.to(DEV)), 0.375, False, ks, st, pad, dil)
    assert torch.equal(y_packed.cpu(), ref), "eval-mode weights (w_pack bytes)"
    for tdt in (torch.float16, torch.bfloat16):
        assert torch.equal(binary_conv2d_cutlass.forward_reference_convention(x.to(tdt).to(DEV), w.to(DEV), 0.375, True, ks, st, pad, dil).cpu(), ref)
    if H == W:  # the layer, like the reference's own test
        layer = BinaryConv2dCutlass(C, OC, ks, stride=st, padding=pad, dilation=dil, reference_convention=True)
        layer.set_weight_data(w.clone())
        layer.to(DEV)
        layer.train()
        with torch.no_grad():
            y_unpacked = layer(x.to(DEV))
        layer.generate_quantized_weight(qweight_only=True)
        layer.eval()
        with torch.no_grad():
            y_pk = layer(x.to(DEV))
        assert tuple(y_pk.shape) == (B, oe, oe, OC) and torch.equal(y_unpacked, y_pk)
        carriers = torch.where((w - w.mean()) >= 0, 1.0, -1.0)
        scale = layer.scale_a.item() * layer.scale_w.item()
        exp = torch.from_numpy(orc.binary_conv2d_cutlass_reference_convention(x.numpy(), carriers.numpy(), scale, ks, st, pad, dil))
        assert torch.equal(y_pk.cpu(), exp)


def test_binary_conv2d_cutlass_reference_convention_refuses_what_the_reference_cannot_run(monkeypatch):
    from bitorch_engine.extensions import binary_conv2d_cutlass
    from bitorch_engine.layers.qconv.binary.cutlass import BinaryConv2dCutlass
    x = torch.randn((1, 512, 4, 4), device=DEV)
    w = torch.randn((8, 512, 3, 3), device=DEV)
    with pytest.raises(RuntimeError, match="multiple of 128"):
        binary_conv2d_cutlass.forward_reference_convention(x, w, 1.0, True, 3, 1, 1, 1)
    assert BinaryConv2dCutlass(64, 8, 3).reference_convention is False  # the A15 convolution stays the default
    monkeypatch.setenv("BIE_BCONV_CUTLASS_CONVENTION", "reference")
    assert BinaryConv2dCutlass(64, 8, 3).reference_convention is True


# ------------------------------------------------------------------------------------------------ functions
def test_functions_cuda_helpers():
    from bitorch_engine.functions.cuda import (tensor_to_packed_uint8, unpack_uint8_tensor, q4_pack_tensor,
                                               q4_unpack_tensor, q4_unpack_and_scaling_tensor)
    d = np.load(os.path.join(GOLDEN, "kat_unpack_uint8.npz"))
    emb = torch.from_numpy(d["bytes"]).reshape(1, 1, 4).expand(2, 16, 4).contiguous()
    scale = torch.rand(2, 16, 1)
    out = unpack_uint8_tensor(emb.to(DEV), scale.to(DEV)).cpu()
    expect = torch.from_numpy(d["expected"]).reshape(1, 1, 32) * scale
    assert torch.equal(out, expect)
    for dtype in (torch.float32, torch.float16, torch.bfloat16, torch.int8):
        v = torch.randn(8, 64)
        v = (v * 10).to(torch.int8) if dtype == torch.int8 else v.to(dtype)
        p = tensor_to_packed_uint8(v.to(DEV)).cpu()
        assert p.shape == (8, 8) and p.dtype == torch.uint8
        assert np.array_equal(p.numpy(), orc.pack_sign_u8(v.float().numpy()))
    a = torch.randint(-8, 8, (10, 10), dtype=torch.int32)
    packed = q4_pack_tensor(a.to(DEV))
    assert packed.dtype == torch.int8 and packed.numel() * 2 == a.numel()
    assert np.array_equal(packed.cpu().numpy(), orc.q4_pack(a.numpy()))
    un = q4_unpack_tensor(packed).cpu()
    assert torch.equal(un & 0xF, a & 0xF)
    assert torch.equal(q4_unpack_and_scaling_tensor(packed, 0.25).cpu(), a.float() * 0.25)


# ------------------------------------------------------------------------------------------------ boundary on device
def test_c_abi_error_codes_on_device():
    from bitorch_engine import _hip
    L = _hip.lib()
    x = torch.zeros(8, device=DEV)
    rc = L.bie_mpq_forward(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), None, None, x.data_ptr(), None, 0, 1, 64, 32, 3, 32, 0, 0, None)
    assert rc == -2 and b"w_bit" in L.bie_last_error()
    rc = L.bie_mpq_forward(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), None, None, x.data_ptr(), None, 0, 1, 4096, 4096, 4, 128, 0, 0, None)
    assert rc == -3 and b"workspace" in L.bie_last_error()


# ------------------------------------------------------------------ W4A4 / W8A8 on the i8 matrix cores
_TDT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
def test_q4_quantize_pack_bit_exact(dt):
    from bitorch_engine.extensions import q_linear_cutlass as qc
    g = torch.Generator().manual_seed(11)
    x = (torch.randn((37, 192), generator=g) * 2).to(_TDT[dt])
    x[0, :8] = torch.tensor([0.5, -0.5, 1.5, -1.5, 2.5, 100.0, -100.0, 0.0]).to(_TDT[dt])  # ties and clamps
    for scale in (1.0, 0.37, 1e-7):
        want = orc.q4_quantize_pack(orc.torch_to_np(x), scale, orc.dt_code(_TDT[dt]))
        got = qc.q4_w_pack(x.to(DEV), scale).cpu().numpy()
        assert np.array_equal(got, want), (dt, scale)
    # transposed weights ([k, n] storage)
    got = qc.q4_w_pack(x.t().contiguous().to(DEV), 0.37, True).cpu().numpy()
    assert np.array_equal(got, orc.q4_quantize_pack(orc.torch_to_np(x), 0.37, orc.dt_code(_TDT[dt])))


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("M,N,K", [(1, 4, 64), (5, 36, 128), (128, 128, 256), (130, 260, 192), (257, 124, 1024), (300, 520, 128), (512, 3072, 1024), (1536, 8192, 384), (130, 260, 4096)])
def test_q4_forward_bit_exact_vs_oracle(dt, M, N, K):  # M, N >= 128 and K % 128 == 0: the pipeline kernel (intgemm_pipe.hip), both tile sizes, 1 / 3 / 8 / 32 K stages
    from bitorch_engine.extensions import q_linear_cutlass as qc
    g = torch.Generator().manual_seed(M * 131 + N)
    x = torch.randn((M, K), generator=g).to(_TDT[dt])
    w = (torch.randn((N, K), generator=g) * 0.05).to(_TDT[dt])
    sa, sw = float(2 * x.float().abs().mean() / 11.269), float(2 * w.float().abs().mean() / 5.6345)
    code = orc.dt_code(_TDT[dt])
    pa, pw = orc.q4_quantize_pack(orc.torch_to_np(x), sa, code), orc.q4_quantize_pack(orc.torch_to_np(w), sw, code)
    want = orc.q4_gemm(pa, pw, K, sa, sw, code)
    out, qa, qw = qc.q4_forward(x.to(DEV), w.to(DEV), torch.tensor(sa), torch.tensor(sw), False, False)
    assert np.array_equal(qa.cpu().numpy(), pa) and np.array_equal(qw.cpu().numpy(), pw)
    assert np.array_equal(orc.torch_to_np(out), want)
    # the packed-weight (inference) call gives the same bits, as the reference's own test asserts (test_nbit_linear.py:69)
    out2 = qc.q4_forward(x.to(DEV), qw, sa, sw, False, False)[0]
    assert torch.equal(out, out2)


@pytest.mark.parametrize("M,N,K", [(1, 4, 64), (33, 100, 320), (256, 384, 512), (129, 132, 4096), (300, 520, 64), (512, 3072, 1024), (1536, 8192, 192)])
def test_q8_forward_bit_exact_vs_oracle(M, N, K):  # M, N >= 128: the pipeline kernel (intgemm_pipe.hip), both tile sizes, 1 / 3 / 16 / 64 K stages
    from bitorch_engine.extensions import q_linear_cutlass as qc
    rng = np.random.default_rng(M + N)
    a = rng.integers(-128, 128, (M, K)).astype(np.int8)
    w = rng.integers(-128, 128, (N, K)).astype(np.int8)
    want = orc.q8_gemm(a, w, 0.013, 0.0021)
    got = qc.q8_forward(torch.from_numpy(a).to(DEV), torch.from_numpy(w).to(DEV), False, torch.tensor(0.013), torch.tensor(0.0021))
    assert np.array_equal(got.cpu().numpy(), want)
    got_t = qc.q8_forward(torch.from_numpy(a).to(DEV), torch.from_numpy(w).t().contiguous().to(DEV), True, 0.013, 0.0021)
    assert torch.equal(got, got_t)


def test_q4_matmul_batched_vs_oracle():
    from bitorch_engine.extensions import q_linear_cutlass as qc
    g = torch.Generator().manual_seed(3)
    B, H, S, D = 2, 3, 40, 64
    x = torch.randn((B, H, S, D), generator=g).half()
    y = torch.randn((B, H, 72, D), generator=g).half()
    out, qx, qy = qc.q4_matmul(x.to(DEV), y.to(DEV), torch.tensor(0.3), torch.tensor(0.4))
    assert out.shape == (B, H, S, 72)
    for b in range(B):
        for h in range(H):
            pa = orc.q4_quantize_pack(orc.torch_to_np(x[b, h]), 0.3, orc.F16)
            pb = orc.q4_quantize_pack(orc.torch_to_np(y[b, h]), 0.4, orc.F16)
            assert np.array_equal(qx[b, h].cpu().numpy(), pa)
            assert np.array_equal(orc.torch_to_np(out[b, h]), orc.q4_gemm(pa, pb, D, 1.0, 1.0, orc.F16))


def test_q4_q8_cutlass_layers_train_eval_equivalence():
    """The reference's own check (tests/layers/test_nbit_linear.py:28-69, 149-190): the layer evaluated from its packed
    weight gives exactly the result computed from the float weight, and tracks the float linear layer."""
    from bitorch_engine.layers.qlinear.nbit.cutlass import Q4LinearCutlass, Q8LinearCutlass
    torch.manual_seed(0)
    x = torch.randn((48, 512)).half()
    for cls, tol in ((Q4LinearCutlass, 0.5), (Q8LinearCutlass, 0.1)):
        layer = cls(in_channels=512, out_channels=256, dtype=torch.half).to(DEV)
        layer.prepare_params()
        wfloat = layer.weight.data.clone()
        with torch.no_grad():
            y = layer.train()(x.to(DEV))  # float weight quantised inside the call
        layer.generate_quantized_weight(qweight_only=True)
        assert layer.weight is None and layer.qweight.dtype == torch.int8
        y2 = layer.eval()(x.to(DEV))
        assert torch.equal(y, y2)
        ref = x.float() @ wfloat.float().cpu().t()
        rel = (y.float().cpu() - ref).norm() / ref.norm()
        assert rel < tol, (cls.__name__, float(rel))


def test_q4_gemm_error_codes():
    from bitorch_engine import _hip
    a = torch.zeros((4, 24), dtype=torch.int8, device=DEV)
    y = torch.zeros((4, 4), dtype=torch.float16, device=DEV)
    rc = _hip.lib().bie_q4_gemm(_hip.ptr(a), _hip.ptr(a), _hip.ptr(y), 4, 4, 48, 1.0, 1.0, 0, 1, 0, 0, 0, None)
    assert rc != 0 and b"multiple of 64" in _hip.lib().bie_last_error()


@pytest.mark.parametrize("dt", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("B,C,H,OC,ks,st,pad,dil", [(2, 64, 14, 64, 3, 1, 1, 1), (1, 32, 9, 36, 3, 2, 1, 1), (3, 8, 7, 4, 1, 1, 0, 1),
                                                     (1, 32, 11, 32, 3, 1, 2, 2), (2, 128, 7, 256, 3, 1, 1, 1)])
def test_q4_conv2d_bit_exact_vs_oracle(dt, B, C, H, OC, ks, st, pad, dil):
    from bitorch_engine.extensions import q4_conv_cutlass as qc
    g = torch.Generator().manual_seed(B * 7 + C)
    x = torch.randn((B, C, H, H), generator=g).to(_TDT[dt])
    w = (torch.randn((OC, C, ks, ks), generator=g) * 0.1).to(_TDT[dt])
    sa, sw = float(2 * x.float().abs().mean() / 11.269), float(2 * w.float().abs().mean() / 5.6345)
    code = orc.dt_code(_TDT[dt])
    # the reference VIEWS the NCHW buffers as NHWC (q4_conv_cutlass_kernel.cu:474-480)
    pa = orc.q4_quantize_pack(orc.torch_to_np(x.view(B, H, H, C)), sa, code)
    pw = orc.q4_quantize_pack(orc.torch_to_np(w.view(OC, ks, ks, C)), sw, code)
    want = orc.q4_conv2d(pa, pw, ks, st, pad, dil, sa, sw, code)
    out, qa, qw = qc.forward(x.to(DEV), w.to(DEV), torch.tensor(sa), torch.tensor(sw), False, ks, st, pad, dil)
    assert np.array_equal(qa.cpu().numpy(), pa) and np.array_equal(qw.cpu().numpy(), pw)
    assert np.array_equal(qc.w_pack(w.to(DEV), sw).cpu().numpy(), pw)
    assert out.shape == want.shape and np.array_equal(orc.torch_to_np(out), want)


def test_q4_conv_layer_train_eval_equivalence():
    """reference check (tests/layers/test_nbit_conv.py): result from the packed weight == result from the float weight"""
    from bitorch_engine.layers.qconv.nbit.cutlass import Q4Conv2dCutlass
    torch.manual_seed(1)
    layer = Q4Conv2dCutlass(in_channels=64, out_channels=32, kernel_size=3, stride=1, padding=1, dilation=1, dtype=torch.half).to(DEV)
    layer.prepare_params()
    x = torch.randn((2, 64, 10, 10)).half().to(DEV)
    with torch.no_grad():
        y = layer.train()(x)
    layer.generate_quantized_weight(qweight_only=True)
    y2 = layer.eval()(x)
    assert y.shape == (2, 32, 10, 10) and torch.equal(y, y2)


def test_exl2_decode_long_k_uses_several_slabs():
    """K = 28672 (a 70B-class down projection): 896 chunks do not fit one LDS metadata slab, so even a wide layer (>= 160 column
    blocks) is cut into K slabs reduced by granules; single band (4-bit rows), random q_perm."""
    from bitorch_engine.extensions import q_linear_cuda
    from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
    K, N, gs = 28672, 10240, 128
    qg, row = [], 0
    for _ in range(K // gs):
        qg += [4, row]
        row += 4 * (gs // 32)
    groups = len(qg) // 2
    q_groups = torch.tensor(qg, dtype=torch.short)
    gmap = make_group_map(q_groups, row)
    rng = np.random.default_rng(3)
    gen = torch.Generator().manual_seed(3)
    qw = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, (row, N), dtype=np.int64).astype(np.int32))
    scales = (torch.rand((groups, N), generator=gen) * 0.02 + 0.001).half()
    zeros = (torch.randn((groups, N), generator=gen) * 0.1).half()
    q_perm = torch.randperm(K, generator=gen).to(torch.short)
    qs, rows = exl2_load(qw, q_groups, K, groups)
    x = torch.randn((1, K), generator=gen).half()
    y = q_linear_cuda.mbwq_exl2_forward(x.to(DEV), qs, scales.to(DEV), zeros.to(DEV), q_perm.to(DEV), gmap.to(DEV), rows, False)
    cols = np.arange(0, N, 257)
    Wo = orc.exl2_dequant(np.ascontiguousarray(qw.numpy()[:, cols]), orc.torch_to_np(scales[:, cols].contiguous()),
                          orc.torch_to_np(zeros[:, cols].contiguous()), q_perm.numpy(), q_groups.numpy(), K)
    ref = t16(orc.gemm(orc.torch_to_np(x), Wo, orc.F16), orc.F16)
    assert_close(y[:, torch.from_numpy(cols).to(DEV)], ref, orc.F16, "exl2 K=28672")


# ------------------------------------------------------------------------------------------------ table-lookup decode GEMV / grouped launch
@pytest.mark.parametrize("K,N,gs,asym,M", [(1024, 200, 64, 0, 1), (768, 520, 256, 0, 2), (640, 64, 32, 1, 1), (1408, 136, 128, 0, 1),
                                            (2048, 1000, 128, 1, 2), (512, 72, 32, 0, 2)])
def test_lut_gemv_group_sizes_ragged_tiles_and_tails(K, N, gs, asym, M):
    """bf16 W4 decode goes through mpq_gemv_lut.hip: every supported group size, column counts that are not multiples of 64,
    group counts that do not divide by the groups-per-workgroup (ragged last K slice), sym and asym, M = 1 and 2.  The table
    holds the exact reference values, so the only difference from the oracle is fp32 summation order."""
    rng = np.random.default_rng(K + N + gs + M)
    qw, scales, zeros, gen = rand_case(rng, K, N, 4, gs, orc.BF16, asym)
    x = torch.randn((M, K), generator=gen).to(torch.bfloat16)
    bias = (torch.randn(N, generator=gen) * 0.1).to(torch.bfloat16)
    y = hip_forward(x, qw, scales, zeros, None, 4, gs, asym, bias)
    ref = oracle_forward(x, qw, scales, zeros, None, 4, gs, asym, orc.BF16, bias)
    assert_close(y, ref, orc.BF16, f"K={K} N={N} g={gs} asym={asym} M={M}")
    assert torch.equal(y, hip_forward(x, qw, scales, zeros, None, 4, gs, asym, bias)), "not reproducible run to run"


@pytest.mark.parametrize("dt", [orc.BF16, orc.F16])
@pytest.mark.parametrize("K,N,gs,asym,M", [(1024, 200, 64, 0, 3), (768, 520, 256, 0, 8), (640, 64, 32, 1, 5), (1408, 136, 128, 0, 4),
                                            (2048, 1000, 128, 1, 7), (4096, 4096, 128, 0, 6), (4096, 12288, 128, 0, 3),
                                            (1024, 200, 128, 0, 1), (1024, 4096, 128, 1, 2), (2048, 328, 64, 0, 12), (4096, 1024, 128, 1, 16),
                                            (512, 68, 32, 0, 9)])
def test_lut_gemv_matrix_pipe_form_small_batches(K, N, gs, asym, M, dt):
    """2 <= M <= 16, W4 (M = 1: the FMA form, bf16 FMAs / fp16 v_fma_mix_f32): mpq_gemv_lutm_kernel (table lookup feeding 16x16x32 MFMAs; rows >= M of the x
    fragment come back as zeros from the buffer descriptor's bounds check).  Same exact table values, so again only the fp32
    summation order differs from the oracle; every group size, ragged column tiles (N % 64 = 4, 8, 40), split groups and
    one-unit-per-wave plans, sym and asym."""
    rng = np.random.default_rng(K + N + gs + M + dt)
    qw, scales, zeros, gen = rand_case(rng, K, N, 4, gs, dt, asym)
    x = torch.randn((M, K), generator=gen).to(TDT[dt])
    bias = (torch.randn(N, generator=gen) * 0.1).to(TDT[dt])
    y = hip_forward(x, qw, scales, zeros, None, 4, gs, asym, bias)
    ref = oracle_forward(x, qw, scales, zeros, None, 4, gs, asym, dt, bias)
    assert_close(y, ref, dt, f"K={K} N={N} g={gs} asym={asym} M={M}")
    assert torch.equal(y, hip_forward(x, qw, scales, zeros, None, 4, gs, asym, bias)), "not reproducible run to run"


@pytest.mark.parametrize("dt", [orc.BF16, orc.F16])
@pytest.mark.parametrize("K,N,gs,asym,M", [(1024, 208, 64, 0, 1), (768, 528, 256, 0, 2), (1408, 144, 128, 1, 1), (2048, 1008, 128, 1, 2),
                                            (4096, 4096, 128, 0, 1), (512, 80, 64, 0, 2)])
def test_lut_gemv_w2_pair_lookup(K, N, gs, asym, M, dt):
    """W2A16 decode (M <= 2): the lookup kernel's pair form -- a nibble of the packed word indexes a 16-entry table of weight PAIRS,
    one v_dot2 per pair against the packed x pair.  Table values are the reference's doubly rounded weights, so only the fp32
    summation order differs from the oracle; group sizes 64 / 128 / 256, sym and asym, ragged column tiles, fp16 and bf16."""
    rng = np.random.default_rng(K + N + gs + M + dt)
    qw, scales, zeros, gen = rand_case(rng, K, N, 2, gs, dt, asym)
    x = torch.randn((M, K), generator=gen).to(TDT[dt])
    bias = (torch.randn(N, generator=gen) * 0.1).to(TDT[dt])
    y = hip_forward(x, qw, scales, zeros, None, 2, gs, asym, bias)
    ref = oracle_forward(x, qw, scales, zeros, None, 2, gs, asym, dt, bias)
    assert_close(y, ref, dt, f"w2 K={K} N={N} g={gs} asym={asym} M={M}")
    assert torch.equal(y, hip_forward(x, qw, scales, zeros, None, 2, gs, asym, bias)), "not reproducible run to run"


@pytest.mark.parametrize("split", ["1", "2", "4"])
def test_lut_gemv_split_groups(split, monkeypatch):
    """Small layers split every quantisation group over 2 or 4 waves (BIE_LUT_H is read once per process: the test forces the
    plan through a subprocess-free path by choosing shapes whose automatic plan takes each branch)."""
    K, gs = 4096, 128
    N = {"1": 12288, "2": 4096, "4": 1024}[split]  # tiles * groups = 6144 / 2048 / 512 -> H = 1 / 2 / 4
    rng = np.random.default_rng(99 + N)
    qw, scales, zeros, gen = rand_case(rng, K, N, 4, gs, orc.BF16, 0)
    x = torch.randn((1, K), generator=gen).to(torch.bfloat16)
    y = hip_forward(x, qw, scales, zeros, None, 4, gs, 0)
    ref = oracle_forward(x, qw, scales, zeros, None, 4, gs, 0, orc.BF16)
    assert_close(y, ref, orc.BF16, f"N={N}")


def test_grouped_forward_matches_separate_calls_and_oracle():
    """bie_mpq_forward_grouped: q/k/v-style sets sharing x in one launch (bf16 W4 -> one grid) and the per-set fallback (fp16)."""
    from bitorch_engine.extensions import q_linear_cuda
    K, gs = 1024, 128
    for dt in (orc.BF16, orc.F16):
        for M in (1, 2, 5):
            sets, refs, x = [], [], None
            for i, N in enumerate((384, 200, 72)):
                rng = np.random.default_rng(1000 * i + M + dt)
                qw, scales, zeros, gen = rand_case(rng, K, N, 4, gs, dt, 0)
                if x is None:
                    x = torch.randn((M, K), generator=gen).to(TDT[dt])
                bias = (torch.randn(N, generator=gen) * 0.1).to(TDT[dt]) if i == 1 else None
                sets.append(tuple(None if t is None else t.to(DEV) for t in (qw, scales, zeros, bias)))
                refs.append(oracle_forward(x, qw, scales, zeros, None, 4, gs, 0, dt, bias))
            ys = q_linear_cuda.mpq_forward_grouped_impl(x.to(DEV), sets, 4, 0, gs)
            for i, (y, r) in enumerate(zip(ys, refs)):
                assert_close(y, r, dt, f"grouped dt={dt} M={M} set {i}")
                single = q_linear_cuda.mpq_forward_impl(x.to(DEV), sets[i][0], sets[i][1], sets[i][2], None, 4, 0, gs, sets[i][3])
                assert torch.equal(y, single) or dt == orc.BF16, "fallback path must equal separate calls bit for bit"


@pytest.mark.parametrize("dt", [orc.F16, orc.BF16])
@pytest.mark.parametrize("K,widths", [(4096, (4096, 4096, 4096)), (4096, (11008, 11008))])
def test_full_size_grouped_launches_fp16_element_wise(K, widths, dt):
    """the grouped decode launches at the Llama-7B shapes (q/k/v, gate/up), fp16 (GreenBit's checkpoint dtype), against
    the oracle norm-wise AND element by element (1e-3 relative + one fp16 ulp on every output with |ref| >= 2^-6 max|ref|); bf16: norm-wise +
    one output ulp (gate/up at M = 1 is 45 MB: the inline list form since round 6's 40 MB threshold)."""
    from bitorch_engine.extensions import q_linear_cuda
    for M in (1, 4):
        sets, refs, x = [], [], None
        for i, N in enumerate(widths):
            rng = np.random.default_rng(77 * i + N + M)
            qw, scales, zeros, gen = rand_case(rng, K, N, 4, 128, dt, 0)
            if x is None:
                x = torch.randn((M, K), generator=gen).to(TDT[dt])
            sets.append(tuple(None if t is None else t.to(DEV) for t in (qw, scales, zeros, None)))
            refs.append(oracle_forward(x, qw, scales, zeros, None, 4, 128, 0, dt))
        ys = q_linear_cuda.mpq_forward_grouped_impl(x.to(DEV), sets, 4, 0, 128)
        for i, (y, r) in enumerate(zip(ys, refs)):
            assert_close(y, r, dt, f"grouped full-size M={M} member {i}")
            if dt == orc.F16:
                assert_close_elementwise_f16(y, r, f"grouped full-size fp16 {K}x{widths[i]} M={M} member {i}")


def test_layer_level_grouped_forward_equals_the_separate_layers():
    """MPQLinearCuda.forward_grouped: q/k/v-style layers sharing x through one grouped launch against each layer's own forward (same
    table values; the K-split plan of the larger grid may differ, hence the summation-order tolerance), and the fallback for a set
    that cannot be grouped (bit-equal: it IS the layers' forward)."""
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQLinearCuda
    K, gs = 1024, 128
    g = torch.Generator().manual_seed(11)
    layers = []
    for N in (256, 384, 128):
        layer = MPQLinearCuda(K, N, w_bit=4, dtype=torch.bfloat16, group_size=gs, dq_group_size=32, use_gba_quant=True, asym=False)
        layer.qweight.data = torch.randint(-2 ** 31, 2 ** 31 - 1, layer.qweight.shape, generator=g, dtype=torch.int64).to(torch.int32)
        layer.prepare_params()
        layer.scales = (torch.rand(layer.scales.shape, generator=g) * 0.01 + 0.005).bfloat16()
        layer.zeros = (layer.scales.float() * torch.rand(layer.scales.shape, generator=g) * 15).bfloat16()
        layers.append(layer.to(DEV).eval())
    for lead in ((1,), (2, 3), (40,)):  # 40 rows: not groupable -> per-layer forward
        x = torch.randn(lead + (K,), generator=g).bfloat16().to(DEV)
        with torch.no_grad():
            ys = MPQLinearCuda.forward_grouped(layers, x)
            for y, l in zip(ys, layers):
                ref = l(x)
                assert y.shape == lead + (l.out_channels,)
                if lead == (40,):
                    assert torch.equal(y, ref)
                else:
                    assert_close(y.reshape(-1, l.out_channels), ref.reshape(-1, l.out_channels), orc.BF16, f"grouped layers {lead}")


@pytest.mark.parametrize("w_bit", [4, 2])
def test_unmodified_module_tree_gets_grouped_launches_after_prepare_bie_layers(w_bit):
    """a caller written against the reference's module API (q_proj(h), k_proj(h), v_proj(h), o_proj(a), gate(h2), up(h2),
    down(...)) is not changed; prepare_bie_layers() registers the sibling candidates, the first forward confirms who really shares an
    input, and from the second forward on q/k/v and gate/up each run as ONE grouped decode launch (counters), with the results of the
    layers' own launches (oracle tolerance: the K-split plan of the larger grid differs)."""
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQLinearCuda
    from bitorch_engine.layers.qlinear.nbit.cuda import mpq_layer
    from bitorch_engine.utils.model_helper import prepare_bie_layers
    H, I, gs = 512, 768, 128
    g = torch.Generator().manual_seed(23)

    def lin(K, N):
        layer = MPQLinearCuda(K, N, w_bit=w_bit, dtype=torch.bfloat16, group_size=gs, dq_group_size=32, use_gba_quant=True, asym=False)
        layer.qweight.data = torch.randint(-2 ** 31, 2 ** 31 - 1, layer.qweight.shape, generator=g, dtype=torch.int64).to(torch.int32)
        return layer

    class Attn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj, self.k_proj, self.v_proj, self.o_proj = lin(H, H), lin(H, H), lin(H, H), lin(H, H)

        def forward(self, h):
            q, k, v = self.q_proj(h), self.k_proj(h), self.v_proj(h)
            return self.o_proj(torch.tanh(q + k) * v)

    class Mlp(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.gate_proj, self.up_proj, self.down_proj = lin(H, I), lin(H, I), lin(I, H)

        def forward(self, h):
            return self.down_proj(torch.sigmoid(self.gate_proj(h)) * self.up_proj(h))

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.self_attn, self.mlp = Attn(), Mlp()

        def forward(self, h):
            h = h + self.self_attn(h)
            return h + self.mlp(h)

    model = torch.nn.Sequential(Block(), Block())
    prepare_bie_layers(model)
    for m in model.modules():
        if isinstance(m, MPQLinearCuda):
            m.scales = (torch.rand(m.scales.shape, generator=g) * 0.004 + 0.002).bfloat16() * (1 if w_bit == 4 else 5)
            m.zeros = (m.scales.float() * (2 ** w_bit - 1) / 2).bfloat16()
    model.to(DEV).eval()
    layers = [m for m in model.modules() if isinstance(m, MPQLinearCuda)]
    assert all(l._bie_group is not None for l in layers if l.in_channels == H) and all(l._bie_group is None for l in layers if l.in_channels == I)
    xs = [torch.randn((1, H), generator=g).bfloat16().to(DEV) for _ in range(4)]
    with torch.no_grad():
        saved = [l._bie_group for l in layers]
        for l in layers:
            l._bie_group = None
        refs = [model(x) for x in xs]            # every layer by itself
        for l, grp in zip(layers, saved):
            l._bie_group = grp
        mpq_layer.GROUP_STATS.update({k: 0 for k in mpq_layer.GROUP_STATS})
        y0 = model(xs[0])                          # observation round: 14 single launches
        assert mpq_layer.GROUP_STATS["grouped_launches"] == 0 and mpq_layer.GROUP_STATS["single_launches"] == 14
        assert_close(y0, refs[0], orc.BF16, "observation round")
        for x, r in zip(xs[1:], refs[1:]):
            before = dict(mpq_layer.GROUP_STATS)
            y = model(x)
            d = {k: mpq_layer.GROUP_STATS[k] - before[k] for k in before}
            # per block: q/k/v = 1 grouped launch + 2 served, gate/up = 1 grouped + 1 served, o and down alone
            assert d["grouped_launches"] == 4 and d["served_from_group"] == 6 and d["single_launches"] == 4, d
            assert_close(y, r, orc.BF16, "grouped rounds")
        assert mpq_layer.GROUP_STATS["groups_confirmed"] == 4 and mpq_layer.GROUP_STATS["groups_dissolved"] == 0
        # prefill (more rows than the grouped launch takes): untouched path, bit-equal to the layers alone
        xp = torch.randn((40, H), generator=g).bfloat16().to(DEV)
        yp = model(xp)
        for l in layers:
            l._bie_group = None
        assert torch.equal(yp, model(xp))


@pytest.mark.parametrize("tdt,odt", [(torch.float16, orc.F16), (torch.bfloat16, orc.BF16)])
def test_module_tree_groups_siblings_at_17_to_32_rows_where_the_library_takes_them(tdt, odt):
    """Round 6: a decode batch of 17 .. 32 sequences through the unchanged module tree.  bie_mpq_grouped_max_rows says which sets the grouped launch takes at that many
    rows (two row blocks per pass: fp16 always, bf16 up to 16384 output columns: profiles/r06_grouped_rb2_probe.txt); the first forward is observed even though it never had
    16 rows or fewer, the following ones run q/k/v (and gate/up where taken) as ONE launch each -- with the numbers of the members' own calls, and straight against the oracle
    for a q/k/v set at 24 and 32 rows through the C entry point."""
    from bitorch_engine import _hip
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQLinearCuda
    from bitorch_engine.layers.qlinear.nbit.cuda import mpq_layer
    from bitorch_engine.utils.model_helper import prepare_bie_layers
    from bitorch_engine.extensions import q_linear_cuda
    L = _hip.lib()
    code = _hip.F16 if tdt == torch.float16 else _hip.BF16
    assert L.bie_mpq_grouped_max_rows(4096, 12288, 4, code) == 32 and L.bie_mpq_grouped_max_rows(4096, 12288, 2, code) == 2
    assert L.bie_mpq_grouped_max_rows(4096, 22016, 4, code) == (32 if tdt == torch.float16 else 16)
    H, I, gs = 512, 9216, 128  # gate + up = 18432 columns: beyond the bf16 limit
    g = torch.Generator().manual_seed(29)

    def lin(K, N):
        layer = MPQLinearCuda(K, N, w_bit=4, dtype=tdt, group_size=gs, dq_group_size=32, use_gba_quant=True, asym=False)
        layer.qweight.data = torch.randint(-2 ** 31, 2 ** 31 - 1, layer.qweight.shape, generator=g, dtype=torch.int64).to(torch.int32)
        return layer

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj, self.k_proj, self.v_proj, self.o_proj = lin(H, H), lin(H, H), lin(H, H), lin(H, H)
            self.gate_proj, self.up_proj, self.down_proj = lin(H, I), lin(H, I), lin(I, H)

        def forward(self, h):
            q, k, v = self.q_proj(h), self.k_proj(h), self.v_proj(h)
            h = h + self.o_proj(torch.tanh(q + k) * v)
            return h + self.down_proj(torch.sigmoid(self.gate_proj(h)) * self.up_proj(h))

    model = torch.nn.Sequential(Block())
    prepare_bie_layers(model)
    for m in model.modules():
        if isinstance(m, MPQLinearCuda):
            m.scales = (torch.rand(m.scales.shape, generator=g) * 0.004 + 0.002).to(tdt)
            m.zeros = (m.scales.float() * 7.5).to(tdt)
    model.to(DEV).eval()
    layers = [m for m in model.modules() if isinstance(m, MPQLinearCuda)]
    xs = [torch.randn((M, H), generator=g).to(tdt).to(DEV) for M in (24, 24, 32, 17)]
    with torch.no_grad():
        saved = [l._bie_group for l in layers]
        for l in layers:
            l._bie_group = None
        refs = [model(x) for x in xs]
        for l, grp in zip(layers, saved):
            l._bie_group = grp
        mpq_layer.GROUP_STATS.update({k: 0 for k in mpq_layer.GROUP_STATS})
        y0 = model(xs[0])  # observed, although it has 24 rows
        assert mpq_layer.GROUP_STATS["grouped_launches"] == 0 and mpq_layer.GROUP_STATS["single_launches"] == 7
        assert_close(y0, refs[0], odt, "observation round")
        want_grouped = 2 if tdt == torch.float16 else 1  # bf16: gate/up (18432 columns) stays member by member at these row counts
        for x, r in zip(xs[1:], refs[1:]):
            before = dict(mpq_layer.GROUP_STATS)
            y = model(x)
            d = {k: mpq_layer.GROUP_STATS[k] - before[k] for k in before}
            assert d["grouped_launches"] == want_grouped and d["served_from_group"] == (3 if want_grouped == 2 else 2) and d["not_groupable"] == 0, (tuple(x.shape), d)
            assert_close(y, r, odt, f"grouped rounds at {x.shape[0]} rows")
        assert mpq_layer.GROUP_STATS["groups_dissolved"] == 0 and mpq_layer.GROUP_STATS["groups_confirmed"] == 2
        y1 = model(torch.randn((1, H), generator=g).to(tdt).to(DEV))  # and one row still groups both sets
        assert torch.isfinite(y1.float()).all()
    # the C entry point against the oracle: a q/k/v set of real width at 24 and 32 rows
    rng = np.random.default_rng(77)
    K, N = 4096, 4096
    members = [rand_case(rng, K, N, 4, 128, odt, 0) for _ in range(3)]
    for M in (24, 32):
        x = torch.randn((M, K), generator=members[0][3]).to(tdt)
        sets = [(qw.to(DEV), sc.to(DEV), ze.to(DEV), None) for (qw, sc, ze, _) in members]
        outs = q_linear_cuda.mpq_forward_grouped_impl(x.to(DEV), sets, 4, False, 128)
        for i, (qw, sc, ze, _) in enumerate(members):
            assert_close(outs[i], oracle_forward(x, qw, sc, ze, None, 4, 128, 0, odt), odt, f"grouped q/k/v member {i} at {M} rows")


def _exl2_layer(K, N, spec, gen, bias=False):
    """An MBWQLinearCuda (exl2) layer with random packed weights; returns (layer, raw qweight, q_groups list)."""
    from bitorch_engine.layers.qlinear.nbit.cuda import MBWQLinearCuda
    qg, row, kk = [], 0, 0
    for bits, k in spec:
        qg += [bits, row]
        row += k * bits // 32
        kk += k
    assert kk == K
    groups = len(spec)
    layer = MBWQLinearCuda(in_channels=K, out_channels=N, w_bit=4, dtype=torch.half, group_size=32, dq_group_size=1, use_gba_quant=True, asym=False,
                           dq_mode=2, use_mbw=True, groups=groups, rows_packed=row, disable_bias=not bias)
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (row, N), generator=gen, dtype=torch.int64).to(torch.int32)
    layer.set_qweight_data(qw.clone())
    layer.set_scales((torch.rand((groups, N), generator=gen) * 0.02 + 0.001).half())
    layer.set_zeros((torch.randn((groups, N), generator=gen) * 0.05).half())
    layer.q_perm = torch.randperm(K, generator=gen).to(torch.short)
    layer.q_groups = torch.tensor(qg, dtype=torch.short)
    if bias:
        layer.bias.data = torch.randn((N,), generator=gen).half() * 0.1
    return layer, qw, qg


def test_exl2_grouped_forward_against_the_oracle_and_the_single_launches():
    """bie_mbwq_exl2_forward_grouped through MBWQLinearCuda.forward_grouped: three mixed-bit layers with their own q_perm, band tables and
    widths (one with a bias) on ONE x -- every output against the oracle; an irregular member (groups of 96) sends the whole set to the
    single launches, same numbers."""
    from bitorch_engine.layers.qlinear.nbit.cuda import MBWQLinearCuda
    gen = torch.Generator().manual_seed(77)
    K = 1024
    specs = [[(4, 64)] * 4 + [(3, 32)] * 16 + [(2, 128)] * 2, [(3, 32)] * 16 + [(2, 32)] * 16, [(6, 32)] * 2 + [(4, 128)] * 3 + [(2, 64)] * 9]
    made = [_exl2_layer(K, N, sp, gen, bias=(i == 1)) for i, (N, sp) in enumerate(zip((328, 512, 200), specs))]
    layers = [m[0] for m in made]
    for l in layers:
        l.eval().to(DEV)
        l.prepare_params()
    x = torch.randn((1, K), generator=gen).half()
    refs = []
    for l, qw, qg in made:
        Wo = orc.exl2_dequant(qw.numpy(), orc.torch_to_np(l.scales.cpu()), orc.torch_to_np(l.zeros.cpu()), l.q_perm.cpu().numpy(), np.array(qg, np.int16), K)
        r = orc.gemm(orc.torch_to_np(x), Wo, orc.F16)
        if not l.disable_bias:
            r = orc.torch_to_np((t16(r, orc.F16).float() + l.bias.float().cpu()).half())
        refs.append(t16(r, orc.F16))
    from bitorch_engine import _hip
    with torch.no_grad():
        outs = MBWQLinearCuda.forward_grouped(layers, x.to(DEV))
        for i, (o, r) in enumerate(zip(outs, refs)):
            assert o.shape == (1, layers[i].out_channels)
            assert_close(o, r, orc.F16, f"exl2 grouped member {i}")
        again = MBWQLinearCuda.forward_grouped(layers, x.to(DEV))
        assert all(torch.equal(a, b) for a, b in zip(outs, again)) and all(a.data_ptr() != b.data_ptr() for a, b in zip(outs, again))
        singles = [l(x.to(DEV)) for l in layers]
        for i, (o, sgl) in enumerate(zip(outs, singles)):
            assert_close(o, sgl.float().cpu().half(), orc.F16, f"exl2 grouped member {i} vs its own launch")
        # two to sixteen rows of x: same two launches, every row against its own single-row result (same sums whatever the instance:
        # 1 / 2 / 4 rows with the slab of x in LDS, 8 / 16 rows with x per wave and chunk)
        for M in (2, 3, 4, 6, 8, 13, 16, 20, 33, 48):  # beyond 16: one two-launch call per slab of 16 rows
            xm = torch.randn((M, K), generator=gen).half()
            xm[0] = x[0]
            outs_m = MBWQLinearCuda.forward_grouped(layers, xm.to(DEV))
            for i, (o, l) in enumerate(zip(outs_m, layers)):
                assert o.shape == (M, l.out_channels)
                assert torch.equal(o[0], outs[i][0]), f"row 0 of member {i} at M={M} differs from the one-row call"
                rows_alone = torch.cat([MBWQLinearCuda.forward_grouped(layers, xm[r:r + 1].to(DEV))[i] for r in range(M)], 0)
                assert torch.equal(o, rows_alone), f"member {i}, M={M}: rows differ from one-row calls"
        # an irregular member: not groupable, the set falls back to one launch per member
        odd, qw, qg = _exl2_layer(K, 136, [(4, 96)] * 4 + [(2, 32)] * 20, gen)
        odd.eval().to(DEV)
        odd.prepare_params()
        assert not (odd.rows[6] & 0x200)
        mixed = MBWQLinearCuda.forward_grouped([layers[0], odd], x.to(DEV))
        assert torch.equal(mixed[0], singles[0])
        Wo = orc.exl2_dequant(qw.numpy(), orc.torch_to_np(odd.scales.cpu()), orc.torch_to_np(odd.zeros.cpu()), odd.q_perm.cpu().numpy(), np.array(qg, np.int16), K)
        assert_close(mixed[1], t16(orc.gemm(orc.torch_to_np(x), Wo, orc.F16), orc.F16), orc.F16, "irregular member alone")


@pytest.mark.parametrize("M", [1, 4, 5, 8, 16])
def test_exl2_grouped_and_list_calls_with_k_slabs_against_the_oracle(M):
    """K = 4096 with few column blocks: the plans cut K into slabs (S > 1), so the tagged-granule reduction across workgroups runs in every
    row form -- the slab of x in LDS (1 / 4 rows) and x per wave and chunk (8 / 16 rows; 5 rows run the 8-row instance with rows repeated).
    Wide and narrow band tables, ragged N, each member its own q_perm.  Grouped call and list launch against the oracle, and against each other."""
    from bitorch_engine.extensions import q_linear_cuda
    from bitorch_engine.layers.qlinear.nbit.cuda import MBWQLinearCuda, MBWQExl2ForwardList
    gen = torch.Generator().manual_seed(4096 + M)
    K = 4096
    specs = [[(4, 128)] * 8 + [(3, 32)] * 48 + [(2, 64)] * 24, [(8, 32)] * 4 + [(6, 64)] * 6 + [(5, 32)] * 16 + [(4, 64)] * 16 + [(3, 32)] * 32 + [(2, 128)] * 8]
    assert all(sum(k for _, k in sp) == K for sp in specs)
    made = [_exl2_layer(K, N, sp, gen) for N, sp in zip((200, 328), specs)]
    layers = [m[0] for m in made]
    for l in layers:
        l.eval().to(DEV)
        l.prepare_params()
        assert l.rows[6] & 0x200  # regular structure
    x = (torch.randn((M, K), generator=gen) * 0.5).half()
    refs = []
    for l, qw, qg in made:
        Wo = orc.exl2_dequant(qw.numpy(), orc.torch_to_np(l.scales.cpu()), orc.torch_to_np(l.zeros.cpu()), l.q_perm.cpu().numpy(), np.array(qg, np.int16), K)
        refs.append(t16(orc.gemm(orc.torch_to_np(x), Wo, orc.F16), orc.F16))
    members = [(l.qweight.data, l.scales, l.zeros, l.q_perm, l.q_group_map, l.rows) for l in layers]
    xd = x.to(DEV)
    outs = q_linear_cuda.mbwq_exl2_forward_grouped(xd, members)
    assert outs is not None
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert_close(o, r, orc.F16, f"grouped call, K slabs, member {i} M={M}")
    again = q_linear_cuda.mbwq_exl2_forward_grouped(xd, members)  # the generation words of the reduction advance from call to call
    assert all(torch.equal(a, b) for a, b in zip(outs, again))
    entries = [{"x": xd, "qweight": l.qweight.data, "scales": l.scales, "zeros": l.zeros, "q_perm": l.q_perm, "q_group_map": l.q_group_map, "rows": l.rows,
                "y": torch.full((M, l.out_channels), float("nan"), dtype=torch.half, device=DEV)} for l in layers]
    plan = MBWQExl2ForwardList(entries)
    for _ in range(3):
        plan()
    torch.cuda.synchronize()
    for i, (e, r) in enumerate(zip(entries, refs)):
        assert_close(e["y"], r, orc.F16, f"list launch, K slabs, entry {i} M={M}")


def test_exl2_grouped_call_under_graph_replay_and_the_limits_of_the_row_counts():
    """The grouped call (two launches, scratch from the per-stream workspace) captured in a HIP graph and replayed on NEW activations written
    into the captured input; and the row limits: a list of three or more rows needs regular groups, a group call never takes more than 16."""
    from bitorch_engine.extensions import q_linear_cuda
    from bitorch_engine.layers.qlinear.nbit.cuda import MBWQLinearCuda, MBWQExl2ForwardList
    gen = torch.Generator().manual_seed(5)
    K = 512
    made = [_exl2_layer(K, N, [(4, 64)] * 2 + [(3, 32)] * 8 + [(2, 64)] * 2, gen) for N in (192, 328)]
    layers = [m[0] for m in made]
    for l in layers:
        l.eval().to(DEV)
        l.prepare_params()
    members = [(l.qweight.data, l.scales, l.zeros, l.q_perm, l.q_group_map, l.rows) for l in layers]
    for M in (1, 4):
        xs = [torch.randn((M, K), generator=gen).half().to(DEV) for _ in range(3)]
        eager = [q_linear_cuda.mbwq_exl2_forward_grouped(x, members) for x in xs]
        xin = xs[0].clone()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            q_linear_cuda.mbwq_exl2_forward_grouped(xin, members)  # warm-up on the capture stream: its workspace exists
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            outs = q_linear_cuda.mbwq_exl2_forward_grouped(xin, members)
        for x, want in zip(xs, eager):
            xin.copy_(x)
            g.replay()
            torch.cuda.synchronize()
            for o, w in zip(outs, want):
                assert torch.equal(o, w), f"graph replay differs from the eager call (M={M})"
    odd = _exl2_layer(K, 136, [(4, 96)] * 4 + [(2, 32)] * 4, gen)[0]
    odd.eval().to(DEV)
    odd.prepare_params()
    ent = lambda l, M: {"x": torch.zeros((M, K), dtype=torch.half, device=DEV), "qweight": l.qweight.data, "scales": l.scales, "zeros": l.zeros, "q_perm": l.q_perm,
                        "q_group_map": l.q_group_map, "rows": l.rows, "y": torch.empty((M, l.out_channels), dtype=torch.half, device=DEV)}
    MBWQExl2ForwardList([ent(odd, 2)])()            # irregular groups: two rows are fine (staged form)
    with pytest.raises(RuntimeError, match="outside the one-launch decode range"):
        MBWQExl2ForwardList([ent(odd, 3)])
    with pytest.raises(RuntimeError, match="outside the one-launch decode range"):
        MBWQExl2ForwardList([ent(layers[0], 17)])
    MBWQExl2ForwardList([ent(layers[0], 16)])()
    assert q_linear_cuda.mbwq_exl2_forward_grouped(torch.zeros((q_linear_cuda.EXL2_GROUP_MAX_ROWS + 1, K), dtype=torch.half, device=DEV), members) is None
    assert q_linear_cuda.mbwq_exl2_forward_grouped(torch.zeros((1, K), dtype=torch.half, device=DEV), members + [(odd.qweight.data, odd.scales, odd.zeros, odd.q_perm, odd.q_group_map, odd.rows)]) is None
    torch.cuda.synchronize()


def test_unmodified_exl2_module_tree_gets_grouped_calls_after_prepare_bie_layers():
    """mixed-bit layers behind the reference's module API -- q_proj(h), k_proj(h), v_proj(h), o_proj(a), gate(h2), up(h2),
    down(..) -- run q/k/v and gate/up as ONE grouped call each from the second forward on (counters), with the numbers of the layers'
    own launches; a channel_scale that is not all ones keeps its layer out."""
    from bitorch_engine.layers.qlinear.nbit.cuda import MBWQLinearCuda
    from bitorch_engine.layers.qlinear.nbit.cuda import mpq_layer
    from bitorch_engine.utils.model_helper import prepare_bie_layers
    H, I = 512, 768
    g = torch.Generator().manual_seed(29)
    spec = lambda K: [(4, 64)] * 2 + [(3, 32)] * 8 + [(2, 64)] * 2 if K == 512 else [(3, 32)] * (K // 64) + [(2, 32)] * (K // 64)
    lin = lambda K, N: _exl2_layer(K, N, spec(K), g)[0]

    class Attn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj, self.k_proj, self.v_proj, self.o_proj = lin(H, H), lin(H, H), lin(H, H), lin(H, H)

        def forward(self, h):
            q, k, v = self.q_proj(h), self.k_proj(h), self.v_proj(h)
            return self.o_proj(torch.tanh(q + k) * v)

    class Mlp(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.gate_proj, self.up_proj, self.down_proj = lin(H, I), lin(H, I), lin(I, H)

        def forward(self, h):
            return self.down_proj(torch.sigmoid(self.gate_proj(h)) * self.up_proj(h))

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.self_attn, self.mlp = Attn(), Mlp()

        def forward(self, h):
            h = h + self.self_attn(h)
            return h + self.mlp(h)

    model = torch.nn.Sequential(Block(), Block())
    for m in model.modules():
        if isinstance(m, MBWQLinearCuda):  # small weights: the activations of the random net stay far from the fp16 range
            m.set_scales(m.scales * 0.2)
            m.set_zeros(m.zeros * 0.2)
    model.to(DEV).eval()
    prepare_bie_layers(model)
    layers = [m for m in model.modules() if isinstance(m, MBWQLinearCuda)]
    assert len(layers) == 14 and all(len(l.rows) == 20 for l in layers)
    assert all(l._bie_group is not None for l in layers if l.in_channels == H) and all(l._bie_group is None for l in layers if l.in_channels == I)
    xs = [torch.randn((1, H), generator=g).half().to(DEV) for _ in range(4)]
    with torch.no_grad():
        saved = [l._bie_group for l in layers]
        for l in layers:
            l._bie_group = None
        refs = [model(x) for x in xs]            # every layer by itself
        for l, grp in zip(layers, saved):
            l._bie_group = grp
        mpq_layer.GROUP_STATS.update({k: 0 for k in mpq_layer.GROUP_STATS})
        y0 = model(xs[0])                          # observation round
        assert mpq_layer.GROUP_STATS["grouped_launches"] == 0
        assert_close(y0, refs[0], orc.F16, "observation round")
        for x, r in zip(xs[1:], refs[1:]):
            before = dict(mpq_layer.GROUP_STATS)
            y = model(x)
            d = {k: mpq_layer.GROUP_STATS[k] - before[k] for k in before}
            assert d["grouped_launches"] == 4 and d["served_from_group"] == 6, d   # per block: q/k/v and gate/up
            assert_close(y, r, orc.F16, "grouped rounds")
        assert mpq_layer.GROUP_STATS["groups_confirmed"] == 4 and mpq_layer.GROUP_STATS["groups_dissolved"] == 0
        # a member whose channel_scale is not all ones does not share its input with the others: the set runs member by member, same numbers
        blk = model[0].self_attn
        blk.k_proj.channel_scale = blk.k_proj.channel_scale * 0.5
        for l in layers:
            l._bie_group = None
        r = model(xs[1])
        for l, grp in zip(layers, saved):
            l._bie_group = grp
        assert_close(model(xs[1]), r, orc.F16, "a scaled member")
        # a few rows (<= 48, slabs of sixteen): still grouped -- another kernel than the members' own at this row count (no per-weight
        # rounding): tolerance; more rows: the layers' own path, untouched
        x5, x12, x20, x52 = torch.cat(xs + xs[:1], 0), torch.cat(xs * 3, 0), torch.cat(xs * 5, 0), torch.cat(xs * 13, 0)
        before = dict(mpq_layer.GROUP_STATS)
        y5, y12, y20, y52 = model(x5), model(x12), model(x20), model(x52)
        # block 0's q/k/v set holds the scaled k_proj: its leader reports "not groupable" (counted as such, ADVICE r4: no inflated launch
        # counter) and after three such calls that set is dissolved; the other three sets group on each of the three small-row forwards
        assert mpq_layer.GROUP_STATS["grouped_launches"] - before["grouped_launches"] == 9
        assert mpq_layer.GROUP_STATS["not_groupable"] >= 3 and mpq_layer.GROUP_STATS["groups_dissolved"] == 1
        for l in layers:
            l._bie_group = None
        assert all(torch.isfinite(y.float()).all() for y in (y5, y12, y20, y52))
        assert_close(y5, model(x5).float().cpu().half(), orc.F16, "five rows, grouped against alone")
        assert_close(y12, model(x12).float().cpu().half(), orc.F16, "twelve rows, grouped against alone")
        assert_close(y20, model(x20).float().cpu().half(), orc.F16, "twenty rows (two slabs), grouped against alone")
        assert torch.equal(y52, model(x52))


@pytest.mark.parametrize("kind", ["mpq_bf16", "mpq_f16", "exl2"])
def test_sibling_projections_fed_freed_temporaries_are_never_grouped(kind):
    """a parent that gives every projection its OWN temporary -- q_proj(h * a); k_proj(h * b);
    v_proj(h.clone()) -- frees each one before the next is made, and the caching allocator places the next at the same address
    (asserted below: without the group the addresses DO repeat on this allocator).  The sibling protocol must never answer k_proj
    with W_k . (h * a): every output equals the layer's own lone launch bit for bit, on every round, and no group is confirmed.  The
    reference keeps no state between calls (layers/qlinear/nbit/cuda/mpq_layer.py:206-224)."""
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQLinearCuda, MBWQLinearCuda
    from bitorch_engine.layers.qlinear.nbit.cuda import mpq_layer
    from bitorch_engine.utils.model_helper import prepare_bie_layers
    H, gs = 512, 128
    g = torch.Generator().manual_seed(31)
    tdt = torch.bfloat16 if kind == "mpq_bf16" else torch.half

    def lin():
        if kind == "exl2":
            return _exl2_layer(H, H, [(4, 64)] * 2 + [(3, 32)] * 8 + [(2, 64)] * 2, g)[0]
        layer = MPQLinearCuda(H, H, w_bit=4, dtype=tdt, group_size=gs, dq_group_size=32, use_gba_quant=True, asym=False)
        layer.qweight.data = torch.randint(-2 ** 31, 2 ** 31 - 1, layer.qweight.shape, generator=g, dtype=torch.int64).to(torch.int32)
        return layer

    class Attn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj, self.k_proj, self.v_proj = lin(), lin(), lin()
            self.addresses = []

        def forward(self, h):
            outs = []
            for proj, f in ((self.q_proj, lambda t: t * 1.0), (self.k_proj, lambda t: t * 10.0), (self.v_proj, lambda t: t.clone())):
                t = f(h)                       # a temporary: dies before the next one is made
                self.addresses.append(t.data_ptr())
                outs.append(proj(t))
                del t
            return outs

    model = Attn()
    if kind == "exl2":
        model.to(DEV).eval()
        prepare_bie_layers(model)
    else:
        prepare_bie_layers(model)
        for m in model.children():
            m.scales = (torch.rand(m.scales.shape, generator=g) * 0.004 + 0.002).to(tdt)
            m.zeros = (m.scales.float() * 7.5).to(tdt)
        model.to(DEV).eval()
    layers = list(model.children())
    assert all(l._bie_group is not None for l in layers) and layers[0]._bie_group is layers[1]._bie_group
    hs = [torch.randn((1, H), generator=g).to(tdt).to(DEV) for _ in range(6)]
    with torch.no_grad():
        saved = [l._bie_group for l in layers]
        for l in layers:
            l._bie_group = None
        refs = [[o.clone() for o in model(h)] for h in hs]    # every layer alone
        recycled = len(set(model.addresses)) < len(model.addresses)
        for l, grp in zip(layers, saved):
            l._bie_group = grp
        mpq_layer.GROUP_STATS.update({k: 0 for k in mpq_layer.GROUP_STATS})
        for rnd, (h, ref) in enumerate(zip(hs, refs)):
            outs = model(h)
            for name, o, r in zip("qkv", outs, ref):
                assert torch.equal(o, r), f"round {rnd}: {name}_proj returned something else than its own launch on its own input"
        # k's input is 10 x q's: had k been served q's parked result the outputs above would differ by that factor
        assert float(refs[0][1].float().abs().max()) > 0
    assert recycled, "the allocator did not recycle the temporaries' block here: the test would not have caught the defect"
    assert mpq_layer.GROUP_STATS["groups_confirmed"] == 0 and mpq_layer.GROUP_STATS["grouped_launches"] == 0 and mpq_layer.GROUP_STATS["served_from_group"] == 0
    assert layers[0]._bie_group.dead     # four rounds without a shared input: the group stopped looking (and holds nothing any more)
    assert not layers[0]._bie_group.trace and not layers[0]._bie_group.parked


def _shared_x_attention(tdt, gen, H=512, gs=128):
    """q/k/v MPQ layers under one parent that all receive the SAME tensor; k_proj is called after `between(x)` ran on it."""
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQLinearCuda

    def lin():
        layer = MPQLinearCuda(H, H, w_bit=4, dtype=tdt, group_size=gs, dq_group_size=32, use_gba_quant=True, asym=False)
        layer.qweight.data = torch.randint(-2 ** 31, 2 ** 31 - 1, layer.qweight.shape, generator=gen, dtype=torch.int64).to(torch.int32)
        return layer

    class Attn(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj, self.k_proj, self.v_proj = lin(), lin(), lin()
            self.between = None

        def forward(self, x):
            q = self.q_proj(x)
            if self.between is not None:
                self.between(x)
            return q, self.k_proj(x), self.v_proj(x)

    return Attn()


def _prepare_shared_x_attention(model, tdt, gen, **kw):
    from bitorch_engine.utils.model_helper import prepare_bie_layers
    prepare_bie_layers(model, **kw)
    for m in model.children():
        m.scales = (torch.rand(m.scales.shape, generator=gen) * 0.004 + 0.002).to(tdt)
        m.zeros = (m.scales.float() * 7.5).to(tdt)
    model.to(DEV).eval()


@pytest.mark.xfail(strict=True, reason="documented contract difference (INTEGRATION.md, 'Sibling grouping holds x by identity'): a write to x through .data "
                                       "between two sibling calls is invisible to the version counter the group keys on, so the second sibling is served "
                                       "the output for the OLD contents; switch: prepare_bie_layers(model, group_siblings=False) / BIE_AUTO_GROUP=0")
def test_sibling_group_cannot_see_a_write_through_dot_data():
    """pins TODAY's behaviour.  The assertion is what a stateless caller (the reference, mpq_layer.py:206-224)
    would see -- k_proj(x) computed on the doubled x -- and it fails, strictly, while the grouping keys on identity."""
    from bitorch_engine.layers.qlinear.nbit.cuda import mpq_layer
    g = torch.Generator().manual_seed(41)
    model = _shared_x_attention(torch.bfloat16, g)
    _prepare_shared_x_attention(model, torch.bfloat16, g)
    with torch.no_grad():
        for _ in range(2):  # observe, confirm
            model(torch.randn((1, 512), generator=g).bfloat16().to(DEV))
        assert mpq_layer.GROUP_STATS["groups_confirmed"] >= 1
        x = torch.randn((1, 512), generator=g).bfloat16().to(DEV)
        model.between = lambda t: t.data.mul_(2)
        _, k, _ = model(x)                       # x now holds 2 * the values q_proj saw
        model.between = None
        for l in model.children():
            l._bie_group = None
        want = model.k_proj(x)                   # its own launch on the CURRENT contents
    assert torch.equal(k, want)


def test_group_siblings_false_is_the_reference_call_pattern_and_sees_every_write():
    """prepare_bie_layers(model, group_siblings=False): no group is attached (and groups of an earlier call are removed); every layer launches
    for itself, so the `.data` write between sibling calls is seen; the same write through a version-bumping in-place op is seen WITH groups."""
    from bitorch_engine.layers.qlinear.nbit.cuda import mpq_layer
    from bitorch_engine.utils.model_helper import prepare_bie_layers
    g = torch.Generator().manual_seed(42)
    model = _shared_x_attention(torch.bfloat16, g)
    _prepare_shared_x_attention(model, torch.bfloat16, g, group_siblings=False)
    layers = list(model.children())
    assert all(l._bie_group is None for l in layers)
    mpq_layer.GROUP_STATS.update({k: 0 for k in mpq_layer.GROUP_STATS})
    with torch.no_grad():
        for _ in range(3):
            model(torch.randn((1, 512), generator=g).bfloat16().to(DEV))
        assert mpq_layer.GROUP_STATS["groups_confirmed"] == 0 and mpq_layer.GROUP_STATS["grouped_launches"] == 0
        x = torch.randn((1, 512), generator=g).bfloat16().to(DEV)
        model.between = lambda t: t.data.mul_(2)
        _, k, _ = model(x)
        model.between = None
        assert torch.equal(k, model.k_proj(x))
        # with groups: a write autograd sees (x.mul_) changes the key -- k_proj leaves the set and runs on the new contents
        mpq_layer.find_sibling_groups(model)
        assert all(l._bie_group is not None for l in layers)
        for _ in range(2):
            model(torch.randn((1, 512), generator=g).bfloat16().to(DEV))
        assert mpq_layer.GROUP_STATS["groups_confirmed"] >= 1
        x = torch.randn((1, 512), generator=g).bfloat16().to(DEV)
        model.between = lambda t: t.mul_(2)
        _, k, v = model(x)
        model.between = None
        prepare_bie_layers(model, layers=[torch.nn.Identity], group_siblings=False)   # detaches; prepares nothing (no Identity below)
        assert all(l._bie_group is None for l in layers)
        assert torch.equal(k, model.k_proj(x)) and torch.equal(v, model.v_proj(x))


@pytest.mark.parametrize("shape", [(1024, 4096, 1024, 4096, 256), (256, 512, 384, 1024, 64), (192, 2048, 3584, 28672, 7168), (40, 512, 256, 768, 512), (4, 512, 256, 768, 0)])
def test_forward_into_a_column_range_of_a_wider_output(shape):
    """bie_mpq_forward_pitched (a column shard's GEMM epilogue stores straight into out[:, lo:hi]): the pitched result is
    bit-equal to the tight one for the dense and the fused MFMA form, the rest of the destination is untouched, and shapes outside the
    pitched range (decode kernels, split-K plans) take the tight buffer + copy path with the same result."""
    from bitorch_engine.extensions import q_linear_cuda
    M, K, N, NT, lo = shape
    rng = np.random.default_rng(M + K + N)
    qw, scales, zeros, _ = rand_case(rng, K, N, 4, 128, orc.BF16, 0)
    qd, sd, zd = qw.to(DEV), scales.to(DEV), zeros.to(DEV)
    xm = torch.randn((M, K), generator=torch.Generator().manual_seed(M)).bfloat16().to(DEV)
    g_idx = (torch.arange(K, dtype=torch.int32) // 128).to(DEV)
    tight = q_linear_cuda.mpq_forward_impl(xm, qd, sd, zd, g_idx, 4, False, 128)
    wide = torch.full((M, NT), 7.0, dtype=torch.bfloat16, device=DEV)
    view = wide[:, lo:lo + N]
    assert not view.is_contiguous() or NT == N
    got = q_linear_cuda.mpq_forward_impl(xm, qd, sd, zd, g_idx, 4, False, 128, out=view)
    assert got.data_ptr() == view.data_ptr()
    assert torch.equal(view, tight), "pitched epilogue differs from the tight one"
    keep = torch.ones(NT, dtype=torch.bool)
    keep[lo:lo + N] = False
    assert bool((wide[:, keep.to(DEV)] == 7.0).all()), "stores outside the column range"
    bias = torch.randn((N,), generator=torch.Generator().manual_seed(1)).bfloat16().to(DEV)
    tb = q_linear_cuda.mpq_forward_impl(xm, qd, sd, zd, g_idx, 4, False, 128, bias)
    q_linear_cuda.mpq_forward_impl(xm, qd, sd, zd, g_idx, 4, False, 128, bias, out=view)
    assert torch.equal(view, tb)


def test_grouped_forward_with_a_column_count_that_is_not_a_multiple_of_4():
    """N % 4 != 0 in one set: the grouped entry point must leave the 16-byte-load kernels and fall back to one launch per set."""
    from bitorch_engine.extensions import q_linear_cuda
    K, gs, dt = 512, 128, orc.BF16
    for M in (1, 3):
        sets, refs, x = [], [], None
        for i, N in enumerate((128, 66)):
            rng = np.random.default_rng(50 * i + M)
            qw, scales, zeros, gen = rand_case(rng, K, N, 4, gs, dt, 0)
            if x is None:
                x = torch.randn((M, K), generator=gen).to(TDT[dt])
            sets.append(tuple(None if t is None else t.to(DEV) for t in (qw, scales, zeros, None)))
            refs.append(oracle_forward(x, qw, scales, zeros, None, 4, gs, 0, dt))
        ys = q_linear_cuda.mpq_forward_grouped_impl(x.to(DEV), sets, 4, 0, gs)
        for y, r in zip(ys, refs):
            assert_close(y, r, dt, f"grouped N%4 M={M}")


def test_decode_gemv_wide_output_more_than_1024_tiles():
    """N > 65536 (quantised lm_head) used to overflow the 1024-counter workspace head; 2004 column tiles now fit the
    16 KiB head (fp16 -> dot2 kernel with tickets, bf16 -> lookup kernel with generation words)."""
    K, N, gs = 1024, 128256, 128
    for dt in (orc.F16, orc.BF16):
        rng = np.random.default_rng(4 + dt)
        qw, scales, zeros, gen = rand_case(rng, K, N, 4, gs, dt, 0)
        x = torch.randn((1, K), generator=gen).to(TDT[dt])
        y = hip_forward(x, qw, scales, zeros, None, 4, gs, 0)
        ref = oracle_forward(x, qw, scales, zeros, None, 4, gs, 0, dt)
        assert_close(y, ref, dt, f"N={N} dt={dt}")
        assert torch.equal(y, hip_forward(x, qw, scales, zeros, None, 4, gs, 0))


def test_binary_conv_does_not_disturb_the_gemv_workspace_head():
    """the conv scratch used to start at offset 0 of the shared per-stream workspace and clobbered the split-K
    tickets; conv -> fp16 decode GEMV (ticketed split-K) -> conv -> GEMV must keep giving the oracle's result."""
    from bitorch_engine.extensions import binary_conv_cpp
    rng = np.random.default_rng(12)
    K, N, gs = 4096, 512, 128
    qw, scales, zeros, gen = rand_case(rng, K, N, 4, gs, orc.F16, 0)
    x = torch.randn((1, K), generator=gen).half()
    ref = oracle_forward(x, qw, scales, zeros, None, 4, gs, 0, orc.F16)
    xi = torch.randn((2, 64, 7, 7), generator=gen).to(DEV)
    wi = torch.randn((32, 64, 3, 3), generator=gen).to(DEV)
    for _ in range(2):
        binary_conv_cpp.forward(xi, wi, 32, 2 * 49, 64 * 9, 3, 1, 1, 1, 7)
        assert_close(hip_forward(x, qw, scales, zeros, None, 4, gs, 0), ref, orc.F16, "GEMV after conv")


# ------------------------------------------------------------------------------------------------ backward (A8) / act-order
@pytest.mark.parametrize("dt", [orc.F16, orc.BF16])
@pytest.mark.parametrize("w_bit,asym,actorder", [(4, 0, False), (4, 1, False), (2, 0, True), (8, 0, False), (4, 0, True)])
def test_mpq_grad_input_vs_oracle(dt, w_bit, asym, actorder):
    """A8: grad_x = grad_y . W^T (bie_mpq_grad_input) against the oracle (dequantised weight, fp32 accumulate)."""
    from bitorch_engine.extensions import q_linear_cuda
    rng = np.random.default_rng(31 * w_bit + asym + 5 * dt + actorder)
    M, K, N, gs = 9, 256, 320, 64
    qw, scales, zeros, gen = rand_case(rng, K, N, w_bit, gs, dt, asym)
    g_idx = torch.arange(K, dtype=torch.int32) // gs
    if actorder:
        g_idx = g_idx[torch.randperm(K, generator=gen)]
    gy = torch.randn((M, N), generator=gen).to(TDT[dt])
    gx = q_linear_cuda.mpq_grad_input(qw.to(DEV), scales.to(DEV), zeros.to(DEV), g_idx.to(DEV), gy.to(DEV), 16, w_bit, bool(asym))
    ref = orc.mpq_grad_input(orc.torch_to_np(gy), qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros) if not asym else zeros.numpy(),
                             g_idx.numpy(), w_bit, gs, asym, dt)
    assert_close(gx, t16(ref, dt), dt, f"grad_input w{w_bit} asym={asym} actorder={actorder}")


def test_mpq_layer_backward_through_autograd():
    """MPQLinearCudaFunction.backward: grad wrt the input equals grad_y . W^T, privileged_grad = x^T . grad_y
    (reference layers/qlinear/nbit/cuda/mpq_layer.py:98-117)."""
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQLinearCuda
    K, N, gs = 256, 128, 64
    layer = MPQLinearCuda(K, N, w_bit=4, dtype=torch.half, group_size=gs, dq_group_size=32, use_gba_quant=True, asym=False)
    g = torch.Generator().manual_seed(3)
    layer.qweight.data = torch.randint(-2 ** 31, 2 ** 31 - 1, layer.qweight.shape, generator=g, dtype=torch.int64).to(torch.int32)
    layer.prepare_params()
    layer.scales = (torch.rand(layer.scales.shape, generator=g) * 0.01 + 0.005).half()
    layer.zeros = (layer.scales.float() * torch.rand(layer.scales.shape, generator=g) * 15).half()
    layer.to(DEV).train()
    x = torch.randn((2, 3, K), generator=g).half().to(DEV).requires_grad_(True)
    y = layer(x)
    gy = torch.randn(y.shape, generator=g).half().to(DEV)
    y.backward(gy)
    W = orc.mpq_dequant(layer.qweight.data.cpu().numpy(), orc.torch_to_np(layer.scales), orc.torch_to_np(layer.zeros), None, 4, gs, 0, orc.F16)
    ref = orc.gemm(orc.torch_to_np(gy.reshape(-1, N)), np.ascontiguousarray(W.T), orc.F16)
    assert_close(x.grad.reshape(-1, K), t16(ref, orc.F16), orc.F16, "autograd grad_input")
    pg = layer.qweight.privileged_grad
    if pg is not None:
        assert tuple(pg.shape) == (K, N)


@pytest.mark.parametrize("dt", [orc.F16, orc.BF16])
@pytest.mark.parametrize("M", [1, 7, 40, 300])
@pytest.mark.parametrize("w_bit", [4, 2])
def test_act_order_gidx_forward(dt, M, w_bit):
    """Explicit (randomly permuted) g_idx, the act-order form GPTQ writes: the packed rows are re-ordered once
    (bie_mpq_sort_rows) and the implicit-group kernels run on x[:, perm] -- same W values per (k, n) as the reference's
    per-weight g_idx lookup (mpq_linear_cuda_kernel.cu:300-317), only the fp32 summation order differs."""
    rng = np.random.default_rng(17 + M + dt + w_bit)
    K, N, gs = 512, 384, 64
    qw, scales, zeros, gen = rand_case(rng, K, N, w_bit, gs, dt, 0)
    g_idx = (torch.arange(K, dtype=torch.int32) // gs)[torch.randperm(K, generator=gen)]
    x = torch.randn((M, K), generator=gen).to(TDT[dt])
    y = hip_forward(x, qw, scales, zeros, g_idx, w_bit, gs, 0)
    ref = oracle_forward(x, qw, scales, zeros, g_idx, w_bit, gs, 0, dt)
    assert_close(y, ref, dt, f"act-order M={M} dt={dt} w{w_bit}")


@pytest.mark.parametrize("M", [1, 32, 33, 40, 300])
@pytest.mark.parametrize("dt,w_bit,asym", [(orc.F16, 4, 0), (orc.BF16, 4, 1), (orc.BF16, 2, 0), (orc.F16, 8, 0)])
def test_gidx_with_unequal_groups_decode_generic_and_prefill_dense_image(M, dt, w_bit, asym):
    """A g_idx that is not a permutation of k // group_size (groups of unequal size) cannot be re-ordered into an implicit-group
    matrix: generic kernel for M <= 32; for M > 32 the per-k dequantise into the MFMA fragment image + the dense kernel
    (mpq_dequant_frag_gidx_kernel; VERDICT r4 next #5: the reference's unpack_qweight + cuBLAS split, mpq_layer.py:59-63, without a
    vendor GEMM).  Same weights as the oracle's dequant bit for bit, fp32 accumulation."""
    from bitorch_engine.extensions import q_linear_cuda
    rng = np.random.default_rng(171 + M + w_bit)
    K, N, gs = 512, 384, 64
    qw, scales, zeros, gen = rand_case(rng, K, N, w_bit, gs, dt, asym)
    g_idx = torch.randint(0, K // gs, (K,), generator=gen, dtype=torch.int32)
    assert q_linear_cuda.act_order_sorted(qw.to(DEV), g_idx.to(DEV), w_bit, gs) is None
    x = torch.randn((M, K), generator=gen).to(TDT[dt])
    bias = torch.randn((N,), generator=gen).to(TDT[dt]) if M in (33, 300) else None
    y = hip_forward(x, qw, scales, zeros, g_idx, w_bit, gs, asym, bias)
    ref = oracle_forward(x, qw, scales, zeros, g_idx, w_bit, gs, asym, dt, bias)
    assert_close(y, ref, dt, f"unequal groups M={M}")


def test_no_forward_path_calls_a_vendor_gemm():
    """VERDICT r4 missing #2: no torch.matmul / mm / bmm / addmm / F.linear anywhere in the extension shims or the layers' forward code
    (the autograd BACKWARD of MBWQ keeps the reference's torch expression, mbwq_layer.py:104-113, and is not a forward path)."""
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bitorch-engine_amd", "bitorch_engine")
    pat = re.compile(r"torch\.(matmul|mm|bmm|addmm|einsum)\(|F\.linear\(|\.matmul\(|\.mm\(")
    hits = []
    for dirpath, _, files in os.walk(os.path.join(root, "extensions")):
        for f in files:
            if f.endswith(".py"):
                for ln, line in enumerate(open(os.path.join(dirpath, f)), 1):
                    if pat.search(line.split("#")[0]):
                        hits.append(f"{f}:{ln}: {line.strip()}")
    assert not hits, hits


@pytest.mark.parametrize("w_bit", [1, 2, 4, 8])
def test_sort_rows_and_gather_cols_are_exact(w_bit):
    """bie_mpq_sort_rows moves whole w-bit fields (field k' of the output = field perm[k'] of the input), bie_gather_cols moves
    whole elements: both bit-exact against numpy, and dequant(sorted)[k'] == dequant(original, g_idx)[perm[k']]."""
    from bitorch_engine.extensions import q_linear_cuda
    from bitorch_engine import _hip
    rng = np.random.default_rng(5 + w_bit)
    K, N, gs = 256, 200, 32
    nb = 32 // w_bit
    qw = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, (K // nb, N), dtype=np.int64).astype(np.int32))
    gen = torch.Generator().manual_seed(w_bit)
    g_idx = (torch.arange(K, dtype=torch.int32) // gs)[torch.randperm(K, generator=gen)]
    perm, qs = q_linear_cuda.act_order_sorted(qw.to(DEV), g_idx.to(DEV), w_bit, gs)
    perm_np = np.argsort(g_idx.numpy(), kind="stable")
    assert np.array_equal(perm.cpu().numpy(), perm_np)
    fields = (qw.numpy().view(np.uint32)[:, None, :] >> (np.arange(nb, dtype=np.uint32) * w_bit)[None, :, None]) & np.uint32((1 << w_bit) - 1)
    fields = fields.reshape(K, N)[perm_np].reshape(K // nb, nb, N)
    want = np.zeros((K // nb, N), np.uint32)
    for i in range(nb):
        want |= fields[:, i, :].astype(np.uint32) << np.uint32(i * w_bit)
    assert np.array_equal(qs.cpu().numpy().view(np.uint32), want)
    for tdt in (torch.float16, torch.float32):
        for rows in (5, 16, 300, 2500):  # >= 16 rows of two-byte elements: the LDS-staged row form (gather_rows_kernel), 2500: rows per workgroup > 1
            x = torch.randn((rows, K), generator=gen).to(tdt)
            assert torch.equal(q_linear_cuda.gather_cols(x.to(DEV), perm).cpu(), x[:, torch.from_numpy(perm_np)])
    assert q_linear_cuda.act_order_sorted(qw.to(DEV), g_idx.to(DEV), w_bit, gs) is not None


def test_full_size_act_order_4096x11008():
    """random-permuted g_idx at 4096x11008, M = 1 (lookup GEMV on the re-ordered matrix) and M = 4096 (MFMA GEMM)."""
    K, N, gs, dt = 4096, 11008, 128, orc.BF16
    rng = np.random.default_rng(77)
    qw, scales, zeros, gen = rand_case(rng, K, N, 4, gs, dt, 0)
    g_idx = (torch.arange(K, dtype=torch.int32) // gs)[torch.randperm(K, generator=gen)]
    x = torch.randn((4096, K), generator=gen).to(TDT[dt])
    y = hip_forward(x, qw, scales, zeros, g_idx, 4, gs, 0)
    rows = torch.tensor([0, 1, 2047, 4095])
    ref = oracle_forward(x[rows], qw, scales, zeros, g_idx, 4, gs, 0, dt)
    assert_close(y[rows.to(DEV)], ref, dt, "act-order 4096x11008 M=4096 sampled rows")
    y1 = hip_forward(x[:1], qw, scales, zeros, g_idx, 4, gs, 0)
    assert_close(y1, ref[:1], dt, "act-order 4096x11008 M=1")


# ------------------------------------------------------------------------------------------------ full-size cases of BASELINE configs
def _sampled_rows_check(K, N, M, w_bit, gs, dt, seed, what):
    rng = np.random.default_rng(seed)
    qw, scales, zeros, gen = rand_case(rng, K, N, w_bit, gs, dt, 0)
    x = torch.randn((M, K), generator=gen).to(TDT[dt])
    y = hip_forward(x, qw, scales, zeros, None, w_bit, gs, 0)
    rows = torch.tensor(sorted({0, 1, 31, 32, 255, M // 2, M - 2, M - 1}))
    ref = t16(orc.mpq_forward(orc.torch_to_np(x[rows]), qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), None, w_bit, gs, 0, dt), dt)
    assert_close(y[rows.to(DEV)], ref, dt, what + " sampled rows vs oracle")
    rel_err_report(y[rows.to(DEV)], ref, what + f" {'bf16' if dt == orc.BF16 else 'f16'} sampled rows")
    lo = M // 4
    y2 = hip_forward(x[lo:lo + 128], qw, scales, zeros, None, w_bit, gs, 0)  # another M tiling / split-K plan
    assert_close(y2, y[lo:lo + 128], dt, what + " row independence")
    y1 = hip_forward(x[5:6], qw, scales, zeros, None, w_bit, gs, 0)          # the decode kernel on one of the rows
    assert_close(y1, y[5:6], dt, what + " decode kernel vs prefill kernel")
    assert torch.isfinite(y.float()).all()


def test_full_size_metric_layer_4096x4096_prefill():
    """BASELINE.json's metric layer at M = 4096 (bf16)."""
    _sampled_rows_check(4096, 4096, 4096, 4, 128, orc.BF16, 501, "4096x4096 M=4096")


def test_full_size_11008x4096_prefill():
    """configs[1]'s down-projection shape at M = 4096 (bf16): sampled rows against the oracle, row independence, decode vs prefill."""
    _sampled_rows_check(11008, 4096, 4096, 4, 128, orc.BF16, 503, "11008x4096 M=4096")


def test_full_size_8192x28672_unsharded_decode_and_prefill_sampled():
    """configs[4]'s layer UN-sharded on one GPU (it is benchmarked that way): M = 1 decode on every column and M = 4096 prefill on a
    sample of rows, both against the oracle on a sample of 2048 columns (the oracle walks all K for them)."""
    K, N = 8192, 28672
    rng = np.random.default_rng(777)
    qw, scales, zeros, gen = rand_case(rng, K, N, 4, 128, orc.BF16, 0)
    cols = np.sort(rng.choice(N, 2048, replace=False))
    qs, ss, zs = np.ascontiguousarray(qw.numpy()[:, cols]), orc.torch_to_np(scales[:, cols].contiguous()), orc.torch_to_np(zeros[:, cols].contiguous())
    tc = torch.from_numpy(cols).to(DEV)
    x1 = torch.randn((1, K), generator=gen).to(torch.bfloat16)
    y1 = hip_forward(x1, qw, scales, zeros, None, 4, 128, 0)
    ref1 = t16(orc.mpq_forward(orc.torch_to_np(x1), qs, ss, zs, None, 4, 128, 0, orc.BF16), orc.BF16)
    assert_close(y1[:, tc], ref1, orc.BF16, "8192x28672 M=1")
    rel_err_report(y1[:, tc], ref1, "decode GEMV 8192x28672 M=1 bf16 (2048 sampled columns)")
    xm = torch.randn((4096, K), generator=gen).to(torch.bfloat16)
    ym = hip_forward(xm, qw, scales, zeros, None, 4, 128, 0)
    rows = torch.tensor([0, 255, 256, 2047, 4095])
    refm = t16(orc.mpq_forward(orc.torch_to_np(xm[rows]), qs, ss, zs, None, 4, 128, 0, orc.BF16), orc.BF16)
    assert_close(ym[rows.to(DEV)][:, tc], refm, orc.BF16, "8192x28672 M=4096 sampled rows / columns")
    rel_err_report(ym[rows.to(DEV)][:, tc], refm, "prefill GEMM 8192x28672 M=4096 bf16 (5 rows x 2048 columns)")
    assert torch.isfinite(ym.float()).all()


def test_full_size_c5_shard_8192x3584_prefill():
    """configs[4]: the per-GPU column shard of the 8192x28672 layer (N / 8 = 3584 columns), M = 4096, bf16."""
    _sampled_rows_check(8192, 3584, 4096, 4, 128, orc.BF16, 502, "C5 shard 8192x3584 M=4096")


@pytest.mark.parametrize("K,N", [(4096, 11008), (11008, 4096)])
def test_full_size_uniform_w2a16_llama_shapes(K, N):
    """configs[2]: uniform W2A16 at the Llama-7B shapes, decode (M = 1, 2) and a prefill sample (fp16)."""
    dt = orc.F16
    rng = np.random.default_rng(K + 7 * N)
    qw, scales, zeros, gen = rand_case(rng, K, N, 2, 128, dt, 0)
    for M in (1, 2):
        x = torch.randn((M, K), generator=gen).to(TDT[dt])
        y = hip_forward(x, qw, scales, zeros, None, 2, 128, 0)
        ref = t16(orc.mpq_forward(orc.torch_to_np(x), qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), None, 2, 128, 0, dt), dt)
        assert_close(y, ref, dt, f"W2A16 {K}x{N} M={M}")
    x = torch.randn((512, K), generator=gen).to(TDT[dt])
    y = hip_forward(x, qw, scales, zeros, None, 2, 128, 0)
    rows = torch.tensor([0, 1, 100, 511])
    ref = t16(orc.mpq_forward(orc.torch_to_np(x[rows]), qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), None, 2, 128, 0, dt), dt)
    assert_close(y[rows.to(DEV)], ref, dt, f"W2A16 {K}x{N} M=512 sampled rows")


@pytest.mark.parametrize("K,N", [(4096, 11008), (11008, 4096)])
def test_full_size_exl2_w3w2_random_perm(K, N):
    """configs[2]: exl2 mixed 3/2-bit at the Llama-7B shapes with a RANDOM q_perm: dequant bit-exact vs the oracle on a column
    slice, decode (M = 1, 2) and a small prefill against oracle GEMMs."""
    from bitorch_engine.extensions import q_linear_cuda
    from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
    qg, row = [], 0
    for b in (3, 2):
        for _ in range(K // 2 // 32):
            qg += [b, row]
            row += b
    groups = len(qg) // 2
    q_groups = torch.tensor(qg, dtype=torch.short)
    gen = torch.Generator().manual_seed(K + N)
    rng = np.random.default_rng(K + N)
    qw = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, (row, N), dtype=np.int64).astype(np.int32))
    scales = (torch.rand((groups, N), generator=gen) * 0.02 + 0.001).half()
    zeros = (torch.randn((groups, N), generator=gen) * 0.05).half()
    q_perm = torch.randperm(K, generator=gen).to(torch.short)
    gmap = make_group_map(q_groups, row)
    qs, rows = exl2_load(qw, q_groups, K, groups)
    d = lambda t: t.to(DEV)
    Wd = q_linear_cuda.mbwq_exl2fp_weight(qs, d(scales), d(zeros), d(q_perm), d(gmap), rows)
    Wo = orc.exl2_dequant(qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), q_perm.numpy(), q_groups.numpy(), K)
    assert np.array_equal(orc.torch_to_np(Wd), Wo), "exl2 dequant at full size not bit-exact"
    for M in (1, 2, 5, 16, 48, 64):
        x = torch.randn((M, K), generator=gen).half()
        y = q_linear_cuda.mbwq_exl2_forward(d(x), qs, d(scales), d(zeros), d(q_perm), d(gmap), rows, False)
        ref = t16(orc.gemm(orc.torch_to_np(x), Wo, orc.F16), orc.F16)
        assert_close(y, ref, orc.F16, f"exl2 w3/w2 {K}x{N} M={M}")
        rel_err_report(y, ref, f"exl2 w3/w2 {K}x{N} M={M}")
        if M >= 49:  # the prefill form multiplies the reference's per-weight-rounded fp16 matrix: element-wise gate.  M <= 48: the decode / matrix-pipe
            # stream kernels sum the exact products q * s - z (the reference itself accumulates in fp16 there, exl2/q_gemm_kernel.cuh): norm-wise gate
            # above, the element-wise distribution is printed by rel_err_report and quoted in DESIGN section 2
            assert_close_elementwise_f16(y, ref, f"exl2 w3/w2 {K}x{N} M={M} (fp16, element-wise)")
    # prefill at BASELINE's M = 4096 (fragment image + x[:, q_perm] + dense MFMA GEMM): sampled rows against the oracle's product with the
    # bit-exact weight matrix, every row finite
    x = torch.randn((4096, K), generator=gen).half()
    y = q_linear_cuda.mbwq_exl2_forward(d(x), qs, d(scales), d(zeros), d(q_perm), d(gmap), rows, False)
    assert bool(torch.isfinite(y.float()).all()) and y.shape == (4096, N)
    pick = [0, 1, 255, 256, 2047, 3000, 4095]
    ref = t16(orc.gemm(orc.torch_to_np(x[pick]), Wo, orc.F16), orc.F16)
    assert_close(y[pick], ref, orc.F16, f"exl2 w3/w2 {K}x{N} M=4096 (sampled rows)")
    assert_close_elementwise_f16(y[pick], ref, f"exl2 w3/w2 {K}x{N} M=4096 prefill form (fp16, element-wise)")


# ------------------------------------------------------------------------------------------------ section 8f: embedding, BMHA, checkpoints
def test_binary_embedding_forward_and_backward():
    """BinaryEmbeddingCuda: pack (HIP sign-pack kernel) + gather + HIP unpack-and-scale == sign(centred row) * mean|row|, exactly."""
    from bitorch_engine.layers.qembedding.binary import BinaryEmbeddingCuda
    torch.manual_seed(5)
    V, D = 50, 37  # D is not a multiple of 8: padded with -1 columns, truncated again on the way out
    emb = BinaryEmbeddingCuda(num_embeddings=V, embedding_dim=D, padding_idx=3)
    w = emb.weight.data.clone()
    emb.to(DEV)
    emb.eval()
    emb.prepare_params()
    assert not hasattr(emb, "weight") and emb.qweight.dtype == torch.uint8 and tuple(emb.qweight.shape) == (V, 5)
    centred = w - w.mean()
    centred[3] = 0
    padded = torch.cat([centred, -torch.ones(V, 3)], dim=1)
    assert np.array_equal(emb.qweight.data.cpu().numpy(), orc.binary_pack_rows(padded.numpy()))
    scale = padded.norm(1, 1, keepdim=True) / 40
    assert torch.allclose(emb.scale_w.cpu(), scale, rtol=1e-6, atol=0)
    idx = torch.tensor([[0, 3, 7, 49], [5, 5, 1, 2]])
    out = emb(idx.to(DEV)).cpu()
    sign = torch.where(padded >= 0, 1.0, -1.0)[:, :D]
    assert out.shape == (2, 4, D) and torch.equal(out, sign[idx] * emb.scale_w.cpu()[idx])


def test_bmha_runs_on_the_xnor_kernels():
    from bitorch_engine.layers.qmha.binary import BMHA
    torch.manual_seed(2)
    m = BMHA(128, 128, 4)
    for lin in (m.q_linear, m.k_linear, m.v_linear, m.out):
        lin.set_weight_data(torch.randn(lin.weight.shape))
    m.eval().to(DEV)
    x = torch.randn(2, 9, 128, device=DEV)
    out, scores = m(x)
    assert out.shape == (2, 9, 128) and scores.shape == (2, 4, 9, 9)
    assert torch.isfinite(out).all() and torch.allclose(scores.sum(-1), torch.ones_like(scores.sum(-1)), atol=1e-5)
    q = m.q_linear(x)  # one projection against the closed form: (sign(x) . sign(wc)^T) * scale_a * scale_w
    ints = orc.binary_linear_rowpacked(orc.binary_pack_rows(x.reshape(-1, 128).cpu().numpy()),
                                       orc.binary_pack_rows(m.q_linear.weight.data.float().cpu().numpy()), 128)
    scale = m.q_linear.scale_a.item() * m.q_linear.scale_w.item()
    assert torch.equal(q.reshape(-1, 128).cpu(), torch.from_numpy(ints.astype(np.float32)) * torch.tensor(scale, dtype=torch.float32))


def test_save_and_load_checkpoint_roundtrip(tmp_path):
    """pack_bie_layers / save_checkpoint / load_checkpoint (reference utils/model_helper.py:199-283): a model with a binary
    linear layer is packed + saved, a fresh model loads the packed weights and reproduces the outputs."""
    from bitorch_engine.layers.qlinear.binary.cuda import BinaryLinearCuda
    from bitorch_engine.utils.model_helper import save_checkpoint, load_checkpoint

    def build():
        net = torch.nn.Sequential()
        net.add_module("fc", BinaryLinearCuda(256, 64))
        return net
    torch.manual_seed(0)
    net = build()
    net.fc.set_weight_data(torch.randn(64, 256))
    net.eval().to(DEV)
    x = torch.randn(5, 256, device=DEV)
    y0 = net(x)
    path = str(tmp_path / "ckpt.pth")
    save_checkpoint(net, path, qweight_only=True)
    sd = torch.load(path)["state_dict"]
    assert "fc.qweight" in sd and sd["fc.qweight"].dtype == torch.uint8 and sd["fc.qweight"].numel() == 64 * 256 // 8
    net2 = build().eval().to(DEV)
    load_checkpoint(net2, path, qweight_only=True)
    net2.to(DEV)
    assert torch.equal(net2(x), y0)


def test_lookup_and_ticket_kernels_share_one_workspace():
    """The bf16 lookup kernel keeps monotonic generation words in the workspace head, the fp16 dot2 kernel zero-returning
    tickets: interleaved launches on one per-stream workspace must not see each other's words (they live in separate halves)."""
    K, N, gs = 4096, 2048, 128
    cases = {}
    for dt in (orc.BF16, orc.F16):
        rng = np.random.default_rng(70 + dt)
        qw, scales, zeros, gen = rand_case(rng, K, N, 4, gs, dt, 0)
        x = torch.randn((1, K), generator=gen).to(TDT[dt])
        cases[dt] = (x, qw, scales, zeros, oracle_forward(x, qw, scales, zeros, None, 4, gs, 0, dt))
    for _ in range(3):
        for dt in (orc.BF16, orc.F16, orc.F16, orc.BF16):
            x, qw, scales, zeros, ref = cases[dt]
            assert_close(hip_forward(x, qw, scales, zeros, None, 4, gs, 0), ref, dt, f"interleaved dt={dt}")


# ------------------------------------------------------------------------------------------------ A13: CUDA-layer packed images
@pytest.mark.parametrize("N,K,bmm,kind", [(64, 256, 3, "btc"), (64, 256, 2, "btc"), (64, 256, 1, "bstc"), (96, 160, 3, "bstc"), (8, 128, 3, "btc")])
def test_binary_cuda_weight_images_bit_exact(N, K, bmm, kind):
    """binary_linear_cuda.w_pack returns the reference's BTC32 / BSTC32 image byte for byte (oracle = restatement of
    binary_linear_cuda_kernel.cu:59-152,186-300,830-882), the inverse kernel restores our row-packed operand, and a forward
    on the image equals sign(x) . sign(w)^T."""
    from bitorch_engine.extensions import binary_linear_cuda
    rng = np.random.default_rng(N + K + bmm)
    w = rng.standard_normal((N, K)).astype(np.float32)
    for tw in (torch.from_numpy(w), torch.from_numpy(np.where(w >= 0, 3, -5).astype(np.int8)), torch.from_numpy(w).half()):
        img = binary_linear_cuda.w_pack(tw.to(DEV), bmm, True)
        want = orc.binary_pack_btc32(w) if kind == "btc" else orc.binary_pack_bstc32(w)
        assert img.dtype == torch.uint8 and img.numel() == N * K // 8 and img.cpu().numpy().tobytes() == want
    rows = binary_linear_cuda.image_to_rows(img, N, K, bmm)
    assert np.array_equal(rows.cpu().numpy(), orc.binary_pack_rows(w))
    x = rng.standard_normal((5, K)).astype(np.float32)
    y = binary_linear_cuda.forward(torch.from_numpy(x).to(DEV), img, bmm, True).cpu().numpy()
    assert np.array_equal(y, np.where(x >= 0, 1.0, -1.0) @ np.where(w >= 0, 1.0, -1.0).T)


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(1, 64, 256), (2, 100, 96), (4, 4096, 4096), (5, 72, 160), (8, 64, 11008), (9, 40, 2048), (16, 33, 96),
                                   (17, 260, 512), (33, 64, 128), (64, 96, 4096),
                                   # xnor_mid_kernel (M >= 5, K % 512 == 0): every (rows, column groups) tile form, ragged M and N
                                   (7, 1000, 1024), (32, 4096, 512), (64, 4096, 512), (64, 1024, 4096), (100, 257, 1536), (512, 4096, 512),
                                   (509, 300, 1024)])
def test_binary_layer_forward_in_one_launch(M, N, K, tdt):
    """bie_binary_linear_fused (M <= 64, or M <= 512 when K % 512 == 0): sign bits of (x + bias_a), XNOR-popcount, `.to(dtype) * scale_a * scale_w`, all in one
    kernel.  Exact against the oracle's integers pushed through the layer's own expression (reference
    layers/qlinear/binary/cuda/layer.py:58-63, 283), every lane mapping (M <= 4 / 8 / 16 / 32 / 64), K not a multiple of 64 or
    128, ragged N; the raw-count form equals the two-launch extension forward."""
    from bitorch_engine.extensions import binary_linear_cuda
    from bitorch_engine.extensions._binary_common import pack_rows, xnor_linear, xnor_linear_fused, fused_ok
    gen = torch.Generator().manual_seed(M * 7 + N + K)
    x = torch.randn((M, K), generator=gen).to(tdt)
    b = (torch.randn(K, generator=gen) * 0.5).to(tdt)
    w = torch.randn((N, K), generator=gen)
    sa = torch.tensor(0.7312, dtype=tdt)
    sw = torch.tensor(0.0131, dtype=tdt)
    assert fused_ok(M, N, K)
    wrows = pack_rows(w.to(DEV))
    y = xnor_linear_fused(x.to(DEV), wrows, b.to(DEV), sa.to(DEV), sw.to(DEV)).cpu()
    xb = (x + b)                                                       # the layer dtype's addition, as set_activation does it
    ints = orc.binary_linear_rowpacked(orc.binary_pack_rows(xb.float().numpy()), orc.binary_pack_rows(w.numpy()), K)
    expect = torch.from_numpy(ints.astype(np.float32)).to(tdt) * sa * sw
    assert y.dtype == tdt and torch.equal(y, expect)
    raw = xnor_linear_fused(x.to(DEV), wrows, raw_counts=True)
    assert raw.dtype == torch.float32
    assert torch.equal(raw, xnor_linear(pack_rows(x.to(DEV)), wrows, M, N, K, 0, 1.0))
    assert torch.equal(raw, binary_linear_cuda.forward(x.to(DEV), w.to(DEV), 3, True))


def test_binary_cuda_layer_uses_the_fused_forward_and_matches_the_composed_one():
    from bitorch_engine.layers.qlinear.binary.cuda import BinaryLinearCuda
    from bitorch_engine.extensions import binary_linear_cuda
    torch.manual_seed(9)
    K, N = 512, 192
    layer = BinaryLinearCuda(K, N, dtype=torch.half)
    layer.set_weight_data(torch.randn(N, K).half())
    layer.bias_a.data = (torch.randn(K) * 0.3).half()
    layer.eval().to(DEV)
    layer.generate_quantized_weight(qweight_only=True)
    calls = []
    orig = binary_linear_cuda.layer_forward
    binary_linear_cuda.layer_forward = lambda *a: calls.append(1) or orig(*a)
    try:
        for lead in ((1,), (3, 5), (80,), (600,)):                      # 600 rows: outside the fused range -> composed path
            x = torch.randn(lead + (K,)).half().to(DEV)
            with torch.no_grad():
                y = layer(x)
            xa = layer.set_activation(x).reshape(-1, K)
            comp = binary_linear_cuda.forward(xa, layer.qweight.data, layer.bmm_type.value, True)
            comp = comp.to(torch.half).view(lead + (N,)) * layer.scale_a * layer.scale_w
            assert torch.equal(y, comp)
    finally:
        binary_linear_cuda.layer_forward = orig
    assert len(calls) == 4


def test_binary_cuda_layer_checkpoint_is_the_reference_image():
    from bitorch_engine.layers.qlinear.binary.cuda import BinaryLinearCuda
    torch.manual_seed(8)
    layer = BinaryLinearCuda(256, 64)
    layer.set_weight_data(torch.randn(64, 256))
    layer.eval().to(DEV)
    layer.generate_quantized_weight(qweight_only=False)
    carriers = layer.weight.data.float().cpu().numpy()
    assert layer.qweight.data.cpu().numpy().tobytes() == orc.binary_pack_btc32(carriers)  # adaptive: K % 128 == 0, N % 8 == 0


# ------------------------------------------------------------------------------------------------ section 8f-2: W4A4 / W8A8 backward entry points
def _nibbles(p):
    """packed int8 [rows, c/2] -> signed 4-bit values [rows, c], first value in the high nibble."""
    u = p.astype(np.uint8)
    hi, lo = (u >> 4).astype(np.int32), (u & 15).astype(np.int32)
    v = np.stack([hi, lo], axis=-1).reshape(p.shape[:-1] + (p.shape[-1] * 2,))
    return np.where(v >= 8, v - 16, v)


def test_q4_backward_reference_arithmetic():
    """q4_backward: gradient quantised with scale_grad, two 4-bit GEMMs on the saved packed operands AS THEY LIE IN MEMORY
    (reference q4_linear_cutlass_kernel.cu:719-743), int32 results times scale_a / scale_w.  Exact integer arithmetic."""
    from bitorch_engine.extensions import q_linear_cutlass as qc
    g = torch.Generator().manual_seed(21)
    m, k, n = 64, 128, 192
    x = torch.randn((m, k), generator=g).half()
    w = (torch.randn((n, k), generator=g) * 0.05).half()
    gy = torch.randn((m, n), generator=g).half()
    sa, sw, sg = 0.31, 0.02, 0.4
    qa, qw = qc.q4_w_pack(x.to(DEV), sa), qc.q4_w_pack(w.to(DEV), sw)
    ga, gw = qc.q4_backward(gy.to(DEV), qa, qw, sa, sw, sg)
    pg = orc.q4_quantize_pack(orc.torch_to_np(gy), sg, orc.F16)                  # [m, n/2]
    G = _nibbles(pg)                                                             # [m, n]
    Wv = _nibbles(qw.cpu().numpy().reshape(k, n // 2))                           # buffer re-read as [k, n]
    want_a = (G @ Wv.T).astype(np.int32)
    Gt = _nibbles(np.ascontiguousarray(pg.T).reshape(n, m // 2))                 # transposed BYTES re-read as [n, m]
    Av = _nibbles(qa.cpu().numpy().reshape(k, m // 2))
    want_w = (Gt @ Av.T).astype(np.int32)
    assert ga.shape == (m, k) and gw.shape == (n, k)
    assert torch.equal(ga.cpu(), torch.from_numpy(want_a) * sa) and torch.equal(gw.cpu(), torch.from_numpy(want_w) * sw)


def test_q4_matmul_backward_and_q8_backward():
    from bitorch_engine.extensions import q_linear_cutlass as qc
    g = torch.Generator().manual_seed(22)
    B, m, n, k = 3, 64, 128, 64
    x = torch.randn((B, m, k), generator=g).half()
    y = torch.randn((B, n, k), generator=g).half()
    gy = torch.randn((B, m, n), generator=g).half()
    out, qx, qy = qc.q4_matmul(x.to(DEV), y.to(DEV), torch.tensor(0.3), torch.tensor(0.4))
    gx, gyy = qc.q4_matmul_backward(gy.to(DEV), qx, qy, 0.3, 0.4, 0.5)
    assert gx.shape == (B, m, k) and gyy.shape == (B, n, k)
    for b in range(B):
        pg = orc.q4_quantize_pack(orc.torch_to_np(gy[b]), 0.5, orc.F16)
        want_x = _nibbles(pg) @ _nibbles(qy[b].cpu().numpy().reshape(k, n // 2)).T
        want_y = _nibbles(np.ascontiguousarray(pg.T).reshape(n, m // 2)) @ _nibbles(qx[b].cpu().numpy().reshape(k, m // 2)).T
        assert torch.equal(gx[b].cpu(), torch.from_numpy(want_x.astype(np.int32)) * 0.3)
        assert torch.equal(gyy[b].cpu(), torch.from_numpy(want_y.astype(np.int32)) * 0.4)
    rng = np.random.default_rng(5)
    m, k, n = 64, 192, 128
    a = rng.integers(-128, 128, (m, k)).astype(np.int8)
    w = rng.integers(-128, 128, (n, k)).astype(np.int8)
    gq = rng.integers(-128, 128, (m, n)).astype(np.int8)
    ga, gw = qc.q8_backward(torch.from_numpy(gq).to(DEV), torch.from_numpy(a).to(DEV), torch.from_numpy(w).to(DEV))
    assert ga.dtype == torch.int32 and gw.dtype == torch.int32
    assert np.array_equal(ga.cpu().numpy(), gq.astype(np.int32) @ w.reshape(k, n).astype(np.int32).T)
    assert np.array_equal(gw.cpu().numpy(), np.ascontiguousarray(gq.T).astype(np.int32) @ a.reshape(k, m).astype(np.int32).T)
    with pytest.raises(RuntimeError):
        qc.q8_backward(torch.from_numpy(gq).to(DEV), torch.from_numpy(a).float().to(DEV), torch.from_numpy(w).to(DEV))


# ------------------------------------------------------------------------------------------------ one launch, many layers (bie_mpq_list_*)
def _list_case(specs, dt, w_bit, gs, asym, M, seed, chain=False):
    """specs: [(K, N, with_bias)].  Returns (entries on DEV, per-entry host tensors for the oracle)."""
    entries, host = [], []
    prev_y = None
    for i, (K, N, with_bias) in enumerate(specs):
        rng = np.random.default_rng(seed + 17 * i)
        qw, scales, zeros, gen = rand_case(rng, K, N, w_bit, gs, dt, asym)
        bias = (torch.randn(N, generator=gen) * 0.1).to(TDT[dt]) if with_bias else None
        if chain and i > 0:
            x_dev, x = prev_y, None
        else:
            x = torch.randn((M, K), generator=gen).to(TDT[dt])
            x_dev = x.to(DEV)
        y = torch.full((M, N), float("nan"), dtype=TDT[dt], device=DEV)
        entries.append({"x": x_dev, "qweight": qw.to(DEV), "scales": scales.to(DEV), "zeros": zeros.to(DEV),
                        "bias": None if bias is None else bias.to(DEV), "y": y, "depends_on": i - 1 if (chain and i > 0) else -1})
        host.append((x, qw, scales, zeros, bias))
        prev_y = y
    return entries, host


@pytest.mark.parametrize("dt,w_bit,gs,asym,M", [(orc.BF16, 4, 128, 0, 1), (orc.BF16, 4, 64, 1, 2), (orc.F16, 4, 128, 0, 1), (orc.F16, 4, 32, 1, 1),
                                                  (orc.BF16, 2, 128, 0, 1), (orc.F16, 2, 64, 1, 2), (orc.BF16, 4, 256, 0, 2),
                                                  # 3 <= M <= 16: the lookup / matrix-pipe kernel in list form (mpq_lutm_list_kernel)
                                                  (orc.BF16, 4, 128, 0, 3), (orc.F16, 4, 64, 1, 8), (orc.BF16, 4, 32, 0, 16), (orc.F16, 4, 256, 0, 5),
                                                  (orc.BF16, 4, 128, 1, 16),
                                                  # 17 <= M <= 32: the same kernel with two 16-row blocks of x per pass over the weights
                                                  (orc.BF16, 4, 128, 0, 17), (orc.F16, 4, 64, 1, 24), (orc.BF16, 4, 32, 1, 32), (orc.F16, 4, 256, 0, 31),
                                                  (orc.BF16, 4, 64, 0, 32)])
def test_list_forward_mixed_shapes_vs_oracle_and_single_calls(dt, w_bit, gs, asym, M):
    """bie_mpq_list_forward: entries with different K / N (ragged column tiles, N % 64 = 8, 40), bias on some, each with its own x;
    small list -> groups split over waves and K sliced over workgroups (tagged-granule reduction inside the launch).  Every entry
    against the oracle; the table holds the reference's doubly rounded values, so only the fp32 summation order differs."""
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQForwardList
    specs = [(1024, 200, True), (512, 520, False), (2048, 64, True), (768, 136, False), (1024, 1000, False)]
    if asym:  # packed qzeros: N must be a multiple of 32 / w_bit
        specs = [(K, (N + 15) // 16 * 16, b) for (K, N, b) in specs]
    entries, host = _list_case(specs, dt, w_bit, gs, asym, M, seed=4000 + w_bit + gs + M)
    plan = MPQForwardList(entries, w_bit=w_bit, group_size=gs, asym=bool(asym))
    assert plan.launches == 1
    plan()
    torch.cuda.synchronize()
    first = [e["y"].clone() for e in entries]
    for i, (e, (x, qw, scales, zeros, bias)) in enumerate(zip(entries, host)):
        ref = oracle_forward(x, qw, scales, zeros, None, w_bit, gs, asym, dt, bias)
        assert_close(e["y"], ref, dt, f"list entry {i} {specs[i]} w{w_bit} g{gs} asym={asym} M={M}")
    plan()  # same plan again: generation words advance, results identical
    torch.cuda.synchronize()
    for i, e in enumerate(entries):
        assert torch.equal(e["y"], first[i]), f"entry {i}: a second launch of the plan differs"


@pytest.mark.parametrize("M", [1, 8])
def test_list_forward_big_list_whole_k_per_workgroup_and_graph_replay(M):
    """A list big enough that every workgroup owns a column tile's whole K range (S = 1, several units per wave, the next unit's rows
    in flight): 48 layers of 4096 -> 1024 (W4 g128 bf16), one row (FMA form) and eight (matrix-pipe form).  Captured in a HIP graph
    and replayed with NEW activations: the plan freezes pointers, not contents."""
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQForwardList
    specs = [(4096, 1024, i % 3 == 0) for i in range(48)]
    entries, host = _list_case(specs, orc.BF16, 4, 128, 0, M, seed=5100)
    plan = MPQForwardList(entries, w_bit=4, group_size=128)
    plan()
    torch.cuda.synchronize()
    for i in (0, 7, 23, 47):
        x, qw, scales, zeros, bias = host[i]
        assert_close(entries[i]["y"], oracle_forward(x, qw, scales, zeros, None, 4, 128, 0, orc.BF16, bias), orc.BF16, f"big list entry {i}")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        plan()
    gen = torch.Generator().manual_seed(5)
    for rep in range(3):
        newx = [torch.randn((M, 4096), generator=gen).to(torch.bfloat16) for _ in entries]
        for e, nx in zip(entries, newx):
            e["x"].copy_(nx.to(DEV))
        g.replay()
        torch.cuda.synchronize()
        for i in (3, 31):
            _, qw, scales, zeros, bias = host[i]
            assert_close(entries[i]["y"], oracle_forward(newx[i], qw, scales, zeros, None, 4, 128, 0, orc.BF16, bias), orc.BF16, f"replay {rep} entry {i}")


_XS_PLANS = {  # knobs of list_xs_plan (mpq_list.hip), re-read per plan under BIE_TUNING (tests/conftest.py sets it)
    "whole_k": {"BIE_LUTM_XS_WANT_WAVES": "1", "BIE_LUTM_XS_MAX_GPW": "48"},                        # every workgroup walks its tiles' whole K
    "sliced": {"BIE_LUTM_XS_WANT_WAVES": "1", "BIE_LUTM_XS_MAX_GPW": "3", "BIE_LUTM_XS_MAX_S": "64"},  # K in slices of <= 3 units: tagged granules
    "split_groups": {"BIE_LUTM_XS_MAX_S": "64"},                                                    # small list: groups split into 2 / 4 units, one unit per workgroup
}


@pytest.mark.parametrize("plan_kind", ["whole_k", "sliced", "split_groups"])
@pytest.mark.parametrize("dt,gs,asym,M", [(orc.F16, 128, 0, 3), (orc.BF16, 128, 0, 16), (orc.F16, 64, 1, 8), (orc.BF16, 32, 0, 12), (orc.F16, 256, 0, 16),
                                           (orc.BF16, 256, 1, 5), (orc.F16, 128, 0, 17), (orc.BF16, 128, 0, 32), (orc.F16, 32, 1, 25), (orc.BF16, 64, 1, 24),
                                           (orc.F16, 256, 1, 32), (orc.BF16, 256, 0, 19), (orc.F16, 128, 0, 1), (orc.F16, 64, 1, 2), (orc.BF16, 128, 1, 1)])
def test_list_forward_x_sharing_form_vs_oracle(plan_kind, dt, gs, asym, M, monkeypatch):
    """The x-sharing matrix-pipe list form (lutm_xs_body: four column tiles per workgroup, x staged once per workgroup through LDS; fp16
    dequantises by packed-fp16 arithmetic, bf16 by table) forced onto a small mixed list -- ragged quads (1, 3, 4, 9, 16 tiles), N % 64 = 8 /
    40, bias on some -- in three plans: whole K per workgroup, K sliced over workgroups (granule reduction), groups split into units.  Every
    entry against the oracle, a second launch bit-identical; bie_mpq_list_form says which kernel ran."""
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQForwardList
    for k, v in _XS_PLANS[plan_kind].items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("BIE_LUTM_XS_MIN_M_BF16", "1")
    monkeypatch.setenv("BIE_LUTM_XS_BF16_WHOLE_PCT", "0")  # bf16 by default takes the form only when most weights sit in unsliced layers
    specs = [(1024, 200, True), (512, 520, False), (2048, 64, True), (768, 136, False), (1024, 1000, False)]
    if gs == 256:
        specs = [(K if K % 256 == 0 else 1024, N, b) for (K, N, b) in specs]
    if asym:
        specs = [(K, (N + 15) // 16 * 16, b) for (K, N, b) in specs]
    entries, host = _list_case(specs, dt, 4, gs, asym, M, seed=8800 + gs + M)
    plan = MPQForwardList(entries, w_bit=4, group_size=gs, asym=bool(asym))
    assert plan.form == 2, f"the x-sharing form was not chosen ({plan_kind}: form {plan.form})"
    assert plan.launches == 1
    plan()
    torch.cuda.synchronize()
    first = [e["y"].clone() for e in entries]
    for i, (e, (x, qw, scales, zeros, bias)) in enumerate(zip(entries, host)):
        ref = oracle_forward(x, qw, scales, zeros, None, 4, gs, asym, dt, bias)
        assert_close(e["y"], ref, dt, f"x-sharing list ({plan_kind}) entry {i} {specs[i]} g{gs} asym={asym} M={M}")
    plan()
    torch.cuda.synchronize()
    for i, e in enumerate(entries):
        assert torch.equal(e["y"], first[i]), f"entry {i}: a second launch of the plan differs"
    # the same list in the k-split form (one row: the lookup + FMA kernel), the default for a list this small: same numbers up to the fp32 summation order
    monkeypatch.setenv("BIE_LUTM_XS_MIN_M", "0")
    entries2, _ = _list_case(specs, dt, 4, gs, asym, M, seed=8800 + gs + M)
    plan2 = MPQForwardList(entries2, w_bit=4, group_size=gs, asym=bool(asym))
    assert plan2.form == (0 if M == 1 else 1)
    plan2()
    torch.cuda.synchronize()
    for i, (e, e2) in enumerate(zip(entries, entries2)):
        assert_close(e["y"], e2["y"], dt, f"x-sharing against k-split, entry {i}")


def test_list_forward_x_sharing_form_is_the_default_for_big_lists_and_fails_loudly(monkeypatch):
    """Defaults: 80 layers of 4096 -> 4096 (a workgroup walks the whole K) take the x-sharing form in fp16 from ONE row and in bf16 from 12 rows
    (below: the k-split form, at one row the lookup + FMA kernel);
    a small list never does.  With K sliced, a reducer that never sees its partial sums (forged tag) returns NaN and raises the status bit."""
    from bitorch_engine import _hip
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQForwardList
    specs = [(4096, 4096, i % 5 == 0) for i in range(80)]
    for dt, M, want in ((orc.F16, 3, 2), (orc.BF16, 8, 1), (orc.BF16, 12, 2), (orc.F16, 1, 2), (orc.BF16, 1, 0)):
        entries, host = _list_case(specs, dt, 4, 128, 0, M, seed=9100 + M)
        plan = MPQForwardList(entries, w_bit=4, group_size=128)
        assert plan.form == want, f"dtype {dt} M {M}: form {plan.form}, expected {want}"
        plan()
        torch.cuda.synchronize()
        for i in (0, 13, 79):
            x, qw, scales, zeros, bias = host[i]
            assert_close(entries[i]["y"], oracle_forward(x, qw, scales, zeros, None, 4, 128, 0, dt, bias), dt, f"default plan, entry {i}, M={M}")
    small, _ = _list_case(specs[:2], orc.F16, 4, 128, 0, 8, seed=9200)
    assert MPQForwardList(small, w_bit=4, group_size=128).form == 1
    for k, v in _XS_PLANS["sliced"].items():
        monkeypatch.setenv(k, v)
    L = _hip.lib()
    entries, host = _list_case([(2048, 256, False)], orc.F16, 4, 128, 0, 8, seed=9300)
    plan = MPQForwardList(entries, w_bit=4, group_size=128)
    assert plan.form == 2
    assert L.bie_device_status(1) == 0
    L.bie_test_forge_reducer(0x5a, 2000)
    try:
        plan()
        torch.cuda.synchronize()
    finally:
        L.bie_test_forge_reducer(0, 0)
    assert torch.isnan(entries[0]["y"].float()).all(), "a timed-out reducer returned numbers"
    assert L.bie_device_status(0) & 1
    with pytest.raises(RuntimeError, match="device status"):
        plan()
    assert L.bie_device_status(0) == 0
    plan()
    torch.cuda.synchronize()
    x, qw, scales, zeros, bias = host[0]
    assert_close(entries[0]["y"], oracle_forward(x, qw, scales, zeros, None, 4, 128, 0, orc.F16, bias), orc.F16, "after the forged timeout")


@pytest.mark.parametrize("dt,gs,asym,M", [(orc.BF16, 128, 0, 33), (orc.F16, 64, 1, 40), (orc.BF16, 128, 0, 48), (orc.BF16, 32, 0, 64)])
def test_list_forward_row_blocks_for_33_to_64_rows(dt, gs, asym, M):
    """32 < M <= 64: MPQForwardList cuts the rows into two balanced blocks of <= 32, one single-launch plan per block over the row slices of
    every x / y; every entry against the oracle."""
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQForwardList
    specs = [(1024, 208, True), (512, 528, False), (2048, 64, True)]
    entries, host = _list_case(specs, dt, 4, gs, asym, M, seed=7100 + M)
    plan = MPQForwardList(entries, w_bit=4, group_size=gs, asym=bool(asym))
    assert plan.launches == 2 and plan.M == M
    plan()
    torch.cuda.synchronize()
    for i, (e, (x, qw, scales, zeros, bias)) in enumerate(zip(entries, host)):
        assert_close(e["y"], oracle_forward(x, qw, scales, zeros, None, 4, gs, asym, dt, bias), dt, f"row-blocked list entry {i} M={M}")
    plan()  # again: identical
    torch.cuda.synchronize()
    for i, (e, (x, qw, scales, zeros, bias)) in enumerate(zip(entries, host)):
        assert_close(e["y"], oracle_forward(x, qw, scales, zeros, None, 4, gs, asym, dt, bias), dt, f"row-blocked list entry {i} M={M}, second launch")
    centries, _ = _list_case([(1024, 512, True), (512, 1024, False)], dt, 4, gs, asym, M, seed=7200 + M, chain=True)
    with pytest.raises(RuntimeError):  # dependent entries are an M <= 2 feature of the list kernels
        MPQForwardList(centries, w_bit=4, group_size=gs, asym=bool(asym))
    with pytest.raises(RuntimeError):
        MPQForwardList(_list_case(specs[:1], dt, 4, gs, asym, 65, seed=1)[0], w_bit=4, group_size=gs, asym=bool(asym))


@pytest.mark.parametrize("dt,M", [(orc.BF16, 1), (orc.F16, 2)])
def test_list_forward_dependent_chain_equals_layer_by_layer(dt, M):
    """y of entry l is x of entry l+1 (depends_on): one launch (+ the memset node of the completion counters) must give exactly
    what layer-by-layer launches of the same kernels give, and match the oracle chain layer by layer (each layer checked on
    the HIP input it actually saw, so the tolerance does not compound)."""
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQForwardList
    specs = [(1024, 2048, True), (2048, 512, False), (512, 1024, True), (1024, 1024, False), (1024, 384, False)]
    entries, host = _list_case(specs, dt, 4, 128, 0, M, seed=6200 + M, chain=True)
    plan = MPQForwardList(entries, w_bit=4, group_size=128)
    assert plan.launches == 2
    gen = torch.Generator().manual_seed(9)
    for rep in range(3):  # several launches: the counters are re-armed every time; NEW input every time (stale activations would show)
        for e in entries:
            e["y"].fill_(float("nan"))
        x = torch.randn(host[0][0].shape, generator=gen).to(TDT[dt])
        entries[0]["x"].copy_(x.to(DEV))
        plan()
        torch.cuda.synchronize()
        for i, (e, (_, qw, scales, zeros, bias)) in enumerate(zip(entries, host)):
            xin = x if i == 0 else entries[i - 1]["y"].cpu()
            ref = oracle_forward(xin, qw, scales, zeros, None, 4, 128, 0, dt, bias)
            assert_close(e["y"], ref, dt, f"chain rep {rep} layer {i}")
            single = hip_forward(xin, qw, scales, zeros, None, 4, 128, 0, bias)
            assert_close(e["y"], single, dt, f"chain rep {rep} layer {i} vs the single-layer launch")


def assert_close_elementwise_f16(y, ref, what, floor=2.0 ** -6):
    """north_star's "within 1e-3 rel for the fp16 accumulate" ELEMENT by element: |y - ref| <= 1e-3 |ref| + one fp16 ulp
    for every output with |ref| >= 2^-6 max|ref| (smaller references are sums with cancellation: their relative error is unbounded by
    construction; those stay under the norm-wise gate of assert_close)."""
    y, ref = to_f32(y).ravel(), to_f32(ref).ravel()
    keep = np.abs(ref) >= np.abs(ref).max() * floor
    err = np.abs(y[keep] - ref[keep])
    tol = 1e-3 * np.abs(ref[keep]) + 2.0 ** -10 * np.abs(ref[keep])
    bad = ~(err <= tol)
    assert not bad.any(), f"{what}: {int(bad.sum())} of {int(keep.sum())} outputs outside 1e-3 relative (+1 ulp); worst {float((err / np.abs(ref[keep])).max()):.3e}"


@pytest.mark.parametrize("K,N,nl,w_bit,tdt", [(4096, 4096, 96, 4, "bf16"), (4096, 11008, 40, 4, "bf16"), (11008, 4096, 40, 4, "bf16"), (8192, 28672, 6, 4, "bf16"), (4096, 4096, 96, 2, "bf16"),
                                              (4096, 4096, 96, 4, "f16"), (4096, 11008, 40, 4, "f16"), (11008, 4096, 40, 4, "f16"), (4096, 4096, 96, 2, "f16")])
def test_the_list_instances_bench_py_times_against_the_oracle(K, N, nl, w_bit, tdt):
    """the TIMED kernel instances at the bench's own shapes -- the headline's 96 x 4096x4096 list, configs[1]'s 40-layer lists,
    configs[4]'s six 8192x28672 layers and the W2A16 list -- built by bench.py's own helper (so: the same plan, the same kernel instance,
    every tile a whole K per workgroup), every entry with its own x, ONE launch.  Sampled layers are checked in full against the oracle's
    forward (orc_mpq_dequant + orc_gemm: the restatement of the reference's CPU path), plus every layer's output must be finite and
    non-zero."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    dev = torch.device(DEV)
    B = bench.Bench(dev)
    gen = torch.Generator(device=dev).manual_seed(4242 + K + N + w_bit)
    dt, ttype = (orc.BF16, torch.bfloat16) if tdt == "bf16" else (orc.F16, torch.float16)
    layers = [bench.make_layer(dev, gen, K, N, w_bit, ttype) for _ in range(nl)]
    y_all = torch.full((nl, N), float("nan"), dtype=ttype, device=dev)
    plan = B.make_list(layers, K, N, gen, w_bit=w_bit, ys=[y_all[i:i + 1] for i in range(nl)])
    assert plan.launches == 1
    plan.forward()
    torch.cuda.synchronize()
    assert torch.isfinite(y_all.float()).all() and (y_all.float().abs().amax(dim=1) > 0).all()
    xs = bench.plan_x(plan)
    for i in sorted({0, 1, nl // 2, nl - 1}):
        qw, sc, ze = (t.cpu() for t in layers[i])
        ref = oracle_forward(xs[i].cpu(), qw, sc, ze, None, w_bit, 128, 0, dt)
        assert_close(y_all[i:i + 1], ref, dt, f"bench list instance {nl} x {K}x{N} w{w_bit} {tdt}, layer {i}")
        if tdt == "f16":  # the DEFAULT fp16 lists (exact 16-bit-table form; the algebraic form is opt-in and has its own test below): north_star's 1e-3 element by element
            rel_err_report(y_all[i:i + 1], ref, f"f16 list {nl} x {K}x{N} w{w_bit} layer {i} (default exact-table form)")
            assert_close_elementwise_f16(y_all[i:i + 1], ref, f"f16 list {nl} x {K}x{N} w{w_bit}, layer {i}")


@pytest.mark.parametrize("w_bit,asym,M,gs", [(4, 0, 1, 128), (4, 1, 1, 64), (2, 0, 2, 128), (2, 1, 1, 64), (4, 0, 1, 32), (2, 0, 1, 256)])
def test_list_forward_algebraic_form_fp16_opt_in(w_bit, asym, M, gs, monkeypatch):
    """The opt-in ALGEBRAIC list form (BIE_LIST_ALG, mpq_list.hip template bit 12; =2 to cover W2 at one row too): fp16, sum_k (c_k + q_k) x_k on
    v_dot2_f32_f16, the group's scale / zero applied once per unit in fp32 -- no per-weight fp16 rounding.  Against the oracle (the
    reference's doubly rounded weights) norm-wise, north_star's 1e-3; against the EXACT real-number product it must be tighter than the
    reference's own arithmetic is (that is the point of the form, and what keeps a wiring mistake from hiding inside the tolerance)."""
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQForwardList
    monkeypatch.setenv("BIE_TUNING", "1")
    monkeypatch.setenv("BIE_LIST_ALG", "2")
    specs = [(1024, 200, True), (512, 520, False), (2048, 64, True), (768, 136, False), (4096, 1000, False)]
    if asym:
        specs = [(K, (N + 15) // 16 * 16, b) for (K, N, b) in specs]
    entries, host = _list_case(specs, orc.F16, w_bit, gs, asym, M, seed=5000 + w_bit + gs + M)
    plan = MPQForwardList(entries, w_bit=w_bit, group_size=gs, asym=bool(asym))
    plan()
    torch.cuda.synchronize()
    for i, (e, (x, qw, scales, zeros, bias)) in enumerate(zip(entries, host)):
        ref = oracle_forward(x, qw, scales, zeros, None, w_bit, gs, asym, orc.F16, bias)
        assert_close(e["y"], ref, orc.F16, f"algebraic list entry {i} {specs[i]} w{w_bit} g{gs} asym={asym} M={M}")
        # the exact product in float64: s * (q - zq1) or s * q - z with the fp16 constants taken as real numbers
        K, N = qw.shape[0] * 32 // w_bit, qw.shape[1]
        fields = ((qw.numpy().view(np.uint32)[:, None, :] >> (np.arange(32 // w_bit, dtype=np.uint32) * w_bit)[None, :, None]) & np.uint32(2 ** w_bit - 1)).reshape(K, N).astype(np.float64)
        sc = np.repeat(scales.double().numpy(), gs, axis=0)[:K]
        if asym:
            zq = ((zeros.numpy().view(np.uint32)[:, :, None] >> (np.arange(32 // w_bit, dtype=np.uint32) * w_bit)[None, None, :]) & np.uint32(2 ** w_bit - 1)).reshape(zeros.shape[0], -1)[:, :N]
            W = sc * (fields - (np.repeat(zq.astype(np.float64), gs, axis=0)[:K] + 1.0))
        else:
            W = sc * fields - np.repeat(zeros.double().numpy(), gs, axis=0)[:K]
        exact = x.double().numpy() @ W + (0 if bias is None else bias.double().numpy())
        got = e["y"].double().cpu().numpy()
        err = np.abs(got - exact).max() / np.abs(exact).max()
        assert err <= 2.0 ** -10, f"algebraic form vs the exact product: {err:.3e} of max|y| (one output ulp is 4.9e-4 at most)"


def _same_nonfinite_and_close(y, ref, dt, what):
    """NaN where the oracle has NaN, the same infinity where it has one, assert_close on the rest."""
    yf, rf = to_f32(y), to_f32(ref)
    assert np.array_equal(np.isnan(yf), np.isnan(rf)), f"{what}: NaN pattern differs ({int(np.isnan(yf).sum())} vs {int(np.isnan(rf).sum())} NaNs)"
    inf = np.isinf(rf)
    assert np.array_equal(np.isinf(yf), inf) and np.array_equal(yf[inf], rf[inf]), f"{what}: infinities differ"
    fin = np.isfinite(rf)
    if fin.any():
        assert_close(torch.from_numpy(np.where(fin, yf, 0.0)), torch.from_numpy(np.where(fin, rf, 0.0)), dt, what)


@pytest.mark.parametrize("dt", [orc.BF16, orc.F16])
@pytest.mark.parametrize("M,path", [(1, "lone"), (1, "list"), (2, "list"), (8, "lone"), (8, "list"), (40, "lone"), (1100, "lone")])
def test_special_values_propagate_like_the_oracle(dt, M, path):
    """NaN / Inf activations, a zero scale, an infinite scale, denormal zeros and all-zero weights through the
    table-lookup decode kernels (lone launch and list form), the lookup / matrix-pipe kernel (M = 8), the fused MFMA GEMM (M = 40)
    and the dequantise-once dense form (M = 1100).  The tables are built from s and z with the reference's expressions, so an
    infinite scale gives NaN for q = 0 (0 * inf) and infinities elsewhere -- exactly as orc_mpq_dequant does; a NaN in x poisons its
    row, an infinity gives inf or (inf - inf) NaN.  Non-finite outputs must MATCH the oracle's class, finite ones its value."""
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQForwardList
    K, N, gs = 1024, 256, 128
    rng = np.random.default_rng(900 + M)
    qw, scales, zeros, gen = rand_case(rng, K, N, 4, gs, dt, 0)
    tdt = TDT[dt]
    scales, zeros = scales.clone(), zeros.clone()
    scales[1, 5] = 0.0                                   # a zero scale: the column's group contributes -z only
    scales[2, 9] = float("inf")                          # an infinite scale: NaN (q = 0) / inf weights in that group and column
    zeros[3, 17] = torch.tensor(2.0 ** -133 if dt == orc.BF16 else 2.0 ** -24).to(tdt)   # denormal in the storage type
    zeros[0, 33] = 0.0
    qw[:, 64:96] = 0                                     # all-zero packed weights in 32 columns
    x = torch.randn((M, K), generator=gen).to(tdt)
    x[0, 3] = float("nan") if M > 1 else x[0, 3]         # row 0 poisoned (only with more than one row: M = 1 keeps finite outputs to compare)
    x[M - 1, 700] = float("inf")                         # an infinity in the last row (group 5)
    ref = oracle_forward(x, qw, scales, zeros, None, 4, gs, 0, dt)
    if path == "lone":
        y = hip_forward(x, qw, scales, zeros, None, 4, gs, 0)
    else:
        d = lambda t: t.to(DEV)
        ent = [{"x": d(x), "qweight": d(qw), "scales": d(scales), "zeros": d(zeros), "y": torch.zeros((M, N), dtype=tdt, device=DEV), "depends_on": -1}]
        MPQForwardList(ent, w_bit=4, group_size=gs, asym=False)()
        torch.cuda.synchronize()
        y = ent[0]["y"]
    _same_nonfinite_and_close(y, ref, dt, f"special values M={M} {path}")
    assert np.isinf(to_f32(ref)).any() or np.isnan(to_f32(ref)).any()   # the case really exercises non-finite outputs


@pytest.mark.parametrize("asym", [0, 1])
def test_lone_decode_launches_of_100_mb_and_more_take_the_inline_list_form(asym):
    """bie_mpq_forward / bie_mpq_forward_grouped at M = 1, W4, bf16 with >= 96 MB of packed weights per launch run the list kernel's D16
    form with the entries in the kernel arguments (mpq_list_inline_launch: block -> (entry, tile, slice) by arithmetic, K sliced over
    workgroups, granules in the caller's workspace): configs[4]'s 8192x28672 layer as one launch, and two 8192x14336 sets sharing x
    as one grouped launch, against the oracle."""
    from bitorch_engine.extensions import q_linear_cuda
    K, gs, dt = 8192, 128, orc.BF16
    rng = np.random.default_rng(77 + asym)
    qw, scales, zeros, gen = rand_case(rng, K, 28672, 4, gs, dt, asym)
    x = torch.randn((1, K), generator=gen).to(TDT[dt])
    y = hip_forward(x, qw, scales, zeros, None, 4, gs, asym)
    assert_close(y, oracle_forward(x, qw, scales, zeros, None, 4, gs, asym, dt), dt, f"inline list form, lone 8192x28672 asym={asym}")
    y2 = hip_forward(x, qw, scales, zeros, None, 4, gs, asym)  # second launch into the same workspace: next granule generation
    assert torch.equal(y, y2)
    sets, refs = [], []
    for i in range(2):
        q2, s2, z2, _ = rand_case(np.random.default_rng(900 + i + asym), K, 14336, 4, gs, dt, asym)
        sets.append(tuple(t.to(DEV) for t in (q2, s2, z2)) + (None,))
        refs.append(oracle_forward(x, q2, s2, z2, None, 4, gs, asym, dt))
    for yg, r in zip(q_linear_cuda.mpq_forward_grouped_impl(x.to(DEV), sets, 4, asym, gs), refs):
        assert_close(yg, r, dt, f"inline list form, grouped 2 x 8192x14336 asym={asym}")


def test_plain_graph_capture_after_a_warm_up_on_another_stream():
    """`with torch.cuda.graph(g):` captures on torch's own capture stream, which has no scratch buffer even after a warm-up on
    the current stream; the first MPQ call inside the capture used to raise.  Now it borrows the device's warmed buffer (bitorch_engine._hip._grow),
    allocates nothing under capture, and the replay computes what the eager call computes -- also after the inputs change in place."""
    from bitorch_engine import _hip
    from bitorch_engine.extensions import q_linear_cuda
    K, N, gs, dt = 2048, 512, 128, orc.BF16   # small: K sliced over workgroups -> the launch uses the workspace's granules and generation words
    rng = np.random.default_rng(321)
    qw, scales, zeros, gen = rand_case(rng, K, N, 4, gs, dt, 0)
    d = lambda t: t.to(DEV)
    qw_d, sc_d, ze_d = d(qw), d(scales), d(zeros)
    x_static = torch.randn((1, K), generator=gen).to(TDT[dt]).to(DEV)
    y_eager = q_linear_cuda.mpq_forward_impl(x_static, qw_d, sc_d, ze_d, None, 4, 0, gs)   # warm-up: allocates the current stream's workspace
    torch.cuda.synchronize()
    n_buffers = len(_hip._WS)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y_static = q_linear_cuda.mpq_forward_impl(x_static, qw_d, sc_d, ze_d, None, 4, 0, gs)
    assert len(_hip._WS) == n_buffers, "a workspace was allocated under capture"
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y_static, y_eager)
    x2 = torch.randn((1, K), generator=gen).to(TDT[dt])
    x_static.copy_(x2.to(DEV))
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    assert_close(y_static, oracle_forward(x2, qw, scales, zeros, None, 4, gs, 0, dt), dt, "graph replay on a borrowed workspace")
    # the buffer MOVED to the capture -- the warming stream has none now and allocates its own at its next eager call, so a
    # replaying graph and eager calls (unordered streams) never share generation words, granules or scratch
    cur = _hip._key(torch.device(DEV))
    assert cur not in _hip._WS, "the capture borrowed a live stream's workspace instead of taking it over"
    graph_bufs = {b.data_ptr() for b in _hip._WS.values()}
    side = torch.cuda.Stream()
    x3 = torch.randn((1, K), generator=gen).to(TDT[dt])
    x3_d = x3.to(DEV)
    ref2, ref3 = oracle_forward(x2, qw, scales, zeros, None, 4, gs, 0, dt), oracle_forward(x3, qw, scales, zeros, None, 4, gs, 0, dt)
    torch.cuda.synchronize()
    for _ in range(20):  # graph replays on a side stream, eager calls on the current one, no ordering between them
        with torch.cuda.stream(side):
            g.replay()
        y3 = q_linear_cuda.mpq_forward_impl(x3_d, qw_d, sc_d, ze_d, None, 4, 0, gs)
    torch.cuda.synchronize()
    assert _hip._WS[cur].data_ptr() not in graph_bufs
    assert_close(y_static, ref2, dt, "graph replay beside eager calls")
    assert_close(y3, ref3, dt, "eager calls beside a replaying graph")


def test_reducer_timeout_fails_loudly():
    """A reducer whose partial sums never arrive (forged: it is told to expect another tag) must not return a number: NaN in y,
    a bit in the status page, and the NEXT launching call fails with BIE_ERR_DEVICE (then the page is clear again)."""
    from bitorch_engine import _hip
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQForwardList
    L = _hip.lib()
    entries, host = _list_case([(2048, 256, False)], orc.BF16, 4, 128, 0, 1, seed=77)  # small: K sliced over workgroups
    plan = MPQForwardList(entries, w_bit=4, group_size=128)
    assert L.bie_device_status(1) == 0
    L.bie_test_forge_reducer(0x5a, 2000)
    try:
        plan()
        torch.cuda.synchronize()
    finally:
        L.bie_test_forge_reducer(0, 0)
    assert torch.isnan(entries[0]["y"].float()).all(), "a timed-out reducer returned numbers"
    assert L.bie_device_status(0) & 1
    with pytest.raises(RuntimeError, match="device status"):
        plan()
    assert L.bie_device_status(0) == 0
    plan()  # healthy again
    torch.cuda.synchronize()
    x, qw, scales, zeros, bias = host[0]
    assert_close(entries[0]["y"], oracle_forward(x, qw, scales, zeros, None, 4, 128, 0, orc.BF16, bias), orc.BF16, "after the forged timeout")
    # the single-layer lookup GEMV and the exl2 decode carry the same reducer: same behaviour
    L.bie_test_forge_reducer(0x33, 2000)
    try:
        y = hip_forward(x, qw, scales, zeros, None, 4, 128, 0)
        torch.cuda.synchronize()
    finally:
        L.bie_test_forge_reducer(0, 0)
    assert torch.isnan(y.float()).all()
    assert L.bie_device_status(1) & 1


def test_dependency_timeout_fails_loudly():
    """A dependent list entry whose producer never finishes (forged: it waits for one tile more than the producer has) must not compute
    from a stale x: NaN in its y AND in every y downstream of it (their completion counts still move, so nobody spins for long), status
    bit 2 raised, the next launching call fails, and the plan is healthy afterwards (the kernel used to set the bit only)."""
    from bitorch_engine import _hip
    from bitorch_engine.layers.qlinear.nbit.cuda import MPQForwardList
    L = _hip.lib()
    entries, host = _list_case([(512, 512, False), (512, 512, True), (512, 256, False)], orc.BF16, 4, 128, 0, 1, seed=91, chain=True)
    plan = MPQForwardList(entries, w_bit=4, group_size=128)
    assert L.bie_device_status(1) == 0
    L.bie_test_forge_reducer(0, 500)  # short spin bound
    L.bie_test_forge_dependency(1)
    try:
        plan()
        torch.cuda.synchronize()
    finally:
        L.bie_test_forge_dependency(0)
        L.bie_test_forge_reducer(0, 0)
    assert torch.isfinite(entries[0]["y"].float()).all(), "the head of the chain has no dependency"
    assert torch.isnan(entries[1]["y"].float()).all(), "a timed-out dependent entry returned numbers"
    assert torch.isnan(entries[2]["y"].float()).all(), "the entry downstream of a poisoned one returned numbers"
    assert L.bie_device_status(0) & 2
    with pytest.raises(RuntimeError, match="device status"):
        plan()
    assert L.bie_device_status(0) == 0
    plan()
    torch.cuda.synchronize()
    xin = host[0][0]
    for i, (e, (_, qw, scales, zeros, bias)) in enumerate(zip(entries, host)):
        assert_close(e["y"], oracle_forward(xin, qw, scales, zeros, None, 4, 128, 0, orc.BF16, bias), orc.BF16, f"after the forged dependency timeout, layer {i}")
        xin = e["y"].cpu()


def test_binary_linear_cutlass_mm_and_batched_matmul_follow_the_reference_signatures():
    """binary_linear_cutlass.mm(x, y, kernel_id) -> int32 XOR-popcount accumulator (binary_linear_cutlass_kernel.cu:293-332,650-666);
    matmul(x, y, scale) batched in ONE launch, float32 exact and bfloat16 with the reference's bf16 arithmetic on the popcount."""
    from bitorch_engine.extensions import binary_linear_cutlass as blc
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn((24, 320), generator=g), torch.randn((40, 320), generator=g)
    sx, sy = torch.where(x >= 0, 1.0, -1.0), torch.where(y >= 0, 1.0, -1.0)
    popc = ((320 - sx @ sy.t()) / 2).to(torch.int32)
    for kid in (1, 3, 9):  # a kernel id never changes the meaning
        out = blc.mm(x.to(DEV), y.to(DEV), kid)
        assert out.dtype == torch.int32 and torch.equal(out.cpu(), popc)
    xb, yb = torch.randn((2, 3, 17, 512), generator=g), torch.randn((2, 3, 33, 512), generator=g)
    ref = (torch.where(xb >= 0, 1.0, -1.0) @ torch.where(yb >= 0, 1.0, -1.0).transpose(-1, -2)) * 0.5
    out = blc.matmul(xb.to(DEV), yb.to(DEV), 0.5)
    assert out.dtype == torch.float32 and torch.equal(out.cpu(), ref)
    outb = blc.matmul(xb.to(DEV).bfloat16(), yb.to(DEV).bfloat16(), 0.5)
    pb = ((512 - 2 * ref) / 2).to(torch.bfloat16)  # signs are unchanged by the bf16 cast of x / y (no zeros in randn)
    refb = (torch.tensor(512.0) - 2 * pb.float()).to(torch.bfloat16) * torch.tensor(0.5, dtype=torch.bfloat16)
    assert outb.dtype == torch.bfloat16 and torch.equal(outb.cpu(), refb)


@pytest.mark.parametrize("M", [1, 2, 3, 4, 7, 8, 13, 16])  # > 2: regular groups only (the golden configs are) -- four rows ride on one v_mfma_f32_4x4x4, up to four of them per four k
def test_exl2_list_forward_equals_the_per_layer_calls(M):
    """bie_mbwq_exl2_list_*: several mixed-bit layers (different K / N / band tables, with and without q_perm) in ONE launch must
    give, entry by entry, exactly what bie_mbwq_exl2_forward gives (same kernel body, same slab plan or not: the fp32 sums
    are added in chunk order inside a wave and in wave / slab order across them, so a different slab plan may differ in the last
    bit -- hence oracle tolerance for both and exact equality between two launches of the same plan)."""
    from bitorch_engine.extensions import q_linear_cuda
    from bitorch_engine.layers.qlinear.nbit.cuda import MBWQExl2ForwardList
    from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
    g = np.load(os.path.join(GOLDEN, "exl2_group_maps.npz"))
    cfgs = sorted(k[:-5] for k in g.files if k.endswith("_meta"))
    entries, refs = [], []
    for i, cfg in enumerate(cfgs + cfgs[:2]):
        K, groups, rows_packed = [int(v) for v in g[cfg + "_meta"]]
        q_groups = torch.from_numpy(g[cfg + "_q_groups"])
        N = (200, 64, 328, 136, 520, 72)[i % 6]
        rng = np.random.default_rng(K + M + i)
        gen = torch.Generator().manual_seed(K + M + i)
        qw = torch.from_numpy(rng.integers(-2 ** 31, 2 ** 31 - 1, (rows_packed, N), dtype=np.int64).astype(np.int32))
        scales = (torch.rand((groups, N), generator=gen) * 0.02 + 0.001).half()
        zeros = (torch.randn((groups, N), generator=gen) * 0.1).half()
        q_perm = torch.randperm(K, generator=gen).to(torch.short) if i % 3 != 2 else None
        gmap = make_group_map(q_groups, rows_packed)
        qs, rows = exl2_load(qw, q_groups, K, groups)
        x = torch.randn((M, K), generator=gen).half()
        Wo = orc.exl2_dequant(qw.numpy(), orc.torch_to_np(scales), orc.torch_to_np(zeros), None if q_perm is None else q_perm.numpy(), q_groups.numpy(), K)
        refs.append(t16(orc.gemm(orc.torch_to_np(x), Wo, orc.F16), orc.F16))
        entries.append({"x": x.to(DEV), "qweight": qs, "scales": scales.to(DEV), "zeros": zeros.to(DEV),
                        "q_perm": None if q_perm is None else q_perm.to(DEV), "q_group_map": gmap.to(DEV), "rows": rows,
                        "y": torch.full((M, N), float("nan"), dtype=torch.float16, device=DEV)})
    plan = MBWQExl2ForwardList(entries)
    plan()
    torch.cuda.synchronize()
    first = [e["y"].clone() for e in entries]
    for i, (e, r) in enumerate(zip(entries, refs)):
        assert_close(e["y"], r, orc.F16, f"exl2 list entry {i} M={M}")
    plan()
    torch.cuda.synchronize()
    for i, e in enumerate(entries):
        assert torch.equal(e["y"], first[i]), f"exl2 list entry {i}: second launch differs"


def test_diodemix_steps_an_mpq_parameter_like_the_reference():
    """The optimiser side of SURVEY 8f-2: bitorch_engine.optim.DiodeMix on the device.  An MPQWeightParameter in its GPTQ form, fed through
    `privileged_grad` for five steps (the fifth also moves the zero points), against the REFERENCE's DiodeMix.step on CPU tensors
    (tests/golden/optim_diodemix.npz, oracle/gen_golden.py optim_diodemix_vectors): the re-packed weights after every step and the re-packed
    zero points at the end, bit for bit; the optimiser's state carries the reference's keys.  A float parameter in the same optimiser takes the
    AdamW branch on the device (1e-5 against the CPU vectors: the GPU kernels contract multiply-adds the CPU ones do not)."""
    from bitorch_engine.layers.qlinear.nbit.layer import MPQWeightParameter
    from bitorch_engine.optim import DiodeMix
    g = np.load(os.path.join(GOLDEN, "optim_diodemix.npz"))
    K, N, w_bit, gs = [int(v) for v in g["mpq_meta"]]
    as_f16 = lambda a: torch.from_numpy(a.view(np.int16).copy()).view(torch.float16)
    p = MPQWeightParameter(torch.from_numpy(g["mpq_qweight0"]).to(DEV), scales=as_f16(g["mpq_scales"]).to(DEV), zeros=torch.from_numpy(g["mpq_qzeros0"]).to(DEV),
                           g_idx=torch.from_numpy(g["mpq_g_idx"]).to(DEV), w_bit=w_bit, asym=True, group_size=gs, layer_type=1)
    opt = DiodeMix([p], lr=2e-3, betas=(0.9, 0.99), eps=1e-6, dtype=torch.float16)
    for it in range(1, 6):
        p.privileged_grad = as_f16(g[f"mpq_grad{it}"]).to(DEV)
        p.grad = torch.zeros_like(p.data)
        opt.step()
        got, want = p.data.cpu().numpy(), g[f"mpq_qweight{it}"]
        assert np.array_equal(got, want), f"packed weights differ after step {it} ({(got != want).sum()} of {want.size} words)"
    assert np.array_equal(p.zeros.cpu().numpy(), g["mpq_qzeros5"]), "re-packed qzeros differ after the fifth step"
    st = opt.state[p]
    assert set(st) == {"step", "exp_avg_l", "exp_avg_s"} and float(st["step"]) == 5.0 and st["exp_avg_l"].dtype == torch.float16 and st["exp_avg_l"].is_cuda
    assert not np.array_equal(p.data.cpu().numpy(), g["mpq_qweight0"])
    # the AdamW branch on the device
    ps = {n: torch.nn.Parameter(torch.from_numpy(g[f"float_{n}_0"]).to(DEV)) for n in ("p1", "p2", "p3")}
    opt = DiodeMix([{"params": [ps["p1"], ps["p2"]], "weight_decay": 0.01}, {"params": [ps["p3"]], "lr": 5e-4, "correct_bias": False}], lr=1e-3, betas=(0.9, 0.99), eps=1e-6)
    for it in range(1, 5):
        for n, q in ps.items():
            q.grad = torch.from_numpy(g[f"float_{n}_grad{it}"]).to(DEV)
        opt.step()
        for n, q in ps.items():
            torch.testing.assert_close(q.detach().cpu(), torch.from_numpy(g[f"float_{n}_{it}"]), rtol=1e-5, atol=1e-6, msg=lambda m: f"{n} after step {it}: {m}")


@pytest.mark.parametrize("key", ["f16_trivial", "f16_actorder", "bf16_trivial", "bf16_actorder"])
def test_qweight_update_step_bit_exact_vs_reference(key):
    """MPQWeightParameter.update -> qweight_update_fn on the device (HIP unpack + Adam moments + HIP re-pack) against
    vectors produced by the REFERENCE's own qweight_update_fn on the CPU (tests/golden/update_step.npz, oracle/gen_golden.py): five
    consecutive steps on a GPTQ-style parameter, bias correction on the even steps, the zero-point update on step 5.  Packed
    weights, both moment tensors and the re-packed qzeros must be identical bit for bit after every step."""
    from bitorch_engine.layers.qlinear.nbit.layer import MPQWeightParameter
    g = np.load(os.path.join(GOLDEN, "update_step.npz"))
    dtype = torch.float16 if key.startswith("f16") else torch.bfloat16
    K, N, w_bit, gs = [int(v) for v in g[key + "_meta"]]
    as_dt = lambda a: torch.from_numpy(a.view(np.int16).copy()).view(dtype)
    p = MPQWeightParameter(torch.from_numpy(g[key + "_qweight0"]).to(DEV), scales=as_dt(g[key + "_scales"]).to(DEV),
                           zeros=torch.from_numpy(g[key + "_qzeros0"]).to(DEV), g_idx=torch.from_numpy(g[key + "_g_idx"]).to(DEV),
                           w_bit=w_bit, asym=True, group_size=gs, layer_type=1)
    exp_l = torch.zeros((K, N), dtype=dtype, device=DEV)
    exp_s = torch.zeros((K, N), dtype=dtype, device=DEV)
    step = torch.tensor(0.0)
    for it in range(1, 6):
        grad = as_dt(g[f"{key}_grad{it}"]).to(DEV)
        MPQWeightParameter.update(p, exp_avg_s=exp_s, exp_avg_l=exp_l, step=step, lr=2e-3, weight_decay=0.0, beta1=0.9, beta2=0.99, eps=1e-6,
                                  dtype=dtype, correct_bias=(it % 2 == 0), projector=None, grad=grad)
        assert np.array_equal(orc.torch_to_np(exp_l), g[f"{key}_exp_l{it}"]), f"{key}: first moment differs after step {it}"
        assert np.array_equal(orc.torch_to_np(exp_s), g[f"{key}_exp_s{it}"]), f"{key}: second moment differs after step {it}"
        got, want = p.data.cpu().numpy(), g[f"{key}_qweight{it}"]
        assert np.array_equal(got, want), f"{key}: packed weights differ after step {it} ({(got != want).sum()} of {want.size} words)"
    assert np.array_equal(p.zeros.cpu().numpy(), g[key + "_qzeros5"]), f"{key}: re-packed qzeros differ after the step-5 zero-point update"
    assert float(step) == 5.0
