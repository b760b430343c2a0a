"""GPU parity of the matrix-pipe form of the binary GEMM (csrc/binary_fp4.hip): +-1 as FP4 (E2M1) operands of
v_mfma_scale_f32_32x32x64_f8f6f4.  Bar: BIT-EXACT -- the same integers K - 2*popcount(x ^ w) as the oracle
(oracle/bie_oracle.c orc_binary_linear_rowpacked, pinned to the outputs of the reference's own binary_linear.cpp by
tests/test_oracle_golden.py) and as the XNOR-popcount kernels; called through the C ABI (ctypes) and through the
reference's extension-level entry points."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _image_ref(bits: np.ndarray, rows: int, K: int) -> np.ndarray:
    """include/bie_hip.h's image definition restated in numpy: 1 KiB fragments (row block of 32, k block of 64), lane l owns the 32
    nibbles of row 32*rb + (l & 31), k 64*kb + 32*(l >> 5) .. +31, element e in bits 4e .. 4e+3; +1 -> 0x2, -1 -> 0xA, padding 0."""
    RB, KT = (rows + 31) // 32, (K + 127) // 128
    sign = np.unpackbits(bits.reshape(rows, K // 8), axis=1, bitorder="little")[:, :K]  # 1 = (v >= 0)
    nib = np.zeros((RB * 32, KT * 128), dtype=np.uint8)
    nib[:rows, :K] = np.where(sign == 1, 0x2, 0xA)
    # [rb, r, kb, h, e] -> [rb, kb, h, r, e]: lane = h * 32 + r
    frag = nib.reshape(RB, 32, KT * 2, 2, 32).transpose(0, 2, 3, 1, 4)
    byt = (frag[..., 0::2] | (frag[..., 1::2] << 4)).astype(np.uint8)
    return np.ascontiguousarray(byt).reshape(-1)


def _fp4_forward(L, _hip, xp, wp, M, N, K, scale=1.0):
    ximg = torch.empty(L.bie_binary_fp4_image_bytes(M, K), dtype=torch.uint8, device=DEV)
    wimg = torch.empty(L.bie_binary_fp4_image_bytes(N, K), dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _hip.check(L.bie_binary_fp4_image(xp.data_ptr(), ximg.data_ptr(), M, K, st), "image x")
    _hip.check(L.bie_binary_fp4_image(wp.data_ptr(), wimg.data_ptr(), N, K, st), "image w")
    y = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    _hip.check(L.bie_binary_linear_forward_fp4(ximg.data_ptr(), wimg.data_ptr(), y.data_ptr(), M, N, K, float(scale), st), "fp4 forward")
    torch.cuda.synchronize()
    return y


@pytest.mark.parametrize("rows,K", [(1, 8), (32, 128), (33, 136), (100, 520), (257, 1000), (64, 4096)])
def test_fp4_image_is_the_documented_layout(rows, K):
    from bitorch_engine import _hip
    L = _hip.lib()
    rng = np.random.default_rng(rows * 7919 + K)
    bits = rng.integers(0, 256, size=(rows, K // 8), dtype=np.uint8)
    ref = _image_ref(bits, rows, K)
    assert L.bie_binary_fp4_image_bytes(rows, K) == ref.size
    img = torch.full((ref.size,), 0x55, dtype=torch.uint8, device=DEV)
    _hip.check(L.bie_binary_fp4_image(torch.from_numpy(bits).to(DEV).data_ptr(), img.data_ptr(), rows, K, torch.cuda.current_stream().cuda_stream), "image")
    assert np.array_equal(img.cpu().numpy(), ref)
    # an unaligned source pointer takes the byte path: same image
    buf = torch.zeros(bits.size + 1, dtype=torch.uint8, device=DEV)
    buf[1:] = torch.from_numpy(bits.reshape(-1)).to(DEV)
    img2 = torch.empty_like(img)
    _hip.check(L.bie_binary_fp4_image(buf.data_ptr() + 1, img2.data_ptr(), rows, K, torch.cuda.current_stream().cuda_stream), "image")
    assert np.array_equal(img2.cpu().numpy(), ref)


@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16, torch.float32, torch.int8])
@pytest.mark.parametrize("rows,K,with_bias", [(70, 200, True), (33, 136, False), (128, 1024, True)])
def test_fp4_image_from_values_equals_pack_rows_then_image(tdt, rows, K, with_bias):
    from bitorch_engine import _hip
    from bitorch_engine.extensions._binary_common import pack_rows, sign_dt
    L = _hip.lib()
    g = torch.Generator().manual_seed(rows + K)
    if tdt == torch.int8:
        v = torch.randint(-3, 4, (rows, K), generator=g, dtype=torch.int8).to(DEV)
        bias = None
    else:
        v = torch.randn((rows, K), generator=g).to(tdt).to(DEV)
        v[0, :8] = 0.0  # zeros (and -0.0) count as +1, as torch's >= does
        v[0, 1] = -0.0
        bias = (torch.randn((K,), generator=g) * 0.5).to(tdt).to(DEV) if with_bias else None
    bits = pack_rows(v if bias is None else v + bias)
    st = torch.cuda.current_stream().cuda_stream
    ref = torch.empty(L.bie_binary_fp4_image_bytes(rows, K), dtype=torch.uint8, device=DEV)
    _hip.check(L.bie_binary_fp4_image(bits.data_ptr(), ref.data_ptr(), rows, K, st), "image")
    img = torch.full_like(ref, 0x55)
    _hip.check(L.bie_binary_fp4_image_from_values(v.data_ptr(), _hip.ptr(bias), img.data_ptr(), rows, K, sign_dt(v), st), "image from values")
    assert torch.equal(img, ref)


@pytest.mark.parametrize("tile", ["64", "128", "256"])
@pytest.mark.parametrize("var", ["0", "1"])
@pytest.mark.parametrize("M,N,K", [(1, 1, 8), (32, 32, 128), (130, 70, 264), (256, 256, 384), (300, 520, 1000), (257, 129, 2048), (512, 384, 4096)])
def test_fp4_gemm_bit_exact_vs_oracle(M, N, K, tile, var, monkeypatch):
    from bitorch_engine import _hip
    L = _hip.lib()
    monkeypatch.setenv("BIE_FP4_TILE", tile)
    monkeypatch.setenv("BIE_FP4_VAR", var)
    rng = np.random.default_rng(M * 31 + N * 17 + K)
    xb = rng.integers(0, 256, size=(M, K // 8), dtype=np.uint8)
    wb = rng.integers(0, 256, size=(N, K // 8), dtype=np.uint8)
    ref = orc.binary_linear_rowpacked(xb, wb, K, 0.5)
    y = _fp4_forward(L, _hip, torch.from_numpy(xb).to(DEV), torch.from_numpy(wb).to(DEV), M, N, K, 0.5)
    assert np.array_equal(y.cpu().numpy(), ref)


def test_fp4_gemm_rows_and_columns_land_where_they_belong():
    """Asymmetric operands (the A = I check of the MFMA playbook): x row m agrees with w row n on exactly the first (m + 2n) % K positions."""
    from bitorch_engine import _hip
    L = _hip.lib()
    M, N, K = 96, 160, 512
    w = np.ones((N, K), dtype=np.float32)
    x = np.ones((M, K), dtype=np.float32)
    rng = np.random.default_rng(5)
    w[:] = np.where(rng.random((N, K)) < 0.5, 1.0, -1.0)
    x[:] = np.where(rng.random((M, K)) < 0.5, 1.0, -1.0)
    x[3] = w[7]          # y[3, 7] = K
    x[64] = -w[130]      # y[64, 130] = -K
    ref = x @ w.T
    y = _fp4_forward(L, _hip, torch.from_numpy(orc.binary_pack_rows(x)).to(DEV), torch.from_numpy(orc.binary_pack_rows(w)).to(DEV), M, N, K)
    assert np.array_equal(y.cpu().numpy(), ref)
    assert y[3, 7].item() == K and y[64, 130].item() == -K


@pytest.mark.parametrize("M", [1024, 4096])
def test_fp4_gemm_4096_equals_the_xnor_kernels_and_the_oracle(M):
    from bitorch_engine import _hip
    L = _hip.lib()
    N = K = 4096
    g = torch.Generator().manual_seed(M)
    xp = torch.randint(0, 256, (M, K // 8), generator=g, dtype=torch.uint8).to(DEV)
    wp = torch.randint(0, 256, (N, K // 8), generator=g, dtype=torch.uint8).to(DEV)
    y = _fp4_forward(L, _hip, xp, wp, M, N, K)
    yx = torch.empty_like(y)
    _hip.check(L.bie_binary_linear_forward(xp.data_ptr(), wp.data_ptr(), yx.data_ptr(), M, N, K, 0, 1.0, torch.cuda.current_stream().cuda_stream), "xnor")
    torch.cuda.synchronize()
    assert torch.equal(y, yx)
    rows = np.array([0, 1, 31, 32, 255, 256, M // 2 + 3, M - 1])
    ref = orc.binary_linear_rowpacked(xp.cpu().numpy()[rows], wp.cpu().numpy(), K)
    assert np.array_equal(y.cpu().numpy()[rows], ref)


def test_extension_forward_takes_the_matrix_pipe_for_large_m_and_matches_the_xnor_path(monkeypatch):
    """binary_linear_cutlass.forward / binary_linear_cuda.forward at M = 512: FP4 form (default threshold) == XNOR form == oracle;
    the weight image is built once per tensor version."""
    from bitorch_engine.extensions import binary_linear_cutlass, binary_linear_cuda
    M, N, K = 512, 768, 1024
    g = torch.Generator().manual_seed(11)
    x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(DEV)
    w = torch.randn((N, K), generator=g).to(torch.bfloat16).to(DEV)
    ref = orc.binary_linear_rowpacked(orc.binary_pack_rows(x.float().cpu().numpy()), orc.binary_pack_rows(w.float().cpu().numpy()), K, 0.25)
    wp = binary_linear_cutlass.w_pack(w, False)
    y1 = binary_linear_cutlass.forward(x, wp, 0.25, False, 0)
    assert ("fp4", N, K) in wp._bie_memo
    img = wp._bie_memo[("fp4", N, K)][1]
    y1b = binary_linear_cutlass.forward(x, wp, 0.25, False, 0)
    assert wp._bie_memo[("fp4", N, K)][1] is img
    monkeypatch.setenv("BIE_FP4_MIN_M", "0")
    y2 = binary_linear_cutlass.forward(x, wp, 0.25, False, 0)
    monkeypatch.delenv("BIE_FP4_MIN_M")
    assert np.array_equal(y1.cpu().numpy(), ref) and torch.equal(y1, y2) and torch.equal(y1, y1b)
    wimg = binary_linear_cuda.w_pack(w, 3, False)
    y3 = binary_linear_cuda.forward(x, wimg, 3, False)
    assert np.array_equal(y3.cpu().numpy() * 0.25, ref)
    wp[0, 0] ^= 0xFF  # in-place edit: the image must follow
    y4 = binary_linear_cutlass.forward(x, wp, 0.25, False, 0)
    wb = wp.cpu().numpy()
    assert np.array_equal(y4.cpu().numpy(), orc.binary_linear_rowpacked(orc.binary_pack_rows(x.float().cpu().numpy()), wb, K, 0.25))


@pytest.mark.parametrize("tile", ["64", "128", "256"])
@pytest.mark.parametrize("tdt", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(300, 264, 1024), (257, 100, 520), (512, 4096, 4096), (64, 72, 128)])
def test_fp4_layer_epilogue_is_the_layer_expression(M, N, K, tdt, tile, monkeypatch):
    """values -> image (bias add + sign) -> matrix-pipe GEMM with the `.to(dtype) * scale_a * scale_w` epilogue == the oracle's integers
    pushed through the layer's own expression (reference layers/qlinear/binary/cuda/layer.py:58-63, 283); N % 8 != 0 takes the
    element-store path, K = 520 the padded tail."""
    from bitorch_engine.extensions._binary_common import pack_rows, xnor_layer_fp4
    monkeypatch.setenv("BIE_FP4_TILE", tile)
    gen = torch.Generator().manual_seed(M * 7 + N + K)
    x = torch.randn((M, K), generator=gen).to(tdt)
    b = (torch.randn(K, generator=gen) * 0.5).to(tdt)
    w = torch.randn((N, K), generator=gen)
    sa = torch.tensor(0.7312, dtype=tdt)
    sw = torch.tensor(0.0131, dtype=tdt)
    wrows = pack_rows(w.to(DEV))
    y = xnor_layer_fp4(x.to(DEV), wrows, b.to(DEV), sa.to(DEV), sw.to(DEV)).cpu()
    ints = orc.binary_linear_rowpacked(orc.binary_pack_rows((x + b).float().numpy()), orc.binary_pack_rows(w.numpy()), K)
    expect = torch.from_numpy(ints.astype(np.float32)).to(tdt) * sa * sw
    assert y.dtype == tdt and torch.equal(y, expect)
    y1 = xnor_layer_fp4(x.to(DEV), wrows).cpu()  # no bias, no scales
    ints1 = orc.binary_linear_rowpacked(orc.binary_pack_rows(x.float().numpy()), orc.binary_pack_rows(w.numpy()), K)
    assert torch.equal(y1, torch.from_numpy(ints1.astype(np.float32)).to(tdt))


def test_binary_cuda_layer_takes_the_matrix_pipe_at_large_m_and_matches_the_composed_forward(monkeypatch):
    from bitorch_engine.layers.qlinear.binary.cuda import BinaryLinearCuda
    from bitorch_engine.extensions import _binary_common
    torch.manual_seed(3)
    K, N = 1024, 384
    layer = BinaryLinearCuda(K, N, dtype=torch.bfloat16)
    layer.set_weight_data(torch.randn(N, K).to(torch.bfloat16))
    wt = layer.weight.data.float().cpu()  # the layer's own sign carriers (init_weight centres the weights before taking the sign)
    layer.bias_a.data = (torch.randn(K) * 0.3).to(torch.bfloat16)
    layer.eval().to(DEV)
    layer.generate_quantized_weight(qweight_only=True)
    x = torch.randn((3, 200, K)).to(torch.bfloat16).to(DEV)  # 600 rows
    calls = []
    orig = _binary_common.xnor_layer_fp4
    monkeypatch.setattr("bitorch_engine.extensions.binary_linear_cuda.xnor_layer_fp4", lambda *a, **k: calls.append(1) or orig(*a, **k))
    with torch.no_grad():
        y = layer(x)
    assert calls == [1] and y.shape == (3, 200, N) and y.dtype == torch.bfloat16
    monkeypatch.setenv("BIE_FP4_MIN_M", "0")  # the XNOR composition: set_activation, pack, popcount GEMM, cast, two multiplies
    with torch.no_grad():
        y0 = layer(x)
    assert calls == [1] and torch.equal(y, y0)
    ints = orc.binary_linear_rowpacked(orc.binary_pack_rows((x + layer.bias_a.detach()).reshape(-1, K).float().cpu().numpy()),
                                       orc.binary_pack_rows(wt.numpy()), K)
    expect = torch.from_numpy(ints.astype(np.float32)).to(torch.bfloat16) * layer.scale_a.detach().cpu() * layer.scale_w.detach().cpu()
    assert torch.equal(y.reshape(-1, N).cpu(), expect)


@pytest.mark.parametrize("tile", ["64", "128", "256"])
@pytest.mark.parametrize("B,C,H,W,OC,ks,st,pad,dil", [(2, 64, 7, 7, 64, 3, 1, 1, 1), (3, 32, 9, 11, 96, 3, 2, 1, 1), (2, 128, 8, 8, 72, 3, 1, 2, 2),
                                                       (1, 64, 6, 5, 64, 1, 1, 0, 1), (4, 96, 5, 5, 130, 5, 1, 2, 1), (33, 512, 7, 7, 512, 3, 1, 1, 1)])
def test_fp4_conv_equals_the_oracle_and_the_tap_form(B, C, H, W, OC, ks, st, pad, dil, tile, monkeypatch):
    """The conv as an FP4 GEMM over (pixel) x (tap, channel) -- padding counted as -1, NCHW epilogue -- == orc_binary_conv2d (pinned to
    the reference's binary_conv.cpp by tests/test_oracle_golden.py) == the XNOR tap form, bit for bit; strides, dilation, 1x1 and 5x5
    kernels, ragged widths, OC not a multiple of 32, the ResNet-18 layer-4 shape."""
    from bitorch_engine.extensions import binary_conv_cpp
    from bitorch_engine.extensions._binary_common import pack_rows
    monkeypatch.setenv("BIE_FP4_TILE", tile)
    g = torch.Generator().manual_seed(B * 131 + C + H * 7 + OC)
    x = torch.randn((B, C, H, W), generator=g)
    w = torch.randn((OC, C, ks, ks), generator=g)
    wp = pack_rows(w.reshape(OC, -1).to(DEV)).contiguous()
    OH = (H + 2 * pad - dil * (ks - 1) - 1) // st + 1
    OW = (W + 2 * pad - dil * (ks - 1) - 1) // st + 1
    args = (wp, OC, B * OH * OW, C * ks * ks, ks, st, pad, dil, OW)
    monkeypatch.setenv("BIE_FP4_CONV_MIN_ROWS", "1")
    y = binary_conv_cpp.forward(x.to(DEV), *args)
    assert ("fp4taps", OC, C, ks) in wp._bie_memo
    yb = binary_conv_cpp.forward(x.to(torch.bfloat16).to(DEV), *args)  # bf16 activations: the signs of the rounded values
    monkeypatch.setenv("BIE_FP4_CONV_MIN_ROWS", "0")
    y0 = binary_conv_cpp.forward(x.to(DEV), *args)
    ref = orc.binary_conv2d(x.numpy(), w.numpy(), st, pad, dil)
    assert y.shape == (B, OC, OH, OW) and torch.equal(y, y0)
    assert np.array_equal(y.cpu().numpy(), ref)
    assert np.array_equal(yb.cpu().numpy(), orc.binary_conv2d(x.to(torch.bfloat16).float().numpy(), w.numpy(), st, pad, dil))


def test_fp4_entry_points_reject_what_they_cannot_take():
    """Argument errors are status codes + bie_last_error(), never a crash or a silent fallback."""
    from bitorch_engine import _hip
    L = _hip.lib()
    st = torch.cuda.current_stream().cuda_stream
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device=DEV)
    p = buf.data_ptr()
    assert L.bie_binary_fp4_image_bytes(0, 128) == 0 and L.bie_binary_fp4_image_bytes(33, 129) == 2 * 2 * 2 * 1024
    assert L.bie_binary_fp4_image(p, p + 4096, 32, 100, st) != 0 and b"K % 8" in L.bie_last_error()      # K not a multiple of 8 (bits form)
    assert L.bie_binary_fp4_image(p, p + 4097, 32, 128, st) != 0 and b"aligned" in L.bie_last_error()    # image not 16-byte aligned
    assert L.bie_binary_fp4_image_from_values(p, p, p + 4096, 32, 128, 3, st) != 0                       # int8 sign carriers take no bias
    assert L.bie_binary_linear_forward_fp4(p, p + 4096, p + 8192, 32, 32, 1 << 24, 1.0, st) != 0 and b"2^24" in L.bie_last_error()
    assert L.bie_binary_linear_layer_fp4(p, p + 4096, None, None, p + 8192, 32, 32, 128, 3, st) != 0       # dtype 3 has no layer epilogue
    assert L.bie_binary_conv2d_fp4_workspace_bytes(1, 48, 7, 7, 3, 1, 1, 1) == 0                          # C % 32 != 0
    assert L.bie_binary_conv2d_forward_fp4(p, p + 4096, p + 8192, p + 16384, 1 << 15, 1, 48, 7, 7, 64, 3, 1, 1, 1, 1.0, 0, st) != 0 and b"multiple of 32" in L.bie_last_error()
    assert L.bie_binary_conv2d_forward_fp4(p, p + 4096, p + 8192, p + 16384, 16, 1, 64, 7, 7, 64, 3, 1, 1, 1, 1.0, 0, st) != 0 and b"workspace" in L.bie_last_error()
    torch.cuda.synchronize()


def test_fp4_paths_replay_in_a_captured_graph():
    """The matrix-pipe forms are stream-ordered and allocation-free on the C side: a captured layer forward replays with new activations."""
    from bitorch_engine.extensions._binary_common import pack_rows, xnor_layer_fp4
    M, N, K = 512, 256, 1024
    g = torch.Generator().manual_seed(4)
    w = torch.randn((N, K), generator=g)
    wrows = pack_rows(w.to(DEV))
    x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(DEV)
    sa = torch.tensor(0.5, dtype=torch.bfloat16, device=DEV)
    y_eager = xnor_layer_fp4(x, wrows, None, sa, None)  # also sizes the scratch buffer and builds the weight image before the capture
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        xnor_layer_fp4(x, wrows, None, sa, None)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            y_cap = xnor_layer_fp4(x, wrows, None, sa, None)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y_cap, y_eager)
    x.copy_(torch.randn((M, K), generator=g).to(torch.bfloat16))
    graph.replay()
    torch.cuda.synchronize()
    ints = orc.binary_linear_rowpacked(orc.binary_pack_rows(x.float().cpu().numpy()), orc.binary_pack_rows(w.numpy()), K)
    assert torch.equal(y_cap.cpu(), torch.from_numpy(ints.astype(np.float32)).to(torch.bfloat16) * sa.cpu())
