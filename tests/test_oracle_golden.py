"""Pins the CPU oracle (oracle/bie_oracle.c) against outputs of the reference itself.

The fixtures under tests/golden/ were produced by oracle/gen_golden.py, which imports the Python
reference (unpack_qweight / pack_fp_weight / MPQLinearCuda CPU branch, cuda/utils.py, mpq_layer.py)
and the reference's compiled binary CPU extensions (binary_linear.cpp / binary_conv.cpp).
Bar: bit-exact for dequant / pack / binary; 1e-3 (norm-wise) for the fp16/bf16 matmul.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MANIFEST = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))


def _dt(name):
    return orc.BF16 if "_bf16" in name else orc.F16


def _to_f32(a, dt):
    out = np.empty(a.shape, np.float32)
    fn = orc.lib().orc_bf16_to_f32 if dt == orc.BF16 else orc.lib().orc_f16_to_f32
    a = np.ascontiguousarray(a)
    import ctypes
    fn(a.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(a.size))
    return out


@pytest.mark.parametrize("name", MANIFEST["mpq_dequant_pack"])
def test_mpq_dequant_and_pack_bit_exact(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    K, N, gs, w_bit, asym, has_gidx = [int(v) for v in d["meta"]]
    dt = _dt(name)
    g_idx = d["g_idx"] if has_gidx else None
    W = orc.mpq_dequant(d["qweight"], d["scales"], d["zeros"], g_idx, w_bit, gs, asym, dt)
    assert np.array_equal(W, d["W"]), f"dequant mismatch: {(W != d['W']).sum()} of {W.size}"
    packed = orc.mpq_pack(d["Wp"], d["scales"], d["zeros"], g_idx, w_bit, gs, asym, dt)
    assert np.array_equal(packed, d["packed"]), f"pack mismatch: {(packed != d['packed']).sum()} words"
    # round trip property: pack(unpack(p)) == p  
    again = orc.mpq_pack(W, d["scales"], d["zeros"], g_idx, w_bit, gs, asym, dt)
    if dt == orc.F16:  # bf16 cannot hold 8 significant bits of q*s, so only fp16 round-trips
        assert np.array_equal(again, d["qweight"])


def test_gptq_zeros_packing():
    d = np.load(os.path.join(GOLDEN, "gptq_zeros_packing.npz"))
    assert np.array_equal(orc.pack_qzeros(d["zq"], 4), d["packed"])


@pytest.mark.parametrize("name", MANIFEST["mpq_layers"])
def test_mpq_layer_forward_vs_reference_cpu_path(name):
    d = np.load(os.path.join(GOLDEN, f"layer_{name}.npz"))
    dt = _dt(name)
    w_bit = 8 if "_w8_" in name else (2 if "_w2_" in name else 4)
    asym = int("asym" in name or "gptq" in name)
    qweight = d["sd_qweight"]
    K = qweight.shape[0] * 32 // w_bit
    scales, zeros = d["prep_scales"], d["prep_zeros"]
    gs = K // scales.shape[0]
    W = orc.mpq_dequant(qweight, scales, zeros, d["sd_g_idx"], w_bit, gs, asym, dt)
    for M in (33, 64):
        y = _to_f32(orc.gemm(d[f"x{M}"], W, dt), dt)
        yr = _to_f32(d[f"y{M}"], dt)
        # torch CPU matmul accumulates in float in an unspecified order; one output ulp is
        # <= 2^-10 (fp16) / 2^-7 (bf16) relative, so compare norm-wise at 1e-3 + 1 ulp of the type
        ulp = 2.0 ** -7 if dt == orc.BF16 else 2.0 ** -10  # one ulp, worst case within a binade
        tol = 1e-3 * np.abs(yr).max() + ulp * np.abs(yr)
        assert np.all(np.abs(y - yr) <= tol), float(np.abs(y - yr).max())
        y2 = _to_f32(orc.mpq_forward(d[f"x{M}"], qweight, scales, zeros, d["sd_g_idx"], w_bit, gs, asym, dt), dt)
        assert np.all(np.abs(y2 - yr) <= tol)


def test_exl2_rows_and_group_map_tables():
    d = np.load(os.path.join(GOLDEN, "exl2_group_maps.npz"))
    for c in ("q_proj", "k_proj", "w3w2", "all6"):
        K, groups, rows = [int(v) for v in d[c + "_meta"]]
        qg = d[c + "_q_groups"]
        r7 = orc.exl2_rows(qg, K)
        assert r7[5] == K, r7
        # group map == (group index, rows left in group) per k : cuda/utils.py:150-187
        gm = d[c + "_group_map"].reshape(-1, 2)
        assert gm.shape[0] == K
        k = 0
        for i in range(groups):
            bits = int(qg[2 * i])
            nxt = int(qg[2 * i + 3]) if i < groups - 1 else rows
            n = (nxt - int(qg[2 * i + 1])) * 32 // bits
            assert np.all(gm[k:k + n, 0] == i)
            assert np.array_equal(gm[k:k + n, 1], np.arange(n, 0, -1))
            k += n


def test_exl2_single_band_equals_mpq_layout():
    """A single 4-bit (or 2-bit) band covering K is the GPTQ layout: cross-check of the bitstream
    reader against the (reference-pinned) MPQ unpacker, modulo the fused rounding."""
    rng = np.random.default_rng(0)
    K, N, gs = 128, 32, 32
    for bits in (2, 4, 8):
        qw = rng.integers(-2 ** 31, 2 ** 31 - 1, (K * bits // 32, N), dtype=np.int64).astype(np.int32)
        groups = K // gs
        qg = np.zeros(2 * groups, np.int16)
        for i in range(groups):
            qg[2 * i], qg[2 * i + 1] = bits, i * gs * bits // 32
        s = orc.np.float32(0.01) * np.ones((groups, N), np.float32)
        import ctypes
        s16 = np.empty(s.shape, np.uint16)
        orc.lib().orc_f32_to_f16(s.ctypes.data_as(ctypes.c_void_p), s16.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(s.size))
        z16 = np.zeros_like(s16)
        a = orc.exl2_dequant(qw, s16, z16, None, qg, K)
        b = orc.mpq_dequant(qw, s16, z16, None, bits, gs, 0, orc.F16)  # z = 0 -> one rounding both ways
        assert np.array_equal(a, b)
        if bits in (2, 4):
            c = orc.mbwq_q4_dequant(qw, s16, z16, None, bits, gs)
            assert np.array_equal(a, c)


def test_binary_linear_vs_reference_cpp():
    d = np.load(os.path.join(GOLDEN, "binary_linear_cpp.npz"))
    for tag in ("M1N64K128", "M4N96K256", "M33N40K64"):
        x, w, y, wp = d[tag + "_x"], d[tag + "_w"], d[tag + "_y"], d[tag + "_wpacked"]
        mine = orc.binary_pack_cols(w)
        assert np.array_equal(mine, wp), "w_pack layout mismatch"
        assert np.array_equal(orc.binary_linear(x, mine, w.shape[0]), y)
        xb, wb = orc.binary_pack_rows(x), orc.binary_pack_rows(w)
        assert np.array_equal(orc.binary_linear_rowpacked(xb, wb, x.shape[1]), y)


def test_binary_conv_vs_reference_cpp():
    d = np.load(os.path.join(GOLDEN, "binary_conv_cpp.npz"))
    tags = sorted({k.rsplit("_", 1)[0] for k in d.files})
    assert len(tags) == 4
    for tag in tags:
        st = int(tag.split("s")[1].split("p")[0])
        pad = int(tag.split("p")[1].split("d")[0])
        dil = int(tag.split("d")[1])
        y = orc.binary_conv2d(d[tag + "_x"], d[tag + "_w"], st, pad, dil)
        assert np.array_equal(y, d[tag + "_y"]), tag


def test_unpack_uint8_known_answer():
    d = np.load(os.path.join(GOLDEN, "kat_unpack_uint8.npz"))
    out = orc.unpack_u8_scaled(d["bytes"].reshape(1, 4), np.array([1.0], np.float32))
    assert np.array_equal(out.reshape(-1), d["expected"])
    assert np.array_equal(orc.pack_sign_u8(d["expected"].reshape(1, 32)).reshape(-1), d["bytes"])


def test_q4_pack_roundtrip_and_sign_extension():
    rng = np.random.default_rng(1)
    a = rng.integers(-8, 8, (10, 10)).astype(np.int32)
    p = orc.q4_pack(a)
    assert p.dtype == np.int8 and p.size * 2 == a.size
    u = orc.q4_unpack(p)
    assert np.array_equal(u & 0xF, a & 0xF)
    assert np.array_equal(orc.q4_unpack_scale(p, 0.5), a.astype(np.float32) * 0.5)
    # first element lands in the HIGH nibble (functions_cuda_kernel.cu:137-159)
    assert orc.q4_pack(np.array([[1, 2]], np.int32)).view(np.uint8)[0, 0] == 0x12


# ------------------------------------------------------------------ W4A4 / W8A8
def _unpack_hi_lo(p):
    u = p.astype(np.uint8)
    hi, lo = (u >> 4).astype(np.int32), (u & 15).astype(np.int32)
    hi[hi > 7] -= 16
    lo[lo > 7] -= 16
    return np.stack([hi, lo], axis=-1).reshape(p.shape[:-1] + (p.shape[-1] * 2,))


def test_q4_quantize_pack_matches_reference_quantiser():
    z = np.load(os.path.join(GOLDEN, "q4_q8_quantization.npz"))
    x = z["x"]
    for scale, want in ((float(z["scale4"]), z["q4_derived"]), (float(z["scale_given"]), z["q4_given"])):
        got = _unpack_hi_lo(orc.q4_quantize_pack(x, scale, orc.F32))
        r = x / np.float32(scale)
        keep = (r - np.floor(r)) != 0.5  # CUDA roundf (half away) vs torch.round (half even) differ only on ties
        assert np.array_equal(got[keep], want.astype(np.int32)[keep])


def test_q4_q8_gemm_oracle_is_exact_integer_gemm():
    rng = np.random.default_rng(5)
    M, N, K = 7, 12, 128
    a4, w4 = rng.integers(-8, 8, (M, K)), rng.integers(-8, 8, (N, K))
    pa = ((a4[:, 0::2] & 15) << 4 | (a4[:, 1::2] & 15)).astype(np.uint8).view(np.int8)
    pw = ((w4[:, 0::2] & 15) << 4 | (w4[:, 1::2] & 15)).astype(np.uint8).view(np.int8)
    y = orc.q4_gemm(pa, pw, K, 0.5, 0.25, orc.F32)
    assert np.array_equal(y, (a4 @ w4.T).astype(np.float32) * np.float32(0.125))
    a8, w8 = rng.integers(-128, 128, (M, K)).astype(np.int8), rng.integers(-128, 128, (N, K)).astype(np.int8)
    y8 = orc.q8_gemm(a8, w8, 0.5, 0.25)
    assert np.array_equal(y8, (a8.astype(np.int64) @ w8.astype(np.int64).T).astype(np.float32) * np.float32(0.125))


def test_q4_conv2d_oracle_equals_torch_conv_on_the_integers():
    import torch
    rng = np.random.default_rng(9)
    for (B, C, H, OC, ks, st, pad, dil) in ((2, 8, 6, 4, 3, 1, 1, 1), (1, 16, 7, 8, 3, 2, 1, 1), (1, 8, 5, 4, 1, 1, 0, 1), (1, 8, 9, 4, 3, 1, 2, 2)):
        a = rng.integers(-8, 8, (B, H, H, C))
        w = rng.integers(-8, 8, (OC, ks, ks, C))
        pa = (((a[..., 0::2] & 15) << 4) | (a[..., 1::2] & 15)).astype(np.uint8).view(np.int8)
        pw = (((w[..., 0::2] & 15) << 4) | (w[..., 1::2] & 15)).astype(np.uint8).view(np.int8)
        y = orc.q4_conv2d(pa, pw, ks, st, pad, dil, 0.5, 0.5, orc.F32)
        ref = torch.nn.functional.conv2d(torch.from_numpy(a).permute(0, 3, 1, 2).double(), torch.from_numpy(w).permute(0, 3, 1, 2).double(),
                                         stride=st, padding=pad, dilation=dil).permute(0, 2, 3, 1).numpy()
        assert np.array_equal(y, (ref * 0.25).astype(np.float32))


def test_btc_bstc_image_oracle_against_a_thread_by_thread_emulation():
    """The vectorised oracle of the CUDA layer's packed-weight images (oracle.binary_pack_btc32 / _bstc32) against a literal
    emulation of the launch geometry the reference uses (binary_linear_cuda_kernel.cu:59-152 BMMA_toBit32Col_new
    <<<(K/128, N/8), (32, 4, 8)>>>, :186-203 ToBit32RowUd<<<(K/32, N/32), 32>>>, :33-41 uint32_to_uint8).  The CUDA kernels
    cannot run here; this pins the restatement to the index arithmetic of the source, thread by thread."""
    rng = np.random.default_rng(3)
    N, K = 64, 256
    w = rng.standard_normal((N, K)).astype(np.float32)
    A = w.T  # the layer packs weight.t(): A[k][n], A_height = K, A_width = N

    def brev_ballot(vals):  # lane i -> bit 31 - i
        v = 0
        for i, f in enumerate(vals):
            v |= int(f >= 0) << (31 - i)
        return v

    def to_bytes(words):  # uint32_to_uint8: out[4j + b] = (in[j] >> ((3 - b) * 8)) & 0xff
        out = bytearray()
        for x in words:
            out += bytes([(x >> 24) & 255, (x >> 16) & 255, (x >> 8) & 255, x & 255])
        return bytes(out)

    btc = [0] * (N * K // 32)
    gx = K // 128
    for bx in range(K // 128):
        for by in range(N // 8):
            for wx in range(4):
                for wy in range(8):
                    btc[(by * gx + bx) * 32 + wy * 4 + wx] = brev_ballot([A[bx * 128 + wx * 32 + lane][by * 8 + wy] for lane in range(32)])
    assert to_bytes(btc) == orc.binary_pack_btc32(w)

    bstc = [0] * (N * K // 32)
    gy = N // 32
    for bx in range(K // 32):
        for by in range(N // 32):
            for lane in range(32):
                val = 0
                for i in range(32):
                    val = ((val << 1) + int(A[bx * 32 + i][by * 32 + lane] >= 0)) & 0xffffffff
                bstc[bx * gy * 32 + by * 32 + lane] = val
    assert to_bytes(bstc) == orc.binary_pack_bstc32(w)


def test_cutlass_conv_reference_convention_restatement_on_a_case_small_enough_to_read():
    """oracle.binary_conv2d_cutlass_reference_convention (parity UNPINNED: the reference kernel needs CUDA + CUTLASS): a 1x1 filter makes the
    convention readable -- output (b, p, q, o) = popcount over the C/8 bits starting at flat bit ((b*H + p)*W + q) * C/8 of the NCHW memory's
    sign bits XOR the C/8 bits starting at o * C/8 of the weight memory's (binary_conv2d_cutlass_kernel.cu:206-228,271,438,453)."""
    rng = np.random.default_rng(5)
    B, C, H, W, OC = 2, 1024, 2, 3, 3
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    w = rng.standard_normal((OC, C, 1, 1)).astype(np.float32)
    y = orc.binary_conv2d_cutlass_reference_convention(x, w, 0.5, 1, 1, 0, 1)
    c8 = C // 8
    oe = W
    xb, wb = (x.reshape(-1) >= 0), (w.reshape(-1) >= 0)
    assert y.shape == (B, oe, oe, OC) and y.dtype == np.float32
    for b in range(B):
        for p in range(H):  # rows p >= H of the (W x W) output read zero bits only
            for q in range(W):
                a = xb[((b * H + p) * W + q) * c8:][:c8]
                for o in range(OC):
                    assert y[b, p, q, o] == 0.5 * np.count_nonzero(a ^ wb[o * c8:(o + 1) * c8])
    for o in range(OC):  # H = 2 < out_edge = 3: the last output row lies outside the image -> popcount of the filter bits alone
        assert np.all(y[:, 2, :, o] == 0.5 * np.count_nonzero(wb[o * c8:(o + 1) * c8]))
