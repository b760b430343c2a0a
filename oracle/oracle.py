"""ctypes front-end of the CPU oracle (oracle/bie_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (bitorch-engine_amd/) never imports this module.

All functions take / return numpy arrays.  16-bit float tensors travel as uint16 bit patterns
(`torch_to_np` / `np_to_torch` convert to and from torch tensors).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("BIE_ORACLE_LIB", os.path.join(_HERE, "libbie_oracle.so"))  # BIE_ORACLE_LIB: e.g. the sanitizer build (make -C oracle asan)
F16, BF16, F32 = 0, 1, 2
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement (and oracle/_ref when /root/reference is present)."""
    src = os.path.join(_HERE, "bie_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, _LIB_PATH])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a)
    if dtype is not None and a.dtype != dtype:
        a = a.astype(dtype)
    return a


def dt_code(torch_dtype) -> int:
    import torch
    return {torch.float16: F16, torch.bfloat16: BF16, torch.float32: F32}[torch_dtype]


def torch_to_np(t):
    """torch tensor -> numpy (fp16/bf16 as uint16 bit patterns)."""
    import torch
    if t is None:
        return None
    t = t.detach().cpu().contiguous()
    if t.dtype in (torch.float16, torch.bfloat16):
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.numpy()


def np_to_torch(a, torch_dtype):
    import torch
    if torch_dtype in (torch.float16, torch.bfloat16):
        return torch.from_numpy(a.view(np.int16).copy()).view(torch_dtype)
    return torch.from_numpy(a.copy())


def _out(shape, dt):
    return np.empty(shape, dtype=np.float32 if dt == F32 else np.uint16)


def mpq_dequant(qweight, scales, zeros, g_idx, w_bit, group_size, asym, dt):
    qweight = _c(qweight, np.int32)
    K = qweight.shape[0] * 32 // w_bit
    N = qweight.shape[1]
    scales, zeros, g_idx = _c(scales), _c(zeros), _c(g_idx, np.int32)
    out = _out((K, N), dt)
    lib().orc_mpq_dequant(_p(qweight), _p(scales), _p(zeros), _p(g_idx), _p(out), K, N, w_bit,
                          group_size, int(asym), dt)
    return out


def mpq_pack(weight, scales, zeros, g_idx, w_bit, group_size, asym, dt):
    weight = _c(weight)
    K, N = weight.shape
    scales, zeros, g_idx = _c(scales), _c(zeros), _c(g_idx, np.int32)
    out = np.empty((K * w_bit // 32, N), dtype=np.int32)
    lib().orc_mpq_pack(_p(weight), _p(scales), _p(zeros), _p(g_idx), _p(out), K, N, w_bit,
                       group_size, int(asym), dt)
    return out


def pack_qzeros(zq, w_bit):
    zq = _c(zq, np.int32)
    G, N = zq.shape
    out = np.empty((G, N * w_bit // 32), dtype=np.int32)
    lib().orc_pack_qzeros(_p(zq), _p(out), G, N, w_bit)
    return out


def gemm(x, W, dt, bias=None):
    x, W, bias = _c(x), _c(W), _c(bias)
    M, K = x.shape
    N = W.shape[1]
    y = _out((M, N), dt)
    lib().orc_gemm(_p(x), _p(W), _p(bias), _p(y), M, K, N, dt)
    return y


def mpq_grad_input(gy, qweight, scales, zeros, g_idx, w_bit, group_size, asym, dt):
    """grad_x[M, K] = grad_y[M, N] . W^T with W the dequantised weight -- the spec line of the reference's backward
    (layers/qlinear/nbit/cuda/mpq_layer.py:107-111: "grad_input = output_gradient.mm(weight)"; the CUDA kernel itself,
    back_quant_mm_kernel mpq_linear_cuda_kernel.cu:635-1049, is not runnable here).  fp32 accumulation, one rounding."""
    W = mpq_dequant(qweight, scales, zeros, g_idx, w_bit, group_size, asym, dt)
    return gemm(gy, np.ascontiguousarray(W.T), dt)


def mpq_forward(x, qweight, scales, zeros, g_idx, w_bit, group_size, asym, dt):
    """Fused dequant+GEMM with float accumulation (the timed CPU baseline)."""
    x, qweight = _c(x), _c(qweight, np.int32)
    M, K = x.shape
    N = qweight.shape[1]
    scales, zeros, g_idx = _c(scales), _c(zeros), _c(g_idx, np.int32)
    y = _out((M, N), dt)
    lib().orc_mpq_forward_f32acc(_p(x), _p(qweight), _p(scales), _p(zeros), _p(g_idx), _p(y), M, K, N,
                                 w_bit, group_size, int(asym), dt)
    return y


class _ListEntry(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("qweight", ctypes.c_void_p), ("scales", ctypes.c_void_p), ("zeros", ctypes.c_void_p),
                ("g_idx", ctypes.c_void_p), ("y", ctypes.c_void_p), ("K", ctypes.c_int), ("N", ctypes.c_int)]


def mpq_forward_list(layers, M, w_bit, group_size, asym, dt):
    """orc_mpq_forward_list_f32acc: several layers (x, qweight, scales, zeros) in ONE statically partitioned OpenMP region -- the CPU twin of
    the GPU's list launch (work items = 64-column blocks of every layer).  Returns the list of outputs."""
    arr = (_ListEntry * len(layers))()
    keep, ys = [], []
    for i, (x, qweight, scales, zeros) in enumerate(layers):
        x, qweight, scales, zeros = _c(x), _c(qweight, np.int32), _c(scales), _c(zeros)
        K, N = x.shape[1], qweight.shape[1]
        y = _out((M, N), dt)
        arr[i] = _ListEntry(x.ctypes.data, qweight.ctypes.data, scales.ctypes.data, zeros.ctypes.data, None, y.ctypes.data, K, N)
        keep.append((x, qweight, scales, zeros))
        ys.append(y)
    lib().orc_mpq_forward_list_f32acc(len(layers), arr, M, w_bit, group_size, int(asym), dt)
    return ys


def mbwq_q4_dequant(qweight, scales, zeros, q_perm, bits, group_size):
    qweight = _c(qweight, np.int32)
    K = qweight.shape[0] * 32 // bits
    N = qweight.shape[1]
    scales, zeros = _c(scales, np.uint16), _c(zeros, np.uint16)
    q_perm = None if q_perm is None else _c(q_perm).view(np.uint16)
    out = np.zeros((K, N), dtype=np.uint16)
    lib().orc_mbwq_q4_dequant(_p(qweight), _p(scales), _p(zeros), _p(q_perm), _p(out), K, N, bits, group_size)
    return out


def exl2_rows(q_groups, K):
    q_groups = _c(q_groups, np.int16)
    rows = (ctypes.c_int * 7)()
    lib().orc_exl2_rows(_p(q_groups), q_groups.size // 2, K, rows)
    return list(rows)


def exl2_dequant(qweight, scales, zeros, q_perm, q_groups, K):
    qweight = _c(qweight, np.int32)
    N = qweight.shape[1]
    scales, zeros = _c(scales, np.uint16), _c(zeros, np.uint16)
    q_groups = _c(q_groups, np.int16)
    q_perm = None if q_perm is None else _c(q_perm).view(np.uint16)
    out = np.zeros((K, N), dtype=np.uint16)
    lib().orc_exl2_dequant(_p(qweight), _p(scales), _p(zeros), _p(q_perm), _p(q_groups), _p(out), K, N,
                           q_groups.size // 2, qweight.shape[0])
    return out


def binary_pack_rows(a):
    a = _c(a, np.float32)
    rows, K = a.shape
    out = np.empty((rows, K // 8), dtype=np.uint8)
    lib().orc_binary_pack_rows(_p(a), _p(out), ctypes.c_long(rows), ctypes.c_long(K))
    return out


def _msb_words(bits_kn):
    """bits [K, N] (k-major, the transposed weight the reference packs) -> uint32 [K/32, N]: bit 31 = first k of the block
    (__brev(__ballot) in BMMA_toBit32Col_new, Bval << 1 in ToBit32RowUd)."""
    K, N = bits_kn.shape
    b = bits_kn.reshape(K // 32, 32, N).astype(np.uint64)
    sh = np.arange(31, -1, -1, dtype=np.uint64).reshape(1, 32, 1)
    return (b << sh).sum(axis=1).astype(np.uint32)


def binary_pack_bstc32(w):
    """Reference BSTC32 image of w [N, K]: ToBit32RowUd<<<(K/32, N/32), 32>>> writes word [k/32][n]
    (binary_linear_cuda_kernel.cu:186-203, B[bx*gridDim.y*32 + by*32 + laneid]); bytes big-endian (uint32_to_uint8 :33-41)."""
    bits = (np.asarray(w, np.float32).T >= 0)
    return _msb_words(bits).astype(">u4").tobytes()


def binary_pack_btc32(w):
    """Reference BTC32 image of w [N, K]: BMMA_toBit32Col_new<<<(K/128, N/8), (32, 4, 8)>>> writes the word of 32 k
    (k = bx*128 + wx*32 ..) and column n = by*8 + wy at (by*gridDim.x + bx)*32 + wy*4 + wx (binary_linear_cuda_kernel.cu:59-152)."""
    bits = (np.asarray(w, np.float32).T >= 0)
    K, N = bits.shape
    words = _msb_words(bits)  # [K/32, N]
    tiles = words.reshape(K // 128, 4, N // 8, 8)      # [bx, wx, by, wy]
    out = np.ascontiguousarray(tiles.transpose(2, 0, 3, 1))  # [by, bx, wy, wx]
    return out.astype(">u4").tobytes()


def binary_pack_cols(w):
    w = _c(w, np.float32)
    N, K = w.shape
    out = np.empty((K // 8) * N, dtype=np.uint8)
    lib().orc_binary_pack_cols(_p(w), _p(out), ctypes.c_long(N), ctypes.c_long(K))
    return out


def binary_linear(x, wpacked, N):
    x = _c(x, np.float32)
    M, K = x.shape
    wpacked = _c(wpacked, np.uint8)
    y = np.empty((M, N), dtype=np.float32)
    lib().orc_binary_linear(_p(x), _p(wpacked), _p(y), ctypes.c_long(M), ctypes.c_long(N), ctypes.c_long(K))
    return y


def binary_linear_rowpacked(xb, wb, K, scale=1.0):
    xb, wb = _c(xb, np.uint8), _c(wb, np.uint8)
    M, N = xb.shape[0], wb.shape[0]
    y = np.empty((M, N), dtype=np.float32)
    lib().orc_binary_linear_rowpacked(_p(xb), _p(wb), _p(y), ctypes.c_long(M), ctypes.c_long(N),
                                      ctypes.c_long(K), ctypes.c_float(scale))
    return y


def binary_conv2d(x, w, stride, pad, dil):
    x, w = _c(x, np.float32), _c(w, np.float32)
    B, C, H, W = x.shape
    OC, _, ksz, _ = w.shape
    OH = (H + 2 * pad - dil * (ksz - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (ksz - 1) - 1) // stride + 1
    y = np.empty((B, OC, OH, OW), dtype=np.float32)
    lib().orc_binary_conv2d(_p(x), _p(w), _p(y), B, C, H, W, OC, ksz, stride, pad, dil)
    return y


def binary_conv2d_cutlass_reference_convention(x, w, scale, ksz, stride, pad, dil):
    """What the reference's BinaryConv2dCutlass kernel computes, read off its source (PARITY UNPINNED: the kernel needs CUDA + CUTLASS, the
    reference's own test only compares packed with unpacked weights after a layer norm, tests/layers/test_binary_conv.py:157-170).
    layers/qconv/binary/cutlass/binary_conv2d_cutlass_kernel.cu:
      :438      the NCHW input is VIEWED (not permuted) as [B, H, W, C]; :430 / :474 the weights as [OC, k, k, C]
      :64-117   bits = (value >= 0), eight consecutive elements of that memory order per byte, LSB first (:325-345: packed shape [.., C/8])
      :206-228  the PACKED tensors' sizes are handed to CUTLASS as the extents of one-bit tensors: it sees C/8 one-bit channels per pixel with
                packed NHWC strides, i.e. it walks the FIRST B*H*W*C/8 (OC*k*k*C/8) bits of each buffer
      :260      Mode::kConvolution: the filter is flipped (tap (r, s) meets input offset (k-1-r, k-1-s) * dilation)
      :142-147,:271  int32 accumulator of popcount(a XOR w), alpha 1 / beta 0, no K - 2*popc; padded positions read as zero bits
      :414      out_edge = (W - k + 2*pad) / stride + 1 for BOTH output extents (the dilation is not in it)
      :419-423,:453  output [B, out_edge, out_edge, OC] (NHWC, not permuted back), int32 * scale -> float32
    x [B, C, H, W], w [OC, C, k, k] values (numpy); returns float32 [B, out_edge, out_edge, OC]."""
    x = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    w = np.ascontiguousarray(np.asarray(w, dtype=np.float32))
    B, C, H, W = x.shape
    OC = w.shape[0]
    assert C % 8 == 0
    C8 = C // 8
    oe = (W - ksz + 2 * pad) // stride + 1
    abits = (x.reshape(-1) >= 0)[: B * H * W * C8].reshape(B, H, W, C8).astype(np.int64)
    wbits = (w.reshape(-1) >= 0)[: OC * ksz * ksz * C8].reshape(OC, ksz, ksz, C8).astype(np.int64)
    out = np.zeros((B, oe, oe, OC), dtype=np.int64)
    for r in range(ksz):
        for s_ in range(ksz):
            f = wbits[:, r, s_, :]                                # [OC, C8]
            plane = np.zeros((B, oe, oe, C8), dtype=np.int64)      # zero bits where the tap falls outside the image
            for p in range(oe):
                h = p * stride - pad + (ksz - 1 - r) * dil
                if h < 0 or h >= H:
                    continue
                for q in range(oe):
                    ww = q * stride - pad + (ksz - 1 - s_) * dil
                    if 0 <= ww < W:
                        plane[:, p, q, :] = abits[:, h, ww, :]
            # popcount(a ^ f) over the channel bits = sum a + sum f - 2 a.f
            out += plane.sum(-1, keepdims=True) + f.sum(-1)[None, None, None, :] - 2 * np.einsum("bpqc,oc->bpqo", plane, f)
    return out.astype(np.float32) * np.float32(scale)


def pack_sign_u8(a):
    a = _c(a, np.float32)
    out = np.empty(a.size // 8, dtype=np.uint8)
    lib().orc_pack_sign_u8(_p(a), _p(out), ctypes.c_long(out.size))
    return out.reshape(a.shape[:-1] + (a.shape[-1] // 8,))


def unpack_u8_scaled(packed, scale):
    packed = _c(packed, np.uint8)
    rows = int(np.prod(packed.shape[:-1]))
    pd = packed.shape[-1]
    scale = _c(np.broadcast_to(np.asarray(scale, dtype=np.float32).reshape(-1), (rows,)), np.float32)
    out = np.empty((rows, pd * 8), dtype=np.float32)
    lib().orc_unpack_u8_scaled(_p(packed), _p(scale), _p(out), ctypes.c_long(rows), ctypes.c_long(pd))
    return out.reshape(packed.shape[:-1] + (pd * 8,))


def q4_pack(a):
    a = _c(a, np.int32)
    out = np.empty(a.size // 2, dtype=np.int8)
    lib().orc_q4_pack(_p(a), _p(out), ctypes.c_long(out.size))
    return out.reshape(a.shape[:-1] + (a.shape[-1] // 2,))


def q4_unpack(p):
    p = _c(p, np.int8)
    out = np.empty(p.size * 2, dtype=np.int32)
    lib().orc_q4_unpack(_p(p), _p(out), ctypes.c_long(p.size))
    return out.reshape(p.shape[:-1] + (p.shape[-1] * 2,))


def q4_unpack_scale(p, scale):
    p = _c(p, np.int8)
    out = np.empty(p.size * 2, dtype=np.float32)
    lib().orc_q4_unpack_scale(_p(p), _p(out), ctypes.c_long(p.size), ctypes.c_float(scale))
    return out.reshape(p.shape[:-1] + (p.shape[-1] * 2,))


def q4_quantize_pack(x, scale, dt):
    x = _c(x)
    out = np.empty(x.shape[:-1] + (x.shape[-1] // 2,), dtype=np.int8)
    lib().orc_q4_quantize_pack(_p(x), _p(out), ctypes.c_long(out.size), ctypes.c_float(scale), dt)
    return out


def q4_gemm(a, w, K, scale_a, scale_w, dt):
    a, w = _c(a, np.int8), _c(w, np.int8)
    M, N = a.shape[0], w.shape[0]
    y = _out((M, N), dt)
    lib().orc_q4_gemm(_p(a), _p(w), _p(y), M, N, K, ctypes.c_float(scale_a), ctypes.c_float(scale_w), dt)
    return y


def q8_gemm(a, w, scale_a, scale_w):
    a, w = _c(a, np.int8), _c(w, np.int8)
    M, K = a.shape
    N = w.shape[0]
    y = np.empty((M, N), dtype=np.float32)
    lib().orc_q8_gemm(_p(a), _p(w), _p(y), M, N, K, ctypes.c_float(scale_a), ctypes.c_float(scale_w))
    return y


def q4_conv2d(a, w, ksize, stride, pad, dil, scale_a, scale_w, dt):
    """a: int8 [B, H, W, C/2], w: int8 [OC, KS, KS, C/2] -> [B, OH, OW, OC]"""
    a, w = _c(a, np.int8), _c(w, np.int8)
    B, H, W, C2 = a.shape
    OC = w.shape[0]
    OH = (H + 2 * pad - dil * (ksize - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (ksize - 1) - 1) // stride + 1
    y = _out((B, OH, OW, OC), dt)
    lib().orc_q4_conv2d(_p(a), _p(w), _p(y), B, H, W, C2 * 2, OC, ksize, stride, pad, dil, ctypes.c_float(scale_a),
                        ctypes.c_float(scale_w), dt)
    return y
