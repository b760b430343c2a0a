"""Drop-in for `q_linear_cutlass` (layers/qlinear/nbit/cutlass/q_linear_cutlass.cpp:368-375): W4A4 / W8A8 linear on the
i8 matrix cores.  q4_forward / q4_w_pack / q4_mm / q4_matmul / q8_forward (the backward entry points run the same GEMM
on the saved packed operands)."""
import torch

from bitorch_engine import _hip


def _f(v):
    return float(v.item()) if torch.is_tensor(v) else float(v)


def q4_w_pack(weight: torch.Tensor, scale, is_transpose: bool = False) -> torch.Tensor:
    """float [.., n, k] -> int8 [.., n, k/2]: clamp(round(w / scale), -8, 7), first value in the high nibble."""
    _hip.need_gpu(weight)
    w = (weight.transpose(-1, -2) if is_transpose else weight).contiguous()
    out = torch.empty(w.shape[:-1] + (w.shape[-1] // 2,), dtype=torch.int8, device=w.device)
    rc = _hip.lib().bie_q4_quantize_pack(_hip.ptr(w), _hip.ptr(out), out.numel(), _f(scale), _hip.dt(w), _hip.stream())
    _hip.check(rc, "bie_q4_quantize_pack")
    return out


def _q4_gemm(pa, pw, M, N, K, sa, sw, dtype, batch=1):
    _hip.need_gpu(pa, pw)
    y = torch.empty((batch, M, N) if batch > 1 else (M, N), dtype=dtype, device=pa.device)
    rc = _hip.lib().bie_q4_gemm(_hip.ptr(pa), _hip.ptr(pw), _hip.ptr(y), M, N, K, sa, sw, _hip._DT[dtype], batch,
                                M * (K // 2), N * (K // 2), M * N, _hip.stream())
    _hip.check(rc, "bie_q4_gemm")
    return y


def q4_forward(input, weight, scale_a, scale_w, transpose, is_train):
    """-> [output (input dtype), packed activations, packed weights]"""
    sa, sw = _f(scale_a), _f(scale_w)
    m, k = input.shape
    packed_w = weight if weight.dtype == torch.int8 else q4_w_pack(weight, sw, transpose)
    packed_a = q4_w_pack(input, sa, False)
    out = _q4_gemm(packed_a, packed_w.contiguous(), m, packed_w.shape[0], k, sa, sw, input.dtype)
    return [out, packed_a, packed_w]


def q4_mm(x, w, scale_a, scale_w):
    return q4_forward(x, w, scale_a, scale_w, False, True)[0]


def q4_matmul(x, y, x_clip, y_clip):
    """Batched x [.., M, K] . y [.., N, K]^T on 4-bit quantised operands; returns [int-valued output, q4_x, q4_y] like the
    reference (the caller multiplies by the clips)."""
    sx, sy = _f(x_clip), _f(y_clip)
    M, K = x.shape[-2:]
    N = y.shape[-2]
    px, py = q4_w_pack(x, sx), q4_w_pack(y, sy)
    batch = x.numel() // (M * K)
    out = _q4_gemm(px, py, M, N, K, 1.0, 1.0, x.dtype, batch)
    return [out.reshape(x.shape[:-2] + (M, N)), px, py]


def q8_forward(q_a, q_w, transpose, scale_a, scale_w):
    _hip.need_gpu(q_a, q_w)
    q_w = (q_w.t() if transpose else q_w).contiguous()
    q_a = q_a.contiguous()
    M, K = q_a.shape
    N = q_w.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=q_a.device)
    rc = _hip.lib().bie_q8_gemm(_hip.ptr(q_a), _hip.ptr(q_w), _hip.ptr(y), M, N, K, _f(scale_a), _f(scale_w), _hip.stream())
    _hip.check(rc, "bie_q8_gemm")
    return y
