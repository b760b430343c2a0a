"""Drop-in for `q_linear_cutlass` (layers/qlinear/nbit/cutlass/q_linear_cutlass.cpp:368-375): W4A4 / W8A8 linear on the
i8 matrix cores.  q4_forward / q4_w_pack / q4_mm / q4_matmul / q8_forward and the backward entry points q4_backward /
q4_matmul_backward / q8_backward (the same integer GEMM on the saved packed operands, raw int32 accumulators)."""
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


# ---------------------------------------------------------------------------------------------- backward entry points
def _int_gemm_i32(a, w, M, N, K, bits, batch=1):
    """int32 [batch, M, N] = a[batch, M, K] . w[batch, N, K]^T on the i8 matrix cores (raw accumulators)."""
    _hip.need_gpu(a, w)
    a, w = a.contiguous(), w.contiguous()
    y = torch.empty((batch, M, N) if batch > 1 else (M, N), dtype=torch.int32, device=a.device)
    row = K // 2 if bits == 4 else K
    rc = _hip.lib().bie_int_gemm_i32(_hip.ptr(a), _hip.ptr(w), _hip.ptr(y), M, N, K, bits, batch, M * row, N * row, M * N, _hip.stream())
    _hip.check(rc, "bie_int_gemm_i32")
    return y


def q4_backward(output_gradient, input_q4, weight_q4, scale_a, scale_w, scale_grad):
    """(grad_a [m, k], grad_w [n, k]) exactly as the reference computes them (q4_linear_cutlass_kernel.cu:719-743): the
    gradient is quantised to 4 bit with scale_grad, then two 4-bit GEMMs run on the SAVED packed operands taken as they lie in
    memory -- q4_gemm(m, k, n, grad, q4_w) reads the [n, k/2] weight buffer as a [k, n/2] operand, q4_gemm(n, k, m, grad^T, q4_a)
    reads the transposed gradient BYTES as [n, m/2] and the [m, k/2] activation buffer as [k, m/2] -- and the int32 results are
    scaled by scale_a / scale_w.  (A reinterpretation, not a transpose: this mirror reproduces the arithmetic, not a derivation.)"""
    m, k = input_q4.shape[0], input_q4.shape[1] * 2
    n = weight_q4.shape[0]
    g = q4_w_pack(output_gradient.reshape(m, n), _f(scale_grad), False)            # [m, n/2]
    grad_a = _int_gemm_i32(g, weight_q4.contiguous().reshape(k, n // 2), m, k, n, 4)
    grad_w = _int_gemm_i32(g.t().contiguous().reshape(n, m // 2), input_q4.contiguous().reshape(k, m // 2), n, k, m, 4)
    return grad_a * _f(scale_a), grad_w * _f(scale_w)


def q4_matmul_backward(output_gradient, x, y, scale_x, scale_y, scale_grad):
    """Batched twin of q4_backward on the packed operands of q4_matmul (reference :901-941): x [.., m, k/2], y [.., n, k/2]."""
    k, m, n = x.shape[-1] * 2, x.shape[-2], y.shape[-2]
    xb = x.contiguous().reshape(-1, m, k // 2)
    bs = xb.shape[0]
    g = q4_w_pack(output_gradient.reshape(bs, m, n), _f(scale_grad), False)      # [bs, m, n/2]
    grad_x = _int_gemm_i32(g, y.contiguous().reshape(bs, k, n // 2), m, k, n, 4, bs)
    grad_y = _int_gemm_i32(g.transpose(-1, -2).contiguous().reshape(bs, n, m // 2), xb.reshape(bs, k, m // 2), n, k, m, 4, bs)
    return grad_x * _f(scale_x), grad_y * _f(scale_y)


def q8_backward(output_gradient, input_q8, weight_q8):
    """(grad_a, grad_w) int32 as the reference returns them (q8_linear_cutlass_kernel.cu:283-308): gradient cast to int8, q8_gemm on
    the saved int8 operands as they lie in memory ([n, k] weight read as [k, n]; [m, k] activations read as [k, m])."""
    if input_q8.dtype != torch.int8 or weight_q8.dtype != torch.int8:
        raise RuntimeError("Error: the dtype of activation or weight tensor must be int8!")
    g = output_gradient if output_gradient.dtype == torch.int8 else output_gradient.to(torch.int8)
    m, k = input_q8.shape
    n = weight_q8.shape[0]
    grad_a = _int_gemm_i32(g.reshape(m, n), weight_q8.contiguous().reshape(k, n), m, k, n, 8)
    grad_w = _int_gemm_i32(g.reshape(m, n).t().contiguous(), input_q8.contiguous().reshape(k, m), n, k, m, 8)
    return grad_a, grad_w
