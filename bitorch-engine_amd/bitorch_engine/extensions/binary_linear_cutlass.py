"""Drop-in for `binary_linear_cutlass` (layers/qlinear/binary/cutlass/binary_linear_cutlass.cpp:206-210), same positional
signatures: forward(input, weight, scale, transpose, kernel_id), w_pack(weight, transpose), mm(x, y, kernel_id),
matmul(x, y, scale), kernel_eval(device_id, m, n, k).  Row-packed uint8 LSB-first operands
(binary_linear_cutlass_kernel.cu:44-90), epilogue (K - 2*popc) * scale (:93-113)."""
import torch

from bitorch_engine import _hip
from ._binary_common import pack_rows, xnor_linear, fp4_ok, xnor_values_fp4


def w_pack(weight: torch.Tensor, transpose: bool) -> torch.Tensor:
    return pack_rows(weight.t() if transpose else weight)


def forward(input: torch.Tensor, weight: torch.Tensor, scale: float, transpose: bool, kernel_id: int) -> torch.Tensor:
    m, k = input.shape
    if weight.dtype == torch.uint8:
        wp = weight.contiguous()
    elif weight.requires_grad and torch.is_grad_enabled():  # under training: `.data` writes (optimiser, clamp) do not advance the version counter -- pack every call
        wp = w_pack(weight, transpose).contiguous()
    else:  # frozen unpacked weight: packed once per tensor version (and with it the FP4 image memoised on the packed tensor)
        from .q_linear_cuda import _cached
        wp = _cached(weight, ("rows_from_values", bool(transpose)), lambda: w_pack(weight, transpose).contiguous())
    if fp4_ok(m, wp.shape[0], k):  # large M: sign-pack folded into the FP4 image pass, GEMM on the matrix pipe
        return xnor_values_fp4(input, wp, scale)
    return xnor_linear(pack_rows(input), wp, m, wp.shape[0], k, 0, scale)


def mm(x: torch.Tensor, y: torch.Tensor, kernel_id: int) -> torch.Tensor:
    """x [m, k], y [n, k] -> int32 [m, n]: the RAW accumulator of the reference's CUTLASS XOR-popcount GEMM, i.e. the number of
    positions where sign(x) and sign(y) differ (binary_mm_function -> binary_forward_cutlass, binary_linear_cutlass_kernel.cu:293-332,
    650-666: no `k - 2 * out` alignment, no scale).  `kernel_id` selected a CUTLASS tiling in the reference; the result does not
    depend on it and this build has one tiling policy, so it is accepted and ignored."""
    m, k = x.shape
    n = y.shape[0]
    dots = xnor_linear(pack_rows(x), pack_rows(y), m, n, k, 0, 1.0)  # k - 2 * popc, exact integers in fp32
    return ((k - dots) * 0.5).to(torch.int32)


def matmul(x: torch.Tensor, y: torch.Tensor, scale: float) -> torch.Tensor:
    """Batched x [..., M, K] . y [..., N, K]^T on sign bits in ONE launch (bie_binary_matmul_batched), returned in x's dtype like
    binary_matmul_function (binary_linear_cutlass_kernel.cu:700-738): float32 -> (K - 2*popc) * scale; bfloat16 -> the popcount is
    first cast to bf16, then fma(popc, -2, K) and one rounded multiply by bf16(scale) (output_scaling_converting, :93-113)."""
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"tensor type not supported: {x.dtype}")
    lead = x.shape[:-2]
    M, K = x.shape[-2:]
    N = y.shape[-2]
    xb, yb = pack_rows(x).reshape(-1, M, K // 8).contiguous(), pack_rows(y).reshape(-1, N, K // 8).contiguous()
    B = xb.shape[0]
    _hip.need_gpu(xb, yb)
    out = torch.empty((B, M, N), dtype=torch.float32, device=x.device)
    bf16 = x.dtype == torch.bfloat16
    if B and M and N:
        rc = _hip.lib().bie_binary_matmul_batched(_hip.ptr(xb), _hip.ptr(yb), _hip.ptr(out), B, M, N, K, M * (K // 8), N * (K // 8), M * N,
                                                  1.0 if bf16 else float(scale), _hip.stream())
        _hip.check(rc, "bie_binary_matmul_batched")
    if bf16:  # the reference's bf16 arithmetic on the popcount, rounding by rounding
        popc = ((K - out) * 0.5).to(torch.bfloat16)
        kb = torch.tensor(float(K), dtype=torch.bfloat16, device=x.device)
        aligned = torch.addcmul(kb.float(), popc.float(), torch.tensor(-2.0, device=x.device)).to(torch.bfloat16)  # one rounding: __hfma
        out = aligned * torch.tensor(float(scale), dtype=torch.bfloat16, device=x.device)
    return out.reshape(lead + (M, N))


def kernel_eval(device_id: int, m: int, n: int, k: int) -> int:
    return 3  # no per-GPU CUTLASS tiling table here; kept for API compatibility
