"""Drop-in for `binary_linear_cutlass` (layers/qlinear/binary/cutlass/binary_linear_cutlass.cpp:206-210):
forward(input, weight, scale, transpose, kernel_id), w_pack(weight, transpose), mm(x, y, transpose),
matmul(x, y, scale), kernel_eval(device_id, m, n, k).  Row-packed uint8 LSB-first operands
(binary_linear_cutlass_kernel.cu:44-90), epilogue (K - 2*popc) * scale (:93-113)."""
import torch

from ._binary_common import pack_rows, xnor_linear


def w_pack(weight: torch.Tensor, transpose: bool) -> torch.Tensor:
    return pack_rows(weight.t() if transpose else weight)


def forward(input: torch.Tensor, weight: torch.Tensor, scale: float, transpose: bool, kernel_id: int) -> torch.Tensor:
    m, k = input.shape
    wp = weight if weight.dtype == torch.uint8 else w_pack(weight, transpose)
    return xnor_linear(pack_rows(input), wp.contiguous(), m, wp.shape[0], k, 0, scale)


def mm(x: torch.Tensor, y: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    return xnor_linear(pack_rows(x), pack_rows(y.t() if transpose else y), x.shape[0], y.shape[0], x.shape[1], 0, 1.0)


def matmul(x: torch.Tensor, y: torch.Tensor, scale: float) -> torch.Tensor:
    """Batched x [..., M, K] . y [..., N, K]^T on sign bits, (K - 2*popc) * scale."""
    lead = x.shape[:-2]
    M, K = x.shape[-2:]
    N = y.shape[-2]
    xb, yb = pack_rows(x).reshape(-1, M, K // 8), pack_rows(y).reshape(-1, N, K // 8)
    outs = [xnor_linear(xb[i], yb[i], M, N, K, 0, scale) for i in range(xb.shape[0])]
    return torch.stack(outs).reshape(lead + (M, N))


def kernel_eval(device_id: int, m: int, n: int, k: int) -> int:
    return 3  # no per-GPU CUTLASS tiling table here; kept for API compatibility
