"""Drop-in for `binary_linear_cuda` (layers/qlinear/binary/cuda/binary_linear_cuda.cpp:119-122):
forward(input, weights, bmm_type, transpose), w_pack(weights, bmm_type, transpose), mm(x, y, bmm_type).
All bmm_type values run the same wave64 XNOR-popcount kernels; packed weights are row-packed uint8
[N, K/8] LSB-first (the reference's per-kernel tile-swizzled BTC/BSTC images are CUDA-WMMA specific
and are not reproduced -- re-pack from the int8 sign carriers)."""
import torch

from ._binary_common import pack_rows, xnor_linear


def w_pack(weights: torch.Tensor, bmm_type: int, transpose: bool) -> torch.Tensor:
    return pack_rows(weights)


def forward(input: torch.Tensor, weights: torch.Tensor, bmm_type: int, transpose: bool) -> torch.Tensor:
    m, k = input.shape
    wp = weights if weights.dtype == torch.uint8 else pack_rows(weights)
    return xnor_linear(pack_rows(input), wp.contiguous(), m, wp.shape[0], k, 0, 1.0)


def mm(x: torch.Tensor, y: torch.Tensor, bmm_type: int) -> torch.Tensor:
    """x [M, K] . y [N, K]^T on sign bits."""
    return xnor_linear(pack_rows(x), pack_rows(y), x.shape[0], y.shape[0], x.shape[1], 0, 1.0)
