"""Drop-in for `binary_linear_cpp` (layers/qlinear/binary/cpp/binary_linear.cpp:515-518):
forward(input, weights, m, n, k), w_pack(weights, n, k).  Packed weights use the reference's column
bit-plane layout (byte[kb*N + n] bit j = (w[n][8kb+j] >= 0)); results are exact integers in fp32."""
import torch

from ._binary_common import pack_rows, pack_cols, xnor_linear


def w_pack(weights: torch.Tensor, n: int, k: int) -> torch.Tensor:
    return pack_cols(weights.reshape(n, k))


def forward(input: torch.Tensor, weights: torch.Tensor, m: int, n: int, k: int) -> torch.Tensor:
    xp = pack_rows(input.reshape(m, k))
    if weights.numel() == k * n // 8 and weights.dtype == torch.uint8:  # already binarised (binary_linear.cpp:507)
        return xnor_linear(xp, weights.contiguous(), m, n, k, 1, 1.0)
    return xnor_linear(xp, pack_rows(weights.reshape(n, k)), m, n, k, 0, 1.0)
