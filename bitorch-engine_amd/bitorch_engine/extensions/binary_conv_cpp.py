"""Drop-in for `binary_conv_cpp` (layers/qconv/binary/cpp/binary_conv.cpp:532-534):
forward(input, weights, m, n, k, kernel_size, stride, padding, dilation, output_edge) ->
[B, m, output_edge, output_edge] fp32 exact integers, padding counted as -1."""
import torch

from ._binary_common import pack_rows, conv2d


def forward(input, weights, m, n, k, kernel_size, stride, padding, dilation, output_edge):
    wp = weights if weights.dtype == torch.uint8 else pack_rows(weights.reshape(m, k))
    return conv2d(input, wp.contiguous(), m, kernel_size, stride, padding, dilation, 1.0)
