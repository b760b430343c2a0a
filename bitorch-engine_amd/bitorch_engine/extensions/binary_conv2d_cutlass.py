"""Drop-in for `binary_conv2d_cutlass` (layers/qconv/binary/cutlass/binary_conv2d_cutlass.cpp:97-99):
forward(input, weight, scale, is_train, kernel_size, stride, padding, dilation), w_pack(data).
NOTE: the reference kernel has undocumented layout quirks (NCHW viewed as NHWC, raw popcount output,
SURVEY.md A16) that its own tests do not pin; this implementation computes the well-defined
convolution of the CPU path (A15): NCHW in, [B, OC, OH, OW] out, (C*k*k - 2*popc) * scale."""
import torch

from ._binary_common import pack_rows, conv2d


def w_pack(data: torch.Tensor) -> torch.Tensor:
    return pack_rows(data.reshape(data.shape[0], -1))


def forward(input, weight, scale, is_train, kernel_size, stride, padding, dilation):
    wp = weight if weight.dtype == torch.uint8 else w_pack(weight)
    return conv2d(input, wp.contiguous(), wp.shape[0], kernel_size, stride, padding, dilation, scale)
