"""Drop-in for `q4_conv_cutlass` (layers/qconv/nbit/cutlass/q4_conv_cutlass.cpp:92-95): W4A4 convolution.
Like the reference host code (q4_conv_cutlass_kernel.cu:470-481, 535-541) the NCHW tensors are *viewed* as NHWC
([B, H, W, C] / [OC, k, k, C]) without a permute, and the output comes back NHWC [B, oe, oe, OC]."""
import torch

from bitorch_engine import _hip
from bitorch_engine.extensions.q_linear_cutlass import q4_w_pack as _quantize_pack, _f


def w_pack(weight: torch.Tensor, scale) -> torch.Tensor:
    oc, c, k, _ = weight.shape
    return _quantize_pack(weight.contiguous().view(oc, k, k, c), _f(scale))


def forward(input, weight, scale_a, scale_w, is_train, kernel_size, stride, padding, dilation):
    """-> [output NHWC (input dtype), packed activations [B,H,W,C/2], packed weights [OC,k,k,C/2]]"""
    _hip.need_gpu(input, weight)
    sa, sw = _f(scale_a), _f(scale_w)
    B, C, H, W = input.shape
    OC = weight.shape[0]
    packed_w = weight if weight.dtype == torch.int8 else w_pack(weight, sw)
    packed_a = _quantize_pack(input.contiguous().view(B, H, W, C), sa)
    oh = (H + 2 * padding - dilation * (kernel_size - 1) - 1) // stride + 1
    ow = (W + 2 * padding - dilation * (kernel_size - 1) - 1) // stride + 1
    out = torch.empty((B, oh, ow, OC), dtype=input.dtype, device=input.device)
    L = _hip.lib()
    need = L.bie_q4_conv2d_workspace_bytes(B, H, W, C, OC, kernel_size, stride, padding, dilation)
    ws = _hip.scratch(max(need, 16), input.device)  # per-(device, stream) buffer, reused across calls (no per-call allocation)
    rc = L.bie_q4_conv2d_forward(_hip.ptr(packed_a), _hip.ptr(packed_w.contiguous()), _hip.ptr(out), _hip.ptr(ws), ws.numel(), B, H, W, C,
                                 OC, kernel_size, stride, padding, dilation, sa, sw, _hip.dt(input), _hip.stream())
    _hip.check(rc, "bie_q4_conv2d_forward")
    return [out, packed_a, packed_w]
