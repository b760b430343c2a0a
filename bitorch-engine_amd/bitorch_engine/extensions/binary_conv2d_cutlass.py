"""Drop-in for `binary_conv2d_cutlass` (layers/qconv/binary/cutlass/binary_conv2d_cutlass.cpp:97-99):
forward(input, weight, scale, is_train, kernel_size, stride, padding, dilation), w_pack(data).

Two output conventions:
* default -- the well-defined convolution of the CPU path: NCHW in, [B, OC, OH, OW] out, (C*k*k - 2*popc) * scale.  The
  reference kernel's own conventions are layout accidents its tests do not pin.
* `forward_reference_convention` (the layer: `reference_convention=True` or BIE_BCONV_CUTLASS_CONVENTION=reference) -- what the reference
  kernel computes, read off its source, for checkpoints trained against that layer: see the function."""
import os

import torch
import torch.nn.functional as F

from ._binary_common import pack_rows, conv2d, xnor_linear


def w_pack(data: torch.Tensor) -> torch.Tensor:
    return pack_rows(data.reshape(data.shape[0], -1))


def forward(input, weight, scale, is_train, kernel_size, stride, padding, dilation):
    wp = weight if weight.dtype == torch.uint8 else w_pack(weight)
    return conv2d(input, wp.contiguous(), wp.shape[0], kernel_size, stride, padding, dilation, scale)


def reference_convention_default() -> bool:
    return os.environ.get("BIE_BCONV_CUTLASS_CONVENTION", "") == "reference"


def forward_reference_convention(input, weight, scale, is_train, kernel_size, stride, padding, dilation):
    """The reference kernel's result, bit for bit as its source defines it (binary_conv2d_cutlass_kernel.cu; no CUDA here, so unpinned by
    any reference output -- the numpy restatement the tests compare with cites the same lines):
      * :438 / :430,:474 -- NCHW input and OIHW weights are VIEWED as [B, H, W, C] / [OC, k, k, C]; :64-117 bit = (value >= 0), eight
        consecutive elements of memory per byte, LSB first;
      * :206-228 -- the packed tensors' sizes ([.., C/8] bytes) go to CUTLASS as extents of ONE-BIT tensors: it sees C/8 one-bit channels
        and walks the first B*H*W*C/8 (OC*k*k*C/8) bits of each buffer;
      * :260 Mode::kConvolution (flipped filter); :271 alpha 1, beta 0 on the int32 popcount(a XOR w) -- no K - 2*popc; taps outside the
        image read zero bits;
      * :414 out_edge from W only, no dilation term; :419-423,:453 output [B, out_edge, out_edge, OC] float32 = int32 * scale.
    Like the reference (the 128-bit operand alignment of its CUTLASS tile: can_implement fails and the process exits otherwise) C/8 must be a
    multiple of 128, i.e. C % 1024 == 0; here that is a RuntimeError.
    The popcounts run on this library's XNOR GEMM (bie_binary_linear_forward gives K - 2*popc exactly; popc = (K - that) / 2) over patches
    gathered as strided views of the packed bytes -- a compatibility mode, not a tuned kernel."""
    if input.dim() != 4:
        raise RuntimeError(f"tensor sizes not supported: {input.dim()}")
    B, C, H, W = input.shape
    k = int(kernel_size)
    if C % 1024:
        raise RuntimeError(f"binary_conv2d_cutlass (reference convention): in_channels / 8 = {C / 8:g} one-bit channels must be a multiple of 128 "
                           f"(the reference's CUTLASS tile cannot be implemented otherwise and exits)")
    OC = weight.shape[0]
    Cb = C // 64  # bytes per pixel of the C/8 one-bit channels
    oe = (W - k + 2 * padding) // stride + 1
    if oe <= 0:
        raise RuntimeError("binary_conv2d_cutlass (reference convention): empty output")
    a = pack_rows(input.contiguous().reshape(1, -1)).reshape(-1)[: B * H * W * Cb].reshape(B, H, W, Cb)
    wbytes = weight.contiguous().reshape(-1) if weight.dtype == torch.uint8 else pack_rows(weight.contiguous().reshape(1, -1)).reshape(-1)
    wq = wbytes[: OC * k * k * Cb].reshape(OC, k, k, Cb).flip(1, 2).reshape(OC, k * k * Cb).contiguous()
    # zero bits around the image: `padding` before, and after as far as the last tap of the last output reaches (the output extent ignores
    # the dilation and takes W for both axes, so it can reach further than `padding`)
    last = (oe - 1) * stride - padding + (k - 1) * dilation
    hi_h, hi_w = max(0, last - (H - 1)), max(0, last - (W - 1))
    ap = F.pad(a, (0, 0, padding, hi_w, padding, hi_h))
    sb, sh, sw, sc = ap.stride()
    patches = ap.as_strided((B, oe, oe, k, k, Cb), (sb, sh * stride, sw * stride, sh * dilation, sw * dilation, sc)).reshape(B * oe * oe, k * k * Cb)
    kbits = k * k * Cb * 8
    y = xnor_linear(patches.contiguous(), wq, B * oe * oe, OC, kbits, 0, 1.0)  # kbits - 2 * popcount, exact in fp32
    popc = (float(kbits) - y) * 0.5
    return (popc * float(scale)).reshape(B, oe, oe, OC)
