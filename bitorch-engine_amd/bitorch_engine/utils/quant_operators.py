"""Quantisation helpers mirroring reference utils/quant_operators.py: nv_tensor_quant (:7-90), get_binary_row / get_binary_col
(python reference bit packers :118-231), gptq_style_zeros_packing (:348-368), q4_quantization (:272-307)."""
import math

import torch


def nv_tensor_quant(inputs, amax=None, num_bits=8, unsigned=False, narrow_range=True):
    """Max-scaled integer quantisation, (quantised tensor, scale) -- reference utils/quant_operators.py:7-90 (itself after
    NVIDIA pytorch-quantization's tensor_quant).  amax defaults to torch.amax(inputs) -- the SIGNED maximum, as the reference
    has it; fp16/bf16 inputs are processed in fp32 and returned in their dtype; amax <= 2^-24 quantises to 0 with scale 1."""
    if isinstance(amax, torch.Tensor) and inputs.dim() != amax.dim():
        raise ValueError(f"amax {tuple(amax.size())} has different shape than inputs {tuple(inputs.size())}. "
                         "Make sure broadcast works as expected!")
    if amax is None:
        amax = torch.amax(inputs, keepdim=True)
    if unsigned and inputs.min() < 0.0:
        raise TypeError("Negative values encountered in unsigned quantization.")
    in_dtype = inputs.dtype
    half_in = in_dtype in (torch.float16, torch.bfloat16)
    x = inputs.float() if half_in else inputs
    amax = amax.float() if amax.dtype in (torch.float16, torch.bfloat16) else amax
    smallest = amax.min()
    if smallest < 0:
        raise ValueError("Negative values in amax")
    hi = torch.tensor(2.0 ** (num_bits - 1 + int(unsigned)) - 1.0, device=x.device)
    lo = 0 if unsigned else (-hi if narrow_range else -hi - 1)
    scale = hi / amax
    q = torch.clamp((x * scale).round_(), lo, hi)
    if smallest <= 2.0 ** -24:
        scale[amax <= 2.0 ** -24] = 1.0
    return (q.to(in_dtype) if half_in else q), scale


def get_binary_row(nd_row, binary_row, nd_size, bits_per_binary_word):
    """Sign-pack a flat float sequence LSB-first: bit j of word i = (nd_row[i*B + j] >= 0)."""
    B = bits_per_binary_word
    v = (torch.as_tensor(nd_row).reshape(-1)[:nd_size] >= 0).to(torch.int64).reshape(-1, B)
    words = (v << torch.arange(B)).sum(dim=1)
    for i, w in enumerate(words.tolist()):
        binary_row[i] = w
    return binary_row


def get_binary_col(nd_col, binary_col, dim_n, dim_k, bits_per_binary_word):
    """Column bit-planes of a row-major [dim_n, dim_k] matrix: word[y*dim_k + x] bit b = (m[y*B+b][x] >= 0)."""
    B = bits_per_binary_word
    m = (torch.as_tensor(nd_col).reshape(dim_n, dim_k) >= 0).to(torch.int64).reshape(dim_n // B, B, dim_k)
    words = (m << torch.arange(B).view(1, B, 1)).sum(dim=1).reshape(-1)
    for i, w in enumerate(words.tolist()):
        binary_col[i] = w
    return binary_col


def gptq_style_zeros_packing(zeros: torch.Tensor, w_bit: int, out_features: int, group_size: int) -> torch.Tensor:
    """Unpacked zero points (zq + 1) [G, N] -> int32 [G, N*w/32]: stores (z - 1) & mask, LSB first along N."""
    per = 32 // w_bit
    z = (zeros.to(torch.int32).reshape(zeros.shape[0], -1, per) - 1) & (2 ** w_bit - 1)
    shifts = torch.arange(0, 32, w_bit, device=zeros.device, dtype=torch.int32)
    return (z << shifts).sum(dim=-1).to(torch.int32)


def q4_quantization(input: torch.Tensor, scale_a: torch.Tensor = None, eps: torch.Tensor = None):
    """Symmetric 4-bit quantisation clamp(round(x / scale), -8, 7); returns (q, scale) when the scale is derived
    here (2*mean|x| / 5.6345), else q (reference :272-307)."""
    derive = scale_a is None
    x = input.to(torch.float32)
    if derive:
        scale_a = 2 * x.abs().mean() / 5.6345
    if eps is None:
        eps = torch.tensor(0.00001, dtype=x.dtype, device=x.device)
    scale_a = torch.where(scale_a > eps, scale_a, eps)
    q = (x / scale_a).round().clamp(-8, 7)
    return (q, scale_a) if derive else q


def q8_quantization(input: torch.Tensor, scale_a: torch.Tensor = None, eps: torch.Tensor = None):
    """Symmetric 8-bit quantisation clamp(round(x / scale), -128, 127); returns (q, scale) when the scale is derived here
    (2*mean|x| / 11.269), else q (reference :234-271)."""
    derive = scale_a is None
    x = input.to(torch.float32)
    if derive:
        scale_a = 2 * x.abs().mean() / 11.269
    if eps is None:
        eps = torch.tensor(0.00001, dtype=x.dtype, device=x.device)
    scale_a = torch.where(scale_a > eps, scale_a, eps.to(scale_a.dtype))
    q = (x / scale_a).round().clamp(-128, 127)
    return (q, scale_a) if derive else q
