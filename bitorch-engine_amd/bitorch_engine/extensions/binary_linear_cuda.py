"""Drop-in for `binary_linear_cuda` (layers/qlinear/binary/cuda/binary_linear_cuda.cpp:119-122):
forward(input, weights, bmm_type, transpose), w_pack(weights, bmm_type, transpose), mm(x, y, bmm_type).

Packed weights are the REFERENCE's images, bit for bit (bmm_type 1 = BSTC32, 2 = BTC32, 3 = adaptive: BTC32 when
K % 128 == 0 and N % 8 == 0, else BSTC32 -- binary_linear_cuda_kernel.cu:847-870), so a BinaryLinearCuda checkpoint packed on
CUDA loads here.  The XNOR-popcount kernels themselves read row-packed words: an image is converted once per tensor version
(bie_binary_unpack_*, a byte permutation with bit reversal) and the result is remembered on the tensor.  Shapes neither image
can hold (K or N not a multiple of 32) fall back to plain row-packed bytes, which only this build produces and consumes."""
import torch

from bitorch_engine import _hip
from ._binary_common import pack_rows, sign_dt, xnor_linear, xnor_linear_fused, fused_ok, fp4_ok, xnor_values_fp4, xnor_layer_fp4
from .q_linear_cuda import _cached

BSTC32, BTC32, ADAPTIVE = 1, 2, 3


def image_kind(bmm_type: int, n: int, k: int):
    """'btc' / 'bstc' / None (no reference image exists for this shape)."""
    if (bmm_type == BTC32 or bmm_type == ADAPTIVE) and k % 128 == 0 and n % 8 == 0:
        return "btc"
    if bmm_type == BTC32:
        raise RuntimeError(f"bmm_type BTC32 needs k % 128 == 0 and n % 8 == 0 (n={n}, k={k})")
    if k % 32 == 0 and n % 32 == 0:
        return "bstc"
    return None


def w_pack(weights: torch.Tensor, bmm_type: int, transpose: bool) -> torch.Tensor:
    """weights [N, K] (float / int8 sign carriers) -> flat uint8 image of N*K/8 bytes."""
    _hip.need_gpu(weights)
    weights = weights.contiguous()
    n, k = weights.shape
    kind = image_kind(bmm_type, n, k)
    if kind is None:
        return pack_rows(weights).reshape(-1)
    out = torch.empty(n * k // 8, dtype=torch.uint8, device=weights.device)
    fn = _hip.lib().bie_binary_pack_btc32 if kind == "btc" else _hip.lib().bie_binary_pack_bstc32
    _hip.check(fn(_hip.ptr(weights), _hip.ptr(out), n, k, sign_dt(weights), _hip.stream()), "bie_binary_pack_" + kind + "32")
    return out


def image_to_rows(image: torch.Tensor, n: int, k: int, bmm_type: int) -> torch.Tensor:
    """image -> row-packed [N, K/8] (remembered on the image tensor until it changes)."""
    kind = image_kind(bmm_type, n, k)
    if kind is None:  # the image IS the row-packed matrix: hand out ONE view object per tensor version, so that conversions memoised on it
        return _cached(image, ("rows", None, n, k), lambda: image.reshape(n, k // 8))  # (the FP4 weight image) are not rebuilt every forward

    def convert():
        _hip.need_gpu(image)
        out = torch.empty((n, k // 8), dtype=torch.uint8, device=image.device)
        fn = _hip.lib().bie_binary_unpack_btc32 if kind == "btc" else _hip.lib().bie_binary_unpack_bstc32
        _hip.check(fn(_hip.ptr(image.contiguous()), _hip.ptr(out), n, k, _hip.stream()), "bie_binary_unpack_" + kind + "32")
        return out
    return _cached(image, ("rows", kind, n, k), convert)


def forward(input: torch.Tensor, weights: torch.Tensor, bmm_type: int, transpose: bool) -> torch.Tensor:
    m, k = input.shape
    if weights.dtype == torch.uint8:
        n = weights.numel() * 8 // k
        wp = image_to_rows(weights, n, k, bmm_type)
    elif weights.requires_grad and torch.is_grad_enabled():  # a weight under training: optimisers and clamps may write through `.data` (no version bump): pack every call
        wp = pack_rows(weights).contiguous()
    else:  # frozen unpacked sign carriers: packed once per tensor version, remembered on the weight tensor.  Writes through `.data` / raw
        # pointers do not advance the version counter: after one, rebind the tensor (or set requires_grad) to have it packed again
        wp = _cached(weights, ("rows_from_values",), lambda: pack_rows(weights).contiguous())
    wp = wp.contiguous()
    if fp4_ok(m, wp.shape[0], k) and (input.dtype in _hip._DT or input.dtype == torch.int8):
        return xnor_values_fp4(input, wp)  # large M: sign-pack folded into the FP4 image pass, GEMM on the matrix pipe (two launches)
    if input.dtype in _hip._DT and fused_ok(m, wp.shape[0], k) and wp.data_ptr() % 16 == 0:
        return xnor_linear_fused(input, wp, raw_counts=True)  # M <= 64 (<= 512 when K % 512 == 0): sign-pack of x inside the XNOR kernel (one launch)
    return xnor_linear(pack_rows(input), wp, m, wp.shape[0], k, 0, 1.0)


def layer_forward(input, bias_a, weights, bmm_type, scale_a, scale_w):
    """BinaryLinearCuda's whole forward: M >= 192 on the matrix pipe in two launches (xnor_layer_fp4); M <= 64 (M <= 512 when K % 512 == 0) in one launch: `((x + bias_a) >= 0)` bits, XNOR-popcount,
    `.to(dtype) * scale_a * scale_w` (reference layers/qlinear/binary/cuda/layer.py:58-63, 283-284).  None when the shape is
    outside the fused range (the caller then composes the separate steps)."""
    m, k = input.shape
    if weights.dtype != torch.uint8 or input.dtype not in _hip._DT:
        return None
    n = weights.numel() * 8 // k
    if fp4_ok(m, n, k):  # large M: two launches (values -> FP4 image with the bias add, matrix-pipe GEMM with the layer epilogue)
        return xnor_layer_fp4(input, image_to_rows(weights, n, k, bmm_type).contiguous(), bias_a, scale_a, scale_w)
    if not fused_ok(m, n, k):
        return None
    wp = image_to_rows(weights, n, k, bmm_type).contiguous()
    if wp.data_ptr() % 16:
        return None
    return xnor_linear_fused(input, wp, bias_a, scale_a, scale_w)


def mm(x: torch.Tensor, y: torch.Tensor, bmm_type: int) -> torch.Tensor:
    """x [M, K] . y [N, K]^T on sign bits."""
    return xnor_linear(pack_rows(x), pack_rows(y), x.shape[0], y.shape[0], x.shape[1], 0, 1.0)
