"""Shared ctypes plumbing of the binary (XNOR-popcount) extension front-ends."""
import os

import torch

from bitorch_engine import _hip

_SIGN_DT = {torch.float16: _hip.F16, torch.bfloat16: _hip.BF16, torch.float32: _hip.F32, torch.int8: _hip.I8}


def sign_dt(t):
    try:
        return _SIGN_DT[t.dtype]
    except KeyError:
        raise RuntimeError(f"tensor type not supported: {t.dtype}")


def pack_rows(a: torch.Tensor) -> torch.Tensor:
    """[..., K] values -> [..., K/8] uint8, bit j of byte b = (a[..., 8b+j] >= 0)."""
    _hip.need_gpu(a)
    a = a.contiguous()
    K = a.shape[-1]
    if K % 8:
        raise RuntimeError(f"bit packing needs the last dimension ({K}) to be a multiple of 8")
    rows = a.numel() // K
    out = torch.empty(a.shape[:-1] + (K // 8,), dtype=torch.uint8, device=a.device)
    if rows:
        rc = _hip.lib().bie_binary_pack_rows_u8(_hip.ptr(a), _hip.ptr(out), rows, K, sign_dt(a), _hip.stream())
        _hip.check(rc, "bie_binary_pack_rows_u8")
    return out


def pack_cols(w: torch.Tensor) -> torch.Tensor:
    """w [N, K] -> flat uint8 [K/8 * N] column bit-planes (binary_linear_cpp.w_pack layout)."""
    _hip.need_gpu(w)
    w = w.contiguous()
    N, K = w.shape
    out = torch.empty((K // 8) * N, dtype=torch.uint8, device=w.device)
    rc = _hip.lib().bie_binary_pack_cols_u8(_hip.ptr(w), _hip.ptr(out), N, K, sign_dt(w), _hip.stream())
    _hip.check(rc, "bie_binary_pack_cols_u8")
    return out


def fp4_min_rows() -> int:
    """Smallest M served by the matrix-pipe (FP4 image) form of the binary GEMM; below it the XNOR kernels win (two extra small
    launches and a 4x larger activation operand).  BIE_FP4_MIN_M overrides (0 switches the form off)."""
    v = os.environ.get("BIE_FP4_MIN_M")
    return int(v) if v else 192


def fp4_image(rowpacked: torch.Tensor, rows: int, K: int, out: torch.Tensor = None) -> torch.Tensor:
    """Row-packed sign bits [rows, K/8] -> FP4 (E2M1 +-1) image in MFMA fragment order (bie_binary_fp4_image)."""
    _hip.need_gpu(rowpacked)
    L = _hip.lib()
    need = L.bie_binary_fp4_image_bytes(rows, K)
    if out is None:
        out = torch.empty(need, dtype=torch.uint8, device=rowpacked.device)
    _hip.check(L.bie_binary_fp4_image(_hip.ptr(rowpacked), _hip.ptr(out), rows, K, _hip.stream()), "bie_binary_fp4_image")
    return out


def fp4_weight_image(wp: torch.Tensor, N: int, K: int) -> torch.Tensor:
    """The weights' image, built once per tensor version and remembered on the packed tensor (4 bits per weight beside the
    1-bit checkpoint tensor)."""
    from .q_linear_cuda import _cached
    return _cached(wp, ("fp4", N, K), lambda: fp4_image(wp.contiguous(), N, K))


def xnor_linear_fp4(xp: torch.Tensor, wp: torch.Tensor, M: int, N: int, K: int, scale: float) -> torch.Tensor:
    """The same y as xnor_linear(w_layout = 0) on the matrix pipe: x image into stream scratch, weight image memoised."""
    _hip.need_gpu(xp, wp)
    L = _hip.lib()
    wimg = fp4_weight_image(wp, N, K)
    ximg = fp4_image(xp, M, K, out=_hip.scratch(L.bie_binary_fp4_image_bytes(M, K), xp.device))
    y = torch.empty((M, N), dtype=torch.float32, device=xp.device)
    _hip.check(L.bie_binary_linear_forward_fp4(_hip.ptr(ximg), _hip.ptr(wimg), _hip.ptr(y), M, N, K, float(scale), _hip.stream()),
               "bie_binary_linear_forward_fp4")
    return y


def fp4_ok(M: int, N: int, K: int) -> bool:
    """Shapes the matrix-pipe form serves by default (it is correct for any shape; below these sizes the XNOR kernels are faster)."""
    m_min = fp4_min_rows()
    return bool(m_min) and M >= m_min and N >= 64 and 256 <= K < (1 << 24)


def _x_image_from_values(x: torch.Tensor, bias_a, M: int, K: int) -> torch.Tensor:
    L = _hip.lib()
    ximg = _hip.scratch(L.bie_binary_fp4_image_bytes(M, K), x.device)
    _hip.check(L.bie_binary_fp4_image_from_values(_hip.ptr(x), _hip.ptr(bias_a), _hip.ptr(ximg), M, K, sign_dt(x), _hip.stream()),
               "bie_binary_fp4_image_from_values")
    return ximg


def xnor_values_fp4(x: torch.Tensor, wp: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """fp32 (K - 2*popc) * scale of sign(x) against row-packed weights, two launches: values -> FP4 image (the sign-pack folded in),
    matrix-pipe GEMM."""
    _hip.need_gpu(x, wp)
    x = x.contiguous()
    M, K = x.shape
    N = wp.shape[0]
    wimg = fp4_weight_image(wp, N, K)
    ximg = _x_image_from_values(x, None, M, K)
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    _hip.check(_hip.lib().bie_binary_linear_forward_fp4(_hip.ptr(ximg), _hip.ptr(wimg), _hip.ptr(y), M, N, K, float(scale), _hip.stream()),
               "bie_binary_linear_forward_fp4")
    return y


def xnor_layer_fp4(x, wrows, bias_a=None, scale_a=None, scale_w=None):
    """BinaryLinearCuda's forward at large M in two launches: `((x + bias_a) >= 0)` -> FP4 image, then the matrix-pipe GEMM with the
    `.to(dtype) * scale_a * scale_w` epilogue (bie_binary_linear_layer_fp4) -- same roundings as xnor_linear_fused."""
    _hip.need_gpu(x, wrows)
    x = x.contiguous()
    M, K = x.shape
    N = wrows.shape[0]
    same = lambda t: None if t is None else t.to(device=x.device, dtype=x.dtype).contiguous()
    bias_a, scale_a, scale_w = same(bias_a), same(scale_a), same(scale_w)
    wimg = fp4_weight_image(wrows, N, K)
    ximg = _x_image_from_values(x, bias_a, M, K)
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    _hip.check(_hip.lib().bie_binary_linear_layer_fp4(_hip.ptr(ximg), _hip.ptr(wimg), _hip.ptr(scale_a), _hip.ptr(scale_w), _hip.ptr(y), M, N, K,
                                                      _hip.dt(x), _hip.stream()), "bie_binary_linear_layer_fp4")
    return y


def xnor_linear(xp: torch.Tensor, wp: torch.Tensor, M: int, N: int, K: int, w_layout: int, scale: float) -> torch.Tensor:
    """y[M, N] fp32 = (K - 2*popc(x ^ w)) * scale from packed operands.  Large M with row-packed weights: the matrix-pipe form
    (identical integers); everything else: the XNOR-popcount kernels."""
    _hip.need_gpu(xp, wp)
    if w_layout == 0 and fp4_ok(M, N, K):
        return xnor_linear_fp4(xp.contiguous(), wp, M, N, K, scale)
    y = torch.empty((M, N), dtype=torch.float32, device=xp.device)
    if M and N:
        rc = _hip.lib().bie_binary_linear_forward(_hip.ptr(xp), _hip.ptr(wp), _hip.ptr(y), M, N, K, w_layout, float(scale), _hip.stream())
        _hip.check(rc, "bie_binary_linear_forward")
    return y


def fused_ok(M: int, N: int, K: int) -> bool:
    return bool(_hip.lib().bie_binary_linear_fused_ok(M, N, K))


def xnor_linear_fused(x, wrows, bias_a=None, scale_a=None, scale_w=None, raw_counts=False):
    """One launch: sign-pack of (x + bias_a), XNOR-popcount against row-packed weights, `.to(dtype) * scale_a * scale_w`
    epilogue (bie_binary_linear_fused).  raw_counts: fp32 K - 2*popc instead (no scales)."""
    _hip.need_gpu(x, wrows)
    x = x.contiguous()
    M, K = x.shape
    N = wrows.shape[0]
    same = lambda t: None if t is None else t.to(device=x.device, dtype=x.dtype).contiguous()
    bias_a, scale_a, scale_w = same(bias_a), same(scale_a), same(scale_w)
    if x.data_ptr() % 16:
        x = x.clone()
    if bias_a is not None and bias_a.data_ptr() % 16:
        bias_a = bias_a.clone()
    y = torch.empty((M, N), dtype=torch.float32 if raw_counts else x.dtype, device=x.device)
    rc = _hip.lib().bie_binary_linear_fused(_hip.ptr(x), _hip.ptr(bias_a), _hip.ptr(wrows), _hip.ptr(scale_a), _hip.ptr(scale_w),
                                            _hip.ptr(y), M, N, K, _hip.dt(x), int(raw_counts), _hip.stream())
    _hip.check(rc, "bie_binary_linear_fused")
    return y


def conv_weight_taps(wpacked, OC, C, ksize):
    """Packed conv weights [OC, C*k*k/8] -> tap-major words [OC, k*k, ceil(C/32)] (bie_binary_conv_weight_taps), remembered on
    the packed tensor until it changes."""
    from .q_linear_cuda import _cached

    def convert():
        out = torch.empty((OC, ksize * ksize, (C + 31) // 32), dtype=torch.int32, device=wpacked.device)
        rc = _hip.lib().bie_binary_conv_weight_taps(_hip.ptr(wpacked), _hip.ptr(out), OC, C, ksize, _hip.stream())
        _hip.check(rc, "bie_binary_conv_weight_taps")
        return out
    return _cached(wpacked, ("taps", OC, C, ksize), convert)


def conv_weight_lanes(wpacked, OC, C, ksize):
    """Lane-major image of the tap words for the one-launch conv (bie_binary_conv_weight_lanes), remembered on the packed tensor."""
    from .q_linear_cuda import _cached

    def convert():
        L = _hip.lib()
        out = torch.empty((L.bie_binary_conv_weight_lanes_bytes(OC, C, ksize) // 4,), dtype=torch.int32, device=wpacked.device)
        rc = L.bie_binary_conv_weight_lanes(_hip.ptr(conv_weight_taps(wpacked, OC, C, ksize)), _hip.ptr(out), OC, C, ksize, _hip.stream())
        _hip.check(rc, "bie_binary_conv_weight_lanes")
        return out
    return _cached(wpacked, ("lanes", OC, C, ksize), convert)


def conv_fused_max_rows() -> int:
    """Largest number of output pixels (B * OH * OW) the one-launch VALU form serves before the one-launch matrix-pipe form takes over
    (BIE_CONV_FUSED_MAX_ROWS; 0 = that form off)."""
    v = os.environ.get("BIE_CONV_FUSED_MAX_ROWS")
    return int(v) if v else 392


def conv_mfma_max_rows() -> int:
    """Largest number of output pixels the one-launch matrix-pipe form serves (BIE_CONV_MFMA_MAX_ROWS; 0 = off: the three-launch FP4 GEMM
    / the tap kernels of round 5)."""
    v = os.environ.get("BIE_CONV_MFMA_MAX_ROWS")
    return int(v) if v else 1 << 30


def conv_fp4_min_rows() -> int:
    """Smallest number of output pixels (B * OH * OW) the matrix-pipe form of the conv serves (BIE_FP4_CONV_MIN_ROWS; 0 = off).
    Measured on the ResNet-18 stage shapes (profiles/r03_fp4_g_conv_ab.txt, r03_fp4_h_tile64.txt): with the 128 x 64 GEMM tile for small
    grids it is ahead of the XNOR tap form from ~1000 pixels up (1.0-2.4x); below, its three launches lose to the tap kernels."""
    v = os.environ.get("BIE_FP4_CONV_MIN_ROWS")
    return int(v) if v else 1024


def conv_weight_fp4_image(wpacked, OC, C, ksize):
    """FP4 image of the tap-major weight words (rows = OC, k = (tap, channel)), remembered on the packed tensor."""
    from .q_linear_cuda import _cached
    return _cached(wpacked, ("fp4taps", OC, C, ksize),
                   lambda: fp4_image(conv_weight_taps(wpacked, OC, C, ksize).view(torch.uint8).reshape(OC, -1), OC, ksize * ksize * C))


def conv2d(x, wpacked, OC, ksize, stride, pad, dil, scale):
    _hip.need_gpu(x, wpacked)
    x = x.contiguous()
    B, C, H, W = x.shape
    OH = (H + 2 * pad - dil * (ksize - 1) - 1) // stride + 1
    OW = (W + 2 * pad - dil * (ksize - 1) - 1) // stride + 1
    y = torch.empty((B, OC, OH, OW), dtype=torch.float32, device=x.device)
    L = _hip.lib()
    if (B * OH * OW <= conv_fused_max_rows() and x.dtype in _hip._DT and (C * ksize * ksize) % 8 == 0
            and L.bie_binary_conv2d_fused_ok(B, C, H, W, OC, ksize, stride, pad, dil)):
        # ONE launch, no workspace: sign-pack into LDS + register-resident weights + XNOR-popcount (binary_conv_fused.hip)
        wl = conv_weight_lanes(wpacked, OC, C, ksize)
        rc = L.bie_binary_conv2d_forward_fused(_hip.ptr(x), _hip.ptr(wl), _hip.ptr(y), B, C, H, W, OC, ksize, stride, pad, dil, float(scale), _hip.dt(x),
                                               _hip.stream())
        _hip.check(rc, "bie_binary_conv2d_forward_fused")
        return y
    if (B * OH * OW <= conv_mfma_max_rows() and x.dtype in _hip._DT and (C * ksize * ksize) % 8 == 0
            and L.bie_binary_conv2d_mfma_ok(B, C, H, W, OC, ksize, stride, pad, dil)):
        # ONE launch on the matrix pipe: FP4 image of the input rows in LDS, pixel fragments gathered tap by tap (binary_conv_fused.hip)
        wimg = conv_weight_fp4_image(wpacked, OC, C, ksize)
        rc = L.bie_binary_conv2d_forward_mfma(_hip.ptr(x), _hip.ptr(wimg), _hip.ptr(y), B, C, H, W, OC, ksize, stride, pad, dil, float(scale), _hip.dt(x),
                                              _hip.stream())
        _hip.check(rc, "bie_binary_conv2d_forward_mfma")
        return y
    rows_min = conv_fp4_min_rows()
    if rows_min and C % 32 == 0 and B * OH * OW >= rows_min and OC >= 64 and x.dtype in _hip._DT:
        # large batch: the conv as a GEMM on the matrix pipe (+-1 as FP4 MFMA operands), bit-identical
        wimg = conv_weight_fp4_image(wpacked, OC, C, ksize)
        need = L.bie_binary_conv2d_fp4_workspace_bytes(B, C, H, W, ksize, stride, pad, dil)
        ws = _hip.scratch(need, x.device)
        rc = L.bie_binary_conv2d_forward_fp4(_hip.ptr(x), _hip.ptr(wimg), _hip.ptr(y), _hip.ptr(ws), ws.numel(), B, C, H, W, OC, ksize, stride, pad,
                                             dil, float(scale), _hip.dt(x), _hip.stream())
        _hip.check(rc, "bie_binary_conv2d_forward_fp4")
        return y
    need = L.bie_binary_conv2d_workspace_bytes(B, C, H, W, OC, ksize, stride, pad, dil)
    ws = _hip.workspace(need, x.device)
    if L.bie_binary_conv2d_taps_ok(C, W, ksize) and (C * ksize * ksize) % 8 == 0:
        # no im2col image: tap-major weights (converted once per tensor version) against channel-minor activation bits
        wt = conv_weight_taps(wpacked, OC, C, ksize)
        rc = L.bie_binary_conv2d_forward_taps(_hip.ptr(x), _hip.ptr(wt), _hip.ptr(y), _hip.ptr(ws), ws.numel(), B, C, H, W, OC,
                                              ksize, stride, pad, dil, float(scale), _hip.dt(x), _hip.stream())
        _hip.check(rc, "bie_binary_conv2d_forward_taps")
        return y
    rc = L.bie_binary_conv2d_forward(_hip.ptr(x), _hip.ptr(wpacked), _hip.ptr(y), _hip.ptr(ws), ws.numel(), B, C, H, W, OC,
                                     ksize, stride, pad, dil, float(scale), _hip.dt(x), _hip.stream())
    _hip.check(rc, "bie_binary_conv2d_forward")
    return y
