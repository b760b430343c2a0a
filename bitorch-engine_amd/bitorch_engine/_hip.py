"""ctypes binding of libbie_hip.so (include/bie_hip.h) -- the only way Python reaches the kernels.

PyTorch is used for device memory and streams only: every call passes `tensor.data_ptr()`, sizes,
a dtype enum and the current HIP stream.  There is NO CPU / eager fallback: if the library is missing
or a call fails, a RuntimeError carrying bie_last_error() is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BIE_HIP_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libbie_hip.so"))

F16, BF16, F32, I8 = 0, 1, 2, 3
_DT = {torch.float16: F16, torch.bfloat16: BF16, torch.float32: F32}

_lib = None
_vp, _i, _l, _sz, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_size_t, ctypes.c_float

# name -> (restype, argtypes); must list every symbol include/bie_hip.h declares (tests check this)
SIGNATURES = {
    "bie_version": (_i, []),
    "bie_last_error": (ctypes.c_char_p, []),
    "bie_mpq_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "bie_mpq_forward": (_i, [_vp] * 8 + [_sz] + [_i] * 7 + [_vp]),
    "bie_mpq_dequant": (_i, [_vp] * 5 + [_i] * 6 + [_vp]),
    "bie_mpq_pack": (_i, [_vp] * 5 + [_i] * 6 + [_vp]),
    "bie_mpq_grad_input": (_i, [_vp] * 6 + [_i] * 7 + [_vp]),
    "bie_mbwq_rows": (_i, [_vp, _i, _i, _vp]),
    "bie_mbwq_q4_dequant": (_i, [_vp] * 5 + [_i] * 4 + [_vp]),
    "bie_mbwq_exl2_dequant": (_i, [_vp] * 7 + [_i] * 3 + [_vp]),
    "bie_mbwq_workspace_bytes": (_sz, [_i, _i, _i]),
    "bie_mbwq_q4_forward": (_i, [_vp] * 7 + [_sz] + [_i] * 5 + [_vp]),
    "bie_mbwq_exl2_forward": (_i, [_vp] * 9 + [_sz] + [_i] * 4 + [_vp]),
    "bie_binary_pack_rows_u8": (_i, [_vp, _vp, _l, _l, _i, _vp]),
    "bie_binary_pack_cols_u8": (_i, [_vp, _vp, _l, _l, _i, _vp]),
    "bie_binary_linear_forward": (_i, [_vp, _vp, _vp, _l, _l, _l, _i, _f, _vp]),
    "bie_binary_conv2d_workspace_bytes": (_sz, [_i] * 9),
    "bie_binary_conv2d_forward": (_i, [_vp, _vp, _vp, _vp, _sz] + [_i] * 9 + [_f, _i, _vp]),
    "bie_pack_sign_u8": (_i, [_vp, _vp, _l, _i, _vp]),
    "bie_unpack_u8_scaled": (_i, [_vp, _vp, _vp, _l, _l, _vp]),
    "bie_q4_pack": (_i, [_vp, _vp, _l, _vp]),
    "bie_q4_unpack": (_i, [_vp, _vp, _l, _vp]),
    "bie_q4_unpack_scale": (_i, [_vp, _vp, _l, _f, _vp]),
    "bie_q4_quantize_pack": (_i, [_vp, _vp, _l, _f, _i, _vp]),
    "bie_q4_gemm": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _i, _l, _l, _l, _vp]),
    "bie_q8_gemm": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _vp]),
    "bie_q4_conv2d_workspace_bytes": (_sz, [_i] * 9),
    "bie_q4_conv2d_forward": (_i, [_vp, _vp, _vp, _vp, _sz] + [_i] * 9 + [_f, _f, _i, _vp]),
}


def lib():
    """Load libbie_hip.so (once).  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"bitorch_engine (MI355X build): {LIB_PATH} not found. Build it with "
                f"`make -C {os.path.dirname(os.path.dirname(LIB_PATH))}` or `python -c 'import __graft_entry__ as g; g.build()'`.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise RuntimeError(f"bitorch_engine: dtype {t.dtype} is not supported by the HIP kernels")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("bitorch_engine (MI355X build): tensors must live on the GPU; "
                               "there is no CPU fallback (move the layer with .to('cuda'))")


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().bie_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed with status {rc}: {msg}")


_WS = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Per-(device, stream) scratch buffer, grown on demand and reused (stream-ordered reuse is safe)."""
    if nbytes == 0:
        return None
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream())
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.zeros(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)  # head = split-K counters, must start at 0
        _WS[key] = buf
    return buf
