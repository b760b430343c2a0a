"""ctypes binding of libbie_hip.so (include/bie_hip.h) -- the only way Python reaches the kernels.

PyTorch is used for device memory and streams only: every call passes `tensor.data_ptr()`, sizes,
a dtype enum and the current HIP stream.  There is NO CPU / eager fallback: if the library is missing
or a call fails, a RuntimeError carrying bie_last_error() is raised.
"""
import ctypes
import os
import threading

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
    "bie_mpq_workspace_bytes_gidx": (_sz, [_i, _i, _i, _i]),
    "bie_mpq_forward": (_i, [_vp] * 8 + [_sz] + [_i] * 7 + [_vp]),
    "bie_mpq_forward_pitched": (_i, [_vp] * 6 + [_i, _vp, _sz] + [_i] * 7 + [_vp]),
    "bie_workspace_init": (_i, [_vp, _sz, _vp]),
    "bie_mpq_grouped_workspace_bytes": (_sz, [_i, _vp, _i, _i, _i]),
    "bie_mpq_forward_grouped": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz] + [_i] * 6 + [_vp]),
    "bie_status_init": (_i, []),
    "bie_device_status": (ctypes.c_uint, [_i]),
    "bie_mpq_list_device_bytes": (_sz, [_i, _vp, _i, _i, _i]),
    "bie_mpq_list_create": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _sz]),
    "bie_mpq_list_forward": (_i, [_vp, _vp]),
    "bie_mpq_list_launches": (_i, [_vp]),
    "bie_mpq_list_form": (_i, [_vp]),
    "bie_mpq_prefill_form": (_i, [_i, _i, _i]),
    "bie_mpq_rows_form": (_i, [_i, _i, _i, _i, _i, _i]),
    "bie_mpq_grouped_max_rows": (_i, [_i, _l, _i, _i]),
    "bie_mpq_list_destroy": (None, [_vp]),
    "bie_mbwq_exl2_list_device_bytes": (_sz, [_i, _vp, _i]),
    "bie_mbwq_exl2_list_create": (_i, [_vp, _i, _vp, _i, _vp, _sz]),
    "bie_mbwq_exl2_list_forward": (_i, [_vp, _vp]),
    "bie_mbwq_exl2_list_destroy": (None, [_vp]),
    "bie_mbwq_exl2_grouped_workspace_bytes": (_sz, [_i, _vp, _i]),
    "bie_mbwq_exl2_forward_grouped": (_i, [_vp, _i, _i, _vp, _vp, _sz, _vp]),
    "bie_mpq_dequant": (_i, [_vp] * 5 + [_i] * 6 + [_vp]),
    "bie_mpq_pack": (_i, [_vp] * 5 + [_i] * 6 + [_vp]),
    "bie_mpq_grad_input": (_i, [_vp] * 6 + [_i] * 7 + [_vp]),
    "bie_mpq_sort_rows": (_i, [_vp] * 3 + [_i] * 3 + [_vp]),
    "bie_gather_cols": (_i, [_vp] * 3 + [_i] * 3 + [_vp]),
    "bie_mbwq_rows": (_i, [_vp, _i, _i, _vp]),
    "bie_mbwq_exl2_shuffle": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "bie_mbwq_exl2_table": (_i, [_vp, _i, _i, _vp]),
    "bie_mbwq_exl2_unshuffle": (_i, [_vp, _vp, _i, _i, _vp]),
    "bie_mbwq_q4_dequant": (_i, [_vp] * 5 + [_i] * 4 + [_vp]),
    "bie_mbwq_exl2_dequant": (_i, [_vp] * 7 + [_i] * 3 + [_vp]),
    "bie_mbwq_workspace_bytes": (_sz, [_i, _i, _i]),
    "bie_mbwq_q4_workspace_bytes": (_sz, [_i, _i, _i]),
    "bie_mbwq_q4_forward": (_i, [_vp] * 7 + [_sz] + [_i] * 5 + [_vp]),
    "bie_mbwq_exl2_forward": (_i, [_vp] * 9 + [_sz] + [_i] * 4 + [_vp]),
    "bie_binary_pack_rows_u8": (_i, [_vp, _vp, _l, _l, _i, _vp]),
    "bie_binary_pack_cols_u8": (_i, [_vp, _vp, _l, _l, _i, _vp]),
    "bie_binary_pack_btc32": (_i, [_vp, _vp, _l, _l, _i, _vp]),
    "bie_binary_pack_bstc32": (_i, [_vp, _vp, _l, _l, _i, _vp]),
    "bie_binary_unpack_btc32": (_i, [_vp, _vp, _l, _l, _vp]),
    "bie_binary_unpack_bstc32": (_i, [_vp, _vp, _l, _l, _vp]),
    "bie_binary_linear_forward": (_i, [_vp, _vp, _vp, _l, _l, _l, _i, _f, _vp]),
    "bie_binary_fp4_image_bytes": (_sz, [_l, _l]),
    "bie_binary_fp4_image": (_i, [_vp, _vp, _l, _l, _vp]),
    "bie_binary_fp4_image_from_values": (_i, [_vp, _vp, _vp, _l, _l, _i, _vp]),
    "bie_binary_linear_forward_fp4": (_i, [_vp, _vp, _vp, _l, _l, _l, _f, _vp]),
    "bie_binary_linear_layer_fp4": (_i, [_vp] * 5 + [_l] * 3 + [_i, _vp]),
    "bie_binary_conv2d_fp4_workspace_bytes": (_sz, [_i] * 8),
    "bie_binary_conv2d_forward_fp4": (_i, [_vp] * 4 + [_sz] + [_i] * 9 + [_f, _i, _vp]),
    "bie_binary_matmul_batched": (_i, [_vp, _vp, _vp] + [_l] * 7 + [_f, _vp]),
    "bie_binary_linear_fused_ok": (_i, [_l] * 3),
    "bie_binary_linear_fused": (_i, [_vp] * 6 + [_l] * 3 + [_i, _i, _vp]),
    "bie_binary_conv2d_workspace_bytes": (_sz, [_i] * 9),
    "bie_binary_conv2d_forward": (_i, [_vp, _vp, _vp, _vp, _sz] + [_i] * 9 + [_f, _i, _vp]),
    "bie_binary_conv2d_taps_ok": (_i, [_i] * 3),
    "bie_binary_conv_weight_taps": (_i, [_vp, _vp] + [_i] * 3 + [_vp]),
    "bie_binary_conv2d_forward_taps": (_i, [_vp] * 4 + [_sz] + [_i] * 9 + [_f, _i, _vp]),
    "bie_binary_conv2d_mfma_ok": (_i, [_i] * 9),
    "bie_binary_conv2d_forward_mfma": (_i, [_vp] * 3 + [_i] * 9 + [_f, _i, _vp]),
    "bie_binary_conv2d_fused_ok": (_i, [_i] * 9),
    "bie_binary_conv_weight_lanes_bytes": (_sz, [_i] * 3),
    "bie_binary_conv_weight_lanes": (_i, [_vp, _vp] + [_i] * 3 + [_vp]),
    "bie_binary_conv2d_forward_fused": (_i, [_vp] * 3 + [_i] * 9 + [_f, _i, _vp]),
    "bie_pack_sign_u8": (_i, [_vp, _vp, _l, _i, _vp]),
    "bie_unpack_u8_scaled": (_i, [_vp, _vp, _vp, _l, _l, _vp]),
    "bie_q4_pack": (_i, [_vp, _vp, _l, _vp]),
    "bie_q4_unpack": (_i, [_vp, _vp, _l, _vp]),
    "bie_q4_unpack_scale": (_i, [_vp, _vp, _l, _f, _vp]),
    "bie_q4_quantize_pack": (_i, [_vp, _vp, _l, _f, _i, _vp]),
    "bie_q4_gemm": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _i, _l, _l, _l, _vp]),
    "bie_q8_gemm": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _f, _vp]),
    "bie_int_gemm_i32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _l, _l, _l, _vp]),
    "bie_q4_conv2d_workspace_bytes": (_sz, [_i] * 9),
    "bie_q4_conv2d_forward": (_i, [_vp, _vp, _vp, _vp, _sz] + [_i] * 9 + [_f, _f, _i, _vp]),
}


# include/bie_hip_testing.h: fault-injection hooks, bound for tests/ only (nothing in this package calls them)
TEST_HOOKS = {
    "bie_test_forge_reducer": (None, [ctypes.c_uint, _i]),
    "bie_test_forge_dependency": (None, [_i]),
}

_HOST_ONLY = ("bie_version", "bie_last_error", "bie_mbwq_rows", "bie_mbwq_exl2_table", "bie_status_init", "bie_device_status", "bie_test_forge_reducer",
              "bie_test_forge_dependency", "bie_mpq_list_launches", "bie_mpq_list_form", "bie_mpq_prefill_form", "bie_mpq_rows_form", "bie_mpq_grouped_max_rows", "bie_mpq_list_destroy", "bie_mbwq_exl2_list_destroy")


class ListEntry(ctypes.Structure):
    """bie_mpq_list_entry (include/bie_hip.h)."""
    _fields_ = [("x", _vp), ("qweight", _vp), ("scales", _vp), ("zeros", _vp), ("bias", _vp), ("y", _vp),
                ("K", _i), ("N", _i), ("depends_on", _i), ("reserved", _i)]


class Exl2ListEntry(ctypes.Structure):
    """bie_exl2_list_entry (include/bie_hip.h)."""
    _fields_ = [("x", _vp), ("qweight", _vp), ("scales", _vp), ("zeros", _vp), ("q_perm", _vp), ("q_group_map", _vp), ("rows7", _vp),
                ("y", _vp), ("K", _i), ("N", _i), ("reserved0", _i), ("reserved1", _i)]


def lib():
    """Load libbie_hip.so (once).  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"bitorch_engine (MI355X build): {LIB_PATH} not found. Build it with "
                f"`make -C {os.path.dirname(os.path.dirname(LIB_PATH))}` or `python -c 'import __graft_entry__ as g; g.build()'`.")
        l = ctypes.CDLL(LIB_PATH)
        ns = _Lib()
        for name, (res, args) in list(SIGNATURES.items()) + list(TEST_HOOKS.items()):
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
            launches = not (name.endswith("_bytes") or name in _HOST_ONLY)
            setattr(ns, name, _guarded(fn) if launches else fn)
        if torch.cuda.is_available():
            l.bie_status_init()  # the host-mapped status page the kernels raise their fail-loud bits in (once, outside any capture)
        _lib = ns
    return _lib


class _Lib:
    """Namespace of the bound entry points (launching ones wrapped in the device guard)."""


_tls = threading.local()


def _guarded(fn):
    """Device guard: the C ABI launches on the CURRENT HIP device, the tensors of a call may live on another one
    (device_map='auto', pipeline-split models).  need_gpu() records the device of the call's tensors; if it is not the
    current device the call is made under torch.cuda.device(...) -- what the reference's
    at::cuda::OptionalCUDAGuard(device_of(x)) does (mpq_linear_cuda_kernel.cu:612)."""
    def call(*args):
        d = getattr(_tls, "device", None)
        if d is not None and d.index is not None and d.index != torch.cuda.current_device():
            with torch.cuda.device(d):
                return fn(*args)
        return fn(*args)
    return call


def dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise RuntimeError(f"bitorch_engine: dtype {t.dtype} is not supported by the HIP kernels")


def ptr(t):
    return None if t is None else t.data_ptr()


def stream() -> int:
    """The current stream of the device the call's tensors live on (see need_gpu)."""
    return torch.cuda.current_stream(getattr(_tls, "device", None)).cuda_stream


def need_gpu(*tensors):
    """Every entry point calls this first: all tensors on ONE GPU (the reference checks the same,
    mpq_layer.py:45-48); remembers that device for stream() / workspace() / the device guard."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("bitorch_engine (MI355X build): tensors must live on the GPU; "
                               "there is no CPU fallback (move the layer with .to('cuda'))")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"bitorch_engine: tensors of one call live on different devices ({dev} and {t.device})")
    _tls.device = dev
    return dev


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().bie_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed with status {rc}: {msg}")


_WS = {}
_WS_RETIRED = []  # buffers outgrown AFTER a capture used them are kept alive: a captured HIP graph has their address baked in
_CAPTURED = set()  # keys of buffers that were handed out while their stream was capturing


def _grow(table, key, nbytes, device, zero):
    """Shared growth policy of workspace() / scratch(): geometric (a serving loop with slowly growing M must not reallocate at
    every new maximum).  An outgrown buffer is retired -- kept alive -- only if a captured graph may hold its address
    (otherwise it is simply dropped); the key's capture mark goes with it.

    Under stream capture nothing is allocated: a buffer created inside a capture would come from the graph's private pool and
    its zero-fill would become a memset node that resets the generation words of the reduction protocol on every replay.
    A plain `with torch.cuda.graph(g):` captures on torch's own capture stream, which has no buffer of its own even after a
    warm-up: the capture then TAKES OVER the largest sufficient buffer another stream of the same device warmed up -- the buffer
    moves to the capturing stream's key and the stream that warmed it up allocates a fresh one at its next eager call, so a
    replaying graph and eager calls never share generation words or scratch (a workspace has ONE owner stream at any time;
    `torch.cuda.graph` synchronises the device before capturing, so nothing enqueued earlier still uses the buffer).  Buffers a
    capture already used are never handed to another stream's capture.  Only when no stream of the device has a free, large
    enough buffer does the call fail, naming the stream."""
    buf = table.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if buf is not None and buf.numel() >= nbytes:
        if capturing:
            _CAPTURED.add((id(table), key))
        return buf
    if capturing:
        free = [(k, b) for k, b in table.items() if k[0] == key[0] and k != key and b.numel() >= nbytes and (id(table), k) not in _CAPTURED]
        if free:
            k, b = max(free, key=lambda kb: kb[1].numel())
            del table[k]  # its stream allocates a new one at its next eager call
            if buf is not None and (id(table), key) in _CAPTURED:
                _WS_RETIRED.append(buf)  # nodes captured earlier on this stream keep their (smaller) buffer
            table[key] = b
            _CAPTURED.add((id(table), key))
            return b
        raise RuntimeError(f"bitorch_engine: stream {key[1]:#x} of device {key[0]} is being captured and no stream of that device has a "
                           f"free scratch buffer of {nbytes} bytes; run the step once (eagerly, on any stream) before capturing, or call "
                           "bitorch_engine._hip.presize_workspace(nbytes, device, stream) with the capturing stream first")
    if buf is not None and (id(table), key) in _CAPTURED:
        _WS_RETIRED.append(buf)
        _CAPTURED.discard((id(table), key))  # the new buffer has not been seen by any capture
    size = max(nbytes, 1 << 20, 2 * buf.numel() if buf is not None else 0)
    buf = (torch.zeros if zero else torch.empty)(size, dtype=torch.uint8, device=device)
    table[key] = buf
    return buf


def _key(device, stream=None):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return (idx, (torch.cuda.current_stream(device) if stream is None else stream).cuda_stream)


def workspace(nbytes: int, device, stream=None) -> torch.Tensor:
    """Per-(device, stream) scratch buffer with the zero-initialised ticket / generation head, grown on demand and reused
    (stream-ordered reuse is safe).  See _grow() for the growth / retirement / capture policy."""
    if nbytes == 0:
        return None
    return _grow(_WS, _key(device, stream), nbytes, device, True)  # head = split-K counters, must start at 0


_SCRATCH = {}


def scratch(nbytes: int, device) -> torch.Tensor:
    """Per-(device, stream) scratch WITHOUT the ticket / generation head (im2col buffers and the like): grown on demand,
    superseded buffers retired like workspace()'s."""
    return _grow(_SCRATCH, _key(device), nbytes, device, False)


def presize_workspace(nbytes: int, device=None, stream=None):
    """Allocate the scratch buffer of `stream` (default: the current stream) up front (e.g. for the largest prefill) so that it
    never regrows after a decode graph has been captured; call it before capturing, on any stream of the device."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    return workspace(nbytes, device, stream)
