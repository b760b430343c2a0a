"""Drop-in for `functions_cuda` (functions/cuda/functions_cuda.cpp:195-202)."""
import torch

from bitorch_engine import _hip
from ._binary_common import pack_rows


def fp32toint4(input):
    raise RuntimeError("fp32toint4 is not provided: the reference kernel is buggy (loop bound 7, stride 4, "
                       "functions_cuda_kernel.cu:63-65) and untested, see SURVEY.md section 2.1")


def tensor_pack_to_uint8(data: torch.Tensor) -> torch.Tensor:
    return pack_rows(data)


def uint8_to_unpacked_tensor(emd: torch.Tensor, scl: torch.Tensor) -> torch.Tensor:
    _hip.need_gpu(emd, scl)
    emd = emd.contiguous()
    scl = scl.to(torch.float32).contiguous()
    pd = emd.shape[-1]
    out = torch.empty(emd.shape[:-1] + (pd * 8,), dtype=torch.float32, device=emd.device)
    if emd.numel():
        rc = _hip.lib().bie_unpack_u8_scaled(_hip.ptr(emd), _hip.ptr(scl), _hip.ptr(out), emd.numel(), pd, _hip.stream())
        _hip.check(rc, "bie_unpack_u8_scaled")
    return out


def q4_pack(data: torch.Tensor, is_transpose: bool) -> torch.Tensor:
    _hip.need_gpu(data)
    if data.dim() not in (2, 3):
        raise RuntimeError(f"tensor sizes not supported: {data.dim()}")
    data = data.contiguous()
    out = torch.empty(data.shape[:-1] + (data.shape[-1] // 2,), dtype=torch.int8, device=data.device)
    rc = _hip.lib().bie_q4_pack(_hip.ptr(data), _hip.ptr(out), out.numel(), _hip.stream())
    _hip.check(rc, "bie_q4_pack")
    return out.transpose(-1, -2).contiguous() if is_transpose else out


def q4_unpack(packed: torch.Tensor, is_transpose: bool) -> torch.Tensor:
    _hip.need_gpu(packed)
    packed = (packed.transpose(-1, -2) if is_transpose else packed).contiguous()
    out = torch.empty(packed.shape[:-1] + (packed.shape[-1] * 2,), dtype=torch.int32, device=packed.device)
    rc = _hip.lib().bie_q4_unpack(_hip.ptr(packed), _hip.ptr(out), packed.numel(), _hip.stream())
    _hip.check(rc, "bie_q4_unpack")
    return out


def q4_unpack_and_scaling(packed: torch.Tensor, scale: float, is_transpose: bool) -> torch.Tensor:
    _hip.need_gpu(packed)
    packed = (packed.transpose(-1, -2) if is_transpose else packed).contiguous()
    out = torch.empty(packed.shape[:-1] + (packed.shape[-1] * 2,), dtype=torch.float32, device=packed.device)
    rc = _hip.lib().bie_q4_unpack_scale(_hip.ptr(packed), _hip.ptr(out), packed.numel(), float(scale), _hip.stream())
    _hip.check(rc, "bie_q4_unpack_scale")
    return out
