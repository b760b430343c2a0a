"""Python wrappers with the reference's names (functions/cuda/functions.py:8-177)."""
import torch

from bitorch_engine.utils.safe_import import import_extension

functions_cuda = import_extension("functions_cuda")


def fp32toint4(input: torch.Tensor) -> torch.Tensor:
    return functions_cuda.fp32toint4(input)


def tensor_to_packed_uint8(input: torch.Tensor) -> torch.Tensor:
    """Sign bits of the last dimension packed 8 per uint8, LSB first."""
    return functions_cuda.tensor_pack_to_uint8(input)


def unpack_uint8_tensor(input: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """uint8 [..., P] -> float [..., 8P] of +-scale (scale broadcast over the last dimension)."""
    return functions_cuda.uint8_to_unpacked_tensor(input, scale)


def q4_pack_tensor(input: torch.Tensor, is_transpose: bool = False) -> torch.Tensor:
    assert input.dtype == torch.int32, "Error: input tensor dtype should be int32"
    return functions_cuda.q4_pack(input, is_transpose)


def q4_unpack_tensor(input: torch.Tensor, is_transpose: bool = False) -> torch.Tensor:
    assert input.dtype == torch.int8, "Error: input tensor dtype should be int8."
    return functions_cuda.q4_unpack(input, is_transpose)


def q4_unpack_and_scaling_tensor(input: torch.Tensor, scale: float, is_transpose: bool = False) -> torch.Tensor:
    assert input.dtype == torch.int8, "Error: input tensor dtype should be int8."
    return functions_cuda.q4_unpack_and_scaling(input, scale, is_transpose)
