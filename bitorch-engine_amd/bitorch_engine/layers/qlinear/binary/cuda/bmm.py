from enum import Enum


class BMM(Enum):
    """Kernel selector of the reference's CUDA layer (layers/qlinear/binary/cuda/bmm.py).  All values map
    to the same wave64 XNOR-popcount kernels on MI355X; kept so that callers' arguments stay valid."""
    BSTC32 = 1
    BTC32 = 2
    ADAPTIVE = 3
