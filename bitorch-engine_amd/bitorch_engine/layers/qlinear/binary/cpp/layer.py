"""BinaryLinearCPP: mirror of reference layers/qlinear/binary/cpp/layer.py:42-125.  Same packed-weight
layout as the reference's CPU extension (column bit-planes), computed by the HIP XNOR kernels."""
import torch

from bitorch_engine.utils.safe_import import import_extension
from bitorch_engine.utils.model_helper import flatten_x, unflatten_x
from ..layer import BinaryLinearBase

binary_linear_cpp = import_extension("binary_linear_cpp")


class BinaryLinearCPP(BinaryLinearBase):
    def __init__(self, input_features: int, out_features: int, device: torch.device = None) -> None:
        super().__init__(input_features, out_features, device)

    def prepare_params(self) -> None:
        pass

    def generate_quantized_weight(self, qweight_only: bool = False) -> None:
        self.qweight = binary_linear_cpp.w_pack(self.weight.data, self.output_features, self.input_features)
        if qweight_only:
            self.weight = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_forward(x)
        x2, lead = flatten_x(x)
        out = binary_linear_cpp.forward(x2, self.opt_weight, x2.size(0), self.output_features, x2.size(1))
        return unflatten_x(out, lead)
