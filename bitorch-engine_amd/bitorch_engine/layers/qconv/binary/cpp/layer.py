"""BinaryConv2dCPP: mirror of reference layers/qconv/binary/cpp/layer.py:63-131.  Weights are sign-packed
row-wise over the flattened [OC, C*k*k] matrix (the reference's get_binary_row), padding counts as -1."""
import torch

from bitorch_engine.utils.safe_import import import_extension
from bitorch_engine.extensions._binary_common import pack_rows
from ..layer import BinaryConv2dBase

binary_conv_cpp = import_extension("binary_conv_cpp")


class BinaryConv2dCPP(BinaryConv2dBase):
    def prepare_params(self) -> None:
        pass

    def generate_quantized_weight(self, qweight_only: bool = False) -> None:
        packed = pack_rows(self.weight.data.reshape(self.out_channels, -1)).reshape(-1)
        self.qweight = torch.nn.Parameter(packed, requires_grad=False)
        if qweight_only:
            self.weight = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_forward(x)
        m = self.out_channels
        k = x.size(1) * self.kernel_size * self.kernel_size
        output_edge = int((x.size(2) - self.kernel_size + 2 * self.padding) / self.stride + 1)
        w = self.opt_weight
        w = w if w.dtype == torch.uint8 else w.data.reshape(m, k)  # packed: the Parameter itself (the tap re-layout is memoised on it)
        return binary_conv_cpp.forward(x, w, m, output_edge * output_edge, k, self.kernel_size, self.stride, self.padding,
                                       self.dilation, output_edge)
