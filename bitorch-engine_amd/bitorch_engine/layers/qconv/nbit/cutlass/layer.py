"""Q4Conv2dCutlass (W4A4 conv): mirror of reference layers/qconv/nbit/cutlass/layer.py:79-146, inference path."""
import torch

from ..layer import nBitConv2dBase
from bitorch_engine.utils.safe_import import import_extension

q4_conv_cutlass = import_extension("q4_conv_cutlass")


class Q4Conv2dCutlass(nBitConv2dBase):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.bias_a = torch.nn.Parameter(torch.zeros(self.in_channels, dtype=self.dtype))
        self.scale_a = torch.nn.Parameter(torch.tensor(0, dtype=self.dtype))
        self.register_buffer("scale_w", torch.tensor(1, dtype=torch.float))
        self.register_buffer("eps", torch.tensor(0.00001).type(self.dtype))

    def prepare_params(self) -> None:
        s = 2 * self.weight.abs().mean() / 5.6345
        self.scale_w.data = torch.where(s > self.eps, s, self.eps).to(self.scale_w.dtype)

    def generate_quantized_weight(self, qweight_only: bool = False) -> None:
        self.qweight = torch.nn.Parameter(q4_conv_cutlass.w_pack(self.weight.data, self.scale_w), requires_grad=False)
        if qweight_only:
            self.weight = None

    def set_activation(self, x: torch.Tensor) -> torch.Tensor:
        if not self.scale_a.is_nonzero():
            self.scale_a.data = (2 * x.abs().mean() / 11.269).to(self.dtype)
        return x + self.bias_a.view(1, -1, 1, 1)

    def _check_forward(self, x: torch.Tensor) -> None:
        assert x.size(dim=1) % 32 == 0, "Input channel dimension must be divisible by 32."
        assert self.out_channels % 32 == 0, "Output channel dimension must be divisible by 32."

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_forward(x)
        x = self.set_activation(x)
        out = q4_conv_cutlass.forward(x, self.opt_weight.data, self.scale_a, self.scale_w, self.training, self.kernel_size, self.stride,
                                      self.padding, self.dilation)[0]
        return out.permute(0, 3, 1, 2)
