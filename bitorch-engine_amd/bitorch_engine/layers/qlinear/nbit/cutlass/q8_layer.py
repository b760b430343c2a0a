"""Q8LinearCutlass (W8A8): mirror of reference layers/qlinear/nbit/cutlass/q8_layer.py, inference path."""
import torch

from ..layer import nBitLinearBase
from bitorch_engine.utils.safe_import import import_extension
from bitorch_engine.utils.model_helper import flatten_x, unflatten_x
from bitorch_engine.utils.quant_operators import q8_quantization

q_linear_cutlass = import_extension("q_linear_cutlass")


class Q8LinearCutlass(nBitLinearBase):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.bias_a = torch.nn.Parameter(torch.zeros(self.in_channels, dtype=self.dtype))
        self.scale_a = torch.nn.Parameter(torch.tensor(0, dtype=torch.float))
        self.register_buffer("scale_w", torch.tensor(1, dtype=torch.float))
        self.register_buffer("eps", torch.tensor(0.00001).type(self.dtype))

    def prepare_params(self) -> None:
        pass

    def generate_quantized_weight(self, qweight_only: bool = False) -> None:
        qw, s = q8_quantization(self.weight.data, None, self.eps)
        self.scale_w.data = s.to(self.scale_w.dtype)
        self.qweight = torch.nn.Parameter(qw.to(torch.int8), requires_grad=False)
        if qweight_only:
            self.weight = None

    def _check_forward(self, x: torch.Tensor) -> None:
        assert x.size(-1) == self.in_channels, "Error: input and weights' dim mismatch."

    def set_activation(self, x: torch.Tensor) -> torch.Tensor:
        if not self.scale_a.is_nonzero():
            self.scale_a.data = (2 * x.abs().mean() / 11.269).to(self.scale_a.dtype)
        return x + self.bias_a.expand_as(x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_forward(x)
        x = self.set_activation(x)
        q_a = q8_quantization(x, self.scale_a, self.eps).to(torch.int8) if x.dtype != torch.int8 else x
        q_a, lead = flatten_x(q_a)
        w, scale_w = self.opt_weight.data, self.scale_w
        if w.dtype != torch.int8:  # float weight (training-mode call): quantise on the fly like Q8LinearFunction.forward :44-53
            w, scale_w = q8_quantization(w, None, self.eps)
            w = w.to(torch.int8)
        out = q_linear_cutlass.q8_forward(q_a, w, False, self.scale_a, scale_w)
        return unflatten_x(out, lead)
