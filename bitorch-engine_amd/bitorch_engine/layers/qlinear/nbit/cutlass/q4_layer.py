"""Q4LinearCutlass / Q4MatMul (W4A4): mirror of reference layers/qlinear/nbit/cutlass/q4_layer.py, inference path.
scale_w = 2*mean|w| / 5.6345 (prepare_params), lazily initialised scale_a = 2*mean|x| / 11.269, learnable bias_a."""
import torch

from ..layer import nBitLinearBase
from bitorch_engine.utils.safe_import import import_extension
from bitorch_engine.utils.model_helper import flatten_x, unflatten_x

q_linear_cutlass = import_extension("q_linear_cutlass")


class Q4LinearCutlass(nBitLinearBase):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.bias_a = torch.nn.Parameter(torch.zeros(self.in_channels, dtype=self.dtype))
        self.scale_a = torch.nn.Parameter(torch.tensor(0, dtype=torch.float))
        self.register_buffer("scale_w", torch.tensor(1, dtype=torch.float))
        self.register_buffer("eps", torch.tensor(0.00001).type(self.dtype))

    def prepare_params(self) -> None:
        s = 2 * self.weight.abs().mean() / 5.6345
        self.scale_w.data = torch.where(s > self.eps, s, self.eps).to(self.scale_w.dtype)

    def generate_quantized_weight(self, qweight_only: bool = False) -> None:
        self.qweight = torch.nn.Parameter(q_linear_cutlass.q4_w_pack(self.weight.data, self.scale_w), requires_grad=False)
        if qweight_only:
            self.weight = None

    def _check_forward(self, x: torch.Tensor) -> None:
        assert x.size(-1) == self.in_channels, "Error: input and weights' dim mismatch."
        assert self.in_channels % 32 == 0, "Input channel dimension must be divisible by 32."

    def set_activation(self, x: torch.Tensor) -> torch.Tensor:
        if not self.scale_a.is_nonzero():
            self.scale_a.data = (2 * x.abs().mean() / 11.269).to(self.scale_a.dtype)
        return x + self.bias_a.expand_as(x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_forward(x)
        x = self.set_activation(x)
        x2, lead = flatten_x(x)
        out = q_linear_cutlass.q4_forward(x2, self.opt_weight.data, self.scale_a, self.scale_w, False, self.training)[0]
        return unflatten_x(out.to(x.dtype), lead)


class Q4MatMul(torch.nn.Module):
    def __init__(self, dtype=torch.float, device=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.device, self.dtype = device, dtype
        self.x_clip = torch.nn.Parameter(torch.tensor(0, dtype=self.dtype))
        self.y_clip = torch.nn.Parameter(torch.tensor(0, dtype=self.dtype))
        self.register_buffer("eps", torch.tensor(0.00001).type(self.dtype))

    def set_activation_scale(self, x: torch.Tensor, y: torch.Tensor) -> None:
        if not self.x_clip.is_nonzero():
            self.x_clip.data = (2 * x.abs().mean() / 11.269).to(self.dtype)
        if not self.y_clip.is_nonzero():
            self.y_clip.data = (2 * x.abs().mean() / 11.269).to(self.dtype)  # the reference derives both from x (:q4_layer)

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        assert x.dim() > 2 and y.dim() > 2, "Expected tensor dim > 2, but got input_dim: '{}', other_dim: {}".format(x.dim(), y.dim())
        self.set_activation_scale(x, y)
        out = q_linear_cutlass.q4_matmul(x, y, self.x_clip, self.y_clip)[0]
        return out.to(x.dtype) * self.x_clip * self.y_clip
