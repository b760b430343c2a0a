"""Q8LinearCutlass (W8A8): mirror of reference layers/qlinear/nbit/cutlass/q8_layer.py, inference path."""
import typing

import torch
from torch.autograd import Function

from ..layer import nBitLinearBase
from bitorch_engine.utils import ste
from bitorch_engine.utils.safe_import import import_extension
from bitorch_engine.utils.model_helper import flatten_x, unflatten_x
from bitorch_engine.utils.quant_operators import q8_quantization

q_linear_cutlass = import_extension("q_linear_cutlass")


class Q8LinearFunction(Function):
    """W8A8 forward on the i8 matrix cores + the reference's straight-through backward (q8_layer.py:64-110): float GEMMs on the saved int8
    operands times their scales, clip range [-128, 127], and -- as the reference has it, :99 -- grad_x additionally times scale_a."""

    @staticmethod
    def forward(ctx, x, weight, scale_a, scale_w, eps, is_train):
        q_a = q8_quantization(x, scale_a, eps).to(torch.int8) if x.dtype != torch.int8 else x
        q_a, lead = flatten_x(q_a)
        if weight.dtype != torch.int8:  # float weight (training): quantised on the fly, per-tensor max scale (:44-53)
            q_w, scale_w = q8_quantization(weight, None, eps)
            q_w = q_w.to(torch.int8)
        else:
            q_w = weight
        out = q_linear_cutlass.q8_forward(q_a, q_w, False, scale_a, scale_w)
        if is_train:
            ctx.save_for_backward(x, q_a, q_w, scale_a, scale_w)
        return unflatten_x(out, lead)

    @staticmethod
    @typing.no_type_check
    def backward(ctx, output_gradient):
        gy2, lead = flatten_x(output_gradient)
        x, q_a, q_w, scale_a, scale_w = ctx.saved_tensors
        grad_a = gy2.mm(q_w.to(gy2.dtype) * scale_w)
        grad_w = gy2.t().mm(q_a.to(gy2.dtype) * scale_a) if ctx.needs_input_grad[1] else None
        x2 = x.reshape(grad_a.shape)
        q, below, above, inside = ste.clip_masks(x2, scale_a, -128.0, 127.0)
        grad_x = grad_a * inside
        grad_x.mul_(scale_a)
        grad_scale_a = ste.nbit_scale_grad(q, below, above, inside, grad_x, -128.0, 127.0)
        return unflatten_x(grad_x.to(x.dtype), lead), grad_w, grad_scale_a.to(scale_a.dtype).reshape(scale_a.shape), None, None, None


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
        ste.refuse_eval_grad(self, x)
        x = self.set_activation(x)
        if ste.wants_grad(self):
            return Q8LinearFunction.apply(x, self.opt_weight, self.scale_a, self.scale_w, self.eps, True)
        q_a = q8_quantization(x, self.scale_a, self.eps).to(torch.int8) if x.dtype != torch.int8 else x
        q_a, lead = flatten_x(q_a)
        w, scale_w = self.opt_weight.data, self.scale_w
        if w.dtype != torch.int8:  # float weight (training-mode call): quantise on the fly like Q8LinearFunction.forward :44-53
            w, scale_w = q8_quantization(w, None, self.eps)
            w = w.to(torch.int8)
        out = q_linear_cutlass.q8_forward(q_a, w, False, self.scale_a, scale_w)
        return unflatten_x(out, lead)
