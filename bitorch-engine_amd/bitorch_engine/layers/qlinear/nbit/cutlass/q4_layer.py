"""Q4LinearCutlass / Q4MatMul (W4A4): mirror of reference layers/qlinear/nbit/cutlass/q4_layer.py, inference path.
scale_w = 2*mean|w| / 5.6345 (prepare_params), lazily initialised scale_a = 2*mean|x| / 11.269, learnable bias_a."""
import typing

import torch
from torch.autograd import Function

from ..layer import nBitLinearBase
from bitorch_engine.utils import ste
from bitorch_engine.utils.safe_import import import_extension
from bitorch_engine.utils.model_helper import flatten_x, unflatten_x
from bitorch_engine.functions.cuda import q4_unpack_and_scaling_tensor

q_linear_cutlass = import_extension("q_linear_cutlass")


class Q4LinearFunction(Function):
    """W4A4 forward on the i8 matrix cores + the reference's straight-through backward (q4_layer.py:60-100): the products of the
    backward are float GEMMs on the DEQUANTISED saved operands, the clip range of the activation is [-8, 7]."""

    @staticmethod
    def forward(ctx, x, weight, scale_a, scale_w, eps, is_train):
        x2, lead = flatten_x(x)
        out, q_a, q_w = q_linear_cutlass.q4_forward(x2, weight, scale_a, scale_w, False, is_train)
        if is_train:
            ctx.save_for_backward(x2, q_a, q_w, scale_a, scale_w)
        return unflatten_x(out.to(x2.dtype), lead)

    @staticmethod
    @typing.no_type_check
    def backward(ctx, output_gradient):
        gy2, lead = flatten_x(output_gradient)
        x2, q_a, q_w, scale_a, scale_w = ctx.saved_tensors
        grad_x = gy2.mm(q4_unpack_and_scaling_tensor(q_w, scale_w).to(gy2.dtype))           # [m, n] . [n, k]
        grad_w = gy2.t().mm(q4_unpack_and_scaling_tensor(q_a, scale_a).to(gy2.dtype)) if ctx.needs_input_grad[1] else None
        q, below, above, inside = ste.clip_masks(x2, scale_a, -8.0, 7.0)
        grad_x.mul_(inside)
        grad_scale_a = ste.nbit_scale_grad(q, below, above, inside, grad_x, -8.0, 7.0)
        return unflatten_x(grad_x, lead), grad_w, grad_scale_a.to(scale_a.dtype).reshape(scale_a.shape), None, None, None


class Q4MatMulFunction(Function):
    """Batched 4-bit x . y^T + backward (q4_layer.py:225-300): the gradient is itself quantised to 4 bit (scale 2*mean|g|/11.269) and both
    products run as 4-bit GEMMs on the saved packed operands (q_linear_cutlass.q4_matmul_backward); the clip range the reference applies
    to the operands in this backward is [-128, 127] (as written there, :282-283, :292-293)."""

    @staticmethod
    def forward(ctx, x, y, x_clip, y_clip, eps, is_train):
        out, q4_x, q4_y = q_linear_cutlass.q4_matmul(x, y, x_clip, y_clip)
        if is_train:
            ctx.save_for_backward(x, y, q4_x, q4_y, x_clip, y_clip)
        return out.to(x.dtype) * x_clip * y_clip

    @staticmethod
    @typing.no_type_check
    def backward(ctx, output_gradient):
        x, y, q4_x, q4_y, x_clip, y_clip = ctx.saved_tensors
        scale_grad = 2 * output_gradient.abs().mean() / 11.269
        grad_x, grad_y = q_linear_cutlass.q4_matmul_backward(output_gradient, q4_x, q4_y, x_clip, y_clip, scale_grad)
        outs = []
        for t, clip, g in ((x, x_clip, grad_x), (y, y_clip, grad_y)):
            q, below, above, inside = ste.clip_masks(t, clip, -128.0, 127.0)
            g = g.to(t.dtype).view(t.shape) * inside
            outs.append((g, ste.nbit_scale_grad(q, below, above, inside, g, -128.0, 127.0).to(clip.dtype).reshape(clip.shape)))
        return outs[0][0], outs[1][0], outs[0][1], outs[1][1], None, None


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
        ste.refuse_eval_grad(self, x)
        x = self.set_activation(x)
        if ste.wants_grad(self):
            return Q4LinearFunction.apply(x, self.opt_weight, self.scale_a, self.scale_w, self.eps, True)
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
        if ste.wants_grad(self):
            return Q4MatMulFunction.apply(x, y, self.x_clip, self.y_clip, self.eps, True)
        ste.refuse_eval_grad(self, x)
        ste.refuse_eval_grad(self, y)
        out = q_linear_cutlass.q4_matmul(x, y, self.x_clip, self.y_clip)[0]
        return out.to(x.dtype) * self.x_clip * self.y_clip
