"""BinaryLinearCutlass / BinaryMatMul: mirror of reference layers/qlinear/binary/cutlass/layer.py.  The
scale is folded into the kernel epilogue ((K - 2*popc) * scale_a * scale_w)."""
import typing

import torch
from torch.autograd import Function

from bitorch_engine.utils import ste
from bitorch_engine.utils.safe_import import import_extension
from bitorch_engine.utils.model_helper import flatten_x, unflatten_x, init_weight
from ..layer import BinaryLinearBase, BinaryLinearParameter

binary_linear_cutlass = import_extension("binary_linear_cutlass")


class BinaryLinearForward(Function):
    """Forward with the scale in the kernel epilogue + straight-through backward (reference layer.py:62-126)."""

    @staticmethod
    def forward(ctx, x, weight, scale_a, scale_w, gemm_kernel_id, is_train):
        x2, lead = flatten_x(x)
        if is_train:
            ctx.save_for_backward(x2, weight, scale_w, scale_a)
        scale = scale_a.item() * scale_w.item()  # host read, like the reference (one sync per call)
        out = binary_linear_cutlass.forward(x2, weight, scale, False, gemm_kernel_id)
        return unflatten_x(out, lead).to(x.dtype)

    @staticmethod
    @typing.no_type_check
    def backward(ctx, output_gradient):
        gy2, lead = flatten_x(output_gradient)
        x2, weight, scale_w, scale_a = ctx.saved_tensors
        grad_x, grad_w, grad_scale_a = ste.binary_linear_backward(gy2, x2, weight, scale_a, scale_w)
        return unflatten_x(grad_x, lead), ste.integer_leaf_grad(weight, grad_w, ctx.needs_input_grad[1]), grad_scale_a, None, None, None


class BinaryMatMulFunction(Function):
    """sign(x) . sign(y)^T * x_clip * y_clip with the straight-through backward of both operands and both clips (reference layer.py:309-362)."""

    @staticmethod
    def forward(ctx, x, y, x_clip, y_clip):
        ctx.save_for_backward(x, y, x_clip, y_clip)
        k = x.size(-1)
        pad = (-k) % 8  # pad K with -1 on x and +1 on y: each padded position contributes -1, corrected below
        if pad:
            x = torch.nn.functional.pad(x, (0, pad), value=-1.0)
            y = torch.nn.functional.pad(y, (0, pad), value=1.0)
        out = binary_linear_cutlass.matmul(x, y, 1.0) + float(pad)
        return out.to(x.dtype) * x_clip * y_clip

    @staticmethod
    @typing.no_type_check
    def backward(ctx, output_gradient):
        x, y, x_clip, y_clip = ctx.saved_tensors
        grad_x = output_gradient.matmul(y.sign() * y_clip) * ste.clip_masks(x, x_clip, -1.0, 1.0)[3]
        grad_y = output_gradient.transpose(-1, -2).matmul(x.sign() * x_clip) * ste.clip_masks(y, y_clip, -1.0, 1.0)[3]
        return grad_x, grad_y, ste.binary_scale_grad(grad_x, x.sign()), ste.binary_scale_grad(grad_y, y.sign())


class BinaryLinearCutlass(BinaryLinearBase):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.gemm_kernel_id = 3
        self.bias_a = torch.nn.Parameter(torch.zeros(self.input_features, dtype=self.dtype))
        self.scale_a = torch.nn.Parameter(torch.tensor(0, dtype=self.dtype))
        self.scale_w = torch.nn.Parameter(torch.tensor(1, dtype=self.dtype), requires_grad=False)

    def prepare_params(self) -> None:
        w, s = init_weight(self.weight, cls=BinaryLinearParameter)
        self.weight = w
        self.scale_w.data = s.to(self.scale_w.dtype).reshape(())

    def generate_quantized_weight(self, qweight_only: bool = False) -> None:
        self.qweight = torch.nn.Parameter(binary_linear_cutlass.w_pack(self.weight.data, False), requires_grad=False)
        if qweight_only:
            self.weight = None

    def set_activation(self, x: torch.Tensor) -> torch.Tensor:
        if not self.scale_a.is_nonzero():
            self.scale_a.data = ((2 if self.symmetric else 4) * x.abs().mean()).to(self.dtype)
        return x + self.bias_a.expand_as(x)

    def set_weight_data(self, x: torch.Tensor):
        super().set_weight_data(x)
        self.prepare_params()

    def select_gemm_kernel(self, x: torch.Tensor) -> None:
        self.gemm_kernel_id = binary_linear_cutlass.kernel_eval(0, x.size(0), self.output_features, x.size(1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_forward(x)
        ste.refuse_eval_grad(self, x)
        x = self.set_activation(x)
        if ste.wants_grad(self):
            return BinaryLinearForward.apply(x, self.opt_weight, self.scale_a, self.scale_w, self.gemm_kernel_id, True)
        x2, lead = flatten_x(x)
        scale = self.scale_a.item() * self.scale_w.item()  # host read, like the reference (one sync per call)
        out = binary_linear_cutlass.forward(x2, self.opt_weight.data, scale, False, self.gemm_kernel_id)
        return unflatten_x(out, lead).to(x.dtype)


class BinaryMatMul(torch.nn.Module):
    """sign(x) . sign(y)^T over the last two dims, scaled by the two learnable clip values."""

    def __init__(self, dtype=torch.float, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.dtype = dtype
        self.x_clip = torch.nn.Parameter(torch.tensor(0, dtype=self.dtype))
        self.y_clip = torch.nn.Parameter(torch.tensor(0, dtype=self.dtype))

    def set_activation_scale(self, x: torch.Tensor, y: torch.Tensor) -> None:
        if not self.x_clip.is_nonzero():
            self.x_clip.data = (2 * x.abs().mean()).to(self.dtype)
        if not self.y_clip.is_nonzero():
            self.y_clip.data = (2 * y.abs().mean()).to(self.dtype)

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        assert x.dim() > 2 and y.dim() > 2, "Expected tensor dim > 2, but got input_dim: '{}', other_dim: {}".format(x.dim(), y.dim())
        self.set_activation_scale(x, y)
        return BinaryMatMulFunction.apply(x, y, self.x_clip, self.y_clip)
