"""BinaryConv2dCutlass: mirror of reference layers/qconv/binary/cutlass/layer.py (activation scale/bias,
int8 sign carriers, packed weights); see extensions/binary_conv2d_cutlass.py for the output convention."""
import typing

import torch
from torch.autograd import Function

from bitorch_engine.utils import ste
from bitorch_engine.utils.safe_import import import_extension
from bitorch_engine.utils.model_helper import init_weight
from ..layer import BinaryConv2dBase, BinaryConvParameter

binary_conv2d_cutlass = import_extension("binary_conv2d_cutlass")


class BinaryConv2dForward(Function):
    """Binary convolution forward (this library's kernel) + straight-through backward (reference layer.py:57-110)."""

    @staticmethod
    def forward(ctx, x, weight, scale_a, scale_w, is_train, kernel_size, stride, padding, dilation, run):
        if is_train:
            ctx.save_for_backward(x, weight, scale_w, scale_a)
            ctx.geometry = (stride, padding, dilation)
        out = run(x, weight, scale_a.item() * scale_w.item(), is_train, kernel_size, stride, padding, dilation)
        return out.to(x.dtype)

    @staticmethod
    @typing.no_type_check
    def backward(ctx, output_gradient):
        x, weight, scale_w, scale_a = ctx.saved_tensors
        stride, padding, dilation = ctx.geometry
        grad_x, grad_w, grad_scale_a = ste.binary_conv_backward(output_gradient, x, weight, scale_a, scale_w, stride, padding, dilation)
        return (grad_x, ste.integer_leaf_grad(weight, grad_w, ctx.needs_input_grad[1]), grad_scale_a) + (None,) * 7


class BinaryConv2dCutlass(BinaryConv2dBase):
    def __init__(self, *args, reference_convention=None, **kwargs):
        super().__init__(*args, **kwargs)
        # the reference kernel's own output convention (viewed layouts, raw popcounts, NHWC result) instead of the A15 convolution: opt-in,
        # see extensions/binary_conv2d_cutlass.forward_reference_convention
        self.reference_convention = binary_conv2d_cutlass.reference_convention_default() if reference_convention is None else bool(reference_convention)
        self.bias_a = torch.nn.Parameter(torch.zeros(self.in_channels, dtype=self.dtype))
        self.scale_a = torch.nn.Parameter(torch.tensor(0, dtype=torch.float))
        self.scale_w = torch.nn.Parameter(torch.tensor(1, dtype=torch.float), requires_grad=False)

    def prepare_params(self) -> None:
        w, s = init_weight(self.weight, cls=BinaryConvParameter)
        self.weight = w
        self.scale_w.data = s.to(self.scale_w.dtype).reshape(())

    def generate_quantized_weight(self, qweight_only: bool = False) -> None:
        self.qweight = torch.nn.Parameter(binary_conv2d_cutlass.w_pack(self.weight.data), requires_grad=False)
        if qweight_only:
            self.weight = None

    def set_activation(self, x: torch.Tensor) -> torch.Tensor:
        if not self.scale_a.is_nonzero():
            self.scale_a.data = ((2 if self.symmetric else 4) * x.abs().mean()).to(self.scale_a.dtype)
        return x + self.bias_a.view(1, -1, 1, 1)

    def set_weight_data(self, x: torch.Tensor) -> None:
        super().set_weight_data(x)
        self.prepare_params()

    def _check_forward(self, x: torch.Tensor) -> None:
        assert x.numel() % self.bits_binary_word == 0, "Input tensor dimension must be divisible by {}.".format(self.bits_binary_word)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_forward(x)
        ste.refuse_eval_grad(self, x)
        x = self.set_activation(x)
        fwd = binary_conv2d_cutlass.forward_reference_convention if self.reference_convention else binary_conv2d_cutlass.forward
        if ste.wants_grad(self):
            if self.reference_convention:
                raise RuntimeError("BinaryConv2dCutlass(reference_convention=True) is an inference compatibility mode: its NHWC / raw-popcount output has no "
                                   "straight-through backward (the reference's own backward, layer.py:80-108, is that of the NCHW convolution)")
            return BinaryConv2dForward.apply(x, self.opt_weight, self.scale_a, self.scale_w, True, self.kernel_size, self.stride, self.padding, self.dilation, fwd)
        scale = self.scale_a.item() * self.scale_w.item()
        out = fwd(x, self.opt_weight, scale, self.training, self.kernel_size, self.stride, self.padding, self.dilation)
        return out.to(x.dtype)
