"""Q4Conv2dCutlass (W4A4 conv): mirror of reference layers/qconv/nbit/cutlass/layer.py:79-146, inference path."""
import typing

import torch
from torch.autograd import Function

from ..layer import nBitConv2dBase
from bitorch_engine.utils import ste
from bitorch_engine.utils.safe_import import import_extension
from bitorch_engine.functions.cuda import q4_unpack_and_scaling_tensor

q4_conv_cutlass = import_extension("q4_conv_cutlass")


class Q4Conv2dCutlassForward(Function):
    """W4A4 convolution forward + the reference's straight-through backward (layer.py:64-112): torch.nn.grad.conv2d_* on the dequantised
    saved operands (stored NHWC-viewed, permuted back), clip range [-8, 7]."""

    @staticmethod
    def forward(ctx, x, weight, scale_a, scale_w, is_train, kernel_size, stride, padding, dilation):
        out, q_a, q_w = q4_conv_cutlass.forward(x, weight, scale_a, scale_w, is_train, kernel_size, stride, padding, dilation)
        if is_train:
            ctx.save_for_backward(x, q_a, q_w, scale_w, scale_a)
            ctx.geometry = (stride, padding, dilation, weight.shape)
        return out.permute(0, 3, 1, 2)

    @staticmethod
    @typing.no_type_check
    def backward(ctx, output_gradient):
        x, q_a, q_w, scale_w, scale_a = ctx.saved_tensors
        stride, padding, dilation, weight_shape = ctx.geometry
        gy = output_gradient.contiguous()
        w_hat = q4_unpack_and_scaling_tensor(q_w, scale_w).permute(0, 3, 1, 2).to(gy.dtype)
        grad_x = torch.nn.grad.conv2d_input(x.shape, w_hat, gy, stride=stride, padding=padding, dilation=dilation)
        grad_w = None
        if ctx.needs_input_grad[1]:
            a_hat = q4_unpack_and_scaling_tensor(q_a, scale_a).permute(0, 3, 1, 2).to(gy.dtype)
            grad_w = torch.nn.grad.conv2d_weight(a_hat, weight_shape, gy, stride=stride, padding=padding, dilation=dilation)
        q, below, above, inside = ste.clip_masks(x, scale_a, -8.0, 7.0)
        grad_x.mul_(inside)
        grad_scale_a = ste.nbit_scale_grad(q, below, above, inside, grad_x, -8.0, 7.0)
        return (grad_x, grad_w, grad_scale_a.to(scale_a.dtype).reshape(scale_a.shape)) + (None,) * 6


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
        ste.refuse_eval_grad(self, x)
        x = self.set_activation(x)
        if ste.wants_grad(self):
            return Q4Conv2dCutlassForward.apply(x, self.opt_weight, self.scale_a, self.scale_w, True, self.kernel_size, self.stride, self.padding, self.dilation)
        out = q4_conv_cutlass.forward(x, self.opt_weight.data, self.scale_a, self.scale_w, self.training, self.kernel_size, self.stride,
                                      self.padding, self.dilation)[0]
        return out.permute(0, 3, 1, 2)
