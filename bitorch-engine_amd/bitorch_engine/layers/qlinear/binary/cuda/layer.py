"""BinaryLinearCuda: mirror of reference layers/qlinear/binary/cuda/layer.py.  out = (K - 2*popc) * scale_a *
scale_w with the lazily initialised activation scale (2*mean|x|) and learnable activation bias."""
import typing

import torch
from torch.autograd import Function

from bitorch_engine.utils import ste
from bitorch_engine.utils.safe_import import import_extension
from bitorch_engine.utils.model_helper import flatten_x, unflatten_x, init_weight
from ..layer import BinaryLinearBase, BinaryLinearParameter
from .bmm import BMM

binary_linear_cuda = import_extension("binary_linear_cuda")


class BinaryLinearForward(Function):
    """Training-mode forward + straight-through backward (reference layer.py:67-120): the XNOR-popcount product is this library's
    kernel, the backward is utils.ste.binary_linear_backward."""

    @staticmethod
    def forward(ctx, x, weight, bmm_type, scale_a, scale_w, is_train):
        x2, lead = flatten_x(x)
        if is_train:
            ctx.save_for_backward(x2, weight, scale_w, scale_a)
        out = binary_linear_cuda.forward(x2, weight, bmm_type, True).to(x.dtype)
        return unflatten_x(out, lead) * scale_a * scale_w

    @staticmethod
    @typing.no_type_check
    def backward(ctx, output_gradient):
        gy2, lead = flatten_x(output_gradient)
        x2, weight, scale_w, scale_a = ctx.saved_tensors
        grad_x, grad_w, grad_scale_a = ste.binary_linear_backward(gy2, x2, weight, scale_a, scale_w)
        return unflatten_x(grad_x, lead), ste.integer_leaf_grad(weight, grad_w, ctx.needs_input_grad[1]), None, grad_scale_a, None, None


class BinaryLinearCuda(BinaryLinearBase):
    def __init__(self, *args, bmm_type: BMM = BMM.ADAPTIVE, **kwargs):
        super().__init__(*args, **kwargs)
        self.bmm_type = bmm_type
        self.bias_a = torch.nn.Parameter(torch.zeros(self.input_features, dtype=self.dtype))
        self.scale_a = torch.nn.Parameter(torch.tensor(0, dtype=self.dtype))
        self.register_buffer("scale_w", torch.tensor(1, dtype=self.dtype))

    def prepare_params(self) -> None:
        w, s = init_weight(self.weight, cls=BinaryLinearParameter)
        self.weight = w
        self.scale_w.data = s.to(self.scale_w.dtype).reshape(())

    def generate_quantized_weight(self, qweight_only: bool = False) -> None:
        self.qweight = torch.nn.Parameter(binary_linear_cuda.w_pack(self.weight.data, self.bmm_type.value, True), requires_grad=False)
        if qweight_only:
            self.weight = None

    @staticmethod
    def w_pack(weights: torch.Tensor, bmm_type: BMM) -> torch.Tensor:
        return binary_linear_cuda.w_pack(weights, bmm_type.value, True)

    def _init_scale_a(self, x: torch.Tensor) -> None:
        # lazily initialised activation scale (reference layer.py set_activation); the reference asks `is_nonzero()` -- a
        # device sync -- on every forward, here the answer is remembered per version of the parameter
        from bitorch_engine.extensions.q_linear_cuda import _cached
        if not _cached(self.scale_a, "nonzero", lambda: bool(self.scale_a.is_nonzero())):
            self.scale_a.data = ((2 if self.symmetric else 4) * x.abs().mean()).to(self.dtype)

    def set_activation(self, x: torch.Tensor) -> torch.Tensor:
        self._init_scale_a(x)
        return x + self.bias_a.expand_as(x)

    def set_weight_data(self, x: torch.Tensor) -> None:
        super().set_weight_data(x)
        self.prepare_params()

    def forward(self, x: torch.Tensor, bmm_type: BMM = BMM.ADAPTIVE) -> torch.Tensor:
        self._check_forward(x)
        self.bmm_type = bmm_type
        ste.refuse_eval_grad(self, x)
        if ste.wants_grad(self):  # training: Function with the straight-through backward (x, bias_a through x, scale_a, the weight's carriers)
            x = self.set_activation(x)
            return BinaryLinearForward.apply(x, self.opt_weight, self.bmm_type.value, self.scale_a, self.scale_w, True)
        if not torch.is_grad_enabled() or not (x.requires_grad or self.bias_a.requires_grad or self.scale_a.requires_grad):
            # no gradient can flow (the fused output is detached from bias_a / scale_a: fine-tuning those under model.eval() must
            # take the differentiable composition below).  M <= 64 (<= 512 when K % 512 == 0): the whole layer (activation bias + sign-pack, XNOR-popcount,
            # cast, both scales) in ONE launch
            self._init_scale_a(x)
            x2, lead = flatten_x(x)
            out = binary_linear_cuda.layer_forward(x2, self.bias_a.data, self.opt_weight, self.bmm_type.value,
                                                   self.scale_a.data, self.scale_w) if x2.dtype == self.bias_a.dtype else None
            if out is not None:
                return unflatten_x(out, lead)
        x = self.set_activation(x)
        x2, lead = flatten_x(x)
        out = binary_linear_cuda.forward(x2, self.opt_weight, self.bmm_type.value, True).to(x.dtype)  # the Parameter itself: conversions are memoised on it
        return unflatten_x(out, lead) * self.scale_a * self.scale_w
