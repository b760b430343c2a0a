"""Binary (1-bit W / 1-bit A) linear base types; mirror of reference layers/qlinear/binary/layer.py
(BinaryLinearParameter :8-60, BinaryLinearBase :63-217)."""
import math

import torch
from torch import nn


class BinaryLinearParameter(nn.Parameter):
    """Parameter holding int8 sign carriers (or packed uint8 words) of a binary layer."""

    def __new__(cls, data: torch.Tensor = None, requires_grad: bool = False):
        if data is not None and not data.is_floating_point():
            requires_grad = False
        return super().__new__(cls, data, requires_grad=requires_grad)

    @staticmethod
    def update(qweight: torch.nn.Parameter, exp_avg_s: torch.Tensor = None, exp_avg_l: torch.Tensor = None, step: torch.Tensor = None,
               lr: float = 1e-4, weight_decay: float = 0.0, beta1: float = 0.99, beta2: float = 0.9999, eps: float = 1e-6, dtype=torch.half,
               correct_bias=None, projector=None, grad: torch.Tensor = None) -> None:
        """The optimiser-side update of this parameter kind (reference layers/qlinear/binary/layer.py:20-60): utils/model_helper.qweight_update_fn on `qweight.grad`."""
        from bitorch_engine.utils.model_helper import qweight_update_fn
        assert isinstance(qweight, BinaryLinearParameter), 'Error: the type of qweight must be BinaryLinearParameter. '
        qweight_update_fn(qweight=qweight, exp_avg_s=exp_avg_s, exp_avg_l=exp_avg_l, step=step, lr=lr, weight_decay=weight_decay, beta1=beta1,
                          beta2=beta2, correct_bias=correct_bias, eps=eps, dtype=dtype, projector=projector, grad=grad)


class BinaryLinearBase(nn.Module):
    def __init__(self, input_features: int, out_features: int, device: torch.device = None,
                 dtype: torch.dtype = torch.float, symmetric: bool = True) -> None:
        super().__init__()
        self.bits_binary_word = 8
        self.input_features, self.output_features = input_features, out_features
        self.qweight = None
        self.device, self.dtype, self.symmetric = device, dtype, symmetric
        self.reset_parameters()

    def reset_parameters(self) -> None:
        w = torch.empty((self.output_features, self.input_features), dtype=self.dtype)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w)

    def set_weight_data(self, x: torch.Tensor) -> None:
        assert self.dtype == x.dtype, "dtype mismatch. Expected: '{}', but '{}' found".format(self.dtype, x.dtype)
        self.weight = nn.Parameter(x)

    def prepare_params(self) -> None:
        raise NotImplementedError("Subclasses should implement this method.")

    def set_quantized_weight_data(self, x: torch.Tensor) -> None:
        self.qweight = nn.Parameter(x, requires_grad=False)

    def generate_quantized_weight(self, qweight_only: bool = False) -> None:
        raise NotImplementedError("Subclasses should implement this method.")

    def _check_forward(self, x: torch.Tensor) -> None:
        packed_in = x.dtype is torch.uint8
        if not packed_in:
            assert x.size(-1) % self.bits_binary_word == 0, \
                "Input tensor dimension ({}) must be divisible by {}.".format(x.size(-1), self.bits_binary_word)
        if self.qweight is not None:
            per = 1 if packed_in else self.bits_binary_word
            expect = x.size(-1) * self.output_features / per
            assert self.qweight.nelement() == expect, "Weight and input tensor mismatch. {}:{}".format(self.qweight.nelement(), expect)
        else:
            k = self.weight.size(1) / self.bits_binary_word if packed_in else self.weight.size(1)
            assert k == x.size(-1), "Weight and input tensor mismatch."

    @property
    def opt_weight(self):
        if not self.training and self.qweight is None:
            self.generate_quantized_weight()
        return self.weight if self.training else self.qweight

    def set_bits_binary_word(self, num_bit: int) -> None:
        self.bits_binary_word = num_bit
