"""Binary conv2d base types; mirror of reference layers/qconv/binary/layer.py:6-137."""
import torch
from torch import nn


class BinaryConvParameter(nn.Parameter):
    def __new__(cls, data: torch.Tensor = None, requires_grad: bool = False):
        if data is not None and not data.is_floating_point():
            requires_grad = False
        return super().__new__(cls, data, requires_grad=requires_grad)

    @staticmethod
    def update(qweight: torch.nn.Parameter, exp_avg_s: torch.Tensor = None, exp_avg_l: torch.Tensor = None, step: torch.Tensor = None,
               lr: float = 1e-4, weight_decay: float = 0.0, beta1: float = 0.99, beta2: float = 0.9999, eps: float = 1e-6, dtype=torch.half,
               correct_bias=None, projector=None, grad: torch.Tensor = None) -> None:
        """The optimiser-side update of this parameter kind (reference layers/qconv/binary/layer.py): utils/model_helper.qweight_update_fn on `qweight.grad`."""
        from bitorch_engine.utils.model_helper import qweight_update_fn
        assert isinstance(qweight, BinaryConvParameter), 'Error: the type of qweight must be BinaryConvParameter. '
        qweight_update_fn(qweight=qweight, exp_avg_s=exp_avg_s, exp_avg_l=exp_avg_l, step=step, lr=lr, weight_decay=weight_decay, beta1=beta1,
                          beta2=beta2, correct_bias=correct_bias, eps=eps, dtype=dtype, projector=projector, grad=grad)


class BinaryConv2dBase(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, padding: int = 0,
                 dilation: int = 1, device=None, dtype: torch.dtype = torch.float, symmetric: bool = True) -> None:
        super().__init__()
        self.bits_binary_word = 8
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.qweight = None
        self.device, self.dtype, self.symmetric = device, dtype, symmetric
        self.reset_parameters()

    def reset_parameters(self) -> None:
        self.weight = nn.Parameter(torch.empty((self.out_channels, self.in_channels, self.kernel_size, self.kernel_size)))

    def set_weight_data(self, x: torch.Tensor) -> None:
        self.weight = nn.Parameter(x, requires_grad=False)

    def set_quantized_weight_data(self, x: torch.Tensor) -> None:
        self.qweight = nn.Parameter(x, requires_grad=False)

    def generate_quantized_weight(self, qweight_only: bool = False) -> None:
        raise NotImplementedError("Subclasses should implement this method.")

    def prepare_params(self) -> None:
        raise NotImplementedError("Subclasses should implement this method.")

    def _check_forward(self, x: torch.Tensor) -> None:
        assert x.size(1) % self.bits_binary_word == 0, "Input tensor dimension must be divisible by {}.".format(self.bits_binary_word)
        assert x.size(1) == self.in_channels, "Dimension mismatch of the input Tensor {}:{}".format(x.size(1), self.in_channels)

    @property
    def opt_weight(self):
        if not self.training and self.qweight is None:
            self.generate_quantized_weight()
        return self.weight if self.training else self.qweight

    def set_bits_binary_word(self, num_bit: int) -> None:
        self.bits_binary_word = num_bit
