"""Base types of the W-n / A16 path, mirroring reference layers/qlinear/nbit/layer.py:8-119 (the
parameter class) and :282-511 (MPQLinearBase): identical constructor arguments, attribute names and
-- the on-disk contract -- identical state_dict keys / shapes / dtypes (tests/golden/
state_dict_tables.json holds the reference's tables)."""
import math

import torch
from torch import nn

_QATTRS = ("privileged_grad", "scales", "zeros", "g_idx", "w_bit", "asym", "group_size", "layer_type", "q_perm",
           "qscales_zeros", "qscales_scales", "qzeros_zeros", "qzeros_scales", "q_group_map", "rows")
_QDEFAULTS = dict(w_bit=-1, asym=False, group_size=-1, layer_type=-1)


class MPQWeightParameter(nn.Parameter):
    """nn.Parameter holding the packed int32 weights plus the quantisation metadata the kernels need
    (scales, zeros, g_idx, w_bit, asym, group_size, layer_type 1=MPQ / 2=MBWQ, q_perm, q_group_map,
    rows, ...) as plain attributes."""

    def __new__(cls, data=None, requires_grad: bool = False, **meta):
        if data is not None and not data.is_floating_point():
            requires_grad = False  # stock torch: integer tensors cannot require grad
        return super().__new__(cls, data, requires_grad=requires_grad)

    def __init__(self, data=None, requires_grad: bool = False, **meta):
        unknown = set(meta) - set(_QATTRS)
        if unknown:
            raise TypeError(f"MPQWeightParameter: unexpected arguments {sorted(unknown)}")
        for name in _QATTRS:
            setattr(self, name, meta.get(name, _QDEFAULTS.get(name)))

    @staticmethod
    def update(qweight, exp_avg_s=None, exp_avg_l=None, step=None, lr=1e-4, weight_decay=0.0, beta1=0.99, beta2=0.9999, eps=1e-6,
               dtype=torch.half, correct_bias=None, projector=None, grad=None) -> None:
        """The optimizer-side re-pack step (reference nbit/layer.py:86-119 -> utils/model_helper.py:363-532), on the device."""
        from bitorch_engine.utils.model_helper import qweight_update_fn
        qweight_update_fn(qweight=qweight, exp_avg_s=exp_avg_s, exp_avg_l=exp_avg_l, step=step, lr=lr, weight_decay=weight_decay,
                          beta1=beta1, beta2=beta2, eps=eps, dtype=dtype, correct_bias=correct_bias, projector=projector, grad=grad)


def _groups(k, g):
    return math.ceil(k / g)


class MPQLinearBase(nn.Module):
    """Allocates the packed weight and the quantisation buffers for GPTQ-style (`use_gba_quant=False`)
    or GBA-style (double-quantised statistics) checkpoints.  Sub-classes implement
    check_parameters / prepare_params / forward."""

    def __init__(self, in_channels: int, out_channels: int, a_bit: int = 16, w_bit: int = 4, dtype=torch.half,
                 group_size=-1, use_gba_quant=True, dq_group_size=-1, dq_mode=2, disable_bias=True, asym=False,
                 requires_grad=False) -> None:
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.a_bit, self.w_bit, self.dtype = a_bit, w_bit, dtype
        self.maxq = 2 ** w_bit - 1
        self.group_size = group_size if group_size > -1 else in_channels
        self.asym, self.disable_bias = asym, disable_bias
        self.use_gba_quant, self.dq_group_size, self.dq_mode = use_gba_quant, dq_group_size, dq_mode
        self.requires_grad = False  # inference build (see package docstring)
        self.privileged_grad = None
        self.initialize()

    # ------------------------------------------------------------------ buffers
    def initialize(self) -> None:
        K, N, w = self.in_channels, self.out_channels, self.w_bit
        self.qweight = MPQWeightParameter(torch.empty((K // 32 * w, N), dtype=torch.int32), requires_grad=False,
                                          w_bit=w, asym=self.asym, group_size=self.group_size)
        self.register_buffer("g_idx", torch.arange(K, dtype=torch.int32) // self.group_size)
        self.register_buffer("bias", torch.zeros(N, dtype=self.dtype))
        self.register_buffer("wf", torch.arange(0, 32, w, dtype=torch.int32).unsqueeze(0))
        (self.init_gba if self.use_gba_quant else self.init_gptq)()

    def init_gptq(self) -> None:
        G, N = _groups(self.in_channels, self.group_size), self.out_channels
        self.register_buffer("qzeros", torch.zeros((G, N // 32 * self.w_bit), dtype=torch.int32))
        self.register_buffer("scales", torch.ones((G, N), dtype=self.dtype))
        self.asym = True

    def init_gba(self) -> None:
        if self.dq_group_size == -1:
            self.dq_group_size = self.out_channels
        G, N, dq = _groups(self.in_channels, self.group_size), self.out_channels, self.dq_group_size
        per_block = (G, math.ceil(N / dq), 1)
        per_value = (G, math.ceil(N / dq), dq)
        if self.asym:
            self.register_buffer("qzeros", torch.zeros((G, N // 32 * self.w_bit), dtype=torch.int32))
            shape = per_value if self.w_bit == 4 else (G, N)
            self.register_buffer("qscales", torch.ones(shape, dtype=torch.uint8))
        else:
            self.register_buffer("qstatistic", torch.ones(per_value, dtype=torch.uint8))
            self.register_buffer("qzeros_zeros", torch.zeros(per_block, dtype=self.dtype))
            self.register_buffer("qzeros_scales", torch.ones(per_block, dtype=self.dtype))
        dq_shape = (1, N, 1) if self.dq_mode == 1 else per_block
        self.register_buffer("qscales_zeros", torch.zeros(dq_shape, dtype=self.dtype))
        self.register_buffer("qscales_scales", torch.ones(dq_shape, dtype=self.dtype))
        self.register_buffer("scales", torch.ones((G, N), dtype=self.dtype))
        self.register_buffer("zeros", torch.zeros((G, N), dtype=self.dtype))

    # ------------------------------------------------------------------ protocol
    def set_qweight_data(self, data: torch.Tensor) -> None:
        self.qweight.data = data

    def generate_quantized_weight(self, qweight_only: bool = False) -> None:
        raise NotImplementedError("this method has not been implemented.")

    def check_parameters(self) -> None:
        raise NotImplementedError("Subclasses should implement this method.")

    def prepare_params(self) -> None:
        raise NotImplementedError("Subclasses should implement this method.")


class nBitLinearParameter(nn.Parameter):
    def __new__(cls, data: torch.Tensor = None, requires_grad: bool = False):
        if data is not None and not data.is_floating_point():
            requires_grad = False
        return super().__new__(cls, data, requires_grad=requires_grad)

    @staticmethod
    def update(qweight: torch.nn.Parameter, exp_avg_s: torch.Tensor = None, exp_avg_l: torch.Tensor = None, step: torch.Tensor = None,
               lr: float = 1e-4, weight_decay: float = 0.0, beta1: float = 0.99, beta2: float = 0.9999, eps: float = 1e-6, dtype=torch.half,
               correct_bias=None, projector=None, grad: torch.Tensor = None) -> None:
        """The optimiser-side update of this parameter kind (reference layers/qlinear/nbit/layer.py:122-170): utils/model_helper.qweight_update_fn on `qweight.grad`."""
        from bitorch_engine.utils.model_helper import qweight_update_fn
        assert isinstance(qweight, nBitLinearParameter), 'Error: the type of qweight must be nBitLinearParameter. '
        qweight_update_fn(qweight=qweight, exp_avg_s=exp_avg_s, exp_avg_l=exp_avg_l, step=step, lr=lr, weight_decay=weight_decay, beta1=beta1,
                          beta2=beta2, correct_bias=correct_bias, eps=eps, dtype=dtype, projector=projector, grad=grad)


class nBitLinearBase(nn.Module):
    """Base of the W-n / A-n integer layers (W4A4, W8A8); mirror of reference nbit/layer.py:122-280."""

    def __init__(self, in_channels: int, out_channels: int, a_bit: int = 4, w_bit: int = 4, device=None, dtype=torch.float) -> None:
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.device, self.dtype, self.a_bit, self.w_bit = device, dtype, a_bit, w_bit
        self.qweight = None
        self.reset_parameters()

    def reset_parameters(self) -> None:
        w = torch.empty((self.out_channels, self.in_channels))
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight = nn.Parameter(w)

    def set_weight_data(self, x: torch.Tensor) -> None:
        self.weight = nn.Parameter(x, requires_grad=False)

    def prepare_params(self) -> None:
        raise NotImplementedError("Subclasses should implement this method.")

    def set_quantized_weight_data(self, x: torch.Tensor) -> None:
        self.qweight = nn.Parameter(x, requires_grad=False)

    def generate_quantized_weight(self, qweight_only: bool = False) -> None:
        raise NotImplementedError("Subclasses should implement this method.")

    def _check_forward(self, x: torch.Tensor) -> None:
        raise NotImplementedError("Subclasses should implement this method.")

    @property
    def opt_weight(self):
        if not self.training and self.qweight is None:
            self.generate_quantized_weight()
        return self.weight if self.training else self.qweight
