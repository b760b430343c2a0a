"""BinaryLinearCutlass / BinaryMatMul: mirror of reference layers/qlinear/binary/cutlass/layer.py.  The
scale is folded into the kernel epilogue ((K - 2*popc) * scale_a * scale_w)."""
import torch

from bitorch_engine.utils.safe_import import import_extension
from bitorch_engine.utils.model_helper import flatten_x, unflatten_x, init_weight
from ..layer import BinaryLinearBase, BinaryLinearParameter

binary_linear_cutlass = import_extension("binary_linear_cutlass")


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
        x = self.set_activation(x)
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
        k = x.size(-1)
        pad = (-k) % 8  # pad K with -1 on x and +1 on y: each padded position contributes -1, corrected below
        if pad:
            x = torch.nn.functional.pad(x, (0, pad), value=-1.0)
            y = torch.nn.functional.pad(y, (0, pad), value=1.0)
        out = binary_linear_cutlass.matmul(x, y, 1.0) + float(pad)
        return out.to(x.dtype) * self.x_clip * self.y_clip
