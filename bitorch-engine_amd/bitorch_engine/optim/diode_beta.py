"""DiodeMix: the optimiser that drives the quantised parameter classes' `update()` (sign-descent for the 1-bit carriers, Adam on the integer values of
W4A4 / W8A8, unpack -> Adam -> re-pack for MPQ) and falls back to AdamW for ordinary float parameters.  API mirror of the reference's
optim/diode_beta.py:37-196 ("Diode", Guo et al. 2024): same constructor arguments and range checks, same per-parameter state keys (`step`,
`exp_avg_l`, `exp_avg_s`, `projector`), same GaLore group options (`rank`, `update_proj_gap`, `scale`, `proj_type`), so optimiser checkpoints
interchange.

Why it is here: round 6 gave the binary / W4A4 / W8A8 / conv layers their straight-through backward; this is the
caller on the other side of that path -- `loss.backward()` leaves the integer gradient on `weight.grad` (MPQ: `privileged_grad`), `step()` hands
it with the moments to `type(p).update`, whose branches are this library's (HIP unpack / pack kernels for MPQ, utils/model_helper.py).
Order of operations = the reference's, op for op (including the ONE `rand_like` drawn per parameter when its state is created, whatever the kind:
a seeded run consumes the generator exactly as the reference does); pinned to the reference's outputs, tests/golden/optim_diodemix.npz."""
import math
from typing import Callable, Iterable, Tuple

import torch
from packaging import version
from torch import nn
from torch.optim import Optimizer

from bitorch_engine.layers.qconv.binary import BinaryConvParameter
from bitorch_engine.layers.qconv.nbit import nBitConvParameter
from bitorch_engine.layers.qembedding.binary import BinaryEmbeddingParameter
from bitorch_engine.layers.qlinear.binary import BinaryLinearParameter
from bitorch_engine.layers.qlinear.nbit import MPQWeightParameter, nBitLinearParameter

from .galore_projector import GaLoreProjector

_SIGN_CARRIERS = (BinaryLinearParameter, BinaryConvParameter)
_QUANTISED = _SIGN_CARRIERS + (BinaryEmbeddingParameter, nBitLinearParameter, nBitConvParameter, MPQWeightParameter)


def check_pytorch_version(required_version):
    """Raise when torch is older than `required_version` (reference :20-34)."""
    if version.parse(torch.__version__) < version.parse(required_version):
        raise Exception(f"Current PyTorch version {torch.__version__} is below the required minimum version {required_version}.")


def _check_range(ok, what, value, rng):
    if not ok:
        raise ValueError(f"Invalid {what}: {value} - should be {rng}")


class DiodeMix(Optimizer):
    """params: iterable of parameters or groups; lr 1e-4; betas (0.99, 0.9999): first / second moment (sign carriers: long / short average);
    eps 1e-6; weight_decay 0.0 (decoupled); correct_bias True; dtype: the type the moments and the quantised updates are computed in."""

    def __init__(self, params: Iterable[nn.parameter.Parameter], lr: float = 1e-4, betas: Tuple[float, float] = (0.99, 0.9999), eps: float = 1e-6,
                 weight_decay: float = 0.0, correct_bias: bool = True, dtype: torch.dtype = torch.float):
        check_pytorch_version("1.5.0")
        _check_range(lr >= 0.0, "learning rate", lr, ">= 0.0")
        _check_range(0.0 <= betas[0] < 1.0, "beta parameter", betas[0], "in [0.0, 1.0)")
        _check_range(0.0 <= betas[1] < 1.0, "beta parameter", betas[1], "in [0.0, 1.0)")
        _check_range(0.0 <= eps, "epsilon value", eps, ">= 0.0")
        self.dtype = dtype
        super().__init__(params, {"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay, "correct_bias": correct_bias})

    def _new_moments(self, p, grad, state):
        """State of a parameter seen for the first time (reference :139-149).  Sign carriers start their short average a hair on the side
        AGAINST their current sign (|.| < 1e-3, random), so that the first flips need real gradient; the draw happens for every parameter."""
        delta = torch.rand_like(p, dtype=self.dtype).mul_(1e-3)
        if isinstance(p, BinaryEmbeddingParameter):
            state["exp_avg_s"] = -(p.data.clone().sign_().to(self.dtype).mul_(delta))
        elif isinstance(p, _SIGN_CARRIERS):
            state["exp_avg_l"] = torch.zeros_like(p, dtype=self.dtype)
            state["exp_avg_s"] = -(p.data.clone().sign_().to(self.dtype).mul_(delta))
        else:
            state["exp_avg_l"] = torch.zeros_like(grad, dtype=self.dtype)
            state["exp_avg_s"] = torch.zeros_like(grad, dtype=self.dtype)

    @staticmethod
    def _adamw(w, grad, state, group, projector):
        """Float parameters: AdamW with the decay applied after the step (reference :158-193); exp_avg_l / exp_avg_s are m / v."""
        beta1, beta2 = group["betas"]
        m, v, step = state["exp_avg_l"], state["exp_avg_s"], state["step"]
        step.add_(1)
        m.mul_(beta1).add_(grad, alpha=(1.0 - beta1))
        v.mul_(beta2).addcmul_(grad, grad, value=1.0 - beta2)
        denom = v.sqrt().add_(group["eps"])
        step_size = group["lr"]
        if group["correct_bias"]:
            step_size = step_size * math.sqrt(1.0 - beta2 ** step.item()) / (1.0 - beta1 ** step.item())
        direction = m / denom
        if projector is not None:
            direction = projector.project_back(direction)
        w.add_(direction, alpha=-step_size)
        if group["weight_decay"] > 0.0:
            w.add_(w, alpha=(-group["lr"] * group["weight_decay"]))

    @torch.no_grad()
    def step(self, closure: Callable = None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                grad = p.privileged_grad if isinstance(p, MPQWeightParameter) else p.grad  # MPQ: the float gradient its backward parked (mpq_layer.py)
                if grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                state = self.state[p]
                if "step" not in state:
                    state["step"] = torch.zeros(1)
                projector = None
                if "rank" in group:  # GaLore group: the moments live in the projected space
                    if "projector" not in state:
                        state["projector"] = GaLoreProjector(group["rank"], update_proj_gap=group["update_proj_gap"], scale=group["scale"],
                                                             proj_type=group["proj_type"])
                    projector = state["projector"]
                    grad = projector.project(grad.to(self.dtype), state["step"].item())
                if "exp_avg_s" not in state:
                    self._new_moments(p, grad, state)
                if isinstance(p, _QUANTISED):
                    beta1, beta2 = group["betas"]
                    type(p).update(qweight=p, exp_avg_s=state["exp_avg_s"], exp_avg_l=state.get("exp_avg_l"), step=state["step"], lr=group["lr"],
                                   weight_decay=group["weight_decay"], beta1=beta1, beta2=beta2, correct_bias=group["correct_bias"], eps=group["eps"],
                                   dtype=self.dtype, projector=projector, grad=grad)
                else:
                    self._adamw(p, grad, state, group, projector)
        return loss
