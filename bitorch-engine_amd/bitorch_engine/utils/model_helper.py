"""Helpers used inside the hot path, mirroring reference utils/model_helper.py: flatten_x / unflatten_x
(:10-51), pad_last_2_dims_to_multiple_of_128 (:85-117), binary_matmul_forward_post_processing
(:120-155), prepare_bie_layers (:158-196), init_weight (:286-327)."""
from typing import Tuple, Type

import torch
import torch.nn.functional as F


def flatten_x(x: torch.Tensor):
    """[..., K] -> ([M, K], leading shape)."""
    lead = list(x.shape[:-1])
    return x.reshape(-1, x.shape[-1]), lead


def unflatten_x(x: torch.Tensor, shape: list):
    return x.view(shape + [x.shape[-1]])


def pad_last_2_dims_to_multiple_of_128(tensor: torch.Tensor) -> torch.Tensor:
    """Pad the last two dims up to multiples of 128 with -1 (== binary 0) on the right / bottom."""
    h, w = tensor.shape[-2], tensor.shape[-1]
    ph, pw = (-h) % 128, (-w) % 128
    if ph == 0 and pw == 0:
        return tensor
    return F.pad(tensor, (0, pw, 0, ph), value=-1)


def binary_matmul_forward_post_processing(tensor, padded_hidden, scale, orig_m, orig_n) -> torch.Tensor:
    """(padded K - 2*popc)-style result -> crop to the original [.., orig_m, orig_n] and scale."""
    out = tensor[..., :orig_m, :orig_n]
    return out * scale


def prepare_bie_layers(model: torch.nn.Module, layers=None) -> None:
    """Call prepare_params() on every BIE layer of `model` (decode double-quantised statistics, build
    band tables, ...).  `layers` optionally restricts the layer classes."""
    from bitorch_engine.layers.qlinear.nbit import MPQLinearBase, nBitLinearBase
    from bitorch_engine.layers.qconv.nbit import nBitConv2dBase
    from bitorch_engine.layers.qlinear.binary import BinaryLinearBase
    from bitorch_engine.layers.qconv.binary import BinaryConv2dBase
    kinds = tuple(layers) if layers else (MPQLinearBase, nBitLinearBase, nBitConv2dBase, BinaryLinearBase, BinaryConv2dBase)
    for module in model.modules():
        if isinstance(module, kinds):
            module.prepare_params()


def init_weight(weight: torch.Tensor, cls: Type[torch.nn.Parameter] = torch.nn.Parameter) -> Tuple[torch.Tensor, torch.Tensor]:
    """Binary weight initialisation (reference :286-327): scale_w = mean|w|; the centred weight is
    quantised to int8 sign carriers in [-127, 127] (amax scaling), exact zeros take sign(w)."""
    w = weight.detach().to(torch.float32)
    scale_w = w.abs().sum().div(w.nelement())
    w = w - w.mean()
    amax = w.abs().max()
    q = torch.clamp((w * (127.0 / amax)).round(), -127, 127) if float(amax) > 2.0 ** -24 else torch.zeros_like(w)
    q = torch.where(q == 0, w.sign(), q)
    return cls(q.to(torch.int8), requires_grad=False), scale_w
