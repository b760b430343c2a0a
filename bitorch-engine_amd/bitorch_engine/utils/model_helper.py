"""Helpers either side of the hot path, mirroring reference utils/model_helper.py: flatten_x / unflatten_x (:10-51),
pad_embedding_dim (:54-82), pad_last_2_dims_to_multiple_of_128 (:85-117), binary_matmul_forward_post_processing (:120-155),
prepare_bie_layers (:158-196), pack_bie_layers / save_checkpoint / load_checkpoint (:199-283), init_weight (:286-327)."""
from typing import Tuple, Type

import torch
import torch.nn.functional as F


def flatten_x(x: torch.Tensor):
    """[..., K] -> ([M, K], leading shape)."""
    lead = list(x.shape[:-1])
    return x.reshape(-1, x.shape[-1]), lead


def unflatten_x(x: torch.Tensor, shape: list):
    return x.view(shape + [x.shape[-1]])


def pad_embedding_dim(weight: torch.Tensor) -> torch.Tensor:
    """Pad the embedding dimension (dim 1) up to a multiple of 8 with -1 (a cleared bit once sign-packed) --
    reference utils/model_helper.py:54-82."""
    extra = (-weight.shape[1]) % 8
    if extra == 0:
        return weight
    fill = torch.full((weight.shape[0], extra), -1.0, dtype=torch.float, device=weight.device)
    return torch.cat([weight, fill], dim=1)


def pad_last_2_dims_to_multiple_of_128(tensor: torch.Tensor):
    """(padded tensor, rows added to dim -2): the last two dims rounded up to multiples of 128, padded with ZEROS on the
    right / bottom -- reference utils/model_helper.py:85-117 (it returns the pair although it is annotated -> Tensor)."""
    rows, cols = tensor.shape[-2], tensor.shape[-1]
    add_rows, add_cols = (-rows) % 128, (-cols) % 128
    if add_rows or add_cols:
        tensor = F.pad(tensor, (0, add_cols, 0, add_rows), mode="constant", value=0)
    return tensor, add_rows


def binary_matmul_forward_post_processing(tensor: torch.Tensor, shape_pre: list, x_pad_sec_last: int, y_pad_sec_last: int,
                                          k: int) -> torch.Tensor:
    """Undo the padding of a batched XOR-popcount product [B, m_pad, n_pad]: drop the padded rows / columns, restore the
    leading dims `shape_pre`, and map the popcount to the +-1 dot product k - 2*popc -- reference :120-155."""
    m_keep = tensor.shape[-2] - x_pad_sec_last
    n_keep = tensor.shape[-1] - y_pad_sec_last
    tensor = tensor[:, :m_keep, :n_keep]
    return k - 2 * tensor.reshape(list(shape_pre) + [m_keep, n_keep])


def _bie_layer_kinds(with_embedding: bool):
    from bitorch_engine.layers.qlinear.nbit import MPQLinearBase, nBitLinearBase
    from bitorch_engine.layers.qconv.nbit import nBitConv2dBase
    from bitorch_engine.layers.qlinear.binary import BinaryLinearBase
    from bitorch_engine.layers.qconv.binary import BinaryConv2dBase
    kinds = [BinaryConv2dBase, nBitConv2dBase, BinaryLinearBase, nBitLinearBase, MPQLinearBase]
    if with_embedding:
        from bitorch_engine.layers.qembedding.binary import BinaryEmbeddingCuda
        kinds.append(BinaryEmbeddingCuda)
    return tuple(kinds)


def prepare_bie_layers(model: torch.nn.Module, layers=None) -> None:
    """Call prepare_params() on every BIE layer below `model` (decode double-quantised statistics, build band tables, ...);
    `layers` optionally restricts the layer classes -- reference :158-196."""
    kinds = tuple(layers) if layers else _bie_layer_kinds(True)
    for i, module in enumerate(model.modules()):
        if i > 0 and isinstance(module, kinds):
            module.prepare_params()


def pack_bie_layers(model: torch.nn.Module, qweight_only: bool = True, layers=None) -> None:
    """generate_quantized_weight(qweight_only) on every BIE layer below `model`: bit-pack the weights before torch.save()
    (qweight_only drops the unpacked training weights) -- reference :199-233."""
    kinds = tuple(layers) if layers else _bie_layer_kinds(False)
    for i, module in enumerate(model.modules()):
        if i > 0 and isinstance(module, kinds):
            module.generate_quantized_weight(qweight_only=qweight_only)


def save_checkpoint(model: torch.nn.Module, name: str, qweight_only: bool = True) -> None:
    """Pack the quantised layers, then torch.save({'state_dict': ...}) -- same file layout as reference :236-262."""
    pack_bie_layers(model, qweight_only)
    torch.save({"state_dict": model.state_dict()}, name)


def load_checkpoint(model: torch.nn.Module, checkpoint_path: str, qweight_only: bool = True) -> None:
    """Pack the model's layers (so the packed buffers exist with the right shapes), then load a save_checkpoint() file
    non-strictly -- reference :265-283."""
    pack_bie_layers(model, qweight_only)
    state = torch.load(checkpoint_path, map_location="cpu")
    model.load_state_dict(state["state_dict"], strict=False)


def init_weight(weight: torch.Tensor, cls: Type[torch.nn.Parameter] = torch.nn.Parameter) -> Tuple[torch.Tensor, torch.Tensor]:
    """Binary weight initialisation (reference :286-327): scale_w = mean|w|; the centred weight goes through nv_tensor_quant
    with its default amax -- the SIGNED maximum of the centred weight, torch.amax, quant_operators.py:52 -- into int8 sign
    carriers, and exact zeros take sign(w)."""
    from bitorch_engine.utils.quant_operators import nv_tensor_quant
    w = weight.detach()
    if w.dtype != torch.float:
        w = w.to(torch.float)
    scale_w = w.norm(p=1).div(w.nelement()).to(weight.device)
    centred = w - w.mean()
    carriers = nv_tensor_quant(centred)[0]
    carriers = torch.where(carriers == 0, centred.sign(), carriers)
    return cls(carriers.to(torch.int8), requires_grad=False), scale_w
