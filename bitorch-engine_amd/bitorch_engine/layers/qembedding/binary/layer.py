"""Binary (1-bit, sign-packed) embedding tables on MI355X -- API mirror of reference layers/qembedding/binary/layer.py
(BinaryEmbeddingParameter :16-83, BinaryEmbeddingForward :87-188, BinaryEmbeddingCuda :191-340, BinaryEmbeddingBag :343-556).

The table is stored as uint8 rows of sign bits (LSB first: bit i of byte j = (w[8j + i] >= 0), the layout of
functions.cuda.tensor_to_packed_uint8) plus one fp scale per row (mean |w| of the centred row).  A lookup gathers the packed
rows and their scales and expands them on the GPU with the HIP unpack kernel (bie_unpack_u8_scaled): 1/32 of the bytes of an
fp32 table cross HBM.  Training-time `update` (the reference's qweight_update_fn, utils/model_helper.py:330-482) belongs to the
optimiser subsystem, which is out of this build's scope: it raises."""
import math
import typing
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn
from torch.autograd import Function
from torch.nn.parameter import Parameter

from bitorch_engine.functions.cuda import tensor_to_packed_uint8, unpack_uint8_tensor
from bitorch_engine.utils.model_helper import pad_embedding_dim


class BinaryEmbeddingParameter(Parameter):
    """Packed binary embedding table; `active_indices` records the token ids of the last batch (sparse update hint)."""

    def __new__(cls, data: torch.Tensor = None, requires_grad: bool = True, active_indices: torch.Tensor = None):
        cls.active_indices = active_indices
        if data is not None and not (data.is_floating_point() or data.is_complex()):
            requires_grad = False  # stock torch refuses gradients on integer tensors (the reference runs a patched torch)
        return super().__new__(cls, data=data, requires_grad=requires_grad)

    @staticmethod
    def update(qweight: torch.nn.Parameter, exp_avg_s: torch.Tensor = None, exp_avg_l: torch.Tensor = None, step: torch.Tensor = None,
               lr: float = 1e-4, weight_decay: float = 0.0, beta1: float = 0.99, beta2: float = 0.9999, eps: float = 1e-6, dtype=torch.half,
               correct_bias=None, projector=None, grad: torch.Tensor = None) -> None:
        """The optimiser-side update of this parameter kind (reference layers/qembedding/binary/layer.py): utils/model_helper.qweight_update_fn on `qweight.grad`."""
        from bitorch_engine.utils.model_helper import qweight_update_fn
        assert isinstance(qweight, BinaryEmbeddingParameter), 'Error: the type of qweight must be BinaryEmbeddingParameter. '
        qweight_update_fn(qweight=qweight, exp_avg_s=exp_avg_s, exp_avg_l=exp_avg_l, step=step, lr=lr, weight_decay=weight_decay, beta1=beta1,
                          beta2=beta2, correct_bias=correct_bias, eps=eps, dtype=dtype, projector=projector, grad=grad)


class BinaryEmbeddingForward(Function):
    """indices [B, S] -> [B, S, D] = sign bits of the selected rows * their row scale."""

    @staticmethod
    def forward(ctx, input, qweight, embed_scale, ori_embedding_dim: int, is_train: bool):
        flat = input.reshape(-1)
        rows = qweight.index_select(0, flat).view(input.size(0), input.size(-1), -1)
        scl = embed_scale.index_select(0, flat).view(input.size(0), input.size(-1), -1)
        if is_train:
            ctx.save_for_backward(input, qweight, scl)
        out = unpack_uint8_tensor(rows, scl).to(embed_scale.dtype)
        return out[:, :, :ori_embedding_dim] if out.size(-1) > ori_embedding_dim else out

    @staticmethod
    @typing.no_type_check
    def backward(ctx, output_gradient):
        input, qweight, scl = ctx.saved_tensors
        width = qweight.size(1) * 8
        if output_gradient.size(-1) < width:
            output_gradient = F.pad(output_gradient, [0, width - output_gradient.size(-1)], mode="constant", value=-1)
        dense = torch.zeros((qweight.size(0), width), dtype=output_gradient.dtype, device=output_gradient.device)
        dense.index_add_(0, input.reshape(-1), (output_gradient * scl.expand_as(output_gradient)).reshape(-1, width))
        qweight.active_indices = input
        return None, tensor_to_packed_uint8(dense), None, None, None


class BinaryEmbeddingCuda(nn.Module):
    """nn.Embedding with a sign-packed table.  Buffers / parameters as in the reference: `weight` (fp, dropped by
    prepare_params), `qweight` uint8 [num_embeddings, ceil(pad8(dim) / 8)], `scale_w` [num_embeddings, 1]."""

    def __init__(self, *args: int, num_embeddings: int, embedding_dim: int, padding_idx: Optional[int] = None,
                 dtype: torch.dtype = torch.float, **kwargs: int) -> None:
        super().__init__()
        self.num_embeddings = num_embeddings
        self.embedding_dim = embedding_dim
        if padding_idx is not None:
            if padding_idx > 0:
                assert padding_idx < num_embeddings, "Padding_idx must be within num_embeddings"
            elif padding_idx < 0:
                assert padding_idx >= -num_embeddings, "Padding_idx must be within num_embeddings"
                padding_idx = num_embeddings + padding_idx
        self.padding_idx = padding_idx
        self.dtype = dtype
        self.init_weight()

    def init_weight(self) -> None:
        self.weight = Parameter(torch.empty((self.num_embeddings, self.embedding_dim)), requires_grad=False)
        nn.init.normal_(self.weight)
        packed_cols = math.ceil(pad_embedding_dim(self.weight).shape[1] / 8)
        self.qweight = BinaryEmbeddingParameter(torch.zeros((self.num_embeddings, packed_cols), dtype=torch.uint8))
        self.register_buffer("scale_w", torch.zeros((self.num_embeddings, 1), dtype=self.dtype))

    def prepare_params(self) -> None:
        """Centre, sign-pack (HIP kernel) and scale the table once; a table that already holds packed data (loaded from a
        checkpoint) is left alone.  Needs the layer on the GPU when it has to pack."""
        if torch.all(self.qweight == 0) or torch.all(self.scale_w == 0):
            centred = self.weight - self.weight.mean()
            self.reset_parameters(centred)
            centred = pad_embedding_dim(centred)
            packed = tensor_to_packed_uint8(centred)
            assert packed.dtype == torch.uint8 and packed.nelement() * 8 == centred.nelement()
            self.scale_w = centred.float().norm(1, 1, keepdim=True).div(centred.shape[1]).to(self.dtype)
            self.qweight = BinaryEmbeddingParameter(packed, requires_grad=False)
        del self.weight

    def reset_parameters(self, weight: torch.Tensor) -> None:
        if self.padding_idx is not None:
            with torch.no_grad():
                weight[self.padding_idx].fill_(0)

    def forward(self, input: Tensor) -> Tensor:
        return BinaryEmbeddingForward.apply(input, self.qweight, self.scale_w, self.embedding_dim, self.training)


class BinaryEmbeddingBagForward(Function):
    """Majority vote over each bag of boolean rows: +1 where at least half of the bag's rows are set, else -1."""

    @staticmethod
    def forward(ctx, input, weight, is_train: bool):
        if is_train:
            ctx.save_for_backward(input, weight)
        bag = weight.index_select(0, input.reshape(-1)).view(input.shape[0], -1, weight.size(1))
        votes = torch.count_nonzero(bag, dim=1)
        hit = votes.ge(math.ceil(input.size(1) / 2))
        return torch.where(hit, torch.ones((), dtype=input.dtype, device=input.device), -torch.ones((), dtype=input.dtype, device=input.device))

    @staticmethod
    @typing.no_type_check
    def backward(ctx, output_gradient):
        input, qweight = ctx.saved_tensors
        acc = torch.zeros_like(qweight, dtype=output_gradient.dtype)
        acc.index_add_(0, input.reshape(-1), output_gradient.repeat(input.shape[1], 1))
        qweight.active_indices = input
        return None, acc >= 0, None


class BinaryEmbeddingBag(nn.Module):
    """Bag-of-words over a boolean table [num_embeddings, embedding_dim] (reference :497-556)."""

    def __init__(self, *args: int, num_embeddings: int, embedding_dim: int, padding_idx: Optional[int] = None, **kwargs: int) -> None:
        super().__init__()
        self.num_embeddings = num_embeddings
        if padding_idx is not None:
            if padding_idx > 0:
                assert padding_idx < num_embeddings, "Padding_idx must be within num_embeddings"
            elif padding_idx < 0:
                assert padding_idx >= -num_embeddings, "Padding_idx must be within num_embeddings"
                padding_idx = num_embeddings + padding_idx
        self.padding_idx = padding_idx
        self.weight = BinaryEmbeddingParameter(torch.rand((num_embeddings, embedding_dim)) > 0.5, requires_grad=True)
        self.reset_parameters()

    def reset_parameters(self) -> None:
        if self.padding_idx is not None:
            with torch.no_grad():
                self.weight[self.padding_idx].fill_(False)

    def forward(self, input: Tensor) -> Tensor:
        return BinaryEmbeddingBagForward.apply(input, self.weight, self.training)
