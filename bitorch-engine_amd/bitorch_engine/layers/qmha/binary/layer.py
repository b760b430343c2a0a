"""Binary multi-head attention -- API mirror of reference layers/qmha/binary/layer.py (LearnableBias :8-46, BMHA :49-181).
The four projections are BinaryLinearCutlass layers (sign-packed weights, XNOR-popcount HIP kernel, bie_binary_linear_forward);
scores / softmax / context stay in the activation dtype, exactly the reference's composition."""
import math
from typing import Tuple

import torch
import torch.nn as nn

from bitorch_engine.layers.qlinear.binary.cutlass import BinaryLinearCutlass


class LearnableBias(nn.Module):
    def __init__(self, out_chn: int):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(out_chn), requires_grad=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x + self.bias.expand_as(x)


class BMHA(nn.Module):
    def __init__(self, input_dim: int, hidden_dim: int, num_heads: int, dtype=torch.float, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if hidden_dim % num_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)" % (hidden_dim, num_heads))
        self.dtype = dtype
        self.num_heads = num_heads
        self.head_dim = hidden_dim // num_heads
        self.hidden_dim = hidden_dim
        self.input_dim = input_dim
        self.q_linear = BinaryLinearCutlass(input_dim, hidden_dim, dtype=dtype)
        self.v_linear = BinaryLinearCutlass(input_dim, hidden_dim, dtype=dtype)
        self.k_linear = BinaryLinearCutlass(input_dim, hidden_dim, dtype=dtype)
        self.dropout = nn.Dropout(0.1)
        self.out = BinaryLinearCutlass(hidden_dim, input_dim, dtype=dtype)

    def _heads(self, t: torch.Tensor, bs: int) -> torch.Tensor:
        return t.view(bs, -1, self.num_heads, self.head_dim).transpose(1, 2)

    def forward(self, hidden_states: torch.Tensor, mask: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        bs = hidden_states.size(0)
        q = self._heads(self.q_linear(hidden_states), bs)
        k = self._heads(self.k_linear(hidden_states), bs)
        v = self._heads(self.v_linear(hidden_states), bs)
        scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(self.head_dim)
        if mask is not None:
            scores = scores.masked_fill(mask.unsqueeze(1) == 0, -1e9)
        scores = self.dropout(nn.functional.softmax(scores, dim=-1))
        ctx = torch.matmul(scores, v).transpose(1, 2).contiguous().view(bs, -1, self.hidden_dim)
        return self.out(ctx), scores
