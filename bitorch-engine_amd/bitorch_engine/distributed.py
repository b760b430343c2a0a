"""Output-column sharding of a prepared MPQ layer across GPUs (one process per GPU, torch.distributed;
backend "nccl" == RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference has no multi-device path at all (no collective call sites); this is the
one place the hot path shards: y[:, n] depends only on column n of qweight / scales / zeros, so rank r keeps
the column block [r*N/W, (r+1)*N/W) of every weight-side tensor (x, g_idx replicated), computes its slice
with the same kernels, and ONE all-gather assembles y.  Shard boundaries are multiples of 128 columns
(covers the 32/w_bit columns per packed qzeros word and the kernels' 64/256-column tiles).
"""
from typing import Callable, Optional

import torch
import torch.distributed as dist

SHARD_MULTIPLE = 128


def column_range(N: int, rank: int, world: int, multiple: int = SHARD_MULTIPLE):
    """[lo, hi) of rank's column block: blocks of `multiple` columns dealt as evenly as possible, in order."""
    if N % multiple:
        raise ValueError(f"out_channels={N} must be a multiple of {multiple} to be column-sharded")
    blocks = N // multiple
    base, extra = divmod(blocks, world)
    lo_b = rank * base + min(rank, extra)
    hi_b = lo_b + base + (1 if rank < extra else 0)
    return lo_b * multiple, hi_b * multiple


class ColumnShardedMPQLinear(torch.nn.Module):
    """Wraps the tensors of a prepared (prepare_params() done) MPQ layer; holds only this rank's columns."""

    def __init__(self, qweight, scales, zeros, g_idx, bias, w_bit: int, group_size: int, asym: bool, rank: int, world: int,
                 group=None, forward_impl: Optional[Callable] = None):
        super().__init__()
        N = qweight.shape[1]
        self.N, self.rank, self.world, self.group = N, rank, world, group
        self.w_bit, self.group_size, self.asym = w_bit, group_size, asym
        self.ranges = [column_range(N, r, world) for r in range(world)]
        lo, hi = self.ranges[rank]
        self.lo, self.hi = lo, hi
        per_word = 32 // w_bit
        self.register_buffer("qweight", qweight[:, lo:hi].contiguous())
        self.register_buffer("scales", scales[:, lo:hi].contiguous())
        if asym:  # packed along N: 32/w_bit columns per int32 word
            self.register_buffer("zeros", zeros[:, lo // per_word:hi // per_word].contiguous())
        else:
            self.register_buffer("zeros", zeros[:, lo:hi].contiguous())
        self.register_buffer("g_idx", g_idx)
        self.register_buffer("bias", None if bias is None else bias[lo:hi].contiguous())
        self._impl = forward_impl

    @classmethod
    def from_layer(cls, layer, rank: int, world: int, group=None, forward_impl=None):
        bias = None if getattr(layer, "disable_bias", True) else layer.bias
        return cls(layer.qweight.data, layer.scales, layer.zeros, layer.g_idx, bias, layer.w_bit, layer.group_size,
                   layer.asym, rank, world, group, forward_impl)

    def local_forward(self, x2: torch.Tensor) -> torch.Tensor:
        if self._impl is not None:
            return self._impl(x2, self.qweight, self.scales, self.zeros, self.g_idx, self.w_bit, self.asym, self.group_size, self.bias)
        from bitorch_engine.extensions import q_linear_cuda
        return q_linear_cuda.mpq_forward_impl(x2, self.qweight, self.scales, self.zeros, self.g_idx, self.w_bit, self.asym,
                                              self.group_size, self.bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        lead = list(x.shape[:-1])
        x2 = x.reshape(-1, x.shape[-1])
        y_local = self.local_forward(x2)  # [M, hi - lo]
        M = x2.shape[0]
        widths = [hi - lo for lo, hi in self.ranges]
        if len(set(widths)) == 1:
            # equal blocks: one all-gather into [W, M, N/W], then interleave the column blocks
            gathered = torch.empty((self.world * M, widths[0]), dtype=y_local.dtype, device=y_local.device)
            dist.all_gather_into_tensor(gathered, y_local.contiguous(), group=self.group)  # rank-major concatenation
            y = gathered.view(self.world, M, widths[0]).permute(1, 0, 2).reshape(M, self.N)
        else:
            parts = [torch.empty((M, w), dtype=y_local.dtype, device=y_local.device) for w in widths]
            dist.all_gather(parts, y_local.contiguous(), group=self.group)
            y = torch.cat(parts, dim=1)
        return y.view(lead + [self.N])
