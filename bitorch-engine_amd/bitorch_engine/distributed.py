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

    def local_forward(self, x2: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """This rank's columns, [M, hi - lo]; written in place into `out` (contiguous) when given."""
        if self._impl is not None:
            y = self._impl(x2, self.qweight, self.scales, self.zeros, self.g_idx, self.w_bit, self.asym, self.group_size, self.bias)
            return y if out is None else out.copy_(y)
        from bitorch_engine.extensions import q_linear_cuda
        return q_linear_cuda.mpq_forward_impl(x2, self.qweight, self.scales, self.zeros, self.g_idx, self.w_bit, self.asym,
                                              self.group_size, self.bias, out=out)

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

    # ------------------------------------------------------------------------------------------------------------------
    # M-tiled schedule: the all-gather of tile t runs on a side stream while the GEMM of tile t+1 computes
    # ------------------------------------------------------------------------------------------------------------------
    def forward_overlapped(self, x: torch.Tensor, m_tile: int = 1024, interleave: bool = True) -> torch.Tensor:
        """Same result as forward().  x is cut into row tiles of `m_tile`; tile t's local GEMM is followed by its all-gather
        on a communication stream, so the exchange of tile t (N/W columns x m_tile rows to every peer over xGMI) hides
        behind the GEMM of tile t+1 -- SURVEY.md section 8e budgets 192 us of exchange against ~192 us of compute for
        configs[4], i.e. un-overlapped the step is twice as long.  Every rank issues the collectives in the same (tile)
        order.  On the CPU (gloo tests) the schedule is identical, just without streams.
        interleave=False returns the collective's natural rank-major layout [W, M, N/W] (what a following row-parallel
        layer consumes directly) and skips the [M, N] interleave copy."""
        lead = list(x.shape[:-1])
        x2 = x.reshape(-1, x.shape[-1])
        M = x2.shape[0]
        widths = [hi - lo for lo, hi in self.ranges]
        if len(set(widths)) != 1:
            if not interleave:
                raise RuntimeError("forward_overlapped(interleave=False): the rank-major [W, M, N/W] layout needs equal shard widths, got "
                                   f"{widths}")
            return self.forward(x)
        if M <= m_tile:  # one tile: nothing to overlap; still honour the requested layout
            y = self.forward(x)
            if interleave:
                return y
            return y.reshape(M, self.world, widths[0]).permute(1, 0, 2).contiguous()
        wdt = widths[0]
        on_gpu = x2.is_cuda
        tiles = [(m0, min(m0 + m_tile, M)) for m0 in range(0, M, m_tile)]
        out_dtype = x2.dtype
        gathered = torch.empty((self.world, M, wdt), dtype=out_dtype, device=x2.device)  # rank-major
        comm = compute = None
        if on_gpu:
            compute = torch.cuda.current_stream(x2.device)
            comm = _comm_stream(x2.device)
            comm.wait_stream(compute)  # `gathered` exists before the first collective writes it
        stage = []
        for (m0, m1) in tiles:
            y_t = self.local_forward(x2[m0:m1]).contiguous()  # [mt, N/W] on the compute stream
            recv = torch.empty((self.world, m1 - m0, wdt), dtype=out_dtype, device=x2.device)
            if on_gpu:
                ev = torch.cuda.Event()
                ev.record(compute)
                with torch.cuda.stream(comm):
                    comm.wait_event(ev)
                    dist.all_gather_into_tensor(recv.view(self.world * (m1 - m0), wdt), y_t, group=self.group)
                    gathered[:, m0:m1].copy_(recv)
                y_t.record_stream(comm)
                recv.record_stream(comm)
            else:
                dist.all_gather_into_tensor(recv.view(self.world * (m1 - m0), wdt), y_t, group=self.group)
                gathered[:, m0:m1].copy_(recv)
            stage.append((y_t, recv))
        if on_gpu:
            compute.wait_stream(comm)
        if not interleave:
            return gathered
        y = gathered.permute(1, 0, 2).reshape(M, self.N)
        return y.view(lead + [self.N])


    # ------------------------------------------------------------------------------------------------------------------
    # Direct exchange: every rank SENDS its tile to each peer and RECEIVES each peer's tile where it finally lives
    # ------------------------------------------------------------------------------------------------------------------
    def forward_direct(self, x: torch.Tensor, m_tile: int = 1024, interleave: bool = True) -> torch.Tensor:
        """Same result as forward() / forward_overlapped(), without the passes those spend on re-arranging:

        * xGMI is point-to-point (7 links per GPU): a grouped send/recv to and from every peer puts one peer's block on each link at
          the same time, where a ring all-gather moves all W - 1 blocks over one link per step (192 us against
          1.34 ms for configs[4]).  One batch_isend_irecv per row tile (ncclGroupStart/End under the RCCL backend; plain isend/irecv
          under gloo), on the communication stream, overlapped with the next tile's GEMM.
        * interleave=False: the destination is the rank-major [W, M, N/W] buffer itself.  This rank's GEMM writes its tile IN PLACE
          (`out=`), every receive lands in the peer's slice `gathered[p, m0:m1]` (contiguous): zero copies.
        * interleave=True: the destination is y [M, N].  A block's rows are strided there (row pitch N), which neither RCCL nor gloo
          can receive into, so remote blocks land in a per-tile staging buffer and are copied ONCE into their column range (one pass
          over the data instead of the two of forward_overlapped: stage -> rank-major -> permute/reshape); shard widths may differ.
          Writing y's strided blocks straight from the GEMM epilogue needs IPC-mapped peer buffers and an output pitch in the C-ABI;
          not built (no multi-GPU hardware was available to validate it).
        Every rank posts the same operations in the same order (peers ascending, tile by tile)."""
        lead = list(x.shape[:-1])
        x2 = x.reshape(-1, x.shape[-1])
        M = x2.shape[0]
        widths = [hi - lo for lo, hi in self.ranges]
        if not interleave and len(set(widths)) != 1:
            raise RuntimeError(f"forward_direct(interleave=False): the rank-major [W, M, N/W] layout needs equal shard widths, got {widths}")
        dev, dt = x2.device, x2.dtype
        on_gpu = x2.is_cuda
        peers = [p for p in range(self.world) if p != self.rank]
        tiles = [(m0, min(m0 + max(1, m_tile), M)) for m0 in range(0, M, max(1, m_tile))] if M else []
        if interleave:
            y = torch.empty((M, self.N), dtype=dt, device=dev)
        else:
            gathered = torch.empty((self.world, M, widths[0]), dtype=dt, device=dev)
        comm = compute = None
        if on_gpu:
            compute = torch.cuda.current_stream(dev)
            comm = _comm_stream(dev)
            comm.wait_stream(compute)  # the destination exists before the first receive writes it
        keep = []
        for (m0, m1) in tiles:
            if interleave:
                mine = self.local_forward(x2[m0:m1])  # [mt, my width], contiguous
                stage = {p: torch.empty((m1 - m0, widths[p]), dtype=dt, device=dev) for p in peers}
                dst = stage
            else:
                mine = self.local_forward(x2[m0:m1], out=gathered[self.rank, m0:m1])
                dst = {p: gathered[p, m0:m1] for p in peers}
            ops = []
            for p in peers:
                ops.append(dist.P2POp(dist.isend, mine, _global_rank(self.group, p), group=self.group))
                ops.append(dist.P2POp(dist.irecv, dst[p], _global_rank(self.group, p), group=self.group))

            def exchange():
                if ops:
                    for req in dist.batch_isend_irecv(ops):
                        req.wait()
                if interleave:
                    y[m0:m1, self.lo:self.hi].copy_(mine)
                    for p in peers:
                        y[m0:m1, self.ranges[p][0]:self.ranges[p][1]].copy_(stage[p])

            if on_gpu:
                ev = torch.cuda.Event()
                ev.record(compute)
                with torch.cuda.stream(comm):
                    comm.wait_event(ev)
                    exchange()
                mine.record_stream(comm)
                for t in dst.values():
                    t.record_stream(comm)
            else:
                exchange()
            keep.append((mine, dst))
        if on_gpu:
            compute.wait_stream(comm)
        if not interleave:
            return gathered
        return y.view(lead + [self.N])


def _global_rank(group, group_rank: int) -> int:
    """P2POp addresses peers by GLOBAL rank; the layer's ranks are ranks of its (sub)group."""
    if group is None:
        return group_rank
    return dist.get_global_rank(group, group_rank)


_COMM_STREAMS = {}


def _comm_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    s = _COMM_STREAMS.get(key)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _COMM_STREAMS[key] = s
    return s
