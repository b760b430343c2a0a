"""bench_sharded.py -- the multi-rank side of bench.py: the launcher, the world proof, the overlapped all-gather of the weak-scaling
headline step, the strong-scaling configs[4] leg and the CPU rehearsal (`--dry-run`).  Measurement code: it lives beside bench.py, not in
the product package (bitorch_engine.distributed holds the layer; this file only times it).

`python bench.py --gpus N` with WORLD_SIZE unset re-executes itself as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
(the command the driver uses itself for N > 1), one rank per GPU, backend nccl (= RCCL over xGMI); rank 0 prints the one JSON line with
`n_gpus: N` and an `rccl` object: the backend, how many ranks the collective really saw, the all-gather's bytes and microseconds alone and
hidden behind the next pass.  The reference has no collective call site anywhere; SURVEY.md section 8e is the spec.
"""
import json
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, argv) -> int:
    """Start `n` ranks of bench.py on this node and hand their stdout / stderr through; returns the launcher's exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL fails with hipIpcGetMemHandle otherwise)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), *argv]
    print("[bench] WORLD_SIZE unset and --gpus %d: starting the ranks myself: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def init_world(backend: str, dev=None):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=dev)
    else:
        dist.init_process_group(backend=backend)


def world_proof(dev, expect: int):
    """What the collective itself saw: every rank contributes (rank, local device index); the gathered table must hold `expect` distinct
    ranks.  A launcher that started N processes which never met (N worlds of one) fails here instead of printing N lines of n_gpus=1."""
    rank = dist.get_rank()
    mine = torch.tensor([rank, dev.index if dev.type == "cuda" and dev.index is not None else -1], dtype=torch.int64, device=dev)
    table = torch.empty((dist.get_world_size(), 2), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(table.view(-1), mine)
    table = table.cpu().tolist()
    seen = sorted({r for r, _ in table})
    if seen != list(range(expect)):
        raise RuntimeError(f"world proof failed: the all-gather saw ranks {seen}, expected 0..{expect - 1}")
    return {"backend": dist.get_backend(), "world_seen": len(seen), "devices": [d for _, d in table]}


def make_barrier(distributed: bool, on_gpu: bool):
    def barrier():
        if distributed:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()
    return barrier


def timed_region(step, steps: int, barrier, distributed: bool, dev, replays=None):
    """EXACTLY `steps` passes between two barriers: (wall seconds, MAX over ranks; GPU milliseconds between events on this rank's
    current stream, 0.0 without a GPU).  `replays`: how many calls of `step` hold those passes (single GPU: one replay of a K-pass graph)."""
    on_gpu = dev.type == "cuda"
    barrier()
    if on_gpu:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if on_gpu:
        ev0.record()
    for _ in range(steps if replays is None else replays):
        step()
    if on_gpu:
        ev1.record()
    barrier()
    el = time.perf_counter() - t0
    ms = ev0.elapsed_time(ev1) if on_gpu else 0.0
    if distributed:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    return el, ms


class OverlappedGather:
    """The exchange of the weak-scaling headline step: after pass i the rank's outputs y_all are snapshotted on the compute stream and
    all-gathered on a communication stream while pass i + 1 streams its weights -- xGMI and HBM are different resources, and RCCL's
    enqueue (tens of microseconds of host time) must not sit between two 150 us list launches.  One bucket per step (all 96 rows of y
    at once: 786 KB per rank), never one collective per layer.  The snapshot is re-used: the compute stream waits for the previous
    gather before overwriting it.  sequential=True: the gather on the compute stream (what `us_step_sequential` times)."""

    def __init__(self, y_all, world: int, sequential: bool = False):
        self.y_all, self.sequential = y_all, sequential
        self.snap = torch.empty_like(y_all)
        self.gathered = torch.empty((world * y_all.shape[0],) + tuple(y_all.shape[1:]), dtype=y_all.dtype, device=y_all.device)  # rank-major
        self.on_gpu = y_all.is_cuda
        if self.on_gpu and not sequential:
            self.comm = torch.cuda.Stream(device=y_all.device)
            self.ready, self.done = torch.cuda.Event(), torch.cuda.Event()
            self.done.record(torch.cuda.current_stream(y_all.device))

    def __call__(self):
        if not self.on_gpu or self.sequential:
            dist.all_gather_into_tensor(self.gathered, self.y_all)
            return
        compute = torch.cuda.current_stream(self.y_all.device)
        compute.wait_event(self.done)        # the previous gather has read the snapshot
        self.snap.copy_(self.y_all, non_blocking=True)
        self.ready.record(compute)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(self.ready)
            dist.all_gather_into_tensor(self.gathered, self.snap)
            self.done.record(self.comm)

    @property
    def bytes_per_rank(self):
        """received + sent payload of one all-gather, per rank"""
        w = self.gathered.shape[0] // self.y_all.shape[0]
        return self.y_all.numel() * self.y_all.element_size() * (w - 1)


def rccl_report(proof, replay_one_pass, y_all, world: int, dev, reps: int = 20, passes_per_step: int = 1):
    """The `rccl` object of the JSON line: the all-gather alone, one replay (`passes_per_step` passes) alone, replay + gather back to back on
    one stream, and the overlapped schedule the timed region uses -- each `reps` times between barriers, MAX over ranks, microseconds per
    REPLAY (divide by passes_per_replay for a step)."""
    barrier = make_barrier(True, dev.type == "cuda")

    def us(step):
        step()
        el, _ = timed_region(step, reps, barrier, True, dev)
        return round(el / reps * 1e6, 2)
    seq, ovl = OverlappedGather(y_all, world, sequential=True), OverlappedGather(y_all, world)
    out = dict(proof)
    out["passes_per_replay"] = passes_per_step
    out["all_gather_bytes_per_rank"] = ovl.bytes_per_rank
    out["us_all_gather_alone"] = us(seq)
    out["us_replay_alone"] = us(replay_one_pass)
    out["us_replay_then_gather"] = us(lambda: (replay_one_pass(), seq()))
    out["us_replay_gather_overlapped"] = us(lambda: (replay_one_pass(), ovl()))
    out["schedule_timed"] = "overlapped (bucket snapshot + ONE all-gather per replay on a communication stream, under the next replay)"
    return out


def bench_column_sharded(world: int, rank: int, dev, M: int = 4096, K: int = 8192, N: int = 28672, reps: int = 5, m_tile: int = 1024,
                         forward_impl=None, dtype=torch.bfloat16):
    """configs[4]: W4A16 K x N g128, M rows, the N output columns sharded over `world` ranks (STRONG scaling: the layer is fixed).
    Times, with a barrier + synchronize on both sides and the MAX over ranks: the local GEMM alone, the all-gather alone,
    GEMM + all-gather back to back, and the M-tiled overlapped schedules.  Aggregate TFLOP/s = 2*M*K*N / time."""
    from bitorch_engine.distributed import ColumnShardedMPQLinear, column_range
    on_gpu = dev.type == "cuda"
    lo, hi = column_range(N, rank, world)
    gen = torch.Generator(device=dev).manual_seed(4242 + rank)
    n_loc = hi - lo
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, n_loc), dtype=torch.int32, generator=gen, device=dev)
    sc = (torch.rand((K // 128, n_loc), generator=gen, device=dev) * 0.01 + 0.005).to(dtype)
    ze = (sc.float() * torch.rand((K // 128, n_loc), generator=gen, device=dev) * 15).to(dtype)
    layer = ColumnShardedMPQLinear.__new__(ColumnShardedMPQLinear)
    torch.nn.Module.__init__(layer)
    layer.N, layer.rank, layer.world, layer.group = N, rank, world, None
    layer.w_bit, layer.group_size, layer.asym = 4, 128, False
    layer.ranges = [column_range(N, r, world) for r in range(world)]
    layer.lo, layer.hi = lo, hi
    for name, t in (("qweight", qw), ("scales", sc), ("zeros", ze), ("g_idx", None), ("bias", None)):
        layer.register_buffer(name, t)
    layer._impl = forward_impl
    x = torch.randn((M, K), generator=torch.Generator(device=dev).manual_seed(7), device=dev).to(dtype)
    y_loc = layer.local_forward(x)
    equal = len({h - l for l, h in layer.ranges}) == 1
    gathered = torch.empty((world * M, n_loc), dtype=dtype, device=dev) if equal else None
    barrier = make_barrier(True, on_gpu)

    def timed(fn):
        fn()
        el, _ = timed_region(fn, reps, barrier, True, dev)
        return el / reps * 1e6

    us_gemm = timed(lambda: layer.local_forward(x))
    us_gather = timed(lambda: dist.all_gather_into_tensor(gathered, y_loc)) if equal else None
    us_seq = timed(lambda: layer.forward(x))
    us_ovl = timed(lambda: layer.forward_overlapped(x, m_tile))
    us_ovl_rm = timed(lambda: layer.forward_overlapped(x, m_tile, interleave=False)) if equal else None
    direct = {}
    if os.environ.get("BIE_BENCH_DIRECT", "0") == "1":  # opt-in: the grouped send/recv exchange has only ever run under gloo (CPU tests)
        direct = {"us_direct_interleaved": round(timed(lambda: layer.forward_direct(x, m_tile)), 1)}
        if equal:
            direct["us_direct_rank_major_zero_copy"] = round(timed(lambda: layer.forward_direct(x, m_tile, interleave=False)), 1)
    # the schedules are the same function of x: prove it on the values the timed calls produced
    agree = bool(torch.equal(layer.forward(x), layer.forward_overlapped(x, m_tile)))
    flops = 2.0 * M * K * N
    rnd = lambda v: None if v is None else round(v, 1)  # noqa: E731
    return {**direct, "workload": f"BASELINE.json configs[4]: W4A16 {K}x{N} g128 {str(dtype).split('.')[-1]}, M={M}, {world} column shards of {n_loc}",
            "scaling": "strong", "us_local_gemm": rnd(us_gemm), "us_all_gather": rnd(us_gather),
            "us_gemm_then_gather": rnd(us_seq), "us_overlapped_m_tiles": rnd(us_ovl),
            "us_overlapped_rank_major_output": rnd(us_ovl_rm), "m_tile": m_tile,
            "TFLOP/s_aggregate_overlapped": round(flops / us_ovl / 1e6, 3), "TFLOP/s_aggregate_local_gemm_only": round(flops / us_gemm / 1e6, 3),
            "gather_bytes_per_rank": M * n_loc * 2 * (world - 1), "schedules_agree": agree}


def _rehearsal_impl(x2, qweight, scales, zeros, g_idx, w_bit, asym, group_size, bias):
    """CPU stand-in for the rank-local GEMM of the rehearsal: plain torch unpack + matmul.  NOT a product path (the product has no CPU
    path at all) -- it exists so that `--dry-run` can drive the real schedules (collective order, tiling, layouts) without a GPU."""
    per = 32 // w_bit
    sh = torch.arange(per, dtype=torch.int32) * w_bit
    q = ((qweight.unsqueeze(1) >> sh.view(1, per, 1)) & (2 ** w_bit - 1)).reshape(-1, qweight.shape[1]).float()
    g = torch.arange(q.shape[0]) // group_size
    w = q * scales.float()[g] - zeros.float()[g]
    return (x2.float() @ w).to(x2.dtype)


def dry_run(args, world: int, rank: int) -> None:
    """`bench.py --gpus N --dry-run`: the launcher, the rendezvous, the world proof, the timed-region protocol (barriers, exactly K steps,
    MAX over ranks) and the configs[4] schedules under gloo on the CPU.  No kernel runs and NOTHING is measured: `value` is null."""
    dev = torch.device("cpu")
    init_world("gloo")
    proof = world_proof(dev, world)
    y_all = torch.full((96, 64), float(rank), dtype=torch.bfloat16)
    gather = OverlappedGather(y_all, world)
    barrier = make_barrier(world > 1, False)
    for _ in range(args.warmup):
        gather()
    el, _ = timed_region(gather, args.steps, barrier, world > 1, dev)
    rows_ok = bool(all(float(gather.gathered[r * 96, 0]) == float(r) for r in range(world)))
    c5 = bench_column_sharded(world, rank, dev, M=48, K=256, N=128 * world * 2, reps=2, m_tile=16, forward_impl=_rehearsal_impl, dtype=torch.float32)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "launcher rehearsal (no kernel ran; nothing measured)", "value": None, "unit": "GB/s", "n_gpus": world, "dry_run": True,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "rccl": dict(proof, gathered_rows_rank_major=rows_ok, all_gather_bytes_per_rank=gather.bytes_per_rank),
                          "c5": c5}))
