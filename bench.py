#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X low-bit Q-Linear engine (driver contract in the task brief).

Workload (BASELINE.json configs[1]): W4A16 qlinear, K=4096 -> N=11008, group 128, bf16 activations,
symmetric GBA-style fp zeros, implicit groups.  One STEP = one decode pass (M=1) over L=64 DISTINCT
layers of that shape (64 x 22.5 MB packed weights = 1.44 GB >> the 256 MiB Infinity Cache, so every
weight byte comes from HBM), captured once in a HIP graph and replayed -- exactly what a token of
batch-1 decoding does to the linear layers.  `value` = algorithmic bytes of the step / time (GB/s).
The compute-bound half (M=4096 prefill GEMM on the same layer) is measured right after the timed
region and reported in the extra "gemm" / "roofline_gemm" objects.

Multi-GPU (--gpus N, one process per GPU, launched by torch.distributed.run): output-column sharding --
rank r owns the 11008-column block r of every layer of a [4096 -> 11008*N] stack (weak scaling: fixed
per-GPU work), x is replicated, and the per-step outputs of all layers are exchanged with ONE
bucketed RCCL all-gather over xGMI (the path's only exchange step).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

K, N, GROUP, WBIT, LAYERS = 4096, 11008, 128, 4, 64
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA


def alg_bytes(M, k=K, n=N, w=WBIT, g=GROUP):
    """packed weights + scales + fp zeros + x + y (implicit g_idx: 0 bytes)."""
    G = k // g
    return k * n * w // 8 + 2 * G * n + 2 * G * n + 2 * M * k + 2 * M * n


def make_layer(dev, gen):
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K * WBIT // 32, N), dtype=torch.int64, generator=gen, device="cpu").to(torch.int32).to(dev)
    scales = (torch.rand((K // GROUP, N), generator=gen) * 0.01 + 0.005).to(torch.bfloat16)
    zeros = (scales.float() * torch.rand((K // GROUP, N), generator=gen) * 15).to(torch.bfloat16)
    return qw, scales.to(dev), zeros.to(dev)


def cpu_baseline(budget_s=12.0):
    """The oracle's fused dequant+GEMV (OpenMP over all host cores) on a bounded sample of the same workload."""
    import numpy as np
    from oracle import oracle as orc
    rng = np.random.default_rng(0)
    qw = rng.integers(-2 ** 31, 2 ** 31 - 1, (K * WBIT // 32, N), dtype=np.int64).astype(np.int32)
    gen = torch.Generator().manual_seed(0)
    sc = (torch.rand((K // GROUP, N), generator=gen) * 0.01 + 0.005).to(torch.bfloat16)
    ze = (sc.float() * torch.rand((K // GROUP, N), generator=gen) * 15).to(torch.bfloat16)
    x = torch.randn((1, K), generator=gen).to(torch.bfloat16)
    sc_n, ze_n, x_n = orc.torch_to_np(sc), orc.torch_to_np(ze), orc.torch_to_np(x)
    orc.mpq_forward(x_n, qw, sc_n, ze_n, None, WBIT, GROUP, 0, orc.BF16)  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        orc.mpq_forward(x_n, qw, sc_n, ze_n, None, WBIT, GROUP, 0, orc.BF16)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= LAYERS:
            break
    cores = os.cpu_count() or 1
    return {"value": round(alg_bytes(1) * n / el / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": f"{n} of {LAYERS} layer GEMVs (M=1, 4096x11008 w4 g128 bf16), oracle/bie_oracle.c orc_mpq_forward_f32acc, OpenMP"}


def pmc_traffic():
    """HBM bytes per GEMV launch from the committed PMC passes (profiles/r01_pmc_gemv.json, produced by tools/gpu_final.sh:
    separate FETCH_SIZE and WRITE_SIZE passes, corrected as DESIGN.md section 6 describes); None when the file is absent."""
    p = os.path.join(ROOT, "profiles", "r01_pmc_gemv.json")
    try:
        return json.load(open(p))["gemv_hbm_bytes_per_launch"]
    except Exception:
        return None


def bench_square(L, _hip, dev, k=4096, n=4096, nlayers=48):
    """M=1 decode GEMV (HIP-graph replay over `nlayers` distinct layers, 400 MB of packed weights) and M=4096 GEMM on the
    4096x4096 g128 W4 bf16 layer; outside the timed region of the headline metric."""
    gen = torch.Generator().manual_seed(99)
    layers = []
    for _ in range(nlayers):
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (k * WBIT // 32, n), dtype=torch.int64, generator=gen).to(torch.int32).to(dev)
        sc = (torch.rand((k // GROUP, n), generator=gen) * 0.01 + 0.005).to(torch.bfloat16)
        ze = (sc.float() * torch.rand((k // GROUP, n), generator=gen) * 15).to(torch.bfloat16)
        layers.append((qw, sc.to(dev), ze.to(dev)))
    res = {}
    for M, reps in ((1, 20), (4096, 2)):
        x = torch.randn((M, k), generator=gen).to(torch.bfloat16).to(dev)
        y = torch.empty((M, n), dtype=torch.bfloat16, device=dev)
        ws = torch.zeros(max(L.bie_mpq_workspace_bytes(M, k, n, WBIT), 16), dtype=torch.uint8, device=dev)

        def run(stream_ptr):
            for (qw, sc, ze) in layers:
                rc = L.bie_mpq_forward(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), ze.data_ptr(), None, None, y.data_ptr(), ws.data_ptr(),
                                       ws.numel(), M, k, n, WBIT, GROUP, 0, _hip.BF16, stream_ptr)
                if rc:
                    raise RuntimeError(L.bie_last_error().decode())
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run(side.cuda_stream)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            run(torch.cuda.current_stream().cuda_stream)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * nlayers)
        if M == 1:
            res["gemv_M1"] = {"us_per_launch": round(us, 2), "GB/s": round(alg_bytes(1, k, n) / us / 1e3, 1),
                              "frac_of_hbm_peak": round(alg_bytes(1, k, n) / us / 1e3 / HBM_PEAK_GBS, 4)}
        else:
            tf = 2.0 * M * k * n / us / 1e6
            res["gemm_M4096"] = {"us_per_launch": round(us, 2), "TFLOP/s": round(tf, 1), "frac_of_mfma_peak": round(tf / MFMA_PEAK_TFLOPS, 4)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)

    from bitorch_engine import _hip
    L = _hip.lib()

    gen = torch.Generator().manual_seed(1234 + rank)
    layers = [make_layer(dev, gen) for _ in range(LAYERS)]
    x = torch.randn((1, K), generator=gen).to(torch.bfloat16).to(dev)
    y_all = torch.empty((LAYERS, N), dtype=torch.bfloat16, device=dev)  # row l = output of layer l
    gathered = torch.empty((world * LAYERS, N), dtype=torch.bfloat16, device=dev) if distributed else None  # rank-major
    ws_bytes = L.bie_mpq_workspace_bytes(1, K, N, WBIT)
    ws = torch.zeros(max(ws_bytes, 16), dtype=torch.uint8, device=dev)  # 4 KiB head of split-K counters starts at zero

    def run_layers(stream_ptr):
        for l, (qw, sc, ze) in enumerate(layers):
            rc = L.bie_mpq_forward(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), ze.data_ptr(), None, None,
                                   y_all[l].data_ptr(), ws.data_ptr(), ws.numel(), 1, K, N, WBIT, GROUP, 0, _hip.BF16, stream_ptr)
            if rc:
                raise RuntimeError(L.bie_last_error().decode())

    # capture one decode pass in a HIP graph (launch-bound otherwise: 128 launches of a few microseconds)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run_layers(side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
        run_layers(torch.cuda.current_stream().cuda_stream)

    def step():
        graph.replay()
        if distributed:
            dist.all_gather_into_tensor(gathered, y_all)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    step_bytes = alg_bytes(1) * LAYERS
    value = step_bytes * world / (elapsed / args.steps) / 1e9

    out = None
    if rank == 0:
        # dominant kernel = the M=1 GEMV; one launch per layer.  Duration from HIP events recorded on the launch
        # stream around the timed region / launches (graph replay only when N=1 -> includes the ~1 us kernel
        # boundaries and the split-K finalize kernel: conservative).
        launches = args.steps * LAYERS
        avg_us = gpu_ms * 1e3 / launches
        achieved = alg_bytes(1) / (avg_us * 1e-6) / 1e9
        out = {
            "metric": "W4A16 decode GEMV weight-streaming throughput (M=1, 4096x11008 g128, bf16)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (seeded random packed weights / scales / zeros, N(0,1) activations)",
            "config": {"workload": "BASELINE.json configs[1]: W4A16 qlinear 4096x11008 g128 bf16, M=1 decode pass over 64 distinct layers (1.44 GB of packed weights, HIP-graph replay)",
                       "layers_per_step": LAYERS, "parallelism": "output-column sharding x%d + 1 all-gather/step" % world if distributed else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(),
                         "kernel": "bie::mpq_gemv3_kernel<bf16,w4,M=1> (in-kernel split-K, no finalize launch)", "avg_launch_us": round(avg_us, 3),
                         "alg_bytes_per_launch": alg_bytes(1)},
        }

    # ---- compute-bound half: M=4096 prefill GEMM on the same layer shape (rank 0 only, outside the timed region)
    # (N = 1 only: with a process group alive, its watchdog thread may touch the runtime while rank 0 captures)
    if rank == 0 and world == 1 and not args.no_gemm:
        M = 4096
        xg = torch.randn((M, K), generator=torch.Generator().manual_seed(7)).to(torch.bfloat16).to(dev)
        yg = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        wsg_b = L.bie_mpq_workspace_bytes(M, K, N, WBIT)
        wsg = torch.zeros(max(wsg_b, 16), dtype=torch.uint8, device=dev)
        st = torch.cuda.current_stream().cuda_stream

        def gemm(l):
            qw, sc, ze = layers[l % LAYERS]
            rc = L.bie_mpq_forward(xg.data_ptr(), qw.data_ptr(), sc.data_ptr(), ze.data_ptr(), None, None, yg.data_ptr(),
                                   wsg.data_ptr(), wsg.numel(), M, K, N, WBIT, GROUP, 0, _hip.BF16, st)
            if rc:
                raise RuntimeError(L.bie_last_error().decode())
        # 16 launches over 16 distinct layers captured in a HIP graph and replayed (same method as the decode pass: no host
        # launch gaps between kernels)
        nl = 16
        gside = torch.cuda.Stream()
        gside.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(gside):
            st = gside.cuda_stream
            for i in range(nl):
                gemm(i)
        torch.cuda.current_stream().wait_stream(gside)
        torch.cuda.synchronize()
        ggraph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(ggraph, stream=gside, capture_error_mode="thread_local"):
            st = torch.cuda.current_stream().cuda_stream
            for i in range(nl):
                gemm(i)
        ggraph.replay()
        torch.cuda.synchronize()
        reps = 4
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ggraph.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (reps * nl)
        tf = 2.0 * M * K * N / (us * 1e-6) / 1e12
        out["gemm"] = {"M": M, "us_per_call": round(us, 2), "TFLOP/s": round(tf, 1), "GB/s_algorithmic": round(alg_bytes(M) / (us * 1e-6) / 1e9, 1)}
        out["roofline_gemm"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(tf / MFMA_PEAK_TFLOPS, 4), "traffic": None, "kernel": "bie::mpq_gemm_kernel<bf16,w4,256x256x64 tile>"}

    # ---- BASELINE.json's metric is worded on the 4096x4096 layer (configs[0], the reference's CPU-runnable case): report it too
    if rank == 0 and world == 1 and not args.no_gemm:
        try:
            out["shape_4096x4096"] = bench_square(L, _hip, dev)
        except Exception as e:  # reporting only
            out["shape_4096x4096"] = {"error": str(e)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline()
        except Exception as e:  # the baseline is reporting only; never fail the bench for it
            out["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
    elif rank == 0:
        out["cpu_baseline"] = None

    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
