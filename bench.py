#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X low-bit Q-Linear engine (driver contract in the task brief).

Headline workload = the configuration BASELINE.json's `metric` is worded on: W4A16 qlinear, K = N = 4096, group 128,
bf16 activations, symmetric (GBA-style fp zeros), implicit groups, M = 1.  One STEP = one decode pass over L = 96
DISTINCT layers of that shape, each with its own activation vector (96 x 8.4 MB of packed weights = 0.8 GB >> the 256 MiB
Infinity Cache, so every weight byte comes from HBM), issued as ONE launch of the layer-list kernel (bie_mpq_list_forward,
csrc/mpq_list.hip) captured in a HIP graph and replayed.  `value` = algorithmic bytes of the step / time (GB/s);
`roofline` = the same quantity per launch from HIP events recorded on the launch stream around the timed region.
The per-layer-launch form of the same pass (one bie_mpq_forward per layer, round 2's headline) is reported beside it
(`per_layer_launches_4096x4096`), as are dependent chains inside one launch (`chain4_4096x4096`).

Also measured live, each in its own event-bracketed region (rank 0, N = 1): the M = 4096 prefill GEMM of the metric's layer
(`roofline_gemm`), the 4096x11008 / 11008x4096 layers of configs[1] (M = 1 and M = 4096), the grouped (shared-x) decode
launches q/k/v and gate/up, configs[2] (exl2 3/2-bit decode), configs[3] (binary linear / conv) and configs[4]'s layer
(8192x28672, M = 4096) -- every object carries its own roofline fraction.  `cpu_baseline`: the oracle's fused dequant+GEMV on
the host cores (all cores, and one thread) on a bounded sample.

Multi-GPU (--gpus N, one process per GPU; started by the driver under torch.distributed.run, or by bench.py itself when WORLD_SIZE is
unset -- bench_sharded.launch_ranks): every rank runs the headline pass over its own 96 layers (independent output-column blocks: weak
scaling) and ONE bucketed RCCL all-gather per step assembles the step's outputs, on a communication stream under the next pass; the
`rccl` object carries the world the collective really saw and the exchange timed alone / sequential / overlapped.  configs[4] itself is
in `c5`: 8192x28672 W4 g128, M = 4096, N/world column shards (strong scaling), the all-gather timed separately and overlapped with the
GEMM by M-tiles (bitorch_engine.distributed; timed by bench_sharded.bench_column_sharded).  `--dry-run --backend gloo` rehearses the
launcher, the rendezvous and the timed-region protocol on the CPU (tests/test_bench_launcher_cpu.py); it measures nothing.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

GROUP, WBIT = 128, 4
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA
I8_MFMA_PEAK_TOPS = 5000.0   # dense i8 MFMA (2x the bf16 rate)
FP4_MFMA_PEAK_TOPS = 10000.0  # dense FP4 MFMA (MI355X_MICROARCH.md: ~10 PF dense; 9.1 measured by its microbenchmark)
XOR_POPC_PEAK_TOPS = 1260.0  # v_xor_b32 + v_bcnt_u32_b32 accumulate: 2 VALU per 32 binary MACs per lane
BF16 = torch.bfloat16


def alg_bytes(M, k, n, w=WBIT, g=GROUP):
    """packed weights + scales + fp zeros + x + y (implicit g_idx: 0 bytes)."""
    G = k // g
    return k * n * w // 8 + 2 * G * n + 2 * G * n + 2 * M * k + 2 * M * n


def make_layer(dev, gen, k, n, w_bit=WBIT, dt=BF16):
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (k * w_bit // 32, n), dtype=torch.int32, generator=gen, device=dev)
    scales = (torch.rand((k // GROUP, n), generator=gen, device=dev) * 0.01 + 0.005).to(dt)
    zeros = (scales.float() * torch.rand((k // GROUP, n), generator=gen, device=dev) * (2 ** w_bit - 1)).to(dt)
    return qw, scales, zeros


def capture(run):
    """Warm `run(stream_ptr)` on a side stream, then capture it in a HIP graph."""
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
    return g


PREROLL_S = float(os.environ.get("BIE_BENCH_PREROLL_S", "0.06"))  # untimed replays in front of every timed region


def preroll(replay, seconds=None):
    """Replay untimed until `seconds` of wall clock have passed: a region of a few milliseconds timed right after host-side set-up
    (layer construction, capture) runs at the clock the idle chip was at, 7-10 % under the clock it holds under load (measured on the
    headline: 0.62-0.64 of the roofline timed cold against 0.67-0.69 after 30 ms of the same launches, profiles/r03_bench_preroll.txt)."""
    seconds = PREROLL_S if seconds is None else seconds
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        replay()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return n


def time_graph(g, reps):
    """Microseconds per replay from HIP events on the current stream (after `preroll`)."""
    preroll(g.replay)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


PMC_FILE = "r06_pmc_gemv.json"


def kernel_source_sha():
    """Hash of the decode kernel sources: profiles/r03_pmc_gemv.json is only trusted for the sources it was collected on."""
    h = hashlib.sha256()
    for f in ("mpq_list.hip", "mpq_gemv_lut.hip", "mpq_gemv.hip", "mpq_dequant.cuh"):
        h.update(open(os.path.join(ROOT, "bitorch-engine_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(shape):
    """HBM bytes per GEMV launch from this round's PMC passes (tools/gpu_pmc_traffic.sh -> profiles/r03_pmc_gemv.json:
    separate FETCH_SIZE and WRITE_SIZE passes, corrected as DESIGN.md section 5 describes).  None when the file is absent,
    was collected on other kernel sources (stale), or has no row for `shape`."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
        if d.get("kernel_source_sha") != kernel_source_sha():
            return None
        e = d["shapes"][shape]
        return e.get("hbm_bytes_per_layer", e.get("hbm_bytes_per_launch"))  # list forms: per LAYER (the caller scales to its launch)
    except Exception:
        return None


def pmc_gemm(shape="M4096_K4096_N4096"):
    """Counters of the dense GEMM kernel at the TIMED shape (tools/gpu_pmc_gemm_r06.sh -> profiles/r06_pmc_gemm.json: MFMA-pipe utilisation =
    SQ_VALU_MFMA_BUSY_CYCLES per SIMD / GRBM_GUI_ACTIVE per XCD; FETCH_SIZE / WRITE_SIZE from their own passes).  None when absent or collected
    on other sources of mpq_dense.hip."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r06_pmc_gemm.json")))
        h = hashlib.sha256()
        for f in ("mpq_dense.hip", "mfma_pipe.cuh", "mpq_frag_dequant.cuh"):
            h.update(open(os.path.join(ROOT, "bitorch-engine_amd", "csrc", f), "rb").read())
        if d.get("gemm_source_sha") != h.hexdigest()[:16]:
            return None
        b = d["shapes"][shape]
        return {"shape": shape, "kernel": b["kernel"], "mfma_pipe_utilisation": b["mfma_pipe_utilisation"], "effective_clock_ghz": b["effective_clock_ghz"],
                "fetch_bytes": b["fetch_bytes_corrected"], "write_bytes": b["write_bytes"], "operand_bytes": b["operand_bytes"],
                "fetch_over_operands": b["fetch_over_operands"], "fetch_over_tiling_floor": b["fetch_over_tiling_floor"]}
    except Exception:
        return None


class Bench:
    def __init__(self, dev):
        from bitorch_engine import _hip
        self._hip = _hip
        self.L = _hip.lib()
        self.dev = dev

    def forward(self, x, layer, y, ws, M, k, n, st, w_bit=WBIT):
        qw, sc, ze = layer
        rc = self.L.bie_mpq_forward(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), ze.data_ptr(), None, None, y.data_ptr(), ws.data_ptr(),
                                    ws.numel(), M, k, n, w_bit, GROUP, 0, self._hip.BF16 if x.dtype == BF16 else self._hip.F16, st)
        if rc:
            raise RuntimeError(self.L.bie_last_error().decode())

    def workspace(self, M, k, n, w_bit=WBIT):
        return torch.zeros(max(self.L.bie_mpq_workspace_bytes(M, k, n, w_bit), 16), dtype=torch.uint8, device=self.dev)

    # ---- M = 1 decode over `nl` distinct layers -> per-launch microseconds
    def gemv(self, k, n, nl, reps, seed, M=1, w_bit=WBIT, dt=BF16):
        gen = torch.Generator(device=self.dev).manual_seed(seed)
        layers = [make_layer(self.dev, gen, k, n, w_bit, dt) for _ in range(nl)]
        x = torch.randn((M, k), generator=gen, device=self.dev).to(dt)
        y = torch.empty((M, n), dtype=dt, device=self.dev)
        ws = self.workspace(M, k, n, w_bit)
        g = capture(lambda st: [self.forward(x, l, y, ws, M, k, n, st, w_bit) for l in layers])
        us = time_graph(g, reps) / nl
        b = alg_bytes(M, k, n, w_bit)
        return {"M": M, "K": k, "N": n, "w_bit": w_bit, "layers": nl, "us_per_launch": round(us, 3), "alg_bytes_per_launch": b,
                "roofline": {"bound": "hbm", "achieved": round(b / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(b / us / 1e3 / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(f"{k}x{n}") if (M == 1 and w_bit == 4) else None}}

    # ---- ONE launch over a list of layers (bie_mpq_list_*): `per_launch` entries per launch, optional dependent chains
    def make_list(self, layers, k, n, gen, M=1, chain=0, w_bit=WBIT, ys=None, xs=None):
        from bitorch_engine.layers.qlinear.nbit.cuda import MPQForwardList
        entries = []
        dt = layers[0][1].dtype
        for i, (qw, sc, ze) in enumerate(layers):
            dep = i - 1 if (chain and i % chain) else -1
            x = entries[-1]["y"] if dep >= 0 else (xs[i] if xs is not None else torch.randn((M, k), generator=gen, device=self.dev).to(dt))
            y = ys[i] if ys is not None else torch.empty((M, n), dtype=dt, device=self.dev)
            entries.append({"x": x, "qweight": qw, "scales": sc, "zeros": ze, "y": y, "depends_on": dep})
        return MPQForwardList(entries, w_bit=w_bit, group_size=GROUP)

    def gemv_list(self, k, n, nl, per_launch, reps, seed, chain=0, w_bit=WBIT, key=None, M=1, dt=BF16):
        gen = torch.Generator(device=self.dev).manual_seed(seed)
        layers = [make_layer(self.dev, gen, k, n, w_bit, dt) for _ in range(nl)]
        plans = [self.make_list(layers[p0:p0 + per_launch], k, n, gen, M=M, chain=chain, w_bit=w_bit) for p0 in range(0, nl, per_launch)]
        inner = 4 if len(plans) == 1 else 1  # a graph replay costs 10-16 us by itself: several passes per replay when a pass is one launch
        g = capture(lambda st: [p.forward(st) for _ in range(inner) for p in plans])
        us = time_graph(g, reps) / (nl * inner)
        b = alg_bytes(M, k, n, w_bit)
        return {"M": M, "K": k, "N": n, "w_bit": w_bit, "dtype": "bf16" if dt == BF16 else "f16", "layers": nl, "layers_per_launch": per_launch, "dependent_chain_length": chain,
                "launches_per_pass": len(plans) * plans[0].launches, "us_per_layer": round(us, 3), "alg_bytes_per_layer": b,
                "roofline": {"bound": "hbm", "achieved": round(b / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(b / us / 1e3 / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(key) if key else None}}

    # ---- grouped decode: `ns` column counts sharing x, `nl` distinct groups of layers
    def grouped(self, k, ns, nl, reps, seed, what):
        import ctypes
        gen = torch.Generator(device=self.dev).manual_seed(seed)
        groups = [[make_layer(self.dev, gen, k, n) for n in ns] for _ in range(nl)]
        x = torch.randn((1, k), generator=gen, device=self.dev).to(BF16)
        ys = [torch.empty((1, n), dtype=BF16, device=self.dev) for n in ns]
        cnt = len(ns)
        Narr = (ctypes.c_int * cnt)(*ns)
        ws = torch.zeros(max(self.L.bie_mpq_grouped_workspace_bytes(cnt, Narr, 1, k, WBIT), 16), dtype=torch.uint8, device=self.dev)
        arr = lambda ts: (ctypes.c_void_p * cnt)(*[t.data_ptr() for t in ts])
        args = [(arr([l[0] for l in gset]), arr([l[1] for l in gset]), arr([l[2] for l in gset])) for gset in groups]
        yarr = arr(ys)

        def run(st):
            for (q, s, z) in args:
                rc = self.L.bie_mpq_forward_grouped(x.data_ptr(), cnt, q, s, z, None, yarr, Narr, ws.data_ptr(), ws.numel(), 1, k, WBIT, GROUP,
                                                    0, self._hip.BF16, st)
                if rc:
                    raise RuntimeError(self.L.bie_last_error().decode())
        g = capture(run)
        us = time_graph(g, reps) / nl
        b = sum(alg_bytes(1, k, n) for n in ns) - 2 * k * (cnt - 1)  # x is read once
        return {"what": what, "K": k, "N": list(ns), "us_per_launch": round(us, 3), "alg_bytes_per_launch": b,
                "roofline": {"bound": "hbm", "achieved": round(b / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(b / us / 1e3 / HBM_PEAK_GBS, 4), "traffic": None}}

    # ---- a dependent chain as per-layer launches: x of layer i = y of layer i-1 inside chains of `chain` (what an unmodified caller issues)
    def chain_launches(self, k, nl, chain, reps, seed):
        gen = torch.Generator(device=self.dev).manual_seed(seed)
        s0 = 1.0 / (21.25 * k) ** 0.5
        layers = []
        for _ in range(nl):
            qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (k * WBIT // 32, k), dtype=torch.int32, generator=gen, device=self.dev)
            sc = (s0 * (0.8 + 0.4 * torch.rand((k // GROUP, k), generator=gen, device=self.dev))).to(BF16)
            layers.append((qw, sc, (sc.float() * 7.5).to(BF16)))
        ys = [torch.empty((1, k), dtype=BF16, device=self.dev) for _ in range(nl)]
        x0 = torch.randn((1, k), generator=gen, device=self.dev).to(BF16)
        ws = self.workspace(1, k, k)

        def run(st):
            for i, l in enumerate(layers):
                self.forward(ys[i - 1] if i % chain else x0, l, ys[i], ws, 1, k, k, st)
        us = time_graph(capture(run), reps) / nl
        b = alg_bytes(1, k, k)
        return {"M": 1, "K": k, "N": k, "layers": nl, "dependent_chain_length": chain, "us_per_layer": round(us, 3),
                "finite": bool(torch.isfinite(ys[chain - 1].float()).all()),
                "roofline": {"bound": "hbm", "achieved": round(b / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b / us / 1e3 / HBM_PEAK_GBS, 4), "traffic": None}}

    # ---- the REAL decode step: Llama-7B's linear layers with true y -> x dependencies
    def decode_step(self, n_layers, reps, seed, hidden=4096, inter=11008):
        """Per transformer layer: q/k/v (ONE grouped launch, shared x) -> o (x = q's output) -> gate/up (ONE grouped launch, x = o's
        output) -> down (x = gate's output) -> next layer's q/k/v (x = down's output): every launch reads what the previous one
        wrote, nothing else runs in between (the element-wise glue of a real model is left out, so this times the Q-Linear path
        alone).  W4 g128 bf16 sym, M = 1, `n_layers` DISTINCT layers (32 x 107.6 MB = 3.4 GB of packed weights: every byte from HBM),
        the whole step captured in one HIP graph.  Scales / zeros are drawn so that activations keep unit variance through the chain
        (s = 1 / sqrt(21.25 K), z = 7.5 s, both jittered): the data stays random-like for all 128 dependent launches."""
        import ctypes
        gen = torch.Generator(device=self.dev).manual_seed(seed)

        def unit_layer(k, n):
            qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (k * WBIT // 32, n), dtype=torch.int32, generator=gen, device=self.dev)
            s0 = 1.0 / (21.25 * k) ** 0.5
            sc = (s0 * (0.8 + 0.4 * torch.rand((k // GROUP, n), generator=gen, device=self.dev))).to(BF16)
            ze = (sc.float() * (7.5 + 0.5 * (torch.rand((k // GROUP, n), generator=gen, device=self.dev) - 0.5))).to(BF16)
            return qw, sc, ze
        h = [torch.randn((1, hidden), generator=gen, device=self.dev).to(BF16)] + [torch.empty((1, hidden), dtype=BF16, device=self.dev) for _ in range(n_layers)]
        steps = []  # (x, [layers], [ys])
        for l in range(n_layers):
            qkv = [unit_layer(hidden, hidden) for _ in range(3)]
            yq = [torch.empty((1, hidden), dtype=BF16, device=self.dev) for _ in range(3)]
            o = unit_layer(hidden, hidden)
            yo = torch.empty((1, hidden), dtype=BF16, device=self.dev)
            gu = [unit_layer(hidden, inter) for _ in range(2)]
            yg = [torch.empty((1, inter), dtype=BF16, device=self.dev) for _ in range(2)]
            dn = unit_layer(inter, hidden)
            steps += [(h[l], qkv, yq), (yq[0], [o], [yo]), (yo, gu, yg), (yg[0], [dn], [h[l + 1]])]
        ws = torch.zeros(1 << 24, dtype=torch.uint8, device=self.dev)
        arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        calls = []
        for (x, ls, ys) in steps:
            calls.append((x, len(ls), arr([l[0] for l in ls]), arr([l[1] for l in ls]), arr([l[2] for l in ls]), arr(ys),
                          (ctypes.c_int * len(ls))(*[l[0].shape[1] for l in ls]), x.shape[1]))

        def run(st):
            for (x, cnt, q, s_, z, y, Narr, k) in calls:
                rc = self.L.bie_mpq_forward_grouped(x.data_ptr(), cnt, q, s_, z, None, y, Narr, ws.data_ptr(), ws.numel(), 1, k, WBIT, GROUP,
                                                    0, self._hip.BF16, st)
                if rc:
                    raise RuntimeError(self.L.bie_last_error().decode())
        g = capture(run)
        us = time_graph(g, reps) / n_layers
        b = 3 * alg_bytes(1, hidden, hidden) - 2 * 2 * hidden + alg_bytes(1, hidden, hidden) + 2 * alg_bytes(1, hidden, inter) - 2 * hidden + alg_bytes(1, inter, hidden)
        fin = h[n_layers].float()
        return {"what": "Llama-7B decode step, linear layers only: per layer grouped q/k/v -> o -> grouped gate/up -> down, true y -> x dependencies",
                "layers": n_layers, "launches_per_layer": 4, "us_per_layer": round(us, 2), "us_per_step": round(us * n_layers, 1), "alg_bytes_per_layer": b,
                "final_activation_rms": round(float(fin.pow(2).mean().sqrt()), 4), "finite": bool(torch.isfinite(fin).all()),
                "roofline": {"bound": "hbm", "achieved": round(b / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(b / us / 1e3 / HBM_PEAK_GBS, 4), "traffic": None}}

    # ---- the same decode step through the UNCHANGED module API (MPQLinearCuda modules, prepare_bie_layers): what a caller of the reference gets
    def decode_step_modules(self, n_layers, reps, seed, auto_group=True, hidden=4096, inter=11008, rows=1, dt=BF16):
        """q_proj(h), k_proj(h), v_proj(h), o_proj(q), gate_proj(o), up_proj(o), down_proj(g) per block, written against the reference's module
        API (layers/qlinear/nbit/cuda/mpq_layer.py:206-224); prepare_bie_layers() is the only model-level call.  With auto_group the sibling
        launches are grouped by the library after one observed forward; without, every layer is its own launch (the reference's call pattern)."""
        from bitorch_engine.layers.qlinear.nbit.cuda import MPQLinearCuda
        from bitorch_engine.layers.qlinear.nbit.cuda import mpq_layer
        from bitorch_engine.utils.model_helper import prepare_bie_layers
        gen = torch.Generator(device=self.dev).manual_seed(seed)

        def lin(k, n):
            l = MPQLinearCuda(k, n, w_bit=WBIT, dtype=dt, group_size=GROUP, dq_group_size=32, use_gba_quant=True, asym=False)
            l.qweight.data = torch.empty(l.qweight.shape, dtype=torch.int32)
            return l

        class Block(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.q_proj, self.k_proj, self.v_proj, self.o_proj = lin(hidden, hidden), lin(hidden, hidden), lin(hidden, hidden), lin(hidden, hidden)
                self.gate_proj, self.up_proj, self.down_proj = lin(hidden, inter), lin(hidden, inter), lin(inter, hidden)

            def forward(self, h):
                q, k, v = self.q_proj(h), self.k_proj(h), self.v_proj(h)   # noqa: F841  (k, v are computed: the attention that eats them is not this library's)
                o = self.o_proj(q)
                g, u = self.gate_proj(o), self.up_proj(o)                  # noqa: F841
                return self.down_proj(g)
        model = torch.nn.Sequential(*[Block() for _ in range(n_layers)])
        prepare_bie_layers(model)
        model.to(self.dev).eval()
        for m in model.modules():
            if isinstance(m, MPQLinearCuda):
                k = m.in_channels
                m.qweight.data = torch.randint(-2 ** 31, 2 ** 31 - 1, m.qweight.shape, dtype=torch.int32, generator=gen, device=self.dev)
                s0 = 1.0 / (21.25 * k) ** 0.5
                m.scales = (s0 * (0.8 + 0.4 * torch.rand(m.scales.shape, generator=gen, device=self.dev))).to(dt)
                m.zeros = (m.scales.float() * 7.5).to(dt)
        old = mpq_layer.AUTO_GROUP
        mpq_layer.AUTO_GROUP = bool(auto_group)
        try:
            h0 = torch.randn((rows, hidden), generator=gen, device=self.dev).to(dt)
            with torch.no_grad():
                model(h0)  # the observation round (sibling groups are confirmed by what they receive)
                model(h0)
                torch.cuda.synchronize()
                before = dict(mpq_layer.GROUP_STATS)
                out = []

                def run(_st):
                    out[:] = [model(h0)]
                g = capture(run)
                stats = {k: mpq_layer.GROUP_STATS[k] - before[k] for k in ("grouped_launches", "served_from_group", "single_launches")}
                us = time_graph(g, reps) / n_layers
        finally:
            mpq_layer.AUTO_GROUP = old
        b = 3 * alg_bytes(1, hidden, hidden) - 2 * 2 * hidden + alg_bytes(1, hidden, hidden) + 2 * alg_bytes(1, hidden, inter) - 2 * hidden + alg_bytes(1, inter, hidden)
        # capture() runs the callable twice (warm-up + capture): launches per forward = half of the counted ones
        return {"what": "the same step through MPQLinearCuda modules + prepare_bie_layers, caller unchanged", "layers": n_layers, "auto_group": bool(auto_group),
                "launches_per_layer": round((stats["grouped_launches"] + stats["single_launches"]) / 2 / n_layers, 2), "us_per_layer": round(us, 2),
                "finite": bool(torch.isfinite(out[0].float()).all()),
                "roofline": {"bound": "hbm", "achieved": round(b / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b / us / 1e3 / HBM_PEAK_GBS, 4), "traffic": None}}

    # ---- M = 4096 prefill GEMM over `nl` distinct layers
    def gemm(self, M, k, n, nl, reps, seed):
        gen = torch.Generator(device=self.dev).manual_seed(seed)
        layers = [make_layer(self.dev, gen, k, n) for _ in range(nl)]
        x = torch.randn((M, k), generator=gen, device=self.dev).to(BF16)
        y = torch.empty((M, n), dtype=BF16, device=self.dev)
        ws = self.workspace(M, k, n)
        g = capture(lambda st: [self.forward(x, l, y, ws, M, k, n, st) for l in layers])
        us = time_graph(g, reps) / nl
        tf = 2.0 * M * k * n / us / 1e6
        return {"M": M, "K": k, "N": n, "layers": nl, "us_per_launch": round(us, 2), "GB/s_algorithmic": round(alg_bytes(M, k, n) / us / 1e3, 1),
                "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(tf / MFMA_PEAK_TFLOPS, 4), "traffic": None}}


def plan_x(plan):
    """The activation tensors of a list plan's entries, in entry order."""
    return [k[0] for k in plan._keep]


def bf16_close(y, ref):
    """The parity bar of tests/test_gpu_parity.py: |y - ref| <= 1e-3 max|ref| + one bf16 ulp of ref.  Returns (ok, worst / bound)."""
    y, ref = y.float(), ref.float()
    tol = 1e-3 * ref.abs().max() + ref.abs() * 2.0 ** -7
    d = (y - ref).abs()
    return bool((d <= tol).all()) and bool(torch.isfinite(y).all()), float((d / tol.clamp_min(1e-30)).max())


def verify_list_outputs(B, plan, layers, y_all, K, N):
    """After the timed region: every row of the timed launch's output against the per-layer entry point (bie_mpq_forward) on the same
    layer and activation.  True / False goes into the bench line; the oracle itself checks two rows in the cpu_baseline leg."""
    ws = B.workspace(1, K, N)
    st = torch.cuda.current_stream().cuda_stream
    ytmp = torch.empty((1, N), dtype=BF16, device=y_all.device)
    worst, ok = 0.0, True
    for i, (l, x) in enumerate(zip(layers, plan_x(plan))):
        B.forward(x, l, ytmp, ws, 1, K, N, st)
        o, w = bf16_close(y_all[i:i + 1], ytmp)
        ok, worst = ok and o, max(worst, w)
    torch.cuda.synchronize()
    return bool(ok and float(y_all.float().abs().max()) > 0)


def bench_exl2_decode_step(dev, n_layers=16, reps=10, hidden=4096, inter=11008):
    """configs[2] as serving sees it: a Llama-7B-shaped decode step over mixed 3/2-bit (exl2, g32 rows, every layer its own random q_perm) layers
    with TRUE y -> x dependencies -- per transformer layer q/k/v in one grouped call (bie_mbwq_exl2_forward_grouped) -> o (x = q's output) ->
    gate/up grouped (x = o's output) -> down (x = gate's output) -> next layer; scales drawn so that the activations keep unit variance."""
    from bitorch_engine.extensions import q_linear_cuda
    from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
    gen = torch.Generator().manual_seed(9)
    meta = {}

    def layer(K, N):
        if K not in meta:
            qg, row = [], 0
            for b in (3, 2):
                for _ in range(K // 2 // 32):
                    qg += [b, row]
                    row += b
            q_groups = torch.tensor(qg, dtype=torch.short)
            meta[K] = (q_groups, row, len(qg) // 2, make_group_map(q_groups, row).to(dev))
        q_groups, row, groups, gmap = meta[K]
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (row, N), dtype=torch.int32, device=dev)
        s0 = 1.0 / (3.0 * K) ** 0.5  # q - zero ~ uniform on a few levels: unit-variance outputs for unit-variance inputs, roughly
        sc = (s0 * (0.8 + 0.4 * torch.rand((groups, N), device=dev))).half()
        zq = torch.cat([torch.full((groups // 2, N), 3.5, device=dev), torch.full((groups - groups // 2, N), 1.5, device=dev)], 0)  # mid level of the 3- / 2-bit halves
        ze = (sc.float() * zq).half()
        perm = torch.randperm(K, generator=gen).to(torch.short).to(dev)
        _, rows = q_linear_cuda.mbwq_trans_qweight(qw, q_groups, True, K, groups, 4)
        return (qw, sc, ze, perm, gmap, rows), row * N * 4 + 4 * groups * N + 2 * K
    blocks, byts = [], 0
    for _ in range(n_layers):
        blk = {}
        for name, (K, N, cnt) in (("qkv", (hidden, hidden, 3)), ("o", (hidden, hidden, 1)), ("gu", (hidden, inter, 2)), ("down", (inter, hidden, 1))):
            made = [layer(K, N) for _ in range(cnt)]
            blk[name] = [m[0] for m in made]
            byts += sum(m[1] for m in made)
        blocks.append(blk)
    h0 = torch.randn((1, hidden), device=dev).half()
    out = []

    def run(_st):
        h = h0
        for blk in blocks:
            q = q_linear_cuda.mbwq_exl2_forward_grouped(h, blk["qkv"])[0]
            o = q_linear_cuda.mbwq_exl2_forward(q, *blk["o"][0], False)
            g = q_linear_cuda.mbwq_exl2_forward_grouped(o, blk["gu"])[0]
            h = q_linear_cuda.mbwq_exl2_forward(g, *blk["down"][0], False)
        out[:] = [h]
    g = capture(run)
    us = time_graph(g, reps) / n_layers
    b = byts / n_layers
    fin = out[0].float()
    return {"op": "exl2 w3/w2 g32 Llama-7B decode step, linear layers only: grouped q/k/v -> o -> grouped gate/up -> down, true y -> x dependencies (6 launches per layer)",
            "M": 1, "K": hidden, "N": inter, "layers": n_layers, "us_per_layer": round(us, 2), "alg_bytes_per_layer": int(b),
            "final_activation_rms": round(float(fin.pow(2).mean().sqrt()), 4), "finite": bool(torch.isfinite(fin).all()),
            "roofline": {"bound": "hbm", "achieved": round(b / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b / us / 1e3 / HBM_PEAK_GBS, 4), "traffic": None}}


def bench_exl2(dev):
    """configs[2]: exl2 mixed 3/2-bit decode (g32 rows, random q_perm), Llama-7B shapes, fp16 (the reference kernel is fp16 only)."""
    from bitorch_engine.extensions import q_linear_cuda
    from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map
    gen = torch.Generator().manual_seed(5)
    out = []
    for (K, N) in ((4096, 4096), (4096, 11008), (11008, 4096)):
        qg, row = [], 0
        for b in (3, 2):
            for _ in range(K // 2 // 32):
                qg += [b, row]
                row += b
        groups = len(qg) // 2
        q_groups = torch.tensor(qg, dtype=torch.short)
        gmap = make_group_map(q_groups, row).to(dev)
        perm = torch.randperm(K, generator=gen).to(torch.short).to(dev)
        nset = max(2, min(32, int(500e6 // (row * N * 4))))
        sets = []
        for _ in range(nset):
            qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (row, N), dtype=torch.int32, device=dev)
            sc = (torch.rand((groups, N), device=dev) * 0.02 + 0.001).half()
            ze = (torch.randn((groups, N), device=dev) * 0.05).half()
            sets.append((qw, sc, ze))
        rows = [q_linear_cuda.mbwq_trans_qweight(s_[0], q_groups, True, K, groups, 4)[1] for s_ in sets][0]  # the load-time step, every tensor
        byts = row * N * 4 + 4 * groups * N + 6 * K + 2 * K + 2 * N
        for M in (1, 2):
            x = torch.randn((M, K), device=dev).half()
            g = capture(lambda st: [q_linear_cuda.mbwq_exl2_forward(x, s_[0], s_[1], s_[2], perm, gmap, rows, False) for s_ in sets])
            us = time_graph(g, 10) / nset
            out.append({"op": "exl2 w3/w2 g32 decode, one launch per layer", "M": M, "K": K, "N": N, "us_per_launch": round(us, 2),
                        "roofline": {"bound": "hbm", "achieved": round(byts / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(byts / us / 1e3 / HBM_PEAK_GBS, 4), "traffic": None}})
        # the same layers, each with its own x, as ONE launch (bie_mbwq_exl2_list_*); four rows of x ride on the same matrix instruction
        from bitorch_engine.layers.qlinear.nbit.cuda import MBWQExl2ForwardList
        for Ml in (1, 2, 4, 8, 16):
            ents = [{"x": torch.randn((Ml, K), device=dev).half(), "qweight": s_[0], "scales": s_[1], "zeros": s_[2], "q_perm": perm, "q_group_map": gmap,
                     "rows": rows, "y": torch.empty((Ml, N), dtype=torch.float16, device=dev)} for s_ in sets]
            plan = MBWQExl2ForwardList(ents)
            us = time_graph(capture(lambda st: [plan.forward(st) for _ in range(4)]), 10) / (nset * 4)
            bl = byts + 2 * (Ml - 1) * (K + N)
            out.append({"op": "exl2 w3/w2 g32 decode, layer list in one launch", "M": Ml, "K": K, "N": N, "layers": nset, "us_per_layer": round(us, 2),
                        "roofline": {"bound": "hbm", "achieved": round(bl / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(bl / us / 1e3 / HBM_PEAK_GBS, 4), "traffic": None}})
            del plan, ents
        # siblings on ONE x (every layer its own q_perm): bie_mbwq_exl2_forward_grouped, two launches per group -- what MBWQLinearCuda layers behind
        # prepare_bie_layers() run for q / k / v and gate / up
        for nmem in ((3, 2) if N == K else (2,)):
            if nset < 2 * nmem:
                continue
            xg = torch.randn((1, K), device=dev).half()
            grps = [[(s_[0], s_[1], s_[2], perm, gmap, rows) for s_ in sets[i:i + nmem]] for i in range(0, nset - nmem + 1, nmem)]
            g = capture(lambda st: [q_linear_cuda.mbwq_exl2_forward_grouped(xg, m) for m in grps])
            us = time_graph(g, 10) / len(grps)
            out.append({"op": f"exl2 w3/w2 g32 decode, {nmem} siblings on one x in one grouped call (2 launches)", "M": 1, "K": K, "N": N, "members": nmem,
                        "us_per_group": round(us, 2), "us_per_layer": round(us / nmem, 2),
                        "roofline": {"bound": "hbm", "achieved": round(nmem * byts / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "frac": round(nmem * byts / us / 1e3 / HBM_PEAK_GBS, 4), "traffic": None}})
        # more rows in one lone call: up to 16 the permute kernel + the pre-permuted decode body (a group of one, two launches); up to 48 the decode stream
        # feeding v_mfma_f32_16x16x32_f16 (x permute launch + exl2_mfma_kernel); beyond, HIP reconstruct + library GEMM (the reference's split for
        # M > 32, mbwq_linear_cuda_kernel.cu:947-957)
        for M in (4, 8, 16, 32, 64, 512, 4096):
            x = torch.randn((M, K), device=dev).half()
            nl = 4 if M <= 512 else 2
            g = capture(lambda st: [q_linear_cuda.mbwq_exl2_forward(x, s_[0], s_[1], s_[2], perm, gmap, rows, False) for s_ in sets[:nl]])
            us = time_graph(g, 5 if M <= 512 else 3) / nl
            tf = 2.0 * M * K * N / us / 1e6
            row = {"op": "exl2 w3/w2 g32, " + ("lone call: permute kernel + pre-permuted decode body" if M <= 16 else "fused matrix-pipe kernel" if M <= q_linear_cuda.EXL2_GEMV_MAX_M
                                               else "prefill form: exl2_dequant_frag_kernel + x[:, q_perm] + mpq_dense_gemm_kernel (three launches, all timed)"),
                   "M": M, "K": K, "N": N, "us_per_launch": round(us, 2), "TFLOP/s": round(tf, 2)}
            if M >= 512:
                row["roofline"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4), "traffic": None}
            else:
                row["roofline"] = {"bound": "hbm", "achieved": round((byts + 2 * M * (K + N)) / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round((byts + 2 * M * (K + N)) / us / 1e3 / HBM_PEAK_GBS, 4), "traffic": None}
            out.append(row)
    return out


def bench_act_order(dev):
    """Random-permuted g_idx (GPTQ act-order) at 4096x11008 against the trivial g_idx, M = 1 and M = 4096: the Python-level call
    (memoised sorted copy + bie_gather_cols + the fast kernel) under graph replay."""
    from bitorch_engine.extensions import q_linear_cuda as q
    K, N = 4096, 11008
    gen = torch.Generator(device=dev).manual_seed(41)
    out = {"K": K, "N": N}
    for M in (1, 4096):
        nl = 8 if M == 1 else 2
        layers = [make_layer(dev, gen, K, N) for _ in range(nl)]
        gidx = [(torch.arange(K, dtype=torch.int32, device=dev) // GROUP)[torch.randperm(K, generator=gen, device=dev)] for _ in range(nl)]
        x = torch.randn((M, K), generator=gen, device=dev).to(BF16)
        for label in ("trivial", "act_order"):
            def run(_st):
                for l, gi in zip(layers, gidx):
                    q.mpq_forward_impl(x, l[0], l[1], l[2], gi if label == "act_order" else None, WBIT, 0, GROUP)
            run(None)
            torch.cuda.synchronize()
            out[f"M{M}_{label}_us"] = round(time_graph(capture(run), 20 if M == 1 else 5) / nl, 2)
        out[f"M{M}_ratio"] = round(out[f"M{M}_act_order_us"] / out[f"M{M}_trivial_us"], 3)
    return out


def bench_binary(dev, L):
    """configs[3]: binary XNOR linear 4096x4096 (row-packed) and the ResNet-18 3x3x512 conv on 7x7 maps."""
    from bitorch_engine.extensions import binary_conv_cpp
    out = []
    K = N = 4096
    wsets = [torch.randint(0, 256, (N, K // 8), dtype=torch.int32, device=dev).to(torch.uint8) for _ in range(16)]
    for M in (1, 64, 512, 4096):
        xp = torch.randint(0, 256, (M, K // 8), dtype=torch.int32, device=dev).to(torch.uint8)
        y = torch.empty((M, N), dtype=torch.float32, device=dev)

        def run(st):
            for w in wsets:
                rc = L.bie_binary_linear_forward(xp.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, 0, 1.0, st)
                if rc:
                    raise RuntimeError(L.bie_last_error().decode())
        us = time_graph(capture(run), 10 if M < 4096 else 3) / len(wsets)
        byts = K * N // 8 + M * K // 8 + 4 * M * N
        tops = 2.0 * M * K * N / us / 1e6
        rf = ({"bound": "hbm", "achieved": round(byts / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(byts / us / 1e3 / HBM_PEAK_GBS, 4)}
              if M < 16 else {"bound": "valu xor+bcnt", "achieved": round(tops, 1), "peak": XOR_POPC_PEAK_TOPS, "unit": "TOP/s", "frac": round(tops / XOR_POPC_PEAK_TOPS, 4)})
        out.append({"op": "binary linear 4096x4096", "M": M, "us_per_launch": round(us, 2), "TOP/s": round(tops, 1), "roofline": dict(rf, traffic=None)})
    # the same product on the matrix pipe (M >= 256): +-1 as FP4 operands of v_mfma_scale_f32_32x32x64_f8f6f4, bit-identical integers.
    # Timed end to end from the same packed operands (x image pass + GEMM; the weight image is per-tensor, built once) and GEMM alone.
    bias_a = torch.randn(K, device=dev).to(BF16)
    sa, sw = torch.tensor(0.7, device=dev).to(BF16), torch.tensor(0.01, device=dev).to(BF16)
    st0 = torch.cuda.current_stream().cuda_stream
    wimgs = []
    for w in wsets[:8]:
        img = torch.empty(L.bie_binary_fp4_image_bytes(N, K), dtype=torch.uint8, device=dev)
        if L.bie_binary_fp4_image(w.data_ptr(), img.data_ptr(), N, K, st0):
            raise RuntimeError(L.bie_last_error().decode())
        wimgs.append(img)
    for M in (192, 512, 4096, 8192):
        xp = torch.randint(0, 256, (M, K // 8), dtype=torch.int32, device=dev).to(torch.uint8)
        xv = torch.randn((M, K), device=dev).to(BF16)
        y = torch.empty((M, N), dtype=torch.float32, device=dev)
        yb = torch.empty((M, N), dtype=BF16, device=dev)
        ximg = torch.empty(L.bie_binary_fp4_image_bytes(M, K), dtype=torch.uint8, device=dev)

        def chk(rc):
            if rc:
                raise RuntimeError(L.bie_last_error().decode())

        def run_e2e(st):
            for wi in wimgs:
                chk(L.bie_binary_fp4_image(xp.data_ptr(), ximg.data_ptr(), M, K, st))
                chk(L.bie_binary_linear_forward_fp4(ximg.data_ptr(), wi.data_ptr(), y.data_ptr(), M, N, K, 1.0, st))

        def run_gemm(st):
            for wi in wimgs:
                chk(L.bie_binary_linear_forward_fp4(ximg.data_ptr(), wi.data_ptr(), y.data_ptr(), M, N, K, 1.0, st))

        def run_layer(st):
            for wi in wimgs:
                chk(L.bie_binary_fp4_image_from_values(xv.data_ptr(), bias_a.data_ptr(), ximg.data_ptr(), M, K, 1, st))
                chk(L.bie_binary_linear_layer_fp4(ximg.data_ptr(), wi.data_ptr(), sa.data_ptr(), sw.data_ptr(), yb.data_ptr(), M, N, K, 1, st))
        reps = 10 if M < 4096 else 3
        us_e, us_g, us_l = (time_graph(capture(f), reps) / len(wimgs) for f in (run_e2e, run_gemm, run_layer))
        ops = 2.0 * M * K * N
        out.append({"op": "binary linear 4096x4096 on the matrix pipe (FP4 +-1 operands, bit-identical): x image pass + GEMM from packed operands", "M": M,
                    "us_per_call": round(us_e, 2), "TOP/s": round(ops / us_e / 1e6, 1), "gemm_only_us": round(us_g, 2),
                    "layer_forward_bf16_two_launches_us": round(us_l, 2),
                    "roofline": {"bound": "mfma fp4", "achieved": round(ops / us_g / 1e6, 1), "peak": FP4_MFMA_PEAK_TOPS, "unit": "TOP/s",
                                 "frac": round(ops / us_g / 1e6 / FP4_MFMA_PEAK_TOPS, 4), "traffic": None,
                                 "kernel": "bie::xnor_fp4_gemm_kernel (v_mfma_scale_f32_32x32x64_f8f6f4), GEMM alone"}})
    # the whole BinaryLinearCuda layer forward (activation bias + sign-pack of bf16 x, XNOR-popcount, cast, both scales) in one launch
    for M in (1, 16, 64, 256):
        x = torch.randn((M, K), device=dev).to(BF16)
        yb = torch.empty((M, N), dtype=BF16, device=dev)

        def run_fused(st):
            for w in wsets:
                rc = L.bie_binary_linear_fused(x.data_ptr(), bias_a.data_ptr(), w.data_ptr(), sa.data_ptr(), sw.data_ptr(), yb.data_ptr(),
                                               M, N, K, 1, 0, st)  # dtype 1 = bf16
                if rc:
                    raise RuntimeError(L.bie_last_error().decode())
        us = time_graph(capture(run_fused), 10) / len(wsets)
        byts = K * N // 8 + 2 * M * K + 2 * K + 2 * M * N
        out.append({"op": "binary layer forward 4096x4096, one launch (bf16 x -> bf16 y)", "M": M, "us_per_launch": round(us, 2),
                    "TOP/s": round(2.0 * M * K * N / us / 1e6, 1),
                    "roofline": ({"bound": "hbm", "achieved": round(byts / us / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(byts / us / 1e3 / HBM_PEAK_GBS, 4), "traffic": None} if M < 16 else
                                 {"bound": "valu xor+bcnt", "achieved": round(2.0 * M * K * N / us / 1e6, 1), "peak": XOR_POPC_PEAK_TOPS,
                                  "unit": "TOP/s", "frac": round(2.0 * M * K * N / us / 1e6 / XOR_POPC_PEAK_TOPS, 4), "traffic": None})})
    for B in (1, 32, 128):  # BASELINE configs[3]
        x = torch.randn((B, 512, 7, 7), device=dev)
        w = torch.randn((512, 512, 3, 3), device=dev)
        from bitorch_engine.extensions._binary_common import pack_rows
        wp = pack_rows(w.reshape(512, -1)).contiguous()  # what BinaryConv2dCPP.generate_quantized_weight stores (eval mode)
        # 16 calls per graph replay: a replay costs ~10 us by itself (MI355X_MICROARCH.md "graph-replay-floor"; an EMPTY kernel timed one per
        # replay reads 10.1 us, profiles/r04_l_conv_dbg.txt), which rounds 1-3 charged to this row at B = 1
        fn = lambda st: [binary_conv_cpp.forward(x, wp, 512, B * 49, 512 * 9, 3, 1, 1, 1, 7) for _ in range(16)]
        try:
            us = time_graph(capture(fn), 20) / 16
            tops = 2.0 * B * 49 * 512 * 4608 / us / 1e6
            from bitorch_engine.extensions import _binary_common as bc
            # the dispatch of extensions/_binary_common.py::conv2d: ONE launch either way (round 6) -- the VALU form (sign-pack into LDS + XNOR-popcount)
            # up to conv_fused_max_rows() output pixels, the matrix-pipe form (FP4 image of the input rows in LDS, fragments gathered tap by tap) beyond
            valu = B * 49 <= bc.conv_fused_max_rows()
            peak = XOR_POPC_PEAK_TOPS if valu else FP4_MFMA_PEAK_TOPS
            out.append({"op": "binary conv 512->512 3x3 on 7x7", "B": B, "us_per_call": round(us, 2), "TOP/s": round(tops, 2), "launches": 1,
                        "kernel": "bie::xnor_conv_fused_kernel (v_xor + v_bcnt)" if valu else "bie::xnor_conv_mfma_kernel (v_mfma_scale_f32_32x32x64_f8f6f4)",
                        "roofline": {"bound": "valu xor+bcnt" if valu else "mfma fp4 (sign-pack and write-out inside the same launch)", "achieved": round(tops, 2),
                                     "peak": peak, "unit": "TOP/s", "frac": round(tops / peak, 5), "traffic": None}})
        except Exception as e:  # reporting only
            out.append({"op": "binary conv 512->512 3x3 on 7x7", "B": B, "error": str(e)[:200]})
    return out


def bench_int_gemm(dev, L):
    """W8A8 (bie_q8_gemm: int8 x int8 -> fp32 with two scales) and W4A4 (bie_q4_gemm, packed nibbles -> bf16) on the i8 matrix
    cores, 4096^3 and the 4096 -> 11008 layer at M = 4096."""
    out = []
    for (M, N, K) in ((4096, 4096, 4096), (4096, 11008, 4096), (256, 4096, 4096)):
        a8 = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
        ws8 = [torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev) for _ in range(4)]
        y = torch.empty((M, N), dtype=torch.float32, device=dev)
        a4 = torch.randint(-128, 128, (M, K // 2), dtype=torch.int8, device=dev)
        ws4 = [torch.randint(-128, 128, (N, K // 2), dtype=torch.int8, device=dev) for _ in range(4)]
        yb = torch.empty((M, N), dtype=BF16, device=dev)

        def run8(st):
            for w in ws8:
                if L.bie_q8_gemm(a8.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, 0.01, 0.02, st):
                    raise RuntimeError(L.bie_last_error().decode())

        def run4(st):
            for w in ws4:
                if L.bie_q4_gemm(a4.data_ptr(), w.data_ptr(), yb.data_ptr(), M, N, K, 0.3, 0.3, 1, 1, 0, 0, 0, st):
                    raise RuntimeError(L.bie_last_error().decode())
        for name, fn, kern in (("W8A8", run8, "bie::i8_pipe_gemm_kernel (v_mfma_i32_32x32x32_i8, ordered-asm pipeline)" if M >= 128 else "bie::int_gemm_kernel"),
                               ("W4A4", run4, "bie::i4_pipe_gemm_kernel (nibbles expanded to i8 beside the MFMAs, ordered-asm pipeline)" if M >= 128 else "bie::int_gemm_kernel")):
            us = time_graph(capture(fn), 5) / 4
            tops = 2.0 * M * N * K / us / 1e6
            out.append({"op": f"{name} GEMM", "M": M, "N": N, "K": K, "us_per_launch": round(us, 2), "TOP/s": round(tops, 1),
                        "roofline": {"bound": "mfma i8", "achieved": round(tops, 1), "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s",
                                     "frac": round(tops / I8_MFMA_PEAK_TOPS, 4), "traffic": None, "kernel": kern}})
    return out


def cpu_baselines(check_layers=(), check_x=(), check_y=(), budget_s=24.0):
    """The CPU restatement (oracle/bie_oracle.c, a "port": proved equal to the reference's CPU path on the golden vectors) timed on the
    host cores, SURVEY.md section 8d: for every workload a bounded sample at the best of {all, 32, 8} OpenMP threads and at one thread.
    Workloads: W4A16 M = 1 fused dequant+GEMV (4096x4096, 4096x11008, 8192x28672), M = 4096 on the metric's layer split into
    dequant only / GEMM only / both (the reference's M > 32 path: unpack_qweight + torch.matmul, mpq_layer.py:59-63), exl2 3/2-bit
    decode, binary linear (M = 1, 64) and the ResNet-18 3x3x512 binary conv."""
    import ctypes
    import numpy as np
    from oracle import oracle as orc
    try:
        omp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        omp = None
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)  # cores this process may run on ...
    try:  # ... capped by the container's CPU quota (cgroup v2 cpu.max "<quota> <period>"): the GPU boxes show 256 logical CPUs and grant 16 --
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # 256 OpenMP threads on 16 cores' worth of time measured 0.7 GB/s against 12 at 32 threads
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    res = []
    t_start = time.perf_counter()

    def timed(fn, max_s, max_n=64):
        fn()  # warm-up
        t0 = time.perf_counter()
        cnt = 0
        while True:
            fn()
            cnt += 1
            el = time.perf_counter() - t0
            if el > max_s or cnt >= max_n:
                return cnt, el

    def run(name, fn, unit_per_call, unit, sample, per_run_s=0.6, thread_sets=None):
        """unit_per_call: bytes (GB/s) or flops (GFLOP/s) or binary ops (GOP/s) of one call."""
        for threads in (thread_sets or default_sets):
            if omp is not None:
                omp.omp_set_num_threads(threads)
            elif threads != 1:
                continue
            if time.perf_counter() - t_start > budget_s * 2.5:  # hard stop: the bench line must finish
                return
            cnt, el = timed(fn, per_run_s)
            res.append({"workload": name, "value": round(unit_per_call * cnt / el / 1e9, 3), "unit": unit, "cores": threads, "kind": "port",
                        "sample": f"{cnt} x {sample}, oracle/bie_oracle.c, " + (f"OpenMP, {threads} threads" if threads > 1 else "one thread")})

    # ---- the oracle as the CHECKER of the timed launch: rows of y_all against orc_mpq_forward on the same layer and x
    if check_layers:
        ok = True
        for (qw_t, sc_t, ze_t), x_t, y_t in zip(check_layers, check_x, check_y):
            ref = orc.mpq_forward(orc.torch_to_np(x_t.cpu()), qw_t.cpu().numpy(), orc.torch_to_np(sc_t.cpu()), orc.torch_to_np(ze_t.cpu()), None, WBIT, GROUP, 0, orc.BF16)
            ok = ok and bf16_close(y_t.reshape(1, -1).cpu(), orc.np_to_torch(ref, BF16).reshape(1, -1))[0]
        res.append({"workload": "oracle_check_of_timed_launch", "verified": bool(ok), "rows": len(check_layers)})
    rng = np.random.default_rng(0)
    gen = torch.Generator().manual_seed(0)
    # thread counts: ALL host cores and one thread, with 32 and 8 between them as the scaling evidence.  The M = 1
    # legs run the list form (orc_mpq_forward_list_f32acc: one statically partitioned OpenMP region over the 64-column blocks of
    # several layers -- the CPU twin of the GPU's list launch), so that every core of a 256-thread host has work
    big = min(cores, 32)
    default_sets = sorted({cores, big, min(cores, 8), 1}, reverse=True)
    few_sets = sorted({cores, big, 1}, reverse=True)

    def w4_layer(k, n):
        qw = rng.integers(-2 ** 31, 2 ** 31 - 1, (k * WBIT // 32, n), dtype=np.int64).astype(np.int32)
        sc = (torch.rand((k // GROUP, n), generator=gen) * 0.01 + 0.005).to(BF16)
        ze = (sc.float() * torch.rand((k // GROUP, n), generator=gen) * 15).to(BF16)
        return qw, orc.torch_to_np(sc), orc.torch_to_np(ze)

    # ---- M = 1 fused dequant + GEMV over a list of layers (each with its own x), like the timed GPU launch
    for (k, n, nl) in ((4096, 4096, 16), (4096, 11008, 6), (8192, 28672, 2)):
        lst = []
        for _ in range(nl):
            qw, sc, ze = w4_layer(k, n)
            lst.append((orc.torch_to_np(torch.randn((1, k), generator=gen).to(BF16)), qw, sc, ze))
        run(f"w4a16_gemv_M1_{k}x{n}", lambda: orc.mpq_forward_list(lst, 1, WBIT, GROUP, 0, orc.BF16), alg_bytes(1, k, n) * nl, "GB/s",
            f"list of {nl} layer GEMVs (M=1, {k}x{n} w4 g128 bf16, orc_mpq_forward_list_f32acc: the reference's two roundings per weight, fp32 accumulate)",
            per_run_s=0.5, thread_sets=default_sets if k * n < 1e8 else few_sets)
    # ---- M = 4096 on the metric's layer: dequant only, GEMM only (sampled rows), both
    k = n = 4096
    qw, sc, ze = w4_layer(k, n)
    run("w4a16_dequant_4096x4096", lambda: orc.mpq_dequant(qw, sc, ze, None, WBIT, GROUP, 0, orc.BF16), k * n * WBIT // 8 + 2 * k * n, "GB/s",
        "unpack_qweight of one 4096x4096 layer (orc_mpq_dequant; bytes = packed in + bf16 out)", per_run_s=0.5, thread_sets=few_sets)
    W = orc.mpq_dequant(qw, sc, ze, None, WBIT, GROUP, 0, orc.BF16)
    for rows, tset in ((512, [big]), (32, [1])):
        xm = orc.torch_to_np(torch.randn((rows, k), generator=gen).to(BF16))
        run("w4a16_gemm_M4096_4096x4096", lambda: orc.gemm(xm, W, orc.BF16), 2.0 * rows * k * n, "GFLOP/s",
            f"{rows} of the 4096 rows of x . W (orc_gemm, bf16 in / fp32 accumulate; the dequant is the row above)", per_run_s=1.0, thread_sets=tset)
    # ---- exl2 3/2-bit decode: reconstruct + GEMV (what the reference's CPU-side restatement has to do per token)
    from bitorch_engine.layers.qlinear.nbit.cuda.utils import make_group_map  # noqa: F401  (host helper; no device work)
    K2 = N2 = 4096
    qg, row = [], 0
    for bts in (3, 2):
        for _ in range(K2 // 2 // 32):
            qg += [bts, row]
            row += bts
    q_groups = np.array(qg, dtype=np.int16)
    groups = len(qg) // 2
    qw2 = rng.integers(-2 ** 31, 2 ** 31 - 1, (row, N2), dtype=np.int64).astype(np.int32)
    sc2 = orc.torch_to_np((torch.rand((groups, N2), generator=gen) * 0.02 + 0.001).half())
    ze2 = orc.torch_to_np((torch.randn((groups, N2), generator=gen) * 0.05).half())
    perm2 = torch.randperm(K2, generator=gen).to(torch.short).numpy()
    x2 = orc.torch_to_np(torch.randn((1, K2), generator=gen).half())
    run("exl2_w3w2_decode_4096x4096", lambda: orc.gemm(x2, orc.exl2_dequant(qw2, sc2, ze2, perm2, q_groups, K2), orc.F16),
        row * N2 * 4 + 4 * groups * N2 + 8 * K2 + 2 * N2, "GB/s", "exl2 3/2-bit g32 layer: orc_exl2_dequant + orc_gemm (M=1)", per_run_s=0.6, thread_sets=few_sets)
    # ---- binary linear 4096x4096 (row-packed operands) and the ResNet-18 conv
    wb = rng.integers(0, 256, (4096, 512), dtype=np.int64).astype(np.uint8)
    for Mb in (1, 64):
        xb = rng.integers(0, 256, (Mb, 512), dtype=np.int64).astype(np.uint8)
        run(f"binary_linear_M{Mb}_4096x4096", lambda: orc.binary_linear_rowpacked(xb, wb, 4096), 2.0 * Mb * 4096 * 4096, "GOP/s",
            f"XNOR-popcount linear M={Mb} (orc_binary_linear_rowpacked)", per_run_s=0.4, thread_sets=few_sets)
    xc = torch.randn((4, 512, 7, 7), generator=gen).numpy()
    wc = torch.randn((512, 512, 3, 3), generator=gen).numpy()
    run("binary_conv_512x512x3x3_7x7_B4", lambda: orc.binary_conv2d(xc, wc, 1, 1, 1), 2.0 * 4 * 49 * 512 * 4608, "GOP/s",
        "binary conv 512->512 3x3 on 7x7, batch 4 (orc_binary_conv2d)", per_run_s=0.5, thread_sets=few_sets)
    if omp is not None:
        omp.omp_set_num_threads(cores)
    return res


def main():
    os.environ.setdefault("BIE_TUNING", "1")  # GEMM plan knobs re-read per launch (one getenv each; the timed regions are captured graphs)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only (used under rocprofv3)")
    ap.add_argument("--only", default="", help="profiling aid: run only this shape's M=1 decode pass, e.g. 4096x11008")
    ap.add_argument("--short", action="store_true", help="first-class rows only (decode step, per-layer launches, GEMM); skips the wide sweep written to gpurun_out/bench_extras.json")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"), help="collective back end of the N > 1 legs (nccl = RCCL; gloo only with --dry-run)")
    ap.add_argument("--dry-run", action="store_true", help="CPU rehearsal of the launcher / rendezvous / timed-region protocol under gloo: no kernel runs, value is null")
    ap.add_argument("--sequential-gather", action="store_true", help="N > 1: all-gather on the compute stream after every pass instead of overlapped under the next pass")
    args = ap.parse_args()

    import bench_sharded as bs
    # The driver's contract names two ways of starting N ranks: `python -m torch.distributed.run ... bench.py --gpus N` (WORLD_SIZE set: this
    # process IS a rank) and a bare `python bench.py --gpus N` (WORLD_SIZE unset: start the ranks here).  Round 5 parsed --gpus and never read it.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(bs.launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world wins; n_gpus reports {world}", file=sys.stderr, flush=True)
    if args.dry_run:
        return bs.dry_run(args, world, rank)
    if args.backend != "nccl":
        raise SystemExit("bench.py measures on GPUs over RCCL (--backend nccl); gloo is for --dry-run only")
    # BIE_BENCH_FORCE_DIST=1: a world of ONE goes through the whole multi-rank code path (RCCL group, overlapped gather, c5 leg) --
    # the only way to execute that code on a one-GPU box; its numbers are not a scaling result
    distributed = world > 1 or os.environ.get("BIE_BENCH_FORCE_DIST", "0") == "1"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible -- one process per GPU")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl = None
    if distributed:
        import torch.distributed as dist
        if "RANK" not in os.environ:  # forced world of one without a launcher
            os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_PORT=str(bs.free_port()))
        bs.init_world("nccl", dev)
        rccl = bs.world_proof(dev, world)

    B = Bench(dev)
    K, N, LAYERS = 4096, 4096, 96
    if args.only:
        K, N = [int(v) for v in args.only.split("x")]
        LAYERS = max(8, min(96, int(800e6 // (K * N // 2))))
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    layers = [make_layer(dev, gen, K, N) for _ in range(LAYERS)]
    # Single GPU: the K timed steps (K decode passes = K list launches) are ONE captured graph, replayed once -- a serving loop captures
    # a whole token step, not one launch, and a replay has a fixed cost of its own (10-16 us on this stack, MI355X_MICROARCH.md
    # "graph-replay-floor"; rocprofv3 shows 194 us of kernel inside a 210 us replay when every pass is its own replay).
    # Multi-GPU: C passes per replay (C = the largest divisor of K up to 10), pass c writing its own 96 rows of a [C x 96, N] bucket, then
    # ONE all-gather of the bucket on a communication stream under the next replay: fewer, larger collectives (7.9 MB per rank at C = 10
    # instead of ten of 786 KB), and RCCL's enqueue cost (tens of microseconds of host time) is paid once per C steps.
    chunk = max(c for c in range(1, 11) if args.steps % c == 0) if distributed else 1
    y_all = torch.empty((chunk * LAYERS, N), dtype=BF16, device=dev)  # row c * LAYERS + l = output of layer l in pass c of a replay
    plan = B.make_list(layers, K, N, gen, ys=[y_all[i:i + 1] for i in range(LAYERS)])  # every layer has its own x; ONE launch per pass
    plans = [plan] + [B.make_list(layers, K, N, gen, ys=[y_all[c * LAYERS + i:c * LAYERS + i + 1] for i in range(LAYERS)], xs=plan_x(plan)) for c in range(1, chunk)]
    passes_per_replay = chunk if distributed else args.steps
    graph = capture(lambda st: [plans[c % chunk].forward(st) for c in range(passes_per_replay)])
    gather = bs.OverlappedGather(y_all, world, sequential=args.sequential_gather) if distributed else None

    def step():
        graph.replay()
        if distributed:
            gather()

    barrier = bs.make_barrier(distributed, True)

    def timed_region():
        """EXACTLY K passes between two barriers: (wall seconds, max over ranks; GPU milliseconds between the events on this rank)."""
        return bs.timed_region(step, args.steps, barrier, distributed, dev, replays=args.steps // chunk if distributed else 1)

    for _ in range(-(-args.warmup // chunk) if distributed else max(1, -(-args.warmup // args.steps))):  # at least W warm-up passes
        step()
    # The W warm-up passes last about a millisecond and the K timed ones a few: timed right here the region runs at the clock of a
    # chip that was idle through the host-side set-up.  That cold figure is reported (`cold_start`), then the same passes are replayed
    # untimed for BIE_BENCH_PRECONDITION_S (0.3 s) and the K passes are timed again: `value` is the sustained rate.
    cold_elapsed, cold_gpu_ms = timed_region()
    pre_s = float(os.environ.get("BIE_BENCH_PRECONDITION_S", "0.3"))
    pre_n = 0
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < pre_s and not distributed:
        step()
        pre_n += 1
        if pre_n % 4 == 0:
            torch.cuda.synchronize()
    if distributed:  # every rank the same number of collectives
        pre_n = max(1, int(pre_s / max(cold_elapsed, 1e-4)))
        for _ in range(pre_n * (args.steps // chunk)):
            step()
    # not one lucky region -- BIE_BENCH_REGIONS (5) timed regions of exactly K passes each, back to back; `value` is the
    # MEDIAN region (wall, max over ranks), the fastest and slowest ones and the cold one are reported beside it
    n_regions = max(1, int(os.environ.get("BIE_BENCH_REGIONS", "5")))
    regions = sorted((timed_region() for _ in range(n_regions)), key=lambda r: r[0])
    elapsed, gpu_ms = regions[len(regions) // 2]

    ms_per_step = elapsed / args.steps * 1e3
    step_bytes = alg_bytes(1, K, N) * LAYERS
    value = step_bytes * world / (elapsed / args.steps) / 1e9

    # ---- self-check of the TIMED launch: every row of y_all against the per-layer entry point on the same layer and x
    verified = None
    if rank == 0:
        verified = all(verify_list_outputs(B, plans[c], layers, y_all[c * LAYERS:(c + 1) * LAYERS], K, N) for c in sorted({0, chunk - 1}))

    out, extras = None, {}
    if rank == 0:
        launches = args.steps  # one list launch per pass
        avg_us = gpu_ms * 1e3 / launches
        achieved = alg_bytes(1, K, N) * LAYERS / (avg_us * 1e-6) / 1e9
        out = {
            "metric": f"W4A16 decode GEMV weight-streaming throughput (M=1, {K}x{N} g128, bf16)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (seeded random packed weights / scales / zeros, N(0,1) activations)",
            "config": {"workload": f"BASELINE.json metric config: W4A16 qlinear {K}x{N} g128 bf16, M=1 decode pass over {LAYERS} distinct layers "
                                   f"({LAYERS * K * N // 2 / 1e9:.2f} GB packed), one layer-list launch per pass, K passes in one HIP graph",
                       "layers_per_step": LAYERS, "launches_per_step": 1,
                       "parallelism": ("%d ranks (one per GPU), each its own %d layers (weak scaling: independent output-column blocks), "
                                       "1 bucketed RCCL all-gather of the outputs of %d steps, %s") % (world, LAYERS, chunk, "sequential" if args.sequential_gather else "overlapped under the next replay")
                       if distributed else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": (lambda t: None if t is None else t * LAYERS)(pmc_traffic(f"list{LAYERS}_{K}x{N}")),
                         "kernel": "bie::mpq_list_kernel<bf16,sym,M=1,rpg=16,w4,D16> (16-bit pair tables, ds_read_u16_d16_hi lookups, v_pk_fma_f32)", "avg_launch_us": round(avg_us, 3), "us_per_layer": round(avg_us / LAYERS, 3),
                         "alg_bytes_per_launch": alg_bytes(1, K, N) * LAYERS},
            # the same K passes timed straight after the W warm-up passes, before the untimed replays that bring the chip to its sustained clock
            "cold_start": {"value": round(step_bytes * world / (cold_elapsed / args.steps) / 1e9, 2), "unit": "GB/s", "ms_per_step": round(cold_elapsed / args.steps * 1e3, 4),
                           "roofline_frac": round(alg_bytes(1, K, N) * LAYERS / (cold_gpu_ms * 1e3 / launches * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                           "untimed_seconds_before_value": pre_s},
            "regions": {"n": n_regions, "value_is": "median region of K passes (wall clock, max over ranks)",
                        "best_GBps": round(step_bytes * world / (regions[0][0] / args.steps) / 1e9, 2),
                        "worst_GBps": round(step_bytes * world / (regions[-1][0] / args.steps) / 1e9, 2),
                        "kernel_frac_min_median_max": [round(alg_bytes(1, K, N) * LAYERS / (r[1] * 1e3 / launches * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                                                       for r in (max(regions, key=lambda r: r[1]), sorted(regions, key=lambda r: r[1])[len(regions) // 2], min(regions, key=lambda r: r[1]))]},
            "verified": verified,
        }

    run_extras = rank == 0 and not distributed and not args.no_extras and not args.only
    if run_extras:
        def guarded(key, fn):
            try:
                extras[key] = fn()
            except Exception as e:  # reporting only: never fail the headline for an extra
                extras[key] = {"error": str(e)[:300]}
            print(f"[bench extra] {key}: {json.dumps(extras[key])[:600]}", file=sys.stderr, flush=True)

        def frac(key):
            v = extras.get(key)
            if isinstance(v, dict) and "roofline" in v:
                r = v["roofline"]
                return {"frac": r["frac"], "us": v.get("us_per_layer", v.get("us_per_launch"))}
            return v if v is None else {"error": str(v.get("error", "?"))[:80]}
        # ---- first-class rows: the real decode step, the drop-in per-layer launches, dependent chains, the prefill GEMM
        guarded("decode_step_llama7b", lambda: B.decode_step(32, 5, 77))
        guarded("decode_step_modules_auto_grouped", lambda: B.decode_step_modules(16, 5, 78, auto_group=True))
        guarded("decode_step_modules_one_launch_per_layer", lambda: B.decode_step_modules(16, 5, 78, auto_group=False))
        # a decode batch of 24 sequences through the same unchanged module tree (round 6: sibling sets and lone calls of 17 .. 32 rows keep the decode kernels where they measured ahead)
        guarded("decode_step_modules_24rows_f16", lambda: B.decode_step_modules(8, 5, 79, auto_group=True, rows=24, dt=torch.float16))
        guarded("decode_step_modules_24rows_bf16", lambda: B.decode_step_modules(8, 5, 79, auto_group=True, rows=24, dt=BF16))
        guarded("gemm", lambda: B.gemm(4096, 4096, 4096, 24, 3, 7))
        if "roofline" in extras.get("gemm", {}):
            out["roofline_gemm"] = dict(extras["gemm"]["roofline"], kernel=("bie::mpq_dequant_frag_kernel + bie::mpq_dense_gemm_kernel<bf16,256x256> (both launches timed)"
                                                if os.environ.get("BIE_GEMM_DENSE", "1") != "0" else "bie::mpq_gemm_kernel<bf16,w4,256x256x64>"),
                                        us_per_launch=extras["gemm"]["us_per_launch"], M=4096, K=4096, N=4096)
            pg = pmc_gemm()
            if pg:
                out["roofline_gemm"]["pmc"] = pg  # the counters of THIS shape (4096^3): fetch = 3.0 x the operands = 1.00 x what 8 private L2s must fetch with 256^2 tiles
                out["roofline_gemm"]["traffic"] = pg["fetch_bytes"] + pg["write_bytes"]  # HBM-side bytes of the GEMM kernel per launch (the dequantise pass adds 8 MB read + 32 MB written)
        guarded("per_layer_launches_4096x4096", lambda: B.gemv(4096, 4096, 96, 10, 1))
        guarded("chain8_4096x4096_launches", lambda: B.chain_launches(4096, 96, 8, 10, 5))
        guarded("c2_gemv_4096x11008", lambda: B.gemv(4096, 11008, 40, 10, 11))
        guarded("c2_gemv_11008x4096", lambda: B.gemv(11008, 4096, 40, 10, 12))
        guarded("c5_gemv_8192x28672", lambda: B.gemv(8192, 28672, 6, 10, 32))
        guarded("grouped_qkv_3x4096x4096", lambda: B.grouped(4096, (4096, 4096, 4096), 32, 10, 21, "q/k/v projections in one launch"))
        guarded("grouped_gate_up_2x4096x11008", lambda: B.grouped(4096, (11008, 11008), 20, 10, 22, "gate/up projections in one launch"))
        guarded("c2_gemm_4096x11008", lambda: B.gemm(4096, 4096, 11008, 16, 3, 14))
        guarded("c2_gemm_11008x4096", lambda: B.gemm(4096, 11008, 4096, 16, 3, 15))
        out["decode_step_llama7b"] = (lambda v: {k: v[k] for k in ("us_per_layer", "launches_per_layer", "layers", "finite") if k in v} | {"roofline_frac": v["roofline"]["frac"]}
                                      if isinstance(v, dict) and "roofline" in v else v)(extras.get("decode_step_llama7b"))
        out["summary"] = {k: frac(k) for k in ("per_layer_launches_4096x4096", "chain8_4096x4096_launches", "c2_gemv_4096x11008", "c2_gemv_11008x4096", "c5_gemv_8192x28672",
                                               "grouped_qkv_3x4096x4096", "grouped_gate_up_2x4096x11008", "c2_gemm_4096x11008", "c2_gemm_11008x4096", "decode_step_modules_auto_grouped",
                                               "decode_step_modules_one_launch_per_layer", "decode_step_modules_24rows_f16", "decode_step_modules_24rows_bf16")}
        if not args.short:
            # the fused form (dequantisation beside the MFMAs) on the same layers, same box: the A/B behind the dense form's dispatch rule
            def fused_gemm():
                old = {k: os.environ.get(k) for k in ("BIE_TUNING", "BIE_GEMM_DENSE")}
                os.environ["BIE_TUNING"], os.environ["BIE_GEMM_DENSE"] = "1", "0"  # BIE_TUNING: the knob is re-read per launch
                try:
                    return dict(B.gemm(4096, 4096, 4096, 24, 3, 7), kernel="bie::mpq_gemm_kernel<bf16,w4,256x256x64 tile> (BIE_GEMM_DENSE=0)")
                finally:
                    for k, v in old.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
            if os.environ.get("BIE_TUNING"):
                guarded("gemm_fused_form_4096x4096", fused_gemm)
            guarded("list_4x24_4096x4096", lambda: B.gemv_list(4096, 4096, 96, 24, 10, 2))
            guarded("chain4_4096x4096", lambda: B.gemv_list(4096, 4096, 96, 4, 10, 3, chain=4))
            guarded("chain8_in_list32_4096x4096", lambda: B.gemv_list(4096, 4096, 96, 32, 10, 4, chain=8))
            guarded("c2_list_4096x11008", lambda: B.gemv_list(4096, 11008, 40, 40, 10, 11, key="list40_4096x11008"))
            guarded("c2_list_11008x4096", lambda: B.gemv_list(11008, 4096, 40, 40, 10, 12, key="list40_11008x4096"))
            guarded("c2_gemv_M2_4096x11008", lambda: B.gemv(4096, 11008, 40, 10, 13, M=2))
            guarded("c2_gemv_M8_4096x11008", lambda: B.gemv(4096, 11008, 40, 10, 16, M=8))
            guarded("c2_gemv_M16_4096x11008", lambda: B.gemv(4096, 11008, 40, 10, 17, M=16))
            guarded("c2_list_M2_4096x11008", lambda: B.gemv_list(4096, 11008, 40, 40, 10, 18, M=2))
            guarded("c2_list_M8_4096x11008", lambda: B.gemv_list(4096, 11008, 40, 40, 10, 19, M=8))
            guarded("c2_list_M16_4096x11008", lambda: B.gemv_list(4096, 11008, 40, 40, 10, 20, M=16))
            guarded("c2_list_M32_4096x11008", lambda: B.gemv_list(4096, 11008, 40, 40, 10, 21, M=32))
            guarded("c2_gemv_M32_4096x11008", lambda: B.gemv(4096, 11008, 40, 10, 22, M=32))
            # the same batched lists in fp16 (the reference kernels' own dtype): arithmetic dequantisation, x shared by the workgroup's four column tiles
            for m_ in (2, 8, 16, 32):
                guarded(f"f16_list_M{m_}_4096x11008", lambda m_=m_: B.gemv_list(4096, 11008, 40, 40, 10, 30 + m_, M=m_, dt=torch.float16))
            guarded("c2_act_order_4096x11008", lambda: bench_act_order(dev))
            guarded("c3_exl2", lambda: bench_exl2(dev))
            guarded("c3_exl2_decode_step_llama7b", lambda: bench_exl2_decode_step(dev))
            if isinstance(extras.get("c3_exl2_decode_step_llama7b"), dict) and "roofline" in extras["c3_exl2_decode_step_llama7b"]:
                out["summary"]["c3_exl2_decode_step_llama7b"] = {"frac": extras["c3_exl2_decode_step_llama7b"]["roofline"]["frac"], "us": extras["c3_exl2_decode_step_llama7b"]["us_per_layer"]}
            if isinstance(extras.get("c3_exl2"), list):  # configs[2] in the short line: the list form and the sibling groups, fraction of the HBM roofline
                for r_ in extras["c3_exl2"]:
                    if ("layer list" in r_["op"] and r_["M"] == 1) or "siblings" in r_["op"]:
                        out["summary"][("c3_exl2_list_" if "layer list" in r_["op"] else f"c3_exl2_group{r_.get('members')}_") + f"{r_['K']}x{r_['N']}"] = r_["roofline"]["frac"]
            # fp16 (GreenBit's checkpoint dtype): the exact 16-bit-table form (default) and the opt-in algebraic form (BIE_LIST_ALG=1: exact products,
            # 6-8e-4 of max|y| away from the reference's doubly rounded weights -- DESIGN.md section 2)
            guarded("f16_list_4096x4096", lambda: B.gemv_list(4096, 4096, 96, 96, 10, 45, dt=torch.float16))
            guarded("f16_list_4096x11008", lambda: B.gemv_list(4096, 11008, 40, 40, 10, 46, dt=torch.float16))

            def alg_list(k, n, nl, seed):
                old = os.environ.get("BIE_LIST_ALG")
                os.environ["BIE_LIST_ALG"] = "1"
                try:
                    return dict(B.gemv_list(k, n, nl, nl, 10, seed, dt=torch.float16), form="algebraic (opt-in, BIE_LIST_ALG=1)")
                finally:
                    if old is None:
                        os.environ.pop("BIE_LIST_ALG", None)
                    else:
                        os.environ["BIE_LIST_ALG"] = old
            if os.environ.get("BIE_TUNING"):
                guarded("f16_list_alg_4096x4096", lambda: alg_list(4096, 4096, 96, 45))
                guarded("f16_list_alg_4096x11008", lambda: alg_list(4096, 11008, 40, 46))
            for k_ in ("f16_list_4096x4096", "f16_list_alg_4096x4096", "f16_list_4096x11008", "f16_list_alg_4096x11008"):
                out["summary"][k_] = frac(k_)
            guarded("c3_w2a16_list_4096x4096", lambda: B.gemv_list(4096, 4096, 96, 96, 10, 44, w_bit=2))
            guarded("c3_w2a16_4096x4096", lambda: B.gemv(4096, 4096, 64, 10, 41, w_bit=2))
            guarded("c3_w2a16_4096x11008", lambda: B.gemv(4096, 11008, 40, 10, 42, w_bit=2))
            guarded("c3_w2a16_11008x4096", lambda: B.gemv(11008, 4096, 40, 10, 43, w_bit=2))
            guarded("c4_binary", lambda: bench_binary(dev, B.L))
            guarded("f1_int_gemm", lambda: bench_int_gemm(dev, B.L))
            guarded("c5_single_gpu_8192x28672", lambda: B.gemm(4096, 8192, 28672, 2, 3, 31))
            guarded("c5_list_8192x28672", lambda: B.gemv_list(8192, 28672, 6, 6, 10, 32))
            guarded("c5_grouped_gate_up_2x8192x28672", lambda: B.grouped(8192, (28672, 28672), 3, 10, 33, "70B-class gate/up projections in one launch"))
    if rank == 0:
        out["kernel_source_sha"] = kernel_source_sha()

    # ---- configs[4] sharded: 8192 x 28672, M = 4096, N / world column shards + all-gather (separate and overlapped)
    if distributed and not args.only:
        try:  # reporting only: the headline line must come out whatever these legs do (they have run under gloo on CPU and under a world of one)
            rccl = bs.rccl_report(rccl, graph.replay, y_all, world, dev, passes_per_step=chunk)
        except Exception as e:
            rccl = dict(rccl or {}, error=str(e)[:300])
        try:
            c5 = bs.bench_column_sharded(world, rank, dev, M=4096, K=8192, N=28672, reps=5)
        except Exception as e:
            c5 = {"error": str(e)[:300]}
        if rank == 0:
            extras["c5"] = c5
            out["rccl"] = rccl
            out["c5"] = {k: c5[k] for k in list(c5)[:14]} if isinstance(c5, dict) else c5

    if rank == 0 and not distributed and not args.no_cpu_baseline and not args.only:
        try:
            bl = cpu_baselines(layers[:2], plan_x(plan)[:2], y_all[:2])  # rows of pass 0
            head = [b for b in bl if b["workload"] == "w4a16_gemv_M1_4096x4096"]
            best = dict(max(head, key=lambda b: b["value"]))  # the headline shape at its best thread count ...
            best["threads_used"] = best["cores"]
            best["nproc"] = os.cpu_count()
            best["cores_granted"] = max(b["cores"] for b in head)  # min(affinity, cgroup CPU quota): "all host cores" as far as this container gets them
            best["by_threads"] = {str(b["cores"]): b["value"] for b in head}  # ... with all cores, 32, 8 and one thread beside it
            out["cpu_baseline"] = best
            out["verified_vs_oracle"] = next((b["verified"] for b in bl if b["workload"] == "oracle_check_of_timed_launch"), None)
            extras["cpu_baselines"] = bl            # every workload of SURVEY.md section 8d at {all, 32, 8, 1} threads (bounded samples)
        except Exception as e:  # the baseline is reporting only; never fail the bench for it
            out["cpu_baseline"] = {"value": None, "unit": "GB/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {str(e)[:200]}"}
    elif rank == 0:
        out["cpu_baseline"] = None

    if distributed:
        # RCCL prints a version banner through C stdio, which is flushed when a process exits -- i.e. AFTER rank 0's JSON line (seen on the
        # world-of-one run: the line was not the last one on stdout).  Every rank flushes its C buffers, then the ranks meet, then rank 0 prints.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # everything beyond the short line: one JSON file under gpurun_out/ (and stderr above, as it is produced)
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_extras.json"), "w") as f:
                json.dump({"headline": out, "extras": extras}, f, indent=1)
            out["extras_file"] = "gpurun_out/bench_extras.json"
        except OSError:
            pass
        line = json.dumps(out)
        if len(line) > 6000:  # the driver's parser gave up on a 26.8 KB line in round 3: never again
            for k in ("summary", "c5", "decode_step_llama7b"):
                if len(line) > 6000 and k in out:
                    out[k] = "see extras_file"
                    line = json.dumps(out)
        print(line, flush=True)


if __name__ == "__main__":
    main()
