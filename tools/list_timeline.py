#!/usr/bin/env python3
"""Per-wave timeline of ONE layer-list launch (96 x 4096x4096 W4 g128 bf16, M = 1) from the LAB build's stamps:
   make -C bitorch-engine_amd lab && BIE_HIP_LIB=$PWD/bitorch-engine_amd/lab/libbie_hip.so python tools/list_timeline.py
Prints, for a few CUs, the workgroups they ran in start order -- layer, tile, [start, rows landed, lookups done] in us from the launch's
first wave -- and chip-wide how many waves were waiting for rows / looking up at sampled instants: the load phase of one workgroup
runs under the lookup phase of its CU neighbours, which is what a per-layer launch cannot do."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd")); sys.path.insert(0, ROOT)
import torch
from bench import Bench, make_layer, BF16, GROUP
from bitorch_engine import _hip

dev = torch.device("cuda", 0)
B = Bench(dev)
gen = torch.Generator(device=dev).manual_seed(1)
L, K, N = 96, 4096, 4096
layers = [make_layer(dev, gen, K, N) for _ in range(L)]
plan = B.make_list(layers, K, N, gen)
for _ in range(3):
    plan.forward()
torch.cuda.synchronize()
raw = ctypes.CDLL(_hip.LIB_PATH)
NWAVES = 6144 * 4
buf = (ctypes.c_ulonglong * (NWAVES * 6))()
assert raw.bie_debug_list_stamps(buf, NWAVES) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(NWAVES, 6).astype(np.int64)
t0 = a[:, 0].min()
tick = 1e-2  # wall_clock64: 100 MHz -> 0.01 us
start, landed, done = (a[:, 0] - t0) * tick, (a[:, 1] - t0) * tick, (a[:, 2] - t0) * tick
entry, tile = a[:, 5] >> 32, a[:, 5] & 0xffffffff
xcc, hwid = a[:, 4] >> 32, a[:, 4] & 0xffffffff
cu = (xcc << 8) | ((hwid >> 8) & 0xff)  # xcc, (se, sh, cu) bits of HW_ID
print(f"launch: {NWAVES} waves, first start 0.0 us, last lookup done {done.max():.1f} us; per wave: wait for rows {np.median(landed - start):.2f} us (median), "
      f"lookups {np.median(done - landed):.2f} us (median)")
for t in np.linspace(done.max() * 0.1, done.max() * 0.9, 9):
    waiting = int(((start <= t) & (landed > t)).sum()); computing = int(((landed <= t) & (done > t)).sum())
    print(f"  t = {t:6.1f} us: {waiting:5d} waves waiting for their rows, {computing:5d} waves looking up / multiplying, layers in flight {entry[(start <= t) & (done > t)].min()}..{entry[(start <= t) & (done > t)].max()}")
for c in list(dict.fromkeys(cu.tolist()))[:2]:
    idx = np.where(cu == c)[0]
    idx = idx[np.argsort(start[idx])][:24]
    print(f"CU {c:#06x} (first 24 waves by start): layer/tile [start, rows landed, lookups done] us")
    print("   " + "  ".join(f"L{entry[i]}/t{tile[i]} [{start[i]:.1f} {landed[i]:.1f} {done[i]:.1f}]" for i in idx))
