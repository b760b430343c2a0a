"""Per-wave timeline of ONE lone decode launch (bie_mpq_forward_grouped, inline form of the list kernel) from the LAB build's stamps:
   make -C bitorch-engine_amd lab && BIE_HIP_LIB=$PWD/bitorch-engine_amd/lab/libbie_hip.so python tools/inl_timeline.py K N [N2 ...]
us since the launch's first wave started; percentiles 0 10 25 50 75 90 100."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import numpy as np, torch
from bitorch_engine import _hip
L = _hip.lib()
raw = ctypes.CDLL(_hip.LIB_PATH)
dev = torch.device("cuda:0")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Ns = [int(v) for v in sys.argv[2:]] or [4096]
gen = torch.Generator(device=dev).manual_seed(0)
sets = []
for _ in range(10):
    grp = []
    for N in Ns:
        qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), dtype=torch.int32, generator=gen, device=dev)
        sc = (torch.rand((K // 128, N), generator=gen, device=dev) * 0.01 + 0.005).to(torch.bfloat16)
        ze = (sc.float() * torch.rand((K // 128, N), generator=gen, device=dev) * 15).to(torch.bfloat16)
        grp.append((qw, sc, ze))
    sets.append(grp)
x = torch.randn((1, K), generator=gen, device=dev).to(torch.bfloat16)
ys = [torch.empty((1, N), dtype=torch.bfloat16, device=dev) for N in Ns]
n = len(Ns)
Narr = (ctypes.c_int * n)(*Ns)
ws = torch.zeros(max(L.bie_mpq_grouped_workspace_bytes(n, Narr, 1, K, 4), 1 << 20), dtype=torch.uint8, device=dev)
arr = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
st = torch.cuda.current_stream().cuda_stream
for grp in sets:  # the last launch's stamps survive: cold weights, warm code
    rc = L.bie_mpq_forward_grouped(x.data_ptr(), n, arr([g[0] for g in grp]), arr([g[1] for g in grp]), arr([g[2] for g in grp]), None, arr(ys), Narr,
                                   ws.data_ptr(), ws.numel(), 1, K, 4, 128, 0, _hip.BF16, st)
    assert rc == 0, L.bie_last_error()
torch.cuda.synchronize()
NW = 4
nw = 65536
buf = (ctypes.c_ulonglong * (nw * 6))()
assert raw.bie_debug_list_stamps(buf, nw) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 6).astype(np.int64)
wid_all = np.arange(nw)
keep0 = a[:, 0] > 0
wid_all, a = wid_all[keep0], a[keep0]
# keep the waves of the LAST launch: stamps within 100 us of the newest start
keep1 = a[:, 0] > a[:, 0].max() - 10000
wid_all, a = wid_all[keep1], a[keep1]
t0 = a[:, 0].min()
tick = 1e-2
start, landed, done = (a[:, 0] - t0) * tick, (a[:, 1] - t0) * tick, (a[:, 2] - t0) * tick
wg = a[a[:, 3] > 0]
wg_end = (wg[:, 3] - t0) * tick
wg_end = wg_end[(wg_end > 0) & (wg_end < 100)]
pct = lambda v: " ".join(f"{np.percentile(v, p):6.2f}" for p in (0, 10, 25, 50, 75, 90, 100))
print(f"K={K} N={Ns}: {len(a)} waves stamped   env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("BIE_INL")))
print("wave start      ", pct(start))
print("rows landed     ", pct(landed))
print("lookups done    ", pct(done))
print("compute duration", pct(done - landed))
print("workgroup end   ", pct(wg_end))

# phases: ramp = launch start -> the first rows have landed; stream = first -> last rows landed; tail = last rows landed -> last lookups done;
# reduce = last lookups done -> last workgroup end.  Reducers are the workgroups of the LAST K slice (block id / tiles): their wave 0 ends the tile.
NWv = int(os.environ.get("BIE_INL_NW", "4"))
tiles = sum((N + 63) // 64 for N in Ns)
blk = wid_all // NWv
S = int(blk.max()) // tiles + 1
sl = blk // tiles
print(f"plan: {tiles} column tiles x {S} K slices = {tiles * S} workgroups of {NWv} waves; bytes {sum(K * N // 2 for N in Ns) / 1e6:.1f} MB")
print(f"phases (us): ramp {landed.min():.2f} | stream {landed.max() - landed.min():.2f} (last rows landed at {landed.max():.2f}) | tail {done.max() - landed.max():.2f} "
      f"(last lookups done at {done.max():.2f}) | reduce + store {wg_end.max() - done.max():.2f} (last workgroup end {wg_end.max():.2f})")
w0 = (wid_all % NWv) == 0
red = w0 & (sl == S - 1) & (a[:, 3] > 0)
pub = w0 & (sl < S - 1) & (a[:, 3] > 0)
if red.any():
    print("reducers   : own lookups done -> tile written", pct(((a[red, 3] - a[red, 2]) * tick)))
if pub.any():
    print("publishers : own lookups done -> granules out ", pct(((a[pub, 3] - a[pub, 2]) * tick)))
for lo in range(0, int(np.ceil(done.max())) + 1, 2):
    m = lambda v: int(((v >= lo) & (v < lo + 2)).sum())
    print(f"  t in [{lo:2d},{lo + 2:2d}) us: waves started {m(start):5d}  rows landed {m(landed):5d}  lookups done {m(done):5d}")
