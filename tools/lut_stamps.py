"""Tuning aid: per-wave timeline of ONE table-lookup GEMV launch (BIE_GEMV_LAB=5 build variant): when each wave started, when its
weights had landed, when its lookups were done.  wall_clock64 = 100 MHz (10 ns)."""
import ctypes, os, sys
os.environ["BIE_GEMV_LAB"] = "5"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import numpy as np, torch
from bitorch_engine import _hip
L = _hip.lib()
raw = ctypes.CDLL(_hip.LIB_PATH)
dev = torch.device("cuda:0")
K, N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 11008
gen = torch.Generator().manual_seed(0)
layers = []
for _ in range(14):
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), dtype=torch.int64, generator=gen).to(torch.int32).to(dev)
    sc = (torch.rand((K // 128, N), generator=gen) * 0.01 + 0.005).to(torch.bfloat16).to(dev)
    ze = (sc.float().cpu() * torch.rand((K // 128, N), generator=gen) * 15).to(torch.bfloat16).to(dev)
    layers.append((qw, sc, ze))
x = torch.randn((1, K), generator=gen).to(torch.bfloat16).to(dev)
y = torch.empty((1, N), dtype=torch.bfloat16, device=dev)
ws = torch.zeros(L.bie_mpq_workspace_bytes(1, K, N, 4), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
for (qw, sc, ze) in layers:  # the last launch is the one whose stamps survive: cold weights, warm code
    rc = L.bie_mpq_forward(x.data_ptr(), qw.data_ptr(), sc.data_ptr(), ze.data_ptr(), None, None, y.data_ptr(), ws.data_ptr(), ws.numel(), 1, K, N, 4, 128, 0, _hip.BF16, st)
    assert rc == 0
torch.cuda.synchronize()
NWv = int(os.environ.get("BIE_LUT_NW", "8"))
G = K // 128
tiles = (N + 63) // 64
S = (G + NWv - 1) // NWv if os.environ.get("BIE_LUT_COOP", "0") == "0" else (G + 7) // 8
nw = tiles * S * NWv
buf = np.zeros(nw * 5, dtype=np.uint64)
raw.bie_debug_lut_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert raw.bie_debug_lut_stamps(buf.ctypes.data, nw) == 0
b = buf.reshape(nw, 5).astype(np.int64)
t0 = b[:, 0].min()
start, landed, done = (b[:, 0] - t0) / 100.0, (b[:, 1] - t0) / 100.0, (b[:, 2] - t0) / 100.0
wg_end = (b[::NWv, 3] - t0) / 100.0
pct = lambda a: " ".join(f"{np.percentile(a, p):6.2f}" for p in (0, 10, 25, 50, 75, 90, 100))
print(f"K={K} N={N}: {nw} waves, NW={NWv}, S={S}   (us since the first wave started; percentiles 0 10 25 50 75 90 100)")
print("wave start      ", pct(start))
print("weights landed  ", pct(landed))
print("lookups done    ", pct(done))
print("compute duration", pct(done - landed))
print("load latency    ", pct(landed - start))
print("workgroup end   ", pct(wg_end))
hw = b[:, 4]
cu = ((hw >> 32) & 0xf) * 1000 + (hw & 0xffffffff >> 0 & 0xf00) // 256 * 16 + ((hw >> 13) & 0x7) * 0  # coarse: xcc + cu_id bits
print("distinct (xcc, hw_id[11:8]) values:", len(np.unique(((hw >> 32) << 8) | ((hw >> 8) & 0xf))))
