"""Per-workgroup timeline of ONE launch of the one-launch binary conv (lab build: tools/build_variant.sh convlab binary_conv_fused.hip -DBIE_CONV_LAB,
BIE_HIP_LIB=.../variants/convlab/libbie_hip.so).  wall_clock64 = 100 MHz (10 ns)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
import numpy as np, torch
from bitorch_engine import _hip
from bitorch_engine.extensions import _binary_common as bc
os.environ.setdefault("BIE_CONV_FUSED_MAX_ROWS", "100000000")
raw = ctypes.CDLL(_hip.LIB_PATH)
raw.bie_debug_conv_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = torch.device("cuda", 0)
for B in [int(v) for v in (sys.argv[1:] or ["1", "32", "128"])]:
    x = torch.randn((B, 512, 7, 7), device=dev)
    w = torch.randn((512, 512, 3, 3), device=dev)
    wp = bc.pack_rows(w.reshape(512, -1)).contiguous()
    for _ in range(int(os.environ.get('CONV_TL_CALLS', '5'))):
        y = bc.conv2d(x, wp, 512, 3, 1, 1, 1, 1.0)
    torch.cuda.synchronize()
    n = 8192
    buf = np.zeros(n * 8, dtype=np.uint64)
    assert raw.bie_debug_conv_stamps(buf.ctypes.data, n) == 0
    b = buf.reshape(n, 8).astype(np.int64)
    b = b[b[:, 0] > 0]
    t0 = b[:, 0].min()
    names = ["start", "weights requested", "image packed", "popcounts done", "sums complete", "end"]
    pct = lambda a: " ".join(f"{np.percentile(a, p):7.2f}" for p in (0, 10, 50, 90, 100))
    print(f"B={B}: {len(b)} workgroups stamped (us since the first workgroup started; percentiles 0 10 50 90 100)")
    for i, nm in enumerate(names):
        print(f"  {nm:18s}", pct((b[:, i] - t0) / 100.0))
    dc, dw = (b[:, 7] - b[:, 6]).astype(float), (b[:, 3] - b[:, 2]).astype(float) * 10.0  # shader cycles / nanoseconds of the contraction phase
    print("  shader clock during the contraction phase (GHz):", pct(dc / np.maximum(dw, 1.0)))
    for i in range(1, 6):
        print(f"  phase {names[i - 1]} -> {names[i]}:", pct((b[:, i] - b[:, i - 1]) / 100.0))
