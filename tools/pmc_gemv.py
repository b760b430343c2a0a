"""PMC workload: M=1 GEMV over 26 distinct 4096x11008 W4 g128 layers (no graph, so every launch is one dispatch row) plus
the dequant kernel on the same layers as a calibration point for FETCH_SIZE/WRITE_SIZE (known bytes: reads K*N/2 + 4*G*N,
writes 2*K*N, qweight read with the same dword-per-lane row-coalesced pattern as the GEMV)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import sweep
from bitorch_engine import _hip
L = _hip.lib()
dt = _hip.BF16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else _hip.F16
KK, NN = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4096, 11008)
MM = int(sys.argv[4]) if len(sys.argv) > 4 else 1   # optional: rows of x
WB = int(sys.argv[5]) if len(sys.argv) > 5 else 4   # optional: weight bits
print(sweep.time_case(MM, KK, NN, dt, layers=26, reps=3, graph=False, w_bit=WB))
if len(sys.argv) > 3:
    sys.exit(0)
K, N, gs = 4096, 11008, 128
tdt = torch.bfloat16 if dt == _hip.BF16 else torch.float16
gen = torch.Generator().manual_seed(1)
out = torch.empty((K, N), dtype=tdt, device="cuda:0")
for i in range(6):
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), dtype=torch.int64, generator=gen).to(torch.int32).cuda()
    sc = torch.rand((K // gs, N), generator=gen).to(tdt).cuda()
    rc = L.bie_mpq_dequant(qw.data_ptr(), sc.data_ptr(), sc.data_ptr(), None, out.data_ptr(), K, N, 4, gs, 0, dt, None)
    assert rc == 0, L.bie_last_error()
torch.cuda.synchronize()
