import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import sweep
from bitorch_engine import _hip
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dt = _hip.BF16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else _hip.F16
K = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
N = int(sys.argv[4]) if len(sys.argv) > 4 else 11008
print(sweep.time_case(M, K, N, dt, layers=2, reps=3, graph=False))
