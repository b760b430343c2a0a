import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import sweep
from bitorch_engine import _hip
for K in (4096, 4160, 4224, 8192, 8320):
    print(sweep.time_case(4096, K, 11008, _hip.BF16, layers=2, reps=3, graph=False, gs=64), flush=True)
