#!/bin/bash
# round 6, pass V: the dense form from 897 rows -- parity suite slice + the band's timing
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "gemm or prefill or dense or full_size or fuzz or randomised or workspace" 2>&1 | tail -4
python - <<'PY'
import os, sys, json
sys.path.insert(0, "bitorch-engine_amd"); sys.path.insert(0, ".")
import torch
from bench import Bench
B = Bench(torch.device("cuda", 0))
for (k, n) in ((4096, 4096), (4096, 11008), (11008, 4096), (4096, 14336), (8192, 8192)):
    for M in (896, 897, 960, 1023, 1024):
        print(json.dumps({"K": k, "N": n, "M": M, "us": B.gemm(M, k, n, 8, 8, 5)["us_per_launch"]}), flush=True)
PY
