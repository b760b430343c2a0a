#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for mf in 0 1; do
echo "== M sweep BIE_LUT_MFMA=$mf"; BIE_LUT_MFMA=$mf timeout 300 python - <<'PY'
import sys, os, json
sys.path.insert(0, "tools"); sys.path.insert(0, "bitorch-engine_amd")
from sweep import time_case
from bitorch_engine import _hip
for (K, N) in ((4096, 11008), (4096, 4096)):
    for M in (1, 2, 3, 4, 8, 12, 16, 17):
        r = time_case(M, K, N, _hip.BF16); print(M, K, N, r["us"], flush=True)
for M in (1, 4, 16):
    r = time_case(M, 4096, 11008, _hip.F16); print("f16", M, r["us"], flush=True)
PY
done
