#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider  > gpurun_out/pytest_x.log 2>&1; tail -2 gpurun_out/pytest_x.log; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_x.log | head -8; grep -E "^E  " gpurun_out/pytest_x.log | head -8
for mf in "2 16"; do
set -- $mf
echo "== M sweep BIE_LUT_MFMA_MIN_M=$1 MAX_M=$2"; BIE_LUT_MFMA_MIN_M=$1 BIE_LUT_MFMA_MAX_M=$2 timeout 300 python - <<'PY'
import sys, os, json
sys.path.insert(0, "tools"); sys.path.insert(0, "bitorch-engine_amd")
from sweep import time_case
from bitorch_engine import _hip
for (K, N) in ((4096, 11008), (4096, 4096)):
    print(K, N, " ".join(f"M{M}:{time_case(M, K, N, _hip.BF16)['us']}" for M in (1, 2, 3, 4, 8, 12, 16)), flush=True)
print("f16 4096x11008", " ".join(f"M{M}:{time_case(M, 4096, 11008, _hip.F16)['us']}" for M in (1, 4, 16)), flush=True)
PY
done
