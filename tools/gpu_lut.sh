#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > gpurun_out/pytest_x.log 2>&1; tail -2 gpurun_out/pytest_x.log; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_x.log | head -8; grep -E "^E  " gpurun_out/pytest_x.log | head -8
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -8
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "bitorch-engine_amd")
from sweep import time_case
from bitorch_engine import _hip
for dt in (_hip.BF16, _hip.F16):
    for (K, N) in ((4096, 4096), (4096, 11008), (11008, 4096)):
        r = time_case(1, K, N, dt, w_bit=2); r2 = time_case(2, K, N, dt, w_bit=2); print("w2", r["dt"], K, N, "M1", r["us"], r["GB/s"], "M2", r2["us"], flush=True)
PY
