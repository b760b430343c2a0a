#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > gpurun_out/pytest_x.log 2>&1; tail -2 gpurun_out/pytest_x.log; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_x.log | head -8; grep -E "^E  " gpurun_out/pytest_x.log | head -8
for lo in 2 1; do echo "fp16 MFMA from M=$lo: $(BIE_LUT_MFMA_MIN_M_F16=$lo timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -1
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "bitorch-engine_amd")
from sweep import time_case
from bitorch_engine import _hip
print(" ".join(f"{K}x{N}:M1 {time_case(1, K, N, _hip.F16)['us']} M2 {time_case(2, K, N, _hip.F16)['us']}" for (K, N) in ((4096, 4096), (4096, 11008), (11008, 4096))))
PY
)"; done
timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -1
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "bitorch-engine_amd")
from sweep import time_case
from bitorch_engine import _hip
print("w2 bf16", " ".join(f"{K}x{N}:M1 {time_case(1, K, N, _hip.BF16, w_bit=2)['us']} M2 {time_case(2, K, N, _hip.BF16, w_bit=2)['us']}" for (K, N) in ((4096, 4096), (4096, 11008), (11008, 4096))))
PY
