#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for bm in 0 128 64; do for dtn in bf16 f16; do
echo "== BM=$bm $dtn"; BIE_GEMM_BM=$bm BIE_GEMM_S=1 timeout 300 python - <<PY 2>&1 | grep -v amdgpu.ids
import sys, json; sys.path.insert(0, "tools")
import sweep
from bitorch_engine import _hip
dt = _hip.BF16 if "$dtn" == "bf16" else _hip.F16
for (M, K, N) in ((4096, 4096, 11008), (4096, 4096, 4096), (2048, 4096, 11008), (4096, 11008, 4096)):
    print(json.dumps(sweep.time_case(M, K, N, dt, layers=4, reps=5)), flush=True)
PY
done; done
