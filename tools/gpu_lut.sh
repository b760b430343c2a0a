#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -k "binary or conv or embedding or bmha or BMHA" > gpurun_out/pytest_x.log 2>&1; tail -2 gpurun_out/pytest_x.log; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_x.log | head -8; grep -E "^E  " gpurun_out/pytest_x.log | head -8
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -9
import sys, json
sys.path.insert(0, "bitorch-engine_amd")
import torch, bench
from bitorch_engine import _hip
dev = torch.device("cuda:0")
for r in bench.bench_binary(dev, _hip.lib()):
    print(r["op"][:40], r.get("M", r.get("B")), r.get("us_per_launch", r.get("us_per_call")), flush=True)
PY
