#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -k "exl2 or mbwq" > gpurun_out/pytest_x.log 2>&1; tail -2 gpurun_out/pytest_x.log; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_x.log | head -8; grep -E "^E  " gpurun_out/pytest_x.log | head -8
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -4
import sys, json
sys.path.insert(0, "bitorch-engine_amd")
import torch, bench
dev = torch.device("cuda:0")
for r in bench.bench_exl2(dev): print(r["K"], r["N"], r["us_per_launch"], r["roofline"]["frac"], flush=True)
PY
