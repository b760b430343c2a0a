#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tail -4
import sys, json
sys.path.insert(0, "bitorch-engine_amd")
import torch, bench
dev = torch.device("cuda:0")
for r in bench.bench_exl2(dev): print(r["K"], r["N"], r["us_per_launch"], flush=True)
PY
