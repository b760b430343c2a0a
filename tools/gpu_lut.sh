#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > gpurun_out/pytest_x.log 2>&1; tail -2 gpurun_out/pytest_x.log; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_x.log | head -8; grep -E "^E  " gpurun_out/pytest_x.log | head -8
timeout 300 python tools/exl2_m_sweep.py 2>&1 | grep -v amdgpu.ids | tail -1
