#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest int"; timeout 600 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider -k "q4 or q8" > gpurun_out/pytest_int.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_int.log | tail -30; grep -E "^E  " gpurun_out/pytest_int.log | head -30
echo "== int gemm bench"; timeout 300 python tools/intgemm_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/intgemm.log
