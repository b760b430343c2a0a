#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== stamps"; BIE_HIP_LIB=$PWD/bitorch-engine_amd/lab7/libbie_hip.so timeout 200 python tools/gemm_stamps.py 2>&1 | grep -v amdgpu.ids | grep "wave 0"
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider -k "mpq or full_size or layer or mbwq" > gpurun_out/pytest.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest.log | tail -30; grep -E "^E  " gpurun_out/pytest.log | head -20
echo "== sweep"; timeout 600 python tools/sweep.py ${SWEEP:-gemm} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sweep.log
