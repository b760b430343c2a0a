#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/pytest.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest.log | tail -30; grep -E "^E  " gpurun_out/pytest.log | head -20
echo "== sweep gemv default (v3 NW16 minblocks160)"; timeout 600 python tools/sweep.py gemv 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sweep_gemv.log
for cfg in "16 256" "16 512" "8 256" "8 512" "16 64"; do set -- $cfg; echo "== gemv3 NW=$1 MIN_BLOCKS=$2"; BIE_GEMV3_NW=$1 BIE_GEMV3_MIN_BLOCKS=$2 timeout 300 python tools/sweep.py gemv 2>&1 | grep '"M": 1' | tee -a gpurun_out/sweep_gemv3_cfg.log; done
echo "== old kernel"; BIE_GEMV_V3=0 timeout 300 python tools/sweep.py gemv 2>&1 | grep '"M": 1'
