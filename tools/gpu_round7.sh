#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== base"; timeout 300 python tools/gemm_only.py 4096 bf16 2>&1 | grep -v amdgpu
for n in 1 2 3 5 6; do echo "== GEMM LAB=$n"; BIE_HIP_LIB=$PWD/bitorch-engine_amd/lab/$n/libbie_hip.so timeout 300 python tools/gemm_only.py 4096 bf16 2>&1 | grep -v amdgpu; done
