#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for lab in 0 1 2 3 4; do echo "== GEMM lab=$lab"; BIE_GEMM_LAB=$lab timeout 300 python tools/gemm_only.py 4096 bf16 2>&1 | grep -v amdgpu; done
