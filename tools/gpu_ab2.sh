#!/bin/bash
# same-box A/B of two library builds on the M=4096 GEMM shapes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for v in old new; do
  echo "== $v (rep $rep)"; BIE_HIP_LIB=$PWD/bitorch-engine_amd/ab/$v.so timeout 300 python tools/sweep.py gemm 2>&1 | grep -v amdgpu.ids | grep '"M": 4096'
done; done
