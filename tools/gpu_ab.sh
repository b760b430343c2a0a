#!/bin/bash
# same-box A/B of two builds of the library (bitorch-engine_amd/ab/{old,new}.so), alternating twice
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest (new)"; BIE_HIP_LIB=$PWD/bitorch-engine_amd/ab/new.so timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do for v in old new; do
  echo "== $v (rep $rep)"; BIE_HIP_LIB=$PWD/bitorch-engine_amd/ab/$v.so timeout 300 python tools/sweep.py ${SWEEP:-ab} 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab_$v.log
done; done
