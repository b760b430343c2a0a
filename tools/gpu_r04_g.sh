#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export BIE_HIP_LIB=$PWD/bitorch-engine_amd/lab/libbie_hip.so
{
for env in "" "BIE_INL_H=1" "BIE_INL_H=4"; do
  env $env timeout 200 python tools/inl_timeline.py 4096 4096 2>&1 | grep -v amdgpu.ids
done
env timeout 200 python tools/inl_timeline.py 4096 11008 2>&1 | grep -v amdgpu.ids
env timeout 200 python tools/inl_timeline.py 4096 4096 4096 4096 2>&1 | grep -v amdgpu.ids
env timeout 200 python tools/inl_timeline.py 4096 11008 11008 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r04_g_inl_timeline.txt 2>&1
cat gpurun_out/r04_g_inl_timeline.txt
