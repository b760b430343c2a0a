#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for env in "BIE_DECODE_INLINE=0" "BIE_INL_NW=8" "BIE_INL_NW=4" "BIE_INL_NW=8 BIE_INL_H=1" "BIE_INL_NW=8 BIE_INL_MAX_WAVES=4096" "BIE_INL_NW=8 BIE_INL_MAX_WAVES=8192" "BIE_INL_NW=4 BIE_INL_MAX_WAVES=8192" "BIE_INL_NW=8 BIE_INL_WANT_WAVES=2048" "BIE_INL_NW=8 BIE_INL_WANT_WAVES=8192"; do
  env $env timeout 300 python tools/inl_sweep.py 2>/dev/null
done
} > gpurun_out/r04_h_inl_sweep.txt 2>&1
cat gpurun_out/r04_h_inl_sweep.txt
