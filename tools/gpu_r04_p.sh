#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for env in "BIE_DECODE_INLINE=0" "BIE_DECODE_INLINE=2 BIE_INL_D16=0 BIE_INL_NW=8" "BIE_DECODE_INLINE=2 BIE_INL_D16=0 BIE_INL_NW=4" "BIE_DECODE_INLINE=2 BIE_INL_D16=1 BIE_INL_NW=4" "BIE_DECODE_INLINE=2 BIE_INL_D16=0 BIE_INL_NW=8 BIE_INL_H=1"; do
  env $env timeout 300 python tools/inl_sweep.py 2>/dev/null
done
} > gpurun_out/r04_p_inl_sweep2.txt 2>&1
cat gpurun_out/r04_p_inl_sweep2.txt
