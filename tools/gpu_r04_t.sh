#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ for env in "BIE_DECODE_INLINE=0" "BIE_INL_NW=4" "BIE_INL_NW=4 BIE_INL_MAX_WAVES=4096" "BIE_INL_NW=4 BIE_INL_MAX_WAVES=8192" "BIE_INL_NW=4 BIE_INL_MAX_WAVES=12288" "BIE_INL_NW=4 BIE_INL_MAX_WAVES=16384" "BIE_INL_NW=4 BIE_INL_MAX_WAVES=28672" "BIE_INL_NW=8 BIE_INL_MAX_WAVES=8192" "BIE_INL_NW=8 BIE_INL_MAX_WAVES=16384"; do env $env timeout 200 python tools/inl_big_sweep.py 2>/dev/null; done; } > gpurun_out/r04_t_inl_big_sweep.txt; cat gpurun_out/r04_t_inl_big_sweep.txt
