#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ for env in "BIE_X=0" "BIE_LUT_WANT_WAVES=8192" "BIE_LUT_WANT_WAVES=16384" "BIE_LUT_H=2" "BIE_LUT_H=4" "BIE_LUT_ROWS=8" "BIE_LUT_ROWS=32" "BIE_LUT_MAX_WG=1024" "BIE_LUT_MAX_WG=4096 BIE_LUT_WANT_WAVES=8192" "BIE_LUT_NW=4"; do env $env timeout 300 python tools/inl_sweep.py --step 2>/dev/null; done; } > gpurun_out/r04_y_lut_plan_sweep.txt; cat gpurun_out/r04_y_lut_plan_sweep.txt
