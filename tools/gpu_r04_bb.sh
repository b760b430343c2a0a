#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{ for env in "BIE_X=0" "BIE_LUT_COOP=1" "BIE_LUT_RD=4" "BIE_LUT_MFMA_MIN_M=1"; do env $env timeout 300 python tools/inl_sweep.py 2>/dev/null; done; } > gpurun_out/r04_bb_lut_forms.txt; cat gpurun_out/r04_bb_lut_forms.txt
