#!/bin/bash
# GPU-side: parity + timing of the table-lookup GEMV and its knobs
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 300 python tools/gemv_lut_check.py $NOPAR 2>&1 | grep -v amdgpu.ids | head -${HEADN:-30}; }
NOPAR=""; run BIE_GEMV_LUT=1
NOPAR="--no-parity"; HEADN=3
run BIE_GEMV_LAB=2
run BIE_LUT_NW=4
run BIE_LUT_NW=16
run BIE_LUT_NW=4 BIE_LUT_ROWS=32
