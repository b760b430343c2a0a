#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 300 python tools/gemv_lut_check.py $NOPAR 2>&1 | grep -v amdgpu.ids | head -${HEADN:-3}; }
NOPAR=""; HEADN=30; run BIE_LUT_H=0; NOPAR="--no-parity"; HEADN=3
run BIE_LUT_H=0
run BIE_LUT_H=1
