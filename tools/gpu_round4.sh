#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 300 python tools/sweep.py gemv 2>&1 | grep '"M": 1' | grep bf16; }
run BIE_GEMV_LAB=1 BIE_GEMV_V3=1
run BIE_GEMV_LAB=1 BIE_GEMV_V3=1 BIE_GEMV3_MIN_BLOCKS=256
run BIE_GEMV_LAB=1 BIE_GEMV_V3=1 BIE_GEMV3_NW=8 BIE_GEMV3_MIN_BLOCKS=512
run BIE_GEMV_LAB=1 BIE_GEMV_V3=0
run BIE_GEMV_LAB=1 BIE_GEMV_V3=0 BIE_GEMV_TARGET_BLOCKS=2048
run BIE_GEMV_LAB=1 BIE_GEMV_V3=0 BIE_GEMV_TARGET_BLOCKS=512
run BIE_GEMV_LAB=0 BIE_GEMV_V3=1
