#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
timeout 600 python tools/gemm_k.py 2>&1 | grep -v amdgpu
