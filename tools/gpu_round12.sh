#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/pytest.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest.log | tail -30; grep -E "^E  " gpurun_out/pytest.log | head -20
