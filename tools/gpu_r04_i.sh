#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=10 > gpurun_out/r04_i_pytest.log 2>&1; tail -5 gpurun_out/r04_i_pytest.log; grep -E "^E  |^FAILED|^ERROR" gpurun_out/r04_i_pytest.log | head -40
