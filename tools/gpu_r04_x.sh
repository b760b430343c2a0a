#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "plain_graph or inline" > gpurun_out/r04_x_pytest.log 2>&1; tail -3 gpurun_out/r04_x_pytest.log; grep -E "^E  " gpurun_out/r04_x_pytest.log | head -20
