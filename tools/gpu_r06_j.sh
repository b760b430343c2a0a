#!/bin/bash
# round 6, pass J: x-sharing list form -- forced-plan parity tests, the list tests, fuzz
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "list or reducer" 2>&1 | tail -15 > gpurun_out/r06_j_pytest.txt
timeout 900 python tests/sweeps/fuzz_mpq_lists.py 300 > gpurun_out/r06_j_fuzz_lists.txt 2>&1
tail -n 6 gpurun_out/r06_j_pytest.txt; tail -n 5 gpurun_out/r06_j_fuzz_lists.txt
