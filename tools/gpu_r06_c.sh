#!/bin/bash
# Round 6, pass C: the one-launch binary conv -- parity tests, A/B against the forms it replaces.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=r06_c
echo "== conv tests"; timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "conv or full_size_exl2 or second_checkpoint" > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log; grep -E "^E  |^FAILED" gpurun_out/${TAG}_pytest.log | head -30
echo "== conv A/B"; timeout 900 python tools/conv_fused_bench.py > gpurun_out/${TAG}_conv_ab.txt 2> gpurun_out/${TAG}_conv_ab.err; cat gpurun_out/${TAG}_conv_ab.txt; tail -3 gpurun_out/${TAG}_conv_ab.err
