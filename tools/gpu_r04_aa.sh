#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "unmodified or grouped" > gpurun_out/r04_aa_pytest.log 2>&1; tail -3 gpurun_out/r04_aa_pytest.log; grep -E "^E  " gpurun_out/r04_aa_pytest.log | head -20
timeout 1200 python bench.py --short --no-cpu-baseline > gpurun_out/r04_aa_bench_short.json 2> gpurun_out/r04_aa_bench.err; grep "decode_step" gpurun_out/r04_aa_bench.err | cut -c1-420
