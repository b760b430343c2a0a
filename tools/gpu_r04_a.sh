#!/bin/bash
# Round 4, GPU call A: cache / instruction probes for the decode redesign, GPU tests after the hygiene commit, the new short bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== probe_prefetch"; timeout 300 tools/probe/bin/probe_prefetch > gpurun_out/r04_a_probe_prefetch.txt 2>&1; tail -80 gpurun_out/r04_a_probe_prefetch.txt
echo "== probe_lut"; timeout 200 tools/probe/bin/probe_lut > gpurun_out/r04_a_probe_lut.txt 2>&1; cat gpurun_out/r04_a_probe_lut.txt
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r04_a_pytest.log 2>&1; tail -5 gpurun_out/r04_a_pytest.log; grep -E "^E  " gpurun_out/r04_a_pytest.log | head -20
echo "== bench --short"; timeout 900 python bench.py --short > gpurun_out/r04_a_bench.json 2> gpurun_out/r04_a_bench.err; cat gpurun_out/r04_a_bench.json; wc -c gpurun_out/r04_a_bench.json; grep "bench extra" gpurun_out/r04_a_bench.err | cut -c1-400
cp gpurun_out/bench_extras.json gpurun_out/r04_a_bench_extras.json 2>/dev/null
