#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/pytest.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest.log | tail -30; grep -E "^E  " gpurun_out/pytest.log | head -20
echo "== sweep default"; timeout 600 python tools/sweep.py all 2>&1 | tee gpurun_out/sweep.log | tail -40
for t in 256 512 2048; do echo "== gemv target $t"; BIE_GEMV_TARGET_BLOCKS=$t timeout 300 python tools/sweep.py gemv 2>&1 | grep '"M": 1' | tee -a gpurun_out/sweep_targets.log; done
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r01 -- python "$OLDPWD/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > /tmp/prof_bench.json 2> /tmp/prof.err)
find /tmp/prof -type f | head; for f in $(find /tmp/prof -name "*kernel_stats*.csv" | head -1); do cp $f gpurun_out/r01_kernel_stats.csv; done
cp /tmp/prof_bench.json gpurun_out/r01_prof_bench.json 2>/dev/null; head -12 gpurun_out/r01_kernel_stats.csv 2>/dev/null
