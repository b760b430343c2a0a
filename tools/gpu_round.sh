#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel trace.  Everything lands in gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4 > gpurun_out/gpu.txt
nproc >> gpurun_out/gpu.txt; lscpu | grep "Model name" >> gpurun_out/gpu.txt
echo "== pytest" ; timeout 1500 python -m pytest tests -m gpu -q -x --maxfail=12 -p no:cacheprovider > gpurun_out/pytest.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest.log | tail -30; grep -E "^E  " gpurun_out/pytest.log | head -20
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
echo "== bench"; timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r01 -- python "$OLDPWD/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > /tmp/prof_bench.json 2> /tmp/prof.err); 
find /tmp/prof -type f | head -20; for f in $(find /tmp/prof -name "*kernel_stats*.csv" | head -1); do cp $f gpurun_out/r01_kernel_stats.csv; done
cp /tmp/prof_bench.json gpurun_out/r01_prof_bench.json 2>/dev/null; tail -3 /tmp/prof.err
head -12 gpurun_out/r01_kernel_stats.csv 2>/dev/null
