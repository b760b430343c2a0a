#!/bin/bash
# Round-3 closing evidence on one MI355X box: the full GPU test suite, smoke(), the default bench line, rocprofv3 kernel stats of the bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 | tee gpurun_out/r03_pytest_gpu_tail.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/r03_smoke.txt
echo "== bench"; timeout 900 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; tail -c 600 gpurun_out/r03_bench.json; python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r03_bench.json') if l.startswith('{')][-1])
print({k: d[k] for k in ('metric', 'value', 'unit', 'ms_per_step')}); print(d['roofline']); print(d.get('roofline_gemm')); print(d['cpu_baseline'])
PY
echo "== rocprof kernel stats: full bench"
(cd /tmp && rm -rf /tmp/prof_full && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o s -- python $R/bench.py --no-cpu-baseline > /tmp/prof_full.json 2> /tmp/prof_full.err)
f=$(find /tmp/prof_full -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r03_kernel_stats_full_bench.csv && head -8 $f | cut -c1-200
