#!/bin/bash
# Round evidence pass: GPU pytest, smoke, bench line, rocprofv3 kernel stats PER SHAPE (separate runs), PMC traffic, per-wave timeline.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; TAG=${TAG:-r02}
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log; grep -E "^(FAILED|ERROR)" gpurun_out/${TAG}_pytest.log | head
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench"; timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; python -c "
import json; d = json.load(open('gpurun_out/${TAG}_bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step')}, d['roofline'])
for k, v in d.items():
    if isinstance(v, dict) and 'roofline' in v: print(k, v.get('us_per_launch'), v['roofline']['frac'])
"
for shape in 4096x4096 4096x11008 11008x4096; do
  echo "== rocprof kernel stats $shape"
  (cd /tmp && rm -rf /tmp/prof_$shape && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$shape -o s -- python $R/bench.py --only $shape --no-cpu-baseline > /tmp/prof_$shape.json 2> /tmp/prof_$shape.err)
  f=$(find /tmp/prof_$shape -name "*kernel_stats*.csv" | head -1)
  if [ -n "$f" ]; then cp $f gpurun_out/${TAG}_kernel_stats_$shape.csv; head -2 $f | cut -c1-200; fi
  cp /tmp/prof_$shape.json gpurun_out/${TAG}_prof_bench_$shape.json 2>/dev/null
done
echo "== pmc traffic"; timeout 400 bash tools/gpu_pmc_traffic.sh > gpurun_out/${TAG}_pmc_traffic.log 2>&1; tail -5 gpurun_out/${TAG}_pmc_traffic.log
echo "== per-wave timeline (4096x11008: one group per wave; the instrumented variant exists for that plan only)"; timeout 120 python tools/lut_stamps.py 4096 11008 2>&1 | grep -v "amdgpu.ids\|distinct\|workgroup end" | tee gpurun_out/${TAG}_lut_timeline.txt
echo "== small-batch sweep"; timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_m_sweep.txt
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "bitorch-engine_amd")
from sweep import time_case
from bitorch_engine import _hip
for (K, N) in ((4096, 11008), (4096, 4096)):
    print("bf16", K, N, " ".join(f"M{M}:{time_case(M, K, N, _hip.BF16)['us']}" for M in (1, 2, 3, 4, 8, 12, 16, 17, 32)), flush=True)
print("f16 4096 11008", " ".join(f"M{M}:{time_case(M, 4096, 11008, _hip.F16)['us']}" for M in (1, 2, 4, 8, 16, 17)), flush=True)
PY
