#!/bin/bash
# Round-3 evidence pass: PMC traffic of the list launch, rocprofv3 kernel stats of the bench command (headline) and per workload family.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
echo "== pmc traffic"; timeout 500 bash tools/gpu_pmc_traffic_r03.sh > gpurun_out/r03_pmc_traffic.log 2>&1; tail -30 gpurun_out/r03_pmc_traffic.log | grep -E "ratio|list" 
echo "== rocprof kernel stats: headline (bench.py --no-extras)"
(cd /tmp && rm -rf /tmp/prof_head && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_head -o s -- python $R/bench.py --no-extras --no-cpu-baseline > /tmp/prof_head.json 2> /tmp/prof_head.err)
f=$(find /tmp/prof_head -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r03_kernel_stats_headline_list96_4096x4096.csv && head -3 $f | cut -c1-220
cp /tmp/prof_head.json gpurun_out/r03_prof_bench_headline.json 2>/dev/null; python -c "
import json; d=json.load(open('/tmp/prof_head.json')); print('bench under rocprof:', d['roofline']['avg_launch_us'], d['roofline']['frac'])"
echo "== rocprof kernel stats: full bench"
(cd /tmp && rm -rf /tmp/prof_full && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o s -- python $R/bench.py --no-cpu-baseline > /tmp/prof_full.json 2> /tmp/prof_full.err)
f=$(find /tmp/prof_full -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r03_kernel_stats_full_bench.csv && head -12 $f | cut -c1-200
