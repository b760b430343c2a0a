#!/bin/bash
# rocprofv3 kernel stats of the whole default bench pass (every extra row): per-kernel averages for the GEMM / exl2 / binary / conv kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
(cd /tmp && rm -rf /tmp/prof_full && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o s -- python $R/bench.py --no-cpu-baseline > /tmp/prof_full.json 2> /tmp/prof_full.err)
f=$(find /tmp/prof_full -name "*kernel_stats*.csv" | head -1)
(head -1 $f; grep '"void bie::\|"bie::' $f) | cut -c1-260 > gpurun_out/r02_kernel_stats_full_bench.csv
wc -l gpurun_out/r02_kernel_stats_full_bench.csv; head -30 gpurun_out/r02_kernel_stats_full_bench.csv | cut -c1-170
