#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
python -m pytest tests -m gpu -q -p no:cacheprovider -k "exl2 or gidx or unequal or dense or prefill" 2>&1 | tail -4
python tools/exl2_prefill_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_exl2_prefill_ab.txt
(cd /tmp && rm -rf /tmp/prof_e && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o s -- python $R/tools/exl2_prefill_one.py 4096 4096 4096 > /tmp/prof_e.out 2>&1)
f=$(find /tmp/prof_e -name "*kernel_stats*.csv" | head -1)
(head -1 $f; grep 'bie::' $f) | cut -c1-330 > gpurun_out/r05_exl2_prefill_kernel_stats.csv; cat gpurun_out/r05_exl2_prefill_kernel_stats.csv | cut -c1-200
