#!/bin/bash
# round 6, pass Q: fragments per wave of the dense form's dequantise pass (8 shipped; 4 / 2 / 1 = more, shorter waves: the store stream of one round under the loads of the next)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_dq_fpw_ab.txt; : > $O
for rep in 1 2; do for f in 8 4 2 1; do
  BIE_DQ_FPW=$f timeout 300 python tools/dq_fpw_ab.py 2>/dev/null >> $O
done; done
cat $O
(cd /tmp && rm -rf /tmp/pq && BIE_DQ_FPW=2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -o q -- python $GRAFT_REPO_ROOT/tools/dq_fpw_ab.py > /dev/null 2>&1)
for f in $(find /tmp/pq -name "*kernel_stats*.csv" | head -1); do grep -i "dequant_frag\|dense_gemm" $f | cut -c1-60,150-330; done
