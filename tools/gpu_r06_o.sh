#!/bin/bash
# round 6, pass O: the shipped list forms after the register fix, both dtypes, all rows
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_lutm_final.txt; : > $O
export LUTM_AB_ROWS=1,2,4,8,16,24,32
for dt in f16 bf16 f16 bf16; do
  echo "== $dt shipped" >> $O
  timeout 600 python tools/lutm_ab.py $dt 2>/dev/null >> $O
done
python tools/lutm_ab_table.py $O
