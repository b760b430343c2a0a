#!/bin/bash
# round 6, pass I: x-sharing form against the k-split form at few rows, both dtypes, final plan
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_lutm_xs_rows.txt; : > $O
export LUTM_AB_ROWS=3,4,6,8,12,16,20,32
for dt in bf16 f16; do
for arm in xs ksplit xs ksplit; do
  if [ $arm = ksplit ]; then export BIE_LUTM_XS_MIN_M=0; else unset BIE_LUTM_XS_MIN_M; fi
  echo "== $dt $arm" >> $O
  timeout 600 python tools/lutm_ab.py $dt 2>/dev/null >> $O
done; done
python tools/lutm_ab_table.py $O
