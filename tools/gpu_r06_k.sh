#!/bin/bash
# round 6, pass K: x-sharing form, four against eight column tiles per workgroup
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_lutm_xs_nw.txt; : > $O
export LUTM_AB_ROWS=8,16,24,32 BIE_LUTM_XS_MIN_M_BF16=3
for dt in bf16 f16; do
for nw in 4 8 4 8; do
  export BIE_LUTM_XS_NW=$nw
  echo "== $dt nw=$nw" >> $O
  timeout 600 python tools/lutm_ab.py $dt 2>/dev/null >> $O
done; done
python tools/lutm_ab_table.py $O
