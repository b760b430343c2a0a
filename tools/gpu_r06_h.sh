#!/bin/bash
# round 6, pass H: x-sharing form, K slices of the K = 11008 layers (units per workgroup), both dtypes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_lutm_xs_slices.txt; : > $O
export LUTM_AB_SHAPES=2 LUTM_AB_ROWS=8,16,24,32 BIE_LUTM_XS_MAX_GPW=96
for dt in bf16 f16; do
for want in 6144 5000 2500 6144 5000 2500; do
  export BIE_LIST_WANT_WAVES=$want
  echo "== $dt want=$want" >> $O
  timeout 600 python tools/lutm_ab.py $dt 2>/dev/null >> $O
done; done
python tools/lutm_ab_table.py $O
