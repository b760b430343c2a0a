#!/bin/bash
# round 6, pass L: x-sharing form, depth of the row ring (units of rows in flight per wave)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_lutm_xs_depth.txt; : > $O
export LUTM_AB_ROWS=4,16,32 BIE_LUTM_XS_MIN_M_BF16=3 LUTM_AB_SHAPES=0,1
for dt in bf16 f16; do
for d in 1 2 3 1 2 3; do
  if [ $d = 1 ]; then unset BIE_HIP_LIB; else export BIE_HIP_LIB=$PWD/bitorch-engine_amd/variants/xsdepth$d/libbie_hip.so; fi
  echo "== $dt depth=$d" >> $O
  timeout 600 python tools/lutm_ab.py $dt 2>/dev/null >> $O
done; done
python tools/lutm_ab_table.py $O
