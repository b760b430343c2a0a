#!/bin/bash
# round 6, pass M: fp16 lists of one / two rows in the x-sharing matrix-pipe form (arithmetic dequantisation) against the lookup + FMA list kernel
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_lutm_xs_m1.txt; : > $O
export LUTM_AB_ROWS=1,2,3
for dt in f16 bf16; do
for arm in xs1 dflt xs1 dflt; do
  if [ $arm = xs1 ]; then export BIE_LUTM_XS_MIN_M=1 BIE_LUTM_XS_MIN_M_BF16=1; else unset BIE_LUTM_XS_MIN_M BIE_LUTM_XS_MIN_M_BF16; fi
  echo "== $dt $arm" >> $O
  timeout 600 python tools/lutm_ab.py $dt 2>/dev/null >> $O
done; done
python tools/lutm_ab_table.py $O
