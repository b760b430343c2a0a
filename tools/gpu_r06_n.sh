#!/bin/bash
# round 6, pass N: fp16 lone launches -- the matrix-pipe form with arithmetic dequantisation from ONE row (BIE_LUT_MFMA_MIN_M_F16=1) against the lookup + v_fma_mix form;
# and the arithmetic matrix-pipe form against its table build (variant f16table) at 2..16 rows
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_lone_f16.txt; : > $O
for arm in dflt mfma1 table dflt mfma1 table; do
  unset BIE_LUT_MFMA_MIN_M_F16 BIE_HIP_LIB
  [ $arm = mfma1 ] && export BIE_LUT_MFMA_MIN_M_F16=1
  [ $arm = table ] && export BIE_HIP_LIB=$PWD/bitorch-engine_amd/variants/f16table/libbie_hip.so
  echo "== f16 $arm" >> $O
  timeout 600 python tools/lone_ab.py f16 2>/dev/null >> $O
done
python tools/lutm_ab_table.py $O
