#!/bin/bash
# round 6, pass P: the folded fp16 sym decode (fma(1024 + q, s, -1024 s)) against the unfolded arm, same box, alternating; parity of the list / lone forms first
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_lutm_fold_ab.txt; : > $O
echo "== pytest (folded)"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "list or special_values or decode or lone or grouped or fuzz or randomised" 2>&1 | tail -4
export LUTM_AB_ROWS=1,2,4,8,16,32
for rep in 1 2; do
  echo "== f16 folded" >> $O; timeout 600 python tools/lutm_ab.py f16 2>/dev/null >> $O
  echo "== f16 nofold" >> $O; BIE_HIP_LIB=$PWD/bitorch-engine_amd/variants/nofold/libbie_hip.so timeout 600 python tools/lutm_ab.py f16 2>/dev/null >> $O
done
python tools/lutm_ab_table.py $O
