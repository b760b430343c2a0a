#!/bin/bash
# round 6, pass F: fp16 matrix-pipe list form, packed-fp16 arithmetic (no table) against the table form (variant f16table)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_lutm_f16_direct.txt; : > $O
for arm in direct table direct table; do
  if [ $arm = table ]; then export BIE_HIP_LIB=$PWD/bitorch-engine_amd/variants/f16table/libbie_hip.so; else unset BIE_HIP_LIB; fi
  echo "== $arm" >> $O
  timeout 600 python tools/lutm_ab.py f16 >> $O 2>&1
done
unset BIE_HIP_LIB
timeout 1500 python -m pytest tests -m gpu -x -q -k "list or lutm or grouped or mpq" 2>&1 | tail -5 > gpurun_out/r06_f_pytest.txt
timeout 900 python tests/sweeps/fuzz_mpq_lists.py 200 > gpurun_out/r06_f_fuzz_lists.txt 2>&1
timeout 900 python tests/sweeps/fuzz_mpq_forward.py 300 > gpurun_out/r06_f_fuzz_fwd.txt 2>&1
tail -3 gpurun_out/r06_f_pytest.txt gpurun_out/r06_f_fuzz_lists.txt gpurun_out/r06_f_fuzz_fwd.txt
