#!/bin/bash
# round 6, pass G: x-sharing matrix-pipe list form against the k-split form (BIE_LUTM_XS_MIN_M=0), both dtypes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "list" 2>&1 | tail -15 > gpurun_out/r06_g_pytest.txt
timeout 900 python tests/sweeps/fuzz_mpq_lists.py 200 > gpurun_out/r06_g_fuzz_lists.txt 2>&1
O=gpurun_out/r06_lutm_xs_ab.txt; : > $O
for dt in bf16 f16; do
for arm in xs ksplit xs ksplit; do
  if [ $arm = ksplit ]; then export BIE_LUTM_XS_MIN_M=0; else unset BIE_LUTM_XS_MIN_M; fi
  echo "== $dt $arm" >> $O
  timeout 600 python tools/lutm_ab.py $dt 2>/dev/null >> $O
done; done
unset BIE_LUTM_XS_MIN_M
tail -n 4 gpurun_out/r06_g_pytest.txt; tail -n 5 gpurun_out/r06_g_fuzz_lists.txt
