#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "list or mpq or forward or layer or chain" > gpurun_out/r04_f_pytest.log 2>&1; tail -3 gpurun_out/r04_f_pytest.log; grep -E "^E  " gpurun_out/r04_f_pytest.log | head -30
for cfg in "BIE_LIST_D16=1 BIE_DECODE_INLINE=0" "BIE_LIST_D16=0 BIE_DECODE_INLINE=0"; do
  echo "== bench headline  $cfg"
  env $cfg timeout 600 python bench.py --no-extras --no-cpu-baseline 2> gpurun_out/r04_f_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('headline frac', d['roofline']['frac'], 'cold', d['cold_start']['roofline_frac'], 'verified', d['verified'])"
done
