#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest (list / forward / grouped subsets first, then everything)"
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r04_e_pytest.log 2>&1; tail -5 gpurun_out/r04_e_pytest.log; grep -E "^E  " gpurun_out/r04_e_pytest.log | head -30
for cfg in "BIE_LIST_D16=0 BIE_DECODE_INLINE=0" "BIE_LIST_D16=1 BIE_DECODE_INLINE=0" "BIE_LIST_D16=1 BIE_DECODE_INLINE=1"; do
  echo "== bench --short  $cfg"
  env $cfg timeout 600 python bench.py --short --no-cpu-baseline 2> gpurun_out/r04_e_bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('headline frac', d['roofline']['frac'], 'cold', d['cold_start']['roofline_frac'], 'verified', d['verified'])
print('decode_step', d.get('decode_step_llama7b'))
print({k:v for k,v in d['summary'].items()})"
done
