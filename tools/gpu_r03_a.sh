#!/bin/bash
# round 3, call A: GPU pytest, list-launch A/B over kernel variants, bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=r03a
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log; grep -E "^(FAILED|ERROR)" gpurun_out/${TAG}_pytest.log | head -40
echo "== list A/B"
for v in 1 0 3 5; do
  echo "-- BIE_LIST_VAR=$v"; BIE_LIST_VAR=$v timeout 300 python tools/list_ab.py $([ $v != 1 ] && echo quick) 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_list_ab.txt
done
for g in 2 8; do echo "-- BIE_LIST_MAX_GPW=$g"; BIE_LIST_MAX_GPW=$g timeout 200 python tools/list_ab.py quick 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_list_ab.txt; done
for w in 3072 12288; do echo "-- BIE_LIST_WANT_WAVES=$w"; BIE_LIST_WANT_WAVES=$w timeout 200 python tools/list_ab.py quick 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_list_ab.txt; done
echo "== bench"; timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -3 gpurun_out/${TAG}_bench.err; python - <<PY
import json
d = json.load(open('gpurun_out/${TAG}_bench.json'))
print({k: d[k] for k in ('value', 'ms_per_step')}, d['roofline'])
for k, v in d.items():
    if isinstance(v, dict) and 'roofline' in v: print(k, v.get('us_per_launch', v.get('us_per_layer')), v['roofline']['frac'])
    if isinstance(v, dict) and 'error' in v: print(k, 'ERROR', v['error'])
PY
