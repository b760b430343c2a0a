#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "list or bench_py or chain or timeout" > gpurun_out/r04_cc_pytest.log 2>&1; tail -3 gpurun_out/r04_cc_pytest.log; grep -E "^E  " gpurun_out/r04_cc_pytest.log | head -20
export LIST_AB_PASSES=4
{ for env in "BIE_LIST_PERSIST_WGS=0" "BIE_LIST_PERSIST_WGS=1536" "BIE_LIST_PERSIST_WGS=1280" "BIE_LIST_PERSIST_WGS=1792" "BIE_LIST_PERSIST_WGS=2048" "BIE_LIST_PERSIST_WGS=3072" "BIE_LIST_PERSIST_WGS=0" "BIE_LIST_PERSIST_WGS=1536"; do
  env $env timeout 300 python tools/list_ab.py quick 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$env', d['list96x1']['us_per_layer'], d['list96x1']['frac'])"
done; } > gpurun_out/r04_cc_list_persist.txt; cat gpurun_out/r04_cc_list_persist.txt
