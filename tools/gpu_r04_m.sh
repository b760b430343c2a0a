#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export LIST_AB_PASSES=4
{ for env in "BIE_LIST_D16=0" "BIE_LIST_D16=1" "BIE_LIST_D16=1 BIE_LIST_WANT_WAVES=3072" "BIE_LIST_D16=1 BIE_LIST_WANT_WAVES=4096" "BIE_LIST_D16=1 BIE_LIST_WANT_WAVES=8192" "BIE_LIST_D16=1 BIE_LIST_WANT_WAVES=12288" "BIE_LIST_D16=1 BIE_LIST_WANT_WAVES=24576" "BIE_LIST_D16=1 BIE_LIST_MAX_GPW=4" "BIE_LIST_D16=1 BIE_LIST_MAX_GPW=32 BIE_LIST_WANT_WAVES=3072" "BIE_LIST_D16=0 BIE_LIST_WANT_WAVES=12288"; do
  env $env timeout 300 python tools/list_ab.py quick 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print('$env', d['list96x1']['us_per_layer'], d['list96x1']['frac'])"
done; } > gpurun_out/r04_m_list_sweep.txt; cat gpurun_out/r04_m_list_sweep.txt
