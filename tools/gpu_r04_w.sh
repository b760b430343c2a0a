#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export BIE_HIP_LIB=$PWD/bitorch-engine_amd/lab/libbie_hip.so LIST_AB_PASSES=4
{ for env in "BIE_LIST_RING2=0" "BIE_LIST_RING2=1" "BIE_LIST_RING2=7" "BIE_LIST_RING2=0" "BIE_LIST_RING2=1" "BIE_LIST_RING2=7"; do
  env $env timeout 300 python tools/list_ab.py quick 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$env', d['list96x1']['us_per_layer'], d['list96x1']['frac'])"
done; } > gpurun_out/r04_w_list_ring.txt; cat gpurun_out/r04_w_list_ring.txt
