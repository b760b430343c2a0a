#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export BIE_HIP_LIB=$PWD/bitorch-engine_amd/lab/libbie_hip.so LIST_AB_PASSES=4
{ for abl in 0 1 2 4 3 5 6 7 0; do
  env BIE_LIST_ABL=$abl timeout 300 python tools/list_ab.py quick 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ABL=$abl', d['list96x1']['us_per_layer'], d['list96x1']['frac'])"
done; } > gpurun_out/r04_v_list_ablation.txt; cat gpurun_out/r04_v_list_ablation.txt
