#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export BIE_HIP_LIB=$PWD/bitorch-engine_amd/lab/libbie_hip.so
for shp in "4096 4096" "4096 11008" "11008 4096"; do
  echo "== lut_stamps $shp"; timeout 300 python tools/lut_stamps.py $shp 2>&1 | tail -9
done > gpurun_out/r04_b_lut_stamps.txt 2>&1
cat gpurun_out/r04_b_lut_stamps.txt
