#!/bin/bash
# decode list launch: per-layer time against the layer shape (row stride N*4 bytes, K groups), same process each
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
for s in "4096 4096 96" "4096 8192 48" "4096 11008 40" "4096 12288 32" "4096 16384 24" "4096 10240 40" "4096 11264 40" "8192 4096 48" "11008 4096 40" "12288 4096 32" "8192 8192 24" "4096 4096 96"; do
  python tools/list_ab.py shape $s 2>&1 | grep -v amdgpu.ids
done
for v in 3; do for s in "4096 4096 96" "4096 11008 40" "11008 4096 40"; do echo "BIE_LIST_VAR=$v"; BIE_LIST_VAR=$v python tools/list_ab.py shape $s 2>&1 | grep -v amdgpu.ids; done; done
} | tee gpurun_out/r03_list_shapes.txt
