#!/bin/bash
# tile order A/B for the W8A8 / W4A4 / FP4 pipeline kernels: product (BIE_PIPE_GM = 4) against bitorch-engine_amd/ab/libbie_gm1.so (-DBIE_PIPE_GM=1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
{
for lib in "" "$PWD/bitorch-engine_amd/ab/libbie_gm1.so" ""; do
  echo "== lib ${lib:-product (gm 4)}"
  if [ -n "$lib" ]; then export BIE_HIP_LIB=$lib; else unset BIE_HIP_LIB; fi
  python tools/intgemm_bench.py 2>&1 | grep -v amdgpu.ids
  python tools/fp4_ab.py 4096 4096 2>&1 | grep -i "gemm\|tile" | head -12
  python tools/fp4_ab.py 8192 8192 2>&1 | grep -i "gemm\|tile" | head -12
done
} 2>&1 | tee gpurun_out/r03_pipe_gm_ab.txt
