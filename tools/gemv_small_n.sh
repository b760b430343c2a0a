#!/bin/bash
# N = 4096 decode shapes: number of waves per block x minimum block count (split-K factor) of the GEMV
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for nw in 16 8; do for mb in 64 128 160 256 384 512; do
  echo -n "NW=$nw MIN_BLOCKS=$mb: "
  BIE_GEMV3_NW=$nw BIE_GEMV3_MIN_BLOCKS=$mb timeout 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tr '\n' ' '
import sys; sys.path.insert(0, "tools")
import sweep
from bitorch_engine import _hip
for (K, N) in ((4096, 4096), (11008, 4096)):
    r = sweep.time_case(1, K, N, _hip.BF16)
    print(f"K={K} {r['us']}us", end="  ")
PY
  echo
done; done
