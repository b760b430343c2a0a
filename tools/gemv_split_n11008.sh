#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for nw in 16 8; do for tgt in 172 344 516 688 1032 1376; do
  echo -n "NW=$nw TARGET=$tgt: "
  BIE_GEMV3_NW=$nw BIE_GEMV3_MIN_BLOCKS=100000 BIE_GEMV3_TARGET_BLOCKS=$tgt timeout 120 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tr '\n' ' '
import sys; sys.path.insert(0, "tools")
import sweep
from bitorch_engine import _hip
for dt in (_hip.BF16, _hip.F16):
    r = sweep.time_case(1, 4096, 11008, dt)
    print(f"{r['dt']} {r['us']}us", end="  ")
PY
  echo
done; done
