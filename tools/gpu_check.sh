#!/bin/bash
# GPU-side check: parity suite (optionally filtered with K=<pytest -k expr>) and an optional timing sweep (SWEEP=<tools/sweep.py mode>)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider ${K:+-k "$K"} > gpurun_out/pytest.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest.log | tail -30; grep -E "^E  " gpurun_out/pytest.log | head -20
if [ -n "$SWEEP" ]; then echo "== sweep $SWEEP"; timeout 600 python tools/sweep.py $SWEEP 2>&1 | grep -v amdgpu.ids | tee gpurun_out/sweep.log; fi
if [ -n "$EXTRA" ]; then echo "== $EXTRA"; timeout 600 python $EXTRA 2>&1 | grep -v amdgpu.ids; fi
