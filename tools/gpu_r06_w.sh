#!/bin/bash
# round 6, pass W: lone 17..32-row calls on the measured shapes through the decode kernel (two row blocks): full GPU suite, the forward fuzz, and the timing on a second box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2000 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3
timeout 900 python tests/sweeps/fuzz_mpq_forward.py 600 211 2>/dev/null | tr '\n' ' ' | cut -c1-400; echo
for e in "BIE_LUT_RB2=0" "" "BIE_LUT_RB2=0" ""; do env $e LONE_SHAPES=1 timeout 600 python tools/lone_rb2_ab.py 2>/dev/null | tail -1; done > gpurun_out/r06_lone_rb2_check.txt
