#!/bin/bash
# round 6, pass R: lone multi-row calls against lists of one entry (k-split default, x-sharing with K sliced)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python tools/midm_probe.py 2> gpurun_out/r06_midm_probe.err | tee gpurun_out/r06_midm_probe.txt
tail -5 gpurun_out/r06_midm_probe.err
