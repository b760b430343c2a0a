#!/bin/bash
# round 6, pass S: closing fuzz sweeps on the final sources, fresh seeds, larger counts
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_fuzz_closing.txt; : > $O
run() { echo "== $*" >> $O; timeout 1500 python "$@" 2>/dev/null | tr '\n' ' ' | cut -c1-1500 >> $O; echo >> $O; }
run tests/sweeps/fuzz_mpq_forward.py 1500 101
run tests/sweeps/fuzz_mpq_lists.py 500 102
run tests/sweeps/fuzz_exl2_forward.py 400 103
run tests/sweeps/fuzz_other_ops.py 200 104
run tests/sweeps/fuzz_sibling_groups.py 80 105
cat $O
