#!/bin/bash
# round 6, pass T: W2 one-row lists with four-wave workgroups (BIE_LIST_W2_NW=4) against eight
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_w2_nw_ab.txt; : > $O
echo "== pytest W2 (nw 4)"; BIE_LIST_W2_NW=4 timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "list or w2 or W2" 2>&1 | tail -3
for rep in 1 2; do for nw in 8 4; do BIE_LIST_W2_NW=$nw timeout 300 python tools/w2_list_ab.py 2>/dev/null >> $O; done; done
cat $O
