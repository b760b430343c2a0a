#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${TAG:-r03c}
echo "== pytest (list tests)"; timeout 600 python -m pytest tests -m gpu -q -k "list or reducer" -p no:cacheprovider 2>&1 | tail -5
for v in ${VARS:-1 0}; do
  echo "-- BIE_LIST_VAR=$v"; BIE_LIST_VAR=$v timeout 300 python tools/list_ab.py ${MODE:-quick} 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_list_ab.txt
done
