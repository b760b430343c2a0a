#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider -k "conv or binary" > gpurun_out/pytest_x.log 2>&1; tail -2 gpurun_out/pytest_x.log; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_x.log | head -8; grep -E "^E  " gpurun_out/pytest_x.log | head -8
(cd /tmp && rm -rf /tmp/prof_conv && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_conv -o s -- python $R/tools/conv_prof.py > /tmp/prof_conv.log 2>&1)
f2=$(find /tmp/prof_conv -name "*kernel_trace*.csv" | head -1); python - "$f2" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    if "bie::" in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:48], r.get("Grid_Size") or r.get("Grid_Size_X"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    v.sort(); print(k, len(v), "median ns", v[len(v)//2], "min", v[0])
PY
echo "== binary bench"; timeout 300 python - <<'PY'
import sys, json
sys.path.insert(0, "bitorch-engine_amd")
import torch, bench
from bitorch_engine import _hip
dev = torch.device("cuda:0")
for r in bench.bench_binary(dev, _hip.lib()):
    if "conv" in r["op"]: print(json.dumps(r), flush=True)
PY
