#!/bin/bash
# SQ instruction counters of the small-batch (matrix-pipe) and W2 (pair) forms of the lookup GEMV at 4096x11008
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; cd /tmp
for cfg in "bf16 4096 11008 8 4" "bf16 4096 11008 1 2" "f16 4096 11008 1 4"; do
  tag=${cfg// /_}; rm -rf /tmp/pq_$tag
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES --output-format csv -d /tmp/pq_$tag -o p -- python $R/tools/pmc_gemv.py $cfg > /tmp/pq_$tag.log 2>&1
  f=$(find /tmp/pq_$tag -name "*counter_collection.csv" | head -1)
  echo "== $cfg"; python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if "bie::mpq_gemv" not in k: continue
    agg[k[:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
done
