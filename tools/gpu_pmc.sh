#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(SQ|TCC|TCP|GRBM|TA|TD)_[A-Z0-9_]+" | sort -u | tr -d ' ' | tr '\n' ' ' > $R/gpurun_out/counters.txt
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -o p -- python $R/tools/gemm_only.py ${M:-4096} ${DT:-bf16} > /tmp/pmc_$name.log 2>&1; f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1); echo "== $name $f"; python - "$f" <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r.get("Kernel_Name", "")[:60]
    if "bie::" not in k: continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
}
pass a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pass b SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS
pass c FETCH_SIZE GRBM_GUI_ACTIVE
pass d TCC_HIT_sum TCC_MISS_sum WRITE_SIZE
pass e TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum
tail -3 /tmp/pmc_a.log
