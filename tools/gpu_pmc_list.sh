#!/bin/bash
# PMC passes over the layer-list launch (96 x 4096x4096 in one launch, no graph): bash tools/gpu_pmc_list.sh [tag]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; TAG=${1:-r03}; cd /tmp
pass() { name=$1; shift; LIST_AB_NOGRAPH=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcl_$name -o p -- python $R/tools/list_ab.py quick > /tmp/pmcl_$name.log 2>&1; f=$(find /tmp/pmcl_$name -name "*counter_collection.csv" | head -1); echo "== $name"; python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if "bie::" not in k: continue
    agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
}
{
pass a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE
pass c SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_WAVES_EQ_64
pass d FETCH_SIZE
pass e WRITE_SIZE
pass f GRBM_GUI_ACTIVE GRBM_COUNT
} 2>&1 | tee $R/gpurun_out/${TAG}_pmc_list.txt
