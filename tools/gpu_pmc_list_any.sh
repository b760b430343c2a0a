#!/bin/bash
# SQ / LDS counters of one list instance: LIST_AB_WBIT / LIST_AB_DT select it.  usage: LIST_AB_WBIT=2 bash tools/gpu_pmc_list_any.sh <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; tag=${1:-w}; cd /tmp
pass() { name=$1; shift; d=/tmp/pmcl_${tag}_$name; rm -rf $d; LIST_AB_NOGRAPH=1 timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $d -o p -- python $R/tools/list_ab.py shape 4096 4096 96 > $d.log 2>&1; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
python - "$tag" <<'PY' | tee $R/gpurun_out/r05_pmc_list_$tag.txt
import csv, glob, sys, collections, json, os
tag = sys.argv[1]
def counters(name):
    acc = collections.defaultdict(list); kern = None
    for f in glob.glob(f"/tmp/pmcl_{tag}_{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "bie::mpq_list_kernel" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"])); kern = r["Kernel_Name"][:70]
    return {k: sum(v) / len(v) for k, v in acc.items()}, kern
s1, k = counters("sq1"); s2, _ = counters("sq2"); g, _ = counters("grbm")
wb = int(os.environ.get("LIST_AB_WBIT", "4"))
ww = 96 * 4096 * 4096 / 64.0
cyc = g["GRBM_GUI_ACTIVE"] / 8
print(json.dumps({"kernel": k, "w_bit": wb, "dt": os.environ.get("LIST_AB_DT", "bf16"), "valu_per_weight": round(s1["SQ_INSTS_VALU"] / ww, 3), "lds_per_weight": round(s2["SQ_INSTS_LDS"] / ww, 3),
 "salu_per_weight": round(s2["SQ_INSTS_SALU"] / ww, 3), "smem_per_weight": round(s2["SQ_INSTS_SMEM"] / ww, 4), "kernel_cycles_per_xcd": round(cyc),
 "valu_busy": round(s1["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, 4), "lds_array_busy": round(s2["SQ_LDS_IDX_ACTIVE"] / 256 / cyc, 4),
 "wave_wait": round(s1["SQ_WAIT_ANY"] / s1["SQ_WAVE_CYCLES"], 4), "issue_stall": round(s1["SQ_WAIT_INST_ANY"] / s1["SQ_WAVE_CYCLES"], 4),
 "lds_conflict_cycles": s2.get("SQ_LDS_BANK_CONFLICT"), "waves": s1["SQ_WAVES"], "avg_resident_waves_per_simd": round(s1["SQ_WAVE_CYCLES"] * 4 / 1024 / cyc, 2)}))
PY
