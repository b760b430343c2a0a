#!/bin/bash
# PMC passes over the one-launch matrix-pipe conv (xnor_conv_mfma_kernel) at B = 128, 512 -> 512 3x3 on 7x7 (tools/conv_fp4_prof.py, 200 eager calls).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; cd /tmp
pass() { name=$1; shift; rm -rf /tmp/pmcc_$name; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcc_$name -o p -- python $R/tools/conv_fp4_prof.py 128 > /tmp/pmcc_$name.log 2>&1; f=$(find /tmp/pmcc_$name -name "*counter_collection.csv" | head -1); echo "== $name"; python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if "xnor_conv_mfma_kernel" not in k: continue
    agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
}
{
echo "# binary conv 512->512 3x3 on 7x7, B = 128, xnor_conv_mfma_kernel<3, 2, 8> (512 workgroups)"
pass a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pass b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD
pass d FETCH_SIZE
pass e WRITE_SIZE
pass f GRBM_GUI_ACTIVE GRBM_COUNT
} 2>&1 | tee $R/gpurun_out/r06_pmc_conv.txt
