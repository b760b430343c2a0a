#!/bin/bash
# PMC passes over the W8A8 GEMM (i8_pipe_gemm_kernel, v_mfma_i32_32x32x32_i8) at 4096^3 -- VERDICT r5 next #9: MFMA busy, LDS busy, fetch.  Counters in their own
# runs (kernel-trace + pmc only), FETCH_SIZE / WRITE_SIZE in separate passes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; cd /tmp
pass() { name=$1; shift; rm -rf /tmp/pmci_$name; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmci_$name -o p -- python $R/tools/intgemm_only.py 4096 > /tmp/pmci_$name.log 2>&1; f=$(find /tmp/pmci_$name -name "*counter_collection.csv" | head -1); echo "== $name"; python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if "i8_pipe_gemm_kernel" not in k: continue
    agg[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
}
{
echo "# W8A8 4096^3, i8_pipe_gemm_kernel"
pass a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pass b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD
pass d FETCH_SIZE
pass e WRITE_SIZE
pass f GRBM_GUI_ACTIVE GRBM_COUNT
rm -rf /tmp/pmci_stats; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmci_stats -o p -- python $R/tools/intgemm_only.py 4096 > /tmp/pmci_stats.log 2>&1
echo "== kernel stats"; f=$(find /tmp/pmci_stats -name "*kernel_stats.csv" | head -1); grep -i "i8_pipe\|Name" "$f" | cut -c1-200
} 2>&1 | tee $R/gpurun_out/r06_pmc_int_gemm.txt
