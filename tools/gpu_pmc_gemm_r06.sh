#!/bin/bash
# PMC passes over the W4A16 MFMA GEMM at the TIMED shape of the bench line (M = 4096, 4096 -> 4096, bf16; tools/gemm_only.py, no graph) and at 4096 -> 11008: bash tools/gpu_pmc_gemm_r06.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; cd /tmp
pass() { name=$1; dt=$2; shift 2; rm -rf /tmp/pmcg_${name}_$dt; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcg_${name}_$dt -o p -- python $R/tools/gemm_only.py 4096 bf16 4096 $NN > /tmp/pmcg_${name}_$dt.log 2>&1; f=$(find /tmp/pmcg_${name}_$dt -name "*counter_collection.csv" | head -1); echo "== $name $dt"; python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if "mpq_gemm_kernel" not in k and "mpq_dense_gemm_kernel" not in k: continue
    agg[k[:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
}
for NN in 4096 11008; do
export NN
{
echo "# M=4096 K=4096 N=$NN bf16 (the dt column below carries N)"
for dt in n$NN; do
pass a $dt SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
pass b $dt SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD
pass d $dt FETCH_SIZE
pass e $dt WRITE_SIZE
pass f $dt GRBM_GUI_ACTIVE GRBM_COUNT
done
rm -rf /tmp/pmcg_stats; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmcg_stats -o p -- python $R/tools/gemm_only.py 4096 bf16 4096 $NN > /tmp/pmcg_stats.log 2>&1
echo "== kernel stats (bf16)"; f=$(find /tmp/pmcg_stats -name "*kernel_stats.csv" | head -1); grep -i "mpq_gemm\|mpq_dense\|mpq_dequant_frag\|Name" "$f" | cut -c1-260
} 2>&1 | tee $R/gpurun_out/r06_pmc_gemm_N$NN.txt
done
