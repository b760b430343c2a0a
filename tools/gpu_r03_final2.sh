#!/bin/bash
# Round-3 (second session) closing evidence on one MI355X box: GPU test suite, smoke(), PMC of the fused GEMM (-> profiles/r03_pmc_gemm.json),
# PMC + kernel stats of the two ordered-asm MFMA kernels (FP4 binary GEMM, dense W4A16 GEMM), the default bench line, rocprofv3 stats of the bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 | tee gpurun_out/r03_pytest_gpu_tail.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== PMC of the GEMM the dispatch picks"; bash tools/gpu_pmc_gemm_r03.sh > /dev/null 2>&1; tail -3 gpurun_out/r03w_pmc_gemm.txt | cut -c1-200
python tools/pmc_gemm_json.py > /dev/null && cp profiles/r03_pmc_gemm.json gpurun_out/r03_pmc_gemm.json  # stamped with the current sources: the bench below attaches it
echo "== PMC new MFMA kernels"
cd /tmp
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcn_$name -o p -- python $R/tools/mfma_new_only.py > /tmp/pmcn_$name.log 2>&1; f=$(find /tmp/pmcn_$name -name "*counter_collection.csv" | head -1); echo "== $name"; python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "")
    if "xnor_fp4_gemm" not in k and "mpq_dense_gemm" not in k and "mpq_dequant_frag" not in k: continue
    agg[k[:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
}
{
echo "# 4096^3: xnor_fp4_gemm_kernel (1024 v_mfma_scale_f32_32x32x64_f8f6f4 per wave, 32 cycles each) and mpq_dense_gemm_kernel (4096 v_mfma_f32_32x32x16_bf16 per wave)"
pass a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY
pass c GRBM_GUI_ACTIVE GRBM_COUNT
pass d FETCH_SIZE
pass e WRITE_SIZE
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmcn_stats -o p -- python $R/tools/mfma_new_only.py > /tmp/pmcn_stats.log 2>&1
echo "== kernel stats"; f=$(find /tmp/pmcn_stats -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200
} 2>&1 | tee $R/gpurun_out/r03_pmc_mfma_new.txt | cut -c1-260
cd $R
echo "== bench (the driver's flags)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r03_bench.json') if l.startswith('{')][-1])
print({k: d[k] for k in ('metric', 'value', 'unit', 'ms_per_step')}); print(d['roofline']); print(d.get('roofline_gemm')); print(d.get('gemm_fused_form_4096x4096')); print(d['cpu_baseline'])
PY
echo "== rocprof kernel stats: full bench"
(cd /tmp && rm -rf /tmp/prof_full && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_full -o s -- python $R/bench.py --no-cpu-baseline > /tmp/prof_full.json 2> /tmp/prof_full.err)
f=$(find /tmp/prof_full -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r03_kernel_stats_full_bench.csv && head -6 $f | cut -c1-200
