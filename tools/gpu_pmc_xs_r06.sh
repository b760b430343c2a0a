#!/bin/bash
# Round 6 counters of the x-sharing matrix-pipe list kernel (40 x 4096x11008 W4 g128, 16 rows, fp16 and bf16) and, beside it, of the k-split form
# (BIE_LUTM_XS_MIN_M=0): separate rocprofv3 --pmc passes (two SQ groups, GRBM, FETCH_SIZE) + kernel-trace for the duration -> gpurun_out/r06_pmc_xs.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; cd /tmp
pass() { name=$1; shift; d=/tmp/pmcx_$name; rm -rf $d; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $d -o p -- python $R/tools/xs_run.py 4096 11008 40 $ROWS $DT > $d.log 2>&1; }
for DT in f16 bf16; do for FORM in xs ksplit; do for ROWS in 16 32; do
  export DT ROWS; if [ $FORM = ksplit ]; then export BIE_LUTM_XS_MIN_M=0; else unset BIE_LUTM_XS_MIN_M; fi
  t=${DT}_${FORM}_${ROWS}
  pass ${t}_sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
  pass ${t}_sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_MFMA
  pass ${t}_grbm GRBM_GUI_ACTIVE GRBM_COUNT
  pass ${t}_fetch FETCH_SIZE
  d=/tmp/pmcx_${t}_trace; rm -rf $d; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $R/tools/xs_run.py 4096 11008 40 $ROWS $DT > $d.log 2>&1
done; done; done
python - "$R" <<'PY'
import csv, glob, sys, os, collections
R = sys.argv[1]
def counters(tag):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"/tmp/pmcx_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "lutm" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
lines = ["x-sharing (xs) against k-split matrix-pipe list kernel, 40 x 4096x11008 W4 g128, rocprofv3 counters (tools/gpu_pmc_xs_r06.sh; one group per pass, 3 launches averaged).",
         "per weight = per 4096 * 11008 * 40 / 64 wave-weights; busy fractions against GRBM_GUI_ACTIVE / 8 cycles per XCD; FETCH_SIZE doubled per MI355X_MICROARCH.md.", ""]
ww = 40 * 4096 * 11008 / 64.0
alg = 40 * (4096 * 11008 // 2 + 4 * 32 * 11008)
for dt in ("f16", "bf16"):
    for form in ("xs", "ksplit"):
        for rows in (16, 32):
            t = f"{dt}_{form}_{rows}"
            s1, s2, g, fe = counters(t + "_sq1"), counters(t + "_sq2"), counters(t + "_grbm"), counters(t + "_fetch")
            us = None
            for f in glob.glob(f"/tmp/pmcx_{t}_trace/**/*kernel_stats.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    if "lutm" in r.get("Name", ""): us = float(r["AverageNs"]) / 1e3
            if not (s1 and s2 and g): lines.append(f"{t}: counters missing"); continue
            cyc = g["GRBM_GUI_ACTIVE"] / 8
            lines.append(f"{dt} {form:6s} {rows:2d} rows: {us and round(us / 40, 2)} us per layer | per weight: VALU {s1['SQ_INSTS_VALU'] / ww:.2f}  LDS {s2['SQ_INSTS_LDS'] / ww:.2f}  SALU {s2['SQ_INSTS_SALU'] / ww:.2f}  "
                         f"VMEM_RD {s2['SQ_INSTS_VMEM_RD'] / ww:.3f}  MFMA {s2.get('SQ_INSTS_MFMA', 0) / ww:.3f} | VALU busy {s1['SQ_ACTIVE_INST_VALU'] * 4 / 1024 / cyc:.2f}  LDS array busy {s2['SQ_LDS_IDX_ACTIVE'] / 256 / cyc:.2f}  "
                         f"bank-conflict cycles / LDS active {s2.get('SQ_LDS_BANK_CONFLICT', 0) / max(s2['SQ_LDS_IDX_ACTIVE'], 1):.3f}  wave wait {s1['SQ_WAIT_ANY'] / s1['SQ_WAVE_CYCLES']:.2f}  waves {s1['SQ_WAVES']:.0f} | "
                         f"HBM fetch {fe.get('FETCH_SIZE', 0) * 2048 / alg:.3f} x algorithmic")
open(os.path.join(R, "gpurun_out", "r06_pmc_xs.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
