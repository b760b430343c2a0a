#!/bin/bash
# Round 6 counters of the headline launch (96 x 4096x4096 W4 g128 bf16, M = 1, ONE list launch; the D16 form): separate rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE, two SQ groups, GRBM) + a kernel-trace pass for the duration -> gpurun_out/r06_pmc_gemv.json (copy to profiles/).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; cd /tmp
pass() { name=$1; shift; d=/tmp/pmc6_$name; rm -rf $d; LIST_AB_NOGRAPH=1 timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $d -o p -- python $R/tools/list_ab.py shape ${SHAPE:-4096 4096 96} > $d.log 2>&1; }
for SHAPE in "4096 4096 96" "4096 11008 40" "11008 4096 40"; do
  export SHAPE; tag=${SHAPE// /x}
  pass ${tag}_fetch FETCH_SIZE
  pass ${tag}_write WRITE_SIZE
  if [ "$SHAPE" = "4096 4096 96" ]; then
    pass ${tag}_sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
    pass ${tag}_sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD
    pass ${tag}_grbm GRBM_GUI_ACTIVE GRBM_COUNT
  fi
done
d=/tmp/pmc6_trace; rm -rf $d; LIST_AB_NOGRAPH=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python $R/tools/list_ab.py shape 4096 4096 96 > $d.log 2>&1
python - "$R" <<'PY'
import csv, glob, json, sys, os, collections, hashlib
R = sys.argv[1]
def counters(tag):
    acc = collections.defaultdict(list); kern = None
    for f in glob.glob(f"/tmp/pmc6_{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "bie::mpq_list_kernel" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"])); kern = r["Kernel_Name"][:80]
    return {k: sum(v) / len(v) for k, v in acc.items()}, kern
h = hashlib.sha256()
for f in ("mpq_list.hip", "mpq_gemv_lut.hip", "mpq_gemv.hip", "mpq_dequant.cuh"):
    h.update(open(os.path.join(R, "bitorch-engine_amd", "csrc", f), "rb").read())
out = {"source": "tools/gpu_pmc_r06.sh: rocprofv3 --kernel-trace --pmc <one group per pass> -- python tools/list_ab.py shape K N L (L distinct layers, ONE list launch, M=1, no graph, 3 launches averaged)",
       "units": "per LAYER = per launch / L; FETCH_SIZE in KiB doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE KiB as reported; SQ_* summed over the chip; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* in quad-cycles; SQ_LDS_IDX_ACTIVE in cycles summed over 256 CUs; GRBM_GUI_ACTIVE summed over 8 XCDs",
       "kernel_source_sha": h.hexdigest()[:16], "shapes": {}}
for (K, N, L) in ((4096, 4096, 96), (4096, 11008, 40), (11008, 4096, 40)):
    tag = f"{K}x{N}x{L}"
    fe, kern = counters(tag + "_fetch"); wr, _ = counters(tag + "_write")
    if "FETCH_SIZE" in fe and "WRITE_SIZE" in wr:
        fb, wb = fe["FETCH_SIZE"] * 2048 / L, wr["WRITE_SIZE"] * 1024 / L
        alg = K * N // 2 + 4 * (K // 128) * N + 2 * K + 2 * N
        out["shapes"][f"list{L}_{K}x{N}"] = {"kernel": kern, "layers_per_launch": L, "fetch_bytes_corrected": round(fb), "write_bytes": round(wb), "hbm_bytes_per_layer": round(fb + wb),
                                            "algorithmic_bytes_per_layer": alg, "ratio": round((fb + wb) / alg, 4)}
s1, _ = counters("4096x4096x96_sq1"); s2, _ = counters("4096x4096x96_sq2"); g, _ = counters("4096x4096x96_grbm")
if s1 and s2 and g:
    ww = 96 * 4096 * 4096 / 64.0  # wave-weights per launch
    cyc = g["GRBM_GUI_ACTIVE"] / 8
    out["headline_sq"] = {"valu_per_weight": round(s1["SQ_INSTS_VALU"] / ww, 3), "lds_per_weight": round(s2["SQ_INSTS_LDS"] / ww, 3), "salu_per_weight": round(s2["SQ_INSTS_SALU"] / ww, 3),
                          "kernel_cycles_per_xcd": round(cyc), "valu_busy_fraction": round(s1["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc, 4), "lds_array_busy_fraction": round(s2["SQ_LDS_IDX_ACTIVE"] / 256 / cyc, 4),
                          "wave_wait_fraction": round(s1["SQ_WAIT_ANY"] / s1["SQ_WAVE_CYCLES"], 4), "issue_stall_fraction": round(s1["SQ_WAIT_INST_ANY"] / s1["SQ_WAVE_CYCLES"], 4),
                          "lds_bank_conflict_cycles": s2.get("SQ_LDS_BANK_CONFLICT"), "waves": s1["SQ_WAVES"], "raw": {**s1, **s2, **g}}
for f in glob.glob("/tmp/pmc6_trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mpq_list_kernel" in r.get("Name", ""):
            out["kernel_trace"] = {"name": r["Name"][:90], "calls": r.get("Calls"), "avg_ns": r.get("AverageNs"), "total_ns": r.get("TotalDurationNs")}
            if "headline_sq" in out: out["headline_sq"]["effective_clock_ghz_profiled"] = round(out["headline_sq"]["kernel_cycles_per_xcd"] / float(r["AverageNs"]), 3)
    import shutil; shutil.copy(f, os.path.join(R, "gpurun_out", "r06_kernel_stats_list96_nograph.csv"))
json.dump(out, open(os.path.join(R, "gpurun_out", "r06_pmc_gemv.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "source" and k != "units"}, indent=1)[:3500])
PY
