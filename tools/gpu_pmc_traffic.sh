#!/bin/bash
# HBM traffic of the decode GEMV per launch (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per layer shape) ->
# gpurun_out/r02_pmc_gemv.json, stamped with the hash of the kernel sources bench.py checks (copy it to profiles/).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; cd /tmp
for shape in "4096 4096" "4096 11008" "11008 4096"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/pmct_${shape// /x}_$ctr; rm -rf $d
    timeout 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $d -o p -- python $R/tools/pmc_gemv.py bf16 $shape > $d.log 2>&1
  done
done
python - "$R" <<'PY'
import csv, glob, json, sys, os, collections
R = sys.argv[1]
sys.path.insert(0, R)
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(R, "bench.py"))
out = {"source": "tools/gpu_pmc_traffic.sh: rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> -- python tools/pmc_gemv.py bf16 K N (26 distinct layers x 6 passes, M=1, no graph)",
       "units": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE in KiB as reported (calibrated exact on the dequant kernel in round 1)",
       "shapes": {}}
import hashlib
h = hashlib.sha256()
for f in ("mpq_gemv_lut.hip", "mpq_gemv.hip", "mpq_dequant.cuh"):
    h.update(open(os.path.join(R, "bitorch-engine_amd", "csrc", f), "rb").read())
out["kernel_source_sha"] = h.hexdigest()[:16]
for shape in ("4096x4096", "4096x11008", "11008x4096"):
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(f"/tmp/pmct_{shape}_{ctr}/**/*counter_collection.csv", recursive=True)
        acc = collections.defaultdict(list)
        for f in fs:
            for r in csv.DictReader(open(f)):
                if "bie::mpq_gemv" in r.get("Kernel_Name", "") and r["Counter_Name"] == ctr:
                    acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
        if acc:
            k = max(acc, key=lambda n: len(acc[n]))
            vals[ctr] = sum(acc[k]) / len(acc[k]); vals["kernel"] = k; vals["n_" + ctr] = len(acc[k])
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        K, N = [int(v) for v in shape.split("x")]
        fetch_b = vals["FETCH_SIZE"] * 1024 * 2
        write_b = vals["WRITE_SIZE"] * 1024
        alg = K * N // 2 + 4 * (K // 128) * N + 2 * K + 2 * N
        out["shapes"][shape] = {"kernel": vals["kernel"], "fetch_bytes_corrected": round(fetch_b), "write_bytes": round(write_b),
                                "hbm_bytes_per_launch": round(fetch_b + write_b), "algorithmic_bytes_per_launch": alg,
                                "ratio": round((fetch_b + write_b) / alg, 4), "dispatches": vals["n_FETCH_SIZE"]}
json.dump(out, open(os.path.join(R, "gpurun_out", "r02_pmc_gemv.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
