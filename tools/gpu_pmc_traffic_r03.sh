#!/bin/bash
# HBM traffic of the layer-list decode launch per LAYER (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per shape) ->
# gpurun_out/r03_pmc_gemv.json, stamped with the hash of the kernel sources bench.py checks (copy it to profiles/).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; cd /tmp
for spec in "4096 4096 96" "4096 11008 40" "11008 4096 40"; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=/tmp/pmcl_${spec// /x}_$ctr; rm -rf $d
    LIST_AB_NOGRAPH=1 timeout 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $d -o p -- python $R/tools/list_ab.py shape $spec > $d.log 2>&1
  done
done
python - "$R" <<'PY'
import csv, glob, json, sys, os, collections, hashlib
R = sys.argv[1]
out = {"source": "tools/gpu_pmc_traffic_r03.sh: rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> -- python tools/list_ab.py shape K N L (L distinct layers in ONE list launch, M=1, no graph, 3 launches)",
       "units": "per LAYER = per launch / L; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE in KiB as reported (calibrated exact on the dequant kernel in round 1)",
       "shapes": {}}
h = hashlib.sha256()
for f in ("mpq_list.hip", "mpq_gemv_lut.hip", "mpq_gemv.hip", "mpq_dequant.cuh"):
    h.update(open(os.path.join(R, "bitorch-engine_amd", "csrc", f), "rb").read())
out["kernel_source_sha"] = h.hexdigest()[:16]
for (K, N, L) in ((4096, 4096, 96), (4096, 11008, 40), (11008, 4096, 40)):
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(f"/tmp/pmcl_{K}x{N}x{L}_{ctr}/**/*counter_collection.csv", recursive=True)
        acc = []
        for f in fs:
            for r in csv.DictReader(open(f)):
                if "bie::mpq_list_kernel" in r.get("Kernel_Name", "") and r["Counter_Name"] == ctr:
                    acc.append(float(r["Counter_Value"])); vals["kernel"] = r["Kernel_Name"][:70]
        if acc:
            vals[ctr] = sum(acc) / len(acc); vals["n_" + ctr] = len(acc)
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        fetch_b = vals["FETCH_SIZE"] * 1024 * 2 / L
        write_b = vals["WRITE_SIZE"] * 1024 / L
        alg = K * N // 2 + 4 * (K // 128) * N + 2 * K + 2 * N
        out["shapes"][f"list{L}_{K}x{N}"] = {"kernel": vals["kernel"], "layers_per_launch": L, "fetch_bytes_corrected": round(fetch_b), "write_bytes": round(write_b),
                                            "hbm_bytes_per_layer": round(fetch_b + write_b), "algorithmic_bytes_per_layer": alg,
                                            "ratio": round((fetch_b + write_b) / alg, 4), "dispatches": vals["n_FETCH_SIZE"]}
json.dump(out, open(os.path.join(R, "gpurun_out", "r03_pmc_gemv.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
