#!/bin/bash
# Round evidence pass: full GPU pytest, bench line, rocprofv3 kernel stats of the bench command, PMC passes for the GEMV.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; TAG=${TAG:-r01}
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/pytest.log 2>&1; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest.log | tail -30; grep -E "^E  " gpurun_out/pytest.log | head -20
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "== bench"; timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/bench.err; cat gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/bench.err
echo "== rocprof"; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o $TAG -- python $R/bench.py --no-cpu-baseline > /tmp/prof_bench.json 2> /tmp/prof.err)
for f in $(find /tmp/prof -name "*kernel_stats*.csv" | head -1); do cp $f gpurun_out/${TAG}_kernel_stats.csv; done
cp /tmp/prof_bench.json gpurun_out/${TAG}_prof_bench.json 2>/dev/null; head -8 gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null
cd /tmp
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -o p -- python $R/tools/pmc_gemv.py bf16 > /tmp/pmc_$name.log 2>&1; f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1); echo "== pmc $name"; python - "$f" "$R/gpurun_out/${TAG}_pmc_gemv_$name.json" <<'PY'
import csv, sys, collections, json
f, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r.get("Kernel_Name", "")
    if "bie::" not in k: continue
    agg[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: dict({c: round(sum(v) / len(v), 2) for c, v in d.items()}, n=len(next(iter(d.values())))) for k, d in agg.items()}
json.dump(res, open(out, "w"), indent=1)
for k, v in res.items(): print(k, v)
PY
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD
pass tcc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum
tail -3 /tmp/pmc_fetch.log
