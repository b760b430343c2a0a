#!/bin/bash
# Round 6 evidence pass: PMC passes (stamped with the kernel-source hashes bench.py checks), GPU tests, smoke, the bench line (full extras), rocprofv3 kernel stats
# of the headline command and of the short bench.  The PMC JSONs are written first so that the bench line of the same box carries `traffic`.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; TAG=${TAG:-r06}
echo "== PMC headline / lists"; bash tools/gpu_pmc_r06.sh > gpurun_out/${TAG}_pmc_gemv.log 2>&1; tail -5 gpurun_out/${TAG}_pmc_gemv.log | cut -c1-300; cp gpurun_out/r06_pmc_gemv.json profiles/r06_pmc_gemv.json
echo "== PMC GEMM"; bash tools/gpu_pmc_gemm_r06.sh > gpurun_out/${TAG}_pmc_gemm.log 2>&1; python tools/pmc_gemm_json_r06.py 2>&1 | tail -3 | cut -c1-300; cp profiles/r06_pmc_gemm.json gpurun_out/r06_pmc_gemm.json
echo "== pytest"; timeout 2400 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log; grep -E "^E  |^FAILED" gpurun_out/${TAG}_pytest.log | head -20
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench (full)"; timeout 1500 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cat gpurun_out/${TAG}_bench.json; wc -c gpurun_out/${TAG}_bench.json; cp gpurun_out/bench_extras.json gpurun_out/${TAG}_bench_extras.json
echo "== bench via the launcher, world of one forced through the multi-rank path"; BIE_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 50 --warmup 5 > gpurun_out/${TAG}_bench_forced_dist.json 2> gpurun_out/${TAG}_bench_forced_dist.err; cut -c1-400 gpurun_out/${TAG}_bench_forced_dist.json
echo "== rocprof headline"; (cd /tmp && rm -rf /tmp/prof1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o $TAG -- python $R/bench.py --no-extras --no-cpu-baseline > /tmp/prof1_bench.json 2> /tmp/prof1.err)
for f in $(find /tmp/prof1 -name "*kernel_stats*.csv" | head -1); do cp $f gpurun_out/${TAG}_kernel_stats_headline.csv; done
cp /tmp/prof1_bench.json gpurun_out/${TAG}_prof_bench_headline.json 2>/dev/null; head -6 gpurun_out/${TAG}_kernel_stats_headline.csv | cut -c1-200
echo "== rocprof short bench"; (cd /tmp && rm -rf /tmp/prof2 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -o $TAG -- python $R/bench.py --short --no-cpu-baseline > /tmp/prof2_bench.json 2> /tmp/prof2.err)
for f in $(find /tmp/prof2 -name "*kernel_stats*.csv" | head -1); do cp $f gpurun_out/${TAG}_kernel_stats_short_bench.csv; done
head -14 gpurun_out/${TAG}_kernel_stats_short_bench.csv | cut -c1-200
