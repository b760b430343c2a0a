#!/bin/bash
# Round 5 evidence pass: GPU tests, smoke, the bench line (full extras), rocprofv3 kernel stats of the headline command and of the short bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD; TAG=${TAG:-r05}
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log; grep -E "^E  |^FAILED" gpurun_out/${TAG}_pytest.log | head -20
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench (full)"; timeout 1200 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cat gpurun_out/${TAG}_bench.json; wc -c gpurun_out/${TAG}_bench.json; cp gpurun_out/bench_extras.json gpurun_out/${TAG}_bench_extras.json
echo "== rocprof headline"; (cd /tmp && rm -rf /tmp/prof1 && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -o $TAG -- python $R/bench.py --no-extras --no-cpu-baseline > /tmp/prof1_bench.json 2> /tmp/prof1.err)
for f in $(find /tmp/prof1 -name "*kernel_stats*.csv" | head -1); do cp $f gpurun_out/${TAG}_kernel_stats_headline.csv; done
cp /tmp/prof1_bench.json gpurun_out/${TAG}_prof_bench_headline.json 2>/dev/null; head -6 gpurun_out/${TAG}_kernel_stats_headline.csv
echo "== rocprof short bench"; (cd /tmp && rm -rf /tmp/prof2 && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -o $TAG -- python $R/bench.py --short --no-cpu-baseline > /tmp/prof2_bench.json 2> /tmp/prof2.err)
for f in $(find /tmp/prof2 -name "*kernel_stats*.csv" | head -1); do cp $f gpurun_out/${TAG}_kernel_stats_short_bench.csv; done
head -14 gpurun_out/${TAG}_kernel_stats_short_bench.csv | cut -c1-200
