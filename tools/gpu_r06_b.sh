#!/bin/bash
# Round 6, pass B: the fixed tests of pass A again + a plan sweep of the inline list form for lone / grouped launches with S = 1 / 2
# (whole K per workgroup, several units per wave: the pipelined body) -- BIE_DECODE_INLINE=2 forces the inline form at every size.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=r06_b
echo "== fixed tests"; timeout 900 python -m pytest tests/test_gpu_backward.py -m gpu -q -p no:cacheprovider > gpurun_out/${TAG}_pytest_new.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_new.log; grep -E "^E  |^FAILED" gpurun_out/${TAG}_pytest_new.log | head -20
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "second_checkpoint or full_size_exl2" > gpurun_out/${TAG}_pytest_parity.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_parity.log; grep -E "^E  |^FAILED" gpurun_out/${TAG}_pytest_parity.log | head -20
echo "== forced dist bench (bucketed gather)"; BIE_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 50 --warmup 5 > gpurun_out/${TAG}_bench_forced_dist.json 2> gpurun_out/${TAG}_bench_forced_dist.err; cut -c1-300 gpurun_out/${TAG}_bench_forced_dist.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_b_bench_forced_dist.json")); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["verified"], d["rccl"], d["config"]["parallelism"])
PY
{
for env in "BIE_DECODE_INLINE=0" "BIE_DECODE_INLINE=2" "BIE_DECODE_INLINE=2 BIE_INL_GPW=8 BIE_INL_NW=4" "BIE_DECODE_INLINE=2 BIE_INL_GPW=4 BIE_INL_NW=8" "BIE_DECODE_INLINE=2 BIE_INL_GPW=4 BIE_INL_NW=4" "BIE_DECODE_INLINE=2 BIE_INL_GPW=2 BIE_INL_NW=8" "BIE_DECODE_INLINE=2 BIE_INL_GPW=2 BIE_INL_NW=4" "BIE_DECODE_INLINE=2 BIE_INL_GPW=16 BIE_INL_NW=4 BIE_INL_H=1" "BIE_DECODE_INLINE=2 BIE_INL_GPW=8 BIE_INL_NW=8 BIE_INL_H=1"; do
  env $env timeout 300 python tools/inl_sweep.py 2>/dev/null
done
} > gpurun_out/${TAG}_inl_plan_sweep.txt 2>&1
cat gpurun_out/${TAG}_inl_plan_sweep.txt
