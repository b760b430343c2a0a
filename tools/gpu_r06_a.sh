#!/bin/bash
# Round 6, pass A: the new tests (STE backward, sibling contract, exl2 reload, element-wise fp16 gates), the forced-distributed bench path on one GPU,
# and this box's baseline numbers (short bench) before any kernel work.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=r06_a
echo "== new tests"; timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_distributed_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/${TAG}_pytest_new.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_new.log; grep -E "^E  |^FAILED" gpurun_out/${TAG}_pytest_new.log | head -40
echo "== touched parity tests"; timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "second_checkpoint or sibling or group_siblings or full_size or the_list_instances or state_dict" > gpurun_out/${TAG}_pytest_parity.log 2>&1; tail -3 gpurun_out/${TAG}_pytest_parity.log; grep -E "^E  |^FAILED" gpurun_out/${TAG}_pytest_parity.log | head -40
echo "== bench, world of one forced through the multi-rank path"; BIE_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_forced_dist.json 2> gpurun_out/${TAG}_bench_forced_dist.err; cat gpurun_out/${TAG}_bench_forced_dist.json; tail -3 gpurun_out/${TAG}_bench_forced_dist.err
echo "== bench, --gpus 1 under torch.distributed.run (WORLD_SIZE=1 set by the launcher)"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/${TAG}_bench_tdr1.json 2> gpurun_out/${TAG}_bench_tdr1.err; cat gpurun_out/${TAG}_bench_tdr1.json | cut -c1-600
echo "== bench short"; timeout 1200 python bench.py --short > gpurun_out/${TAG}_bench_short.json 2> gpurun_out/${TAG}_bench_short.err; cat gpurun_out/${TAG}_bench_short.json; cp gpurun_out/bench_extras.json gpurun_out/${TAG}_bench_extras.json
