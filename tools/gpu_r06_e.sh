#!/bin/bash
# Round 6, pass E: the whole GPU suite + smoke + the short bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${TAG:-r06_e}
echo "== pytest"; timeout 2400 python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log; grep -E "^E  |^FAILED" gpurun_out/${TAG}_pytest.log | head -40
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench short"; timeout 1200 python bench.py --short > gpurun_out/${TAG}_bench_short.json 2> gpurun_out/${TAG}_bench_short.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_e_bench_short.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d.get("roofline_gemm", {}).get("frac"), d.get("decode_step_llama7b"))
print(json.dumps(d.get("summary"), indent=0)[:1500])
PY
