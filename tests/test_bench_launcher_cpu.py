"""bench.py's multi-rank launcher, rehearsed on the CPU (gloo): VERDICT r5 found `--gpus N` parsed and never read, so a bare
`python bench.py --gpus N` would have printed N single-GPU lines.  Both ways the driver's contract starts N ranks are run here with
`--dry-run` (no kernel, nothing measured): the one JSON line must say n_gpus == N and carry the world the collective itself saw."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith("{") and l.rstrip().endswith("}")]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def _check(line, n):
    assert line["n_gpus"] == n and line["dry_run"] is True and line["value"] is None
    assert line["rccl"]["backend"] == "gloo" and line["rccl"]["world_seen"] == n
    assert line["rccl"]["gathered_rows_rank_major"] is True
    assert line["c5"]["schedules_agree"] is True and line["c5"]["scaling"] == "strong"
    assert f"{n} column shards" in line["c5"]["workload"]


@pytest.mark.parametrize("n", [2, 3])
def test_bare_invocation_starts_the_ranks_itself(n):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-run", "--backend", "gloo", "--steps", "3", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1, f"exactly ONE JSON line (rank 0), got {len(lines)}: {p.stdout[-1000:]}"
    _check(lines[0], n)
    assert "starting the ranks myself" in p.stderr


def test_under_torch_distributed_run_as_the_driver_launches_it():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run", "--backend", "gloo"],
                       env=_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = _json_lines(p.stdout)
    assert len(lines) == 1
    _check(lines[0], 2)
    assert "starting the ranks myself" not in p.stderr  # WORLD_SIZE was set: this process IS a rank, no second launcher


def test_gpus_one_never_launches_or_joins_a_world():
    """N = 1 must stay the plain single-process path (today's line): no launcher, no process group.  Without a GPU it refuses loudly."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1"], env=_env(), capture_output=True, text=True, timeout=600)
    import torch
    if not torch.cuda.is_available():
        assert p.returncode != 0 and "needs a GPU" in p.stderr and "starting the ranks" not in p.stderr


def test_measurement_helper_is_not_in_the_product_package():
    src = open(os.path.join(ROOT, "bitorch-engine_amd", "bitorch_engine", "distributed.py")).read()
    assert "bench_column_sharded" not in src and "perf_counter" not in src
