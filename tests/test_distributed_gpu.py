"""RCCL variant of tests/test_distributed_cpu.py: one process per visible GPU, backend "nccl" (= RCCL on ROCm), the
per-rank compute is the HIP path itself.  forward (one all-gather), forward_overlapped (M-tiled, exchange under compute) and
forward_direct (grouped send / recv into the final place) against the unsharded HIP forward and the oracle.  Skips on a one-GPU box:
the first multi-GPU lease exercises it."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(asym, K=1024, N=4096, gs=128, w_bit=4):
    g = torch.Generator().manual_seed(17)
    qweight = torch.randint(-2 ** 31, 2 ** 31 - 1, (K * w_bit // 32, N), generator=g, dtype=torch.int64).to(torch.int32)
    scales = (torch.rand((K // gs, N), generator=g) * 0.01 + 0.005).bfloat16()
    if asym:
        zeros = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // gs, N * w_bit // 32), generator=g, dtype=torch.int64).to(torch.int32)
    else:
        zeros = (scales.float() * torch.rand((K // gs, N), generator=g) * 15).bfloat16()
    bias = torch.randn(N, generator=g).bfloat16()
    g_idx = torch.arange(K, dtype=torch.int32) // gs
    return qweight, scales, zeros, g_idx, bias, w_bit, gs, g


def _worker(rank, world, port, asym, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from bitorch_engine.distributed import ColumnShardedMPQLinear
    from bitorch_engine.extensions import q_linear_cuda
    # the first multi-GPU lease must PROVE the exchange ran over RCCL with every rank in it
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    ids = torch.empty(world, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(ids, torch.tensor([rank], dtype=torch.int32, device=dev))
    saw_world = (dist.get_backend() == "nccl" and dist.get_world_size() == world and int(ones.item()) == world
                 and ids.cpu().tolist() == list(range(world)))
    qweight, scales, zeros, g_idx, bias, w_bit, gs, g = _make(asym)
    d = lambda t: t.to(dev)
    layer = ColumnShardedMPQLinear(d(qweight), d(scales), d(zeros), d(g_idx), d(bias), w_bit, gs, asym, rank, world)
    ok = True
    for M in (1, 7, 300):
        x = torch.randn((M, qweight.shape[0] * 32 // w_bit), generator=g).bfloat16().to(dev)
        full = q_linear_cuda.mpq_forward_impl(x, d(qweight), d(scales), d(zeros), None, w_bit, asym, gs, d(bias))
        tol = 1e-3 * full.float().abs().max() + full.float().abs() * 2.0 ** -7
        for name, fn in (("forward", lambda: layer(x)), ("overlapped", lambda: layer.forward_overlapped(x, m_tile=128)), ("direct", lambda: layer.forward_direct(x, m_tile=128))):
            y = fn()
            torch.cuda.synchronize()
            good = y.shape == full.shape and bool(((y.float() - full.float()).abs() <= tol).all())
            if not good:
                print(f"rank {rank}: {name} M={M} differs", flush=True)
            ok = ok and good
    torch.save({"ok": ok, "saw_world": saw_world, "world": world}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("asym", [False, True])
def test_column_sharded_layer_rccl(tmp_path, asym):
    world = torch.cuda.device_count()
    if world < 2:
        pytest.skip("needs at least two GPUs (the RCCL exchange); covered under gloo by tests/test_distributed_cpu.py")
    world = min(world, 8)
    port = _free_port()
    mp.spawn(_worker, args=(world, port, asym, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(os.path.join(tmp_path, f"r{r}.pt"))
        assert res["saw_world"] and res["world"] == world, f"rank {r}: the RCCL group did not hold all {world} ranks"
        assert res["ok"], f"rank {r}: sharded output differs from the unsharded HIP forward"
