"""Exercise bench.py's configs[4] leg (column-sharded 8192x28672, M=4096) on ONE GPU: RCCL process group of size 1, so the whole
code path (local GEMM, all_gather_into_tensor, tiled overlap on the communication stream) runs; the numbers are not a scaling result."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29641")
import torch, torch.distributed as dist
import bench
from bench_sharded import bench_column_sharded
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1)
B = bench.Bench(dev)
print(json.dumps(bench_column_sharded(1, 0, dev, M=4096, K=8192, N=28672, reps=3)))
dist.barrier(); dist.destroy_process_group()
