"""W8A8 GEMM 4096^3 (bie_q8_gemm -> i8_pipe_gemm_kernel), 30 eager launches over four weight matrices: the target of the PMC passes of tools/gpu_pmc_int_r06.sh."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bitorch-engine_amd"))
import torch
from bitorch_engine.extensions import q_linear_cutlass as qc
dev = "cuda:0"
M = N = K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a8 = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
ws = [torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev) for _ in range(4)]
for i in range(30):
    qc.q8_forward(a8, ws[i % 4], False, 0.01, 0.01)
torch.cuda.synchronize()
