#!/usr/bin/env python3
"""bie_mpq_forward over the whole row range (W4 g128 bf16): us per call and the fraction of BOTH rooflines (HBM for the algorithmic bytes, MFMA for 2MKN), so that a cliff
   between dispatch regimes (lookup <= 16 rows, fused MFMA kernel, dense two-launch form from 1024) shows.  usage: python tools/m_sweep_r06.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bitorch-engine_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import Bench, alg_bytes  # noqa: E402

B = Bench(torch.device("cuda", 0))
for (k, n) in ((4096, 11008), (4096, 4096), (11008, 4096)):
    for M in (1, 2, 8, 16, 17, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1023, 1024, 1536, 2048, 4096):
        nl = 12 if M >= 256 else 40
        r = B.gemm(M, k, n, nl, 6 if M >= 1024 else 10, 5)
        us = r["us_per_launch"]
        print(json.dumps({"K": k, "N": n, "M": M, "us": us, "hbm_frac": round(alg_bytes(M, k, n) / us / 1e3 / 8000.0, 3), "mfma_frac": round(2.0 * M * k * n / us / 1e6 / 2500.0, 3)}), flush=True)
