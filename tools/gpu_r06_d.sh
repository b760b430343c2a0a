#!/bin/bash
# Round 6, pass D: the MFMA-pipe probe of the dense GEMM (variants dense_lab8 / dense_lab9: the loop's own instruction stream on operands that stay in LDS,
# no global traffic, no stores) against the product and the no-epilogue-stores build (dense_lab1), both launches timed through bie_mpq_forward.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/dense_lab.py product dense_lab1 dense_lab8 dense_lab9 product dense_lab8 > gpurun_out/r06_d_dense_probe.txt 2>&1; cat gpurun_out/r06_d_dense_probe.txt
