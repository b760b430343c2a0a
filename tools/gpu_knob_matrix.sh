#!/bin/bash
# Every dispatch knob's OTHER setting against the parity suite (the forms behind the knobs are all product code): one pytest run per setting over the
# tests that exercise the form.  -> gpurun_out/<TAG>_knob_matrix.txt (TAG defaults to r06)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; out=gpurun_out/${TAG:-r06}_knob_matrix.txt; : > $out
run() { envs="$1"; sel="$2"; echo "== $envs   -k \"$sel\"" >> $out; env $envs timeout 900 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "$sel" 2>&1 | tail -2 >> $out; }
run "BIE_LUTM_NW16=8 BIE_LUTM_NW32=8" "list_forward or lut"
run "BIE_LIST_D16=0" "list_forward or special"
run "BIE_LIST_ALG=1" "list_forward and not x_sharing"   # (the x-sharing tests assert the DEFAULT routing: the algebraic opt-in takes precedence at one / two rows)
run "BIE_DECODE_INLINE=2" "gemv or forward_sizes or grouped or special or module"
run "BIE_DECODE_INLINE=0" "gemv or forward_sizes or grouped or special or module"
run "BIE_GEMM_DENSE=0" "gemm or prefill or forward_sizes or special"
run "BIE_GEMM_DENSE=2" "gemm or prefill or forward_sizes or special"
run "BIE_FP4_MIN_M=0 BIE_FP4_CONV_MIN_ROWS=0" "binary"
# round 6 knobs
run "BIE_GEMM_DENSE_TABLE=0" "gemm or prefill or forward_sizes or special"
run "BIE_GEMM_PLAN_TABLE=0" "gemm or prefill or forward_sizes or special"
run "BIE_LUT_RB2=0" "mpq_forward_vs_oracle or forward_sizes or special or gemm"   # (test_lone_calls_of_17_to_32_rows asserts the DEFAULT routing)
run "BIE_DQ_FPW=8" "gemm or prefill or forward_sizes or special"
run "BIE_LIST_W2_NW=8" "list_forward or w2 or W2"
run "BIE_LUTM_XS_MIN_M=0" "(list_forward or list_instances) and not x_sharing"
run "BIE_DECODE_INLINE_MIN_MB=96" "gemv or forward_sizes or grouped or special or module"
run "BIE_CONV_FUSED_MAX_ROWS=0 BIE_CONV_MFMA_MAX_ROWS=0" "conv"
# (BIE_AUTO_GROUP=0, BIE_EXL2_DIRECT=0 and BIE_EXL2_XP=0 are not rows: the tests that exercise those forms assert the DEFAULT routing -- launch counters,
#  multi-row lists that only the direct form serves -- and fail on the expectation, not on a value; their value tests run in the lines below)
run "BIE_AUTO_GROUP=0" "module_tree_matches or state_dict or forward_sizes or layer_forward"
run "BIE_EXL2_DIRECT=0" "mbwq_exl2_dequant_and_forward or full_size_exl2 or exl2_decode_long_k or group_structures"
run "BIE_EXL2_XP=0" "mbwq_exl2_dequant_and_forward or full_size_exl2 or exl2_decode_long_k or group_structures"
run "BIE_EXL2_DENSE_MIN_M=1000000" "exl2 and not mbwq_exl2_dequant_and_forward"  # (that test moves the same switch itself and compares the two forms bit for bit)
run "BIE_EXL2_MFMA=0" "exl2"
run "BIE_I8_PIPE=0" "q4 or q8 or int"
run "BIE_ACT_ORDER_SORTED=0" "act_order or gidx or g_idx"
cat $out
