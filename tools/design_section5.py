#!/usr/bin/env python3
"""Print the rows of DESIGN.md section 5 from profiles/r06_bench.json + r06_bench_extras.json + the rocprofv3 kernel stats of the same box, so that the
   table is the evidence set and nothing else.  usage: python tools/design_section5.py"""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
d = json.load(open(os.path.join(P, "r06_bench.json")))
ex = json.load(open(os.path.join(P, "r06_bench_extras.json")))["extras"]
S = d["summary"]


def fr(k):
    return ex[k]["roofline"]["frac"]


def us(k, key="us_per_layer"):
    return ex[k][key]


prof = {}
for f, tag in (("r06_kernel_stats_headline.csv", "headline"), ("r06_kernel_stats_short_bench.csv", "short")):
    for r in csv.DictReader(open(os.path.join(P, f))):
        n = r["Name"]
        if "mpq_list_kernel<1, 0, 1, 16, 4, 73, false>" in n and tag == "headline":
            prof["headline_us"] = float(r["AverageNs"]) / 1e3
        if "mpq_dequant_frag_kernel" in n and tag == "short":
            prof["dequant_min_us"] = float(r["MinNs"]) / 1e3
        if "mpq_dense_gemm_kernel<1, 4, 4>" in n and tag == "short":
            prof["gemm_min_us"] = float(r["MinNs"]) / 1e3
r, g = d["roofline"], d["roofline_gemm"]
reg = d["regions"]["kernel_frac_min_median_max"]
e2 = {(e["M"], e["K"], e["N"]): e for e in ex["c3_exl2"] if "prefill" in e.get("op", "")}
el = {(e.get("K"), e.get("N")): e for e in ex["c3_exl2"] if "layer list" in e.get("op", "") and e.get("M") == 1}
cb = d["cpu_baseline"]["by_threads"]
rows = [
    ("**headline: 96 × 4096² W4 g128 bf16, M = 1, ONE list launch**", f"{r['avg_launch_us']:.1f} µs per launch ({r['us_per_layer']:.2f} µs per layer); {prof['headline_us']:.1f} µs under the profiler",
     f"**{r['frac']:.3f} of 8 TB/s** ({r['achieved']:,.0f} GB/s); cold {d['cold_start']['roofline_frac']:.3f}; five regions {reg[0]:.3f} / {reg[1]:.3f} / {reg[2]:.3f}",
     f"traffic {r['traffic'] / 1e6:.1f} MB = {r['traffic'] / r['alg_bytes_per_launch']:.3f} × algorithmic; stream-only floor of the same launch 0.80"),
    ("fp16 lists, one row (exact; round 6: arithmetic dequantisation on the matrix pipe) 96 × 4096² / 40 × 4096→11008", f"{S['f16_list_4096x4096']['us']:.2f} / {S['f16_list_4096x11008']['us']:.2f} µs per layer",
     f"{S['f16_list_4096x4096']['frac']:.3f} / {S['f16_list_4096x11008']['frac']:.3f}", "GreenBit checkpoints are fp16"),
    ("fp16 lists, algebraic form (opt-in)", f"{S['f16_list_alg_4096x4096']['us']:.2f} / {S['f16_list_alg_4096x11008']['us']:.2f} µs per layer",
     f"**{S['f16_list_alg_4096x4096']['frac']:.3f} / {S['f16_list_alg_4096x11008']['frac']:.3f}**", "§2 for what it costs in agreement with the oracle"),
    ("bf16 lists 40 × 4096→11008 / 40 × 11008→4096 / 6 × 8192→28672", f"{us('c2_list_4096x11008'):.2f} / {us('c2_list_11008x4096'):.2f} / {us('c5_list_8192x28672'):.1f} µs per layer",
     f"{fr('c2_list_4096x11008'):.3f} / {fr('c2_list_11008x4096'):.3f} / {fr('c5_list_8192x28672'):.3f}", ""),
    ("**batched decode lists, 40 × 4096→11008 at 2 / 8 / 16 / 32 rows** (matrix-pipe lookup form)",
     " / ".join(f"{us(f'c2_list_M{m}_4096x11008'):.2f}" for m in (2, 8, 16, 32)) + " µs per layer",
     " / ".join(f"{fr(f'c2_list_M{m}_4096x11008'):.3f}" for m in (2, 8, 16, 32)) + " (round 4: 0.51 / 0.50 / 0.46 / 0.30)",
     "bf16; round 6: x-sharing workgroups from 12 rows (`r06_lutm_xs.txt`); round 5: four-wave workgroups, column-pair table entries"),
    ("**the same lists in fp16** (round 6: packed-fp16 arithmetic instead of the table; the four column tiles of a workgroup share one LDS stage of x)",
     " / ".join(f"{us(f'f16_list_M{m}_4096x11008'):.2f}" for m in (2, 8, 16, 32)) + " µs per layer",
     "**" + " / ".join(f"{fr(f'f16_list_M{m}_4096x11008'):.3f}" for m in (2, 8, 16, 32)) + "**",
     "`profiles/r06_lutm_xs.txt`; bf16 takes the x-sharing form from 12 rows (table form, instruction-bound: 4–9 %)"),
    ("W2A16 list 96 × 4096²", f"{us('c3_w2a16_list_4096x4096'):.2f} µs per layer", f"{fr('c3_w2a16_list_4096x4096'):.3f}",
     "VALU-bound (1.61 VALU per weight, 0.73 busy); round 6: four-wave workgroups (+1–2 %, `r06_w2_nw_ab.txt`)"),
    ("exl2 3/2-bit lists 4096² / 4096→11008 / 11008→4096", "—", f"{S['c3_exl2_list_4096x4096']:.3f} / {S['c3_exl2_list_4096x11008']:.3f} / {S['c3_exl2_list_11008x4096']:.3f}", ""),
    ("lone launch 4096² / 4096→11008 / 11008→4096 / 8192→28672",
     f"{S['per_layer_launches_4096x4096']['us']:.2f} / {S['c2_gemv_4096x11008']['us']:.2f} / {S['c2_gemv_11008x4096']['us']:.2f} / {S['c5_gemv_8192x28672']['us']:.1f} µs",
     f"{S['per_layer_launches_4096x4096']['frac']:.2f} / {S['c2_gemv_4096x11008']['frac']:.2f} / {S['c2_gemv_11008x4096']['frac']:.2f} / {S['c5_gemv_8192x28672']['frac']:.2f}",
     "a lone launch is NOT memory-bound: with its weights resident in the Infinity Cache it still takes 5.34 µs (round 5 probe)"),
    ("grouped q/k/v (3 × 4096²) / gate+up (2 × 4096→11008)", f"{S['grouped_qkv_3x4096x4096']['us']:.2f} / {S['grouped_gate_up_2x4096x11008']['us']:.2f} µs",
     f"{S['grouped_qkv_3x4096x4096']['frac']:.2f} / {S['grouped_gate_up_2x4096x11008']['frac']:.2f}", ""),
    ("**decode step, Llama-7B linears, true y → x dependencies** (C ABI / unchanged module tree / one launch per layer)",
     f"{d['decode_step_llama7b']['us_per_layer']:.1f} / {S['decode_step_modules_auto_grouped']['us']:.1f} / {S['decode_step_modules_one_launch_per_layer']['us']:.1f} µs per layer",
     f"{d['decode_step_llama7b']['roofline_frac']:.2f} / {S['decode_step_modules_auto_grouped']['frac']:.2f} / {S['decode_step_modules_one_launch_per_layer']['frac']:.2f}",
     "4 launches per layer; per-wave timelines with ramp / stream / tail / reduce phases: `profiles/r06_inl_timeline.txt`"),
    ("**the same step for a batch of 24 sequences, unchanged module tree** (round 6: sibling sets and lone calls of 17 … 32 rows keep the decode kernels where they measured ahead) fp16 / bf16",
     f"{S['decode_step_modules_24rows_f16']['us']:.1f} / {S['decode_step_modules_24rows_bf16']['us']:.1f} µs per layer (round-5 routing, `BIE_LUT_RB2=0`: 115.6 / 123.3)",
     f"{S['decode_step_modules_24rows_f16']['frac']:.2f} / {S['decode_step_modules_24rows_bf16']['frac']:.2f}", "4 / 5 launches per layer instead of 7; `profiles/r06_module_step_rows.txt`, `r06_grouped_rb2_probe.txt`, `r06_lone_rb2_check.txt`"),
    ("**GEMM M = 4096, 4096² (both launches timed)**", f"{g['us_per_launch']:.1f} µs (under the profiler: dequant {prof['dequant_min_us']:.1f} + GEMM {prof['gemm_min_us']:.1f})",
     f"**{g['frac']:.3f} of 2.5 PF**",
     f"MFMA-pipe probe (`profiles/r06_dense_mfma_probe.txt`): MFMAs + barriers only 0.76, + fragment reads 0.60, the loop 0.52, + stores 0.50, + dequantise launch 0.47; counters: MFMA pipe {g['pmc']['mfma_pipe_utilisation']:.3f} busy at {g['pmc']['effective_clock_ghz']:.2f} GHz, fetch = {g['pmc']['fetch_over_tiling_floor']:.3f} × the tiling floor"),
    ("GEMM M = 4096, 4096→11008 / 11008→4096 / 8192→28672", f"{S['c2_gemm_4096x11008']['us']:.1f} / {S['c2_gemm_11008x4096']['us']:.1f} / {us('c5_single_gpu_8192x28672', 'us_per_launch'):.0f} µs",
     f"{S['c2_gemm_4096x11008']['frac']:.2f} / {S['c2_gemm_11008x4096']['frac']:.2f} / {fr('c5_single_gpu_8192x28672'):.2f}",
     f"688 tiles = 2.69 rounds of 256 CUs; long launches clock lower (1.53 GHz under the profiler); act-order (`g_idx` permutation) at M = 4096: {ex['c2_act_order_4096x11008']['M4096_ratio']:.2f} × the plain time (was 1.13 ×)"),
    ("exl2 prefill form, M = 4096: 4096² / 4096→11008 / 11008→4096",
     " / ".join(f"{e2[(4096, k, n)]['us_per_launch']:.0f}" for k, n in ((4096, 4096), (4096, 11008), (11008, 4096))) + " µs",
     " / ".join(f"{e2[(4096, k, n)]['roofline']['frac']:.2f}" for k, n in ((4096, 4096), (4096, 11008), (11008, 4096))),
     "= 0.98–1.11 × the MPQ W4 dense form of the same shape and box (`profiles/r05_exl2_prefill_ab.txt`); M = 64: "
     + "–".join(f"{v:.0f}" for v in sorted(e2[(64, k, n)]['us_per_launch'] for k, n in ((4096, 4096), (4096, 11008), (11008, 4096)))[::2]) + " µs (was 39–59 with the vendor GEMM)"),
    ("**binary conv2d 512→512 3×3 on 7×7 (configs[3]), ONE launch** B = 1 / 32 / 128",
     " / ".join(f"{e['us_per_call']:.1f}" for e in ex["c4_binary"] if "conv" in e["op"]) + " µs (round 5: 8.9 / 20.7 / 29.8 in two / two / three launches)",
     " / ".join(f"{e['roofline']['frac']:.3f}" for e in ex["c4_binary"] if "conv" in e["op"]) + " (VALU xor+bcnt peak at B = 1, FP4 MFMA peak beyond)",
     "B ≤ 8: `xnor_conv_fused_kernel`; beyond: `xnor_conv_mfma_kernel` (FP4 image of the input rows in LDS, no im2col image); history `profiles/r06_conv_timelines.txt`, counters `r06_pmc_conv.txt`"),
    ("binary linear 4096² M = 1 / 64 / 512 / 4096 (XNOR kernels) · matrix pipe M = 512 / 4096",
     " / ".join(f"{e['us_per_launch']:.1f}" for e in ex["c4_binary"] if e["op"] == "binary linear 4096x4096") + " · " + " / ".join(f"{e['us_per_call']:.1f}" for e in ex["c4_binary"] if "matrix pipe" in e["op"] and e["M"] in (512, 4096)) + " µs",
     " / ".join(f"{e['roofline']['frac']:.2f}" for e in ex["c4_binary"] if e["op"] == "binary linear 4096x4096") + " · " + " / ".join(f"{e['roofline']['frac']:.2f}" for e in ex["c4_binary"] if "matrix pipe" in e["op"] and e["M"] in (512, 4096)), "unchanged this round"),
    ("W8A8 / W4A4 GEMM 4096³ · 4096→11008 at M = 4096",
     " · ".join(" / ".join(f"{e['us_per_launch']:.1f}" for e in ex["f1_int_gemm"] if (e["M"], e["N"]) == mn) for mn in ((4096, 4096), (4096, 11008))) + " µs",
     " · ".join(" / ".join(f"{e['roofline']['frac']:.2f}" for e in ex["f1_int_gemm"] if (e["M"], e["N"]) == mn) for mn in ((4096, 4096), (4096, 11008))) + " of 5 POP/s",
     "counters `profiles/r06_pmc_int_gemm.txt`: the i8 pipe 0.43 busy at 32 cycles per instruction; the loop runs at the bf16 kernel's stage time, the fp32 output of W8A8 (67 MB) is what separates it"),
    (f"CPU baseline (oracle port; the box grants {d['cpu_baseline']['cores_granted']} of its {d['cpu_baseline']['nproc']} logical CPUs)",
     " / ".join(f"{cb[k]:.1f}" for k in sorted(cb, key=int, reverse=True)) + " GB/s at " + " / ".join(sorted(cb, key=int, reverse=True)) + " threads", "—",
     "list of 16 layer GEMVs, statically partitioned"),
]
print("| row | time | fraction of its roofline | note |\n|---|---|---|---|")
for row in rows:
    print("| " + " | ".join(row) + " |")
