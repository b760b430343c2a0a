#!/usr/bin/env python3
"""gpurun_out/r06_pmc_gemm_N{4096,11008}.txt (tools/gpu_pmc_gemm_r06.sh) -> profiles/r06_pmc_gemm.json, stamped with the hash of the dense GEMM's
sources so that bench.py only trusts it for the kernel it was collected on.  usage: python tools/pmc_gemm_json_r06.py"""
import ast, hashlib, json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for f in ("mpq_dense.hip", "mfma_pipe.cuh", "mpq_frag_dequant.cuh"):
    h.update(open(os.path.join(ROOT, "bitorch-engine_amd", "csrc", f), "rb").read())
out = {"source": "tools/gpu_pmc_gemm_r06.sh: rocprofv3 --kernel-trace --pmc <one pass per counter group> -- python tools/gemm_only.py 4096 bf16 4096 N (no graph, 14 launches "
                 "averaged; separate FETCH_SIZE / WRITE_SIZE passes); converted by tools/pmc_gemm_json_r06.py",
       "units": "SQ_VALU_MFMA_BUSY_CYCLES sums 32 cycles per v_mfma_f32_32x32x16 over all 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_WAVE_CYCLES / SQ_WAIT_ANY in "
                "quad-cycles; FETCH_SIZE KiB doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE KiB as reported; profiled passes run at a "
                "lower clock than un-profiled ones",
       "gemm_source_sha": h.hexdigest()[:16], "shapes": {}}
for N in (4096, 11008):
    v, stats = {}, {}
    for line in open(os.path.join(ROOT, "gpurun_out", f"r06_pmc_gemm_N{N}.txt")):
        if "mpq_dense_gemm_kernel" in line and "{" in line:
            v.update(ast.literal_eval(line[line.index("{"):line.rindex("}") + 1]))
        m = re.match(r'"void bie::(mpq_dense_gemm_kernel|mpq_dequant_frag_kernel)[^"]*",(\d+),(\d+),([\d.]+)', line)
        if m:
            stats[m.group(1)] = round(float(m.group(4)) / 1e3, 2)
    M = K = 4096
    operands = 2 * M * K + 2 * K * N  # what the GEMM kernel reads: x and the dequantised fragment image
    floor = 8 * (4 * 256 * K * 2 + 8 * 256 * K * 2)  # 8 XCD-private L2s, each walks 4 x panels + 8 image panels of 256 rows (32 tiles per XCD at 4096^2; the floor of THIS tiling)
    out["shapes"][f"M4096_K4096_N{N}"] = {
        "kernel": "bie::mpq_dense_gemm_kernel<bf16, 4, 4>", "us_gemm_kernel_profiled": stats.get("mpq_dense_gemm_kernel"), "us_dequant_kernel_profiled": stats.get("mpq_dequant_frag_kernel"),
        "instructions": {"mfma": v["SQ_INSTS_MFMA"], "valu": v["SQ_INSTS_VALU"], "lds": v["SQ_INSTS_LDS"], "salu": v["SQ_INSTS_SALU"], "vmem_rd": v["SQ_INSTS_VMEM_RD"]},
        "kernel_cycles_per_xcd": round(v["GRBM_GUI_ACTIVE"] / 8), "effective_clock_ghz": round(v["GRBM_GUI_ACTIVE"] / 8 / (stats.get("mpq_dense_gemm_kernel", 1) * 1e3), 3),
        "mfma_busy_cycles_per_simd": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024), "mfma_pipe_utilisation": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (v["GRBM_GUI_ACTIVE"] / 8), 4),
        "wave_wait_fraction": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4), "issue_stall_fraction": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 4),
        "lds_bank_conflict_cycles": v["SQ_LDS_BANK_CONFLICT"],
        "fetch_bytes_corrected": round(v["FETCH_SIZE"] * 2048), "write_bytes": round(v["WRITE_SIZE"] * 1024), "operand_bytes": operands, "y_bytes": 2 * M * N,
        "fetch_over_operands": round(v["FETCH_SIZE"] * 2048 / operands, 3),
        "tiling_floor_bytes_4096x4096": floor if N == 4096 else None, "fetch_over_tiling_floor": round(v["FETCH_SIZE"] * 2048 / floor, 3) if N == 4096 else None}
json.dump(out, open(os.path.join(ROOT, "profiles", "r06_pmc_gemm.json"), "w"), indent=1)
print(json.dumps(out["shapes"], indent=1)[:2500])
