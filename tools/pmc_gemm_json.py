#!/usr/bin/env python3
"""gpurun_out/r03w_pmc_gemm.txt (tools/gpu_pmc_gemm_r03.sh) -> profiles/r03_pmc_gemm.json, stamped with the hash of the fused GEMM's
sources so that bench.py only trusts it for the kernel it was collected on.  usage: python tools/pmc_gemm_json.py [txt] [json]"""
import ast, hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r03w_pmc_gemm.txt")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r03_pmc_gemm.json")


def gemm_source_sha(kernel):
    """Hash of the sources of the kernel the counters belong to (bench.py recomputes it the same way)."""
    files = ("mpq_dense.hip", "mfma_pipe.cuh", "mpq_frag_dequant.cuh") if "mpq_dense_gemm_kernel" in (kernel or "") else ("mpq_gemm.hip", "mpq_frag_dequant.cuh")
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(ROOT, "bitorch-engine_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


vals, cur = {"bf16": {}, "f16": {}}, None
for line in open(src):
    m = re.match(r"== \w+ (bf16|f16)", line)
    if m:
        cur = m.group(1)
    elif cur and ("mpq_gemm_kernel" in line or "mpq_dense_gemm_kernel" in line) and "{" in line:
        vals[cur].update(ast.literal_eval(line[line.index("{"):line.rindex("}") + 1]))
        vals[cur]["kernel"] = line[:line.index("(")].replace("void ", "").strip()
out = {"source": "tools/gpu_pmc_gemm_r03.sh: rocprofv3 --kernel-trace --pmc <one pass per counter group> -- python tools/gemm_only.py 4096 <dtype> "
                 "(M=4096, K=4096, N=11008, no graph; counters of the GEMM kernel the dispatch picks for this shape, named per dtype); converted by tools/pmc_gemm_json.py",
       "units": "SQ_VALU_MFMA_BUSY_CYCLES sums 32 cycles per v_mfma_f32_32x32x16 over all 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_WAVE_CYCLES / "
                "SQ_WAIT_ANY in quad-cycles; FETCH_SIZE KiB doubled per MI355X_MICROARCH.md; profiled passes run at a lower clock than un-profiled ones",
       "gemm_source_sha": gemm_source_sha(vals["bf16"].get("kernel")), "shape": "M4096_K4096_N11008"}
for dt, v in vals.items():
    if not v:
        continue
    out[dt] = {"kernel": v.get("kernel"), "instructions": {"mfma": v["SQ_INSTS_MFMA"], "valu": v["SQ_INSTS_VALU"], "lds": v["SQ_INSTS_LDS"], "salu": v["SQ_INSTS_SALU"],
                                "vmem_rd": v["SQ_INSTS_VMEM_RD"], "valu_per_mfma": round(v["SQ_INSTS_VALU"] / v["SQ_INSTS_MFMA"], 2)},
               "kernel_cycles_per_xcd": round(v["GRBM_GUI_ACTIVE"] / 8), "mfma_busy_cycles_per_simd": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024),
               "mfma_pipe_utilisation": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (v["GRBM_GUI_ACTIVE"] / 8), 4),
               "wave_wait_fraction": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 4),
               "fetch_bytes_corrected": round(v["FETCH_SIZE"] * 2048), "write_bytes": round(v["WRITE_SIZE"] * 1024), "algorithmic_bytes": 147685376}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out)[:400])
