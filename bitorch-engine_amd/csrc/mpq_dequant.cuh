// Register-level dequantisation of packed W{1,2,4,8} words into 16-bit pairs, shared by the GEMV
// and the MFMA GEMM.  The rounding sequence is part of the contract (it reproduces the reference's
// CPU path bit for bit, layers/qlinear/nbit/cuda/utils.py:36-51):
//     ZM_SYM  : w = fl16( fl16(q * s) - z )      ZM_ASYM : w = fl16( s * (q - (zq + 1)) )
//     ZM_FUSED: w = fl16( q * s - z )  (one rounding: the MBWQ kernels' __hfma2,
//               mbwq_linear_cuda_kernel.cu:388, exl2/q_gemm_kernel_gptq.cuh:17,28)
// where fl16 is round-to-nearest-even to the layer dtype (fp16 or bf16).
//
// A packed word holds NB = 32/WBIT consecutive-k values of one output column.  Dequantisation
// emits NP = NB/2 dwords, each holding two 16-bit weights.  WHICH two k offsets share a dword is
// chosen per (dtype, WBIT) so that the extraction is cheapest:
//   fp16 : dword i = (k=i, k=i+NB/2)  -- one v_and_or_b32 per pair with the 0x6400 magic
//          ((q | 0x6400) is the fp16 number 1024+q), then v_pk_add/mul/add_f16.
//   bf16 : no packed bf16 ALU on gfx950, so values go through fp32.  The byte-sliced fields (0..15) are read as
//          fp8 e4m3: bytes 0x00..0x0f are the subnormals and the first normal binade, i.e. exactly q * 2^-9, so
//          v_cvt_pk_f32_fp8 turns two fields into two floats in ONE instruction (measured on gfx950,
//          tools/probe/probe_cvt.hip); the factor 2^9 is folded into the scale (exact).  Then v_pk_mul_f32,
//          v_cvt_pk_bf16_f32 (RNE), and for the second rounding of ZM_SYM v_dot2_f32_bf16 with a (1,0) / (0,1)
//          selector and C = -z as a fused unpack-and-subtract.  W8 keeps v_cvt_f32_ubyteN.
//          dword index p = 2*i + ph holds values (2ph)*(8/WBIT)+i and (2ph+1)*(8/WBIT)+i.
// The activation vector is staged in the same order (`pair_src_k`), so the order is invisible
// outside the kernel.
#pragma once
#include "bie_common.h"

namespace bie {

constexpr int ZM_SYM = 0, ZM_ASYM = 1, ZM_FUSED = 2;

// k offset (0..NB-1) of the 16-bit slot `p` (0..NB-1) in pair order
template <int DT, int WBIT>
__host__ __device__ constexpr int pair_src_k(int p) {
    constexpr int NB = 32 / WBIT;
    if (DT == BIE_F16) {
        return (p >> 1) + (p & 1) * (NB / 2);
    } else {
        constexpr int VPB = 8 / WBIT;  // values per byte
        const int h = p & 1, ph = (p >> 1) & 1, i = p >> 2;
        return (2 * ph + h) * VPB + i;
    }
}

// bf16 pair (1, 0) or (0, 1) in a VGPR, opaque to the optimiser (as immediates hipcc mis-encodes them, probe_cvt.hip)
template <int HI>
__device__ __forceinline__ uint32_t sel_lo_hi() {
    uint32_t v;
    if constexpr (HI == 0) asm("v_mov_b32 %0, 0x3f80" : "=v"(v));
    else asm("v_mov_b32 %0, 0x3f800000" : "=v"(v));
    return v;
}

// fl32(t - z) of the four bf16 values in two packed pairs, ONE instruction each: 1*t.lo + 0*t.hi + (-z) on the dot unit
// (v_dot2_f32_bf16 with the (1,0) / (0,1) selectors) instead of unpack (2 ops) + v_pk_add_f32 (1 op) per pair.
// The VOP3P form is spelled out because hipcc otherwise picks v_dot2c (accumulator tied to the destination) and pays a
// v_mov_b32 per use to seed it.  gfx940+ needs 3 wait states between a DOT write and a different VALU reading it, and
// the hazard recogniser cannot see into inline asm -- so the block carries its own s_nop (3 cycles per 4 values).
__device__ __forceinline__ void bf16_pairs_sub(uint32_t t0, uint32_t t1, uint32_t sel0, uint32_t sel1, float neg_z, float (&r)[4]) {
    asm("v_dot2_f32_bf16 %0, %4, %6, %8\n\t"
        "v_dot2_f32_bf16 %1, %4, %7, %8\n\t"
        "v_dot2_f32_bf16 %2, %5, %6, %8\n\t"
        "v_dot2_f32_bf16 %3, %5, %7, %8\n\t"
        "s_nop 2"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
        : "v"(t0), "v"(t1), "v"(sel0), "v"(sel1), "v"(neg_z));
}

// byte B of v -> float (v_cvt_f32_ubyteB).  hipcc folds the shift/mask forms back into per-value v_bfe + ubyte0,
// so the instruction is named explicitly (plain asm: schedulable, no side effects).
template <int B>
__device__ __forceinline__ float cvt_ubyte(uint32_t v) {
    float f;
    if constexpr (B == 0) asm("v_cvt_f32_ubyte0_e32 %0, %1" : "=v"(f) : "v"(v));
    else if constexpr (B == 1) asm("v_cvt_f32_ubyte1_e32 %0, %1" : "=v"(f) : "v"(v));
    else if constexpr (B == 2) asm("v_cvt_f32_ubyte2_e32 %0, %1" : "=v"(f) : "v"(v));
    else asm("v_cvt_f32_ubyte3_e32 %0, %1" : "=v"(f) : "v"(v));
    return f;
}

// Per-column dequant constants held in registers for the current group.
template <int DT, int ZM> struct ColParams;

template <> struct ColParams<BIE_F16, ZM_SYM> { half2_t s2, z2; };
template <> struct ColParams<BIE_F16, ZM_FUSED> { half2_t s2, z2; };
template <> struct ColParams<BIE_F16, ZM_ASYM> { half2_t s2, zoff2; };  // zoff = 1024 + (zq + 1)
template <> struct ColParams<BIE_BF16, ZM_SYM> { float s, z, nz; uint32_t sel0, sel1; };  // s carries 2^9 when FP8READ
template <> struct ColParams<BIE_BF16, ZM_FUSED> { float s, z; };
template <> struct ColParams<BIE_BF16, ZM_ASYM> { float s, zq1; };

// FP8READ: the consumer reads the fields as fp8 (q * 2^-9, see dequant_word), so the factor 2^9 goes into s / zq1
template <int DT, int WBIT, int ZM, bool FP8READ = (WBIT < 8)>
__device__ __forceinline__ ColParams<DT, ZM> make_col_params(uint32_t s_bits, uint32_t z_bits_or_zq1) {
    ColParams<DT, ZM> p;
    if constexpr (DT == BIE_F16) {
        const half_t s = __builtin_bit_cast(half_t, (uint16_t)s_bits);
        p.s2 = half2_t{s, s};
        if constexpr (ZM == ZM_ASYM) {
            const half_t zo = (half_t)(1024.0f + (float)z_bits_or_zq1);  // exact: < 2048
            p.zoff2 = half2_t{zo, zo};
        } else {
            const half_t z = __builtin_bit_cast(half_t, (uint16_t)z_bits_or_zq1);
            p.z2 = half2_t{z, z};
        }
    } else {
        constexpr float QS = FP8READ ? 512.0f : 1.0f;
        p.s = bf16_bits_to_f32(s_bits) * QS;
        if constexpr (ZM == ZM_ASYM) p.zq1 = (float)z_bits_or_zq1 * (1.0f / QS);
        else p.z = bf16_bits_to_f32(z_bits_or_zq1);
        if constexpr (ZM == ZM_SYM) { p.nz = -p.z; p.sel0 = sel_lo_hi<0>(); p.sel1 = sel_lo_hi<1>(); }
    }
    return p;
}

// word -> NP dwords of dequantised pairs (see header comment for the order)
template <int DT, int WBIT, int ZM>
__device__ __forceinline__ void dequant_word(uint32_t w, const ColParams<DT, ZM>& cp, uint32_t (&out)[16 / WBIT]) {
    constexpr int NB = 32 / WBIT;
    constexpr int NP = NB / 2;
    if constexpr (DT == BIE_F16) {
        constexpr uint32_t M1 = (1u << WBIT) - 1u;
        constexpr uint32_t PAIRMASK = M1 | (M1 << 16);
        const half2_t k1024 = half2_t{(half_t)1024.0f, (half_t)1024.0f};
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const uint32_t P = ((w >> (WBIT * i)) & PAIRMASK) | 0x64006400u;  // (1024+q_i, 1024+q_{i+NB/2})
            const half2_t q = __builtin_bit_cast(half2_t, P);
            half2_t r;
            if constexpr (ZM == ZM_ASYM) {
                r = (q - cp.zoff2) * cp.s2;  // exact integer difference, one rounding
            } else if constexpr (ZM == ZM_FUSED) {
                r = __builtin_elementwise_fma(q - k1024, cp.s2, -cp.z2);  // v_pk_fma_f16: one rounding
            } else {
                r = (q - k1024) * cp.s2;  // exact, then fl16(q*s)
                r = r - cp.z2;            // fl16(. - z)
            }
            out[i] = __builtin_bit_cast(uint32_t, r);
        }
    } else {
        constexpr int VPB = 8 / WBIT;
        constexpr uint32_t M1 = (1u << WBIT) - 1u;
        constexpr uint32_t BMASK = M1 * 0x01010101u;
#pragma unroll
        for (int i = 0; i < VPB; i++) {
            const uint32_t t = (w >> (WBIT * i)) & BMASK;  // bytes b=0..3 hold value index b*VPB + i
            float q[4];
            if constexpr (WBIT < 8) {  // two fields per instruction, values q * 2^-9
                const float2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(t, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8(t, true);
                q[0] = lo.x; q[1] = lo.y; q[2] = hi.x; q[3] = hi.y;
            } else {
                q[0] = cvt_ubyte<0>(t);
                q[1] = cvt_ubyte<1>(t);
                q[2] = cvt_ubyte<2>(t);
                q[3] = cvt_ubyte<3>(t);
            }
            if constexpr (ZM == ZM_SYM) {
                const uint32_t t0 = pack_bf16x2(q[0] * cp.s, q[1] * cp.s), t1 = pack_bf16x2(q[2] * cp.s, q[3] * cp.s);  // fl16(q*s)
                float d[4];
                bf16_pairs_sub(t0, t1, cp.sel0, cp.sel1, cp.nz, d);
                out[2 * i] = pack_bf16x2(d[0], d[1]);  // fl16(. - z)
                out[2 * i + 1] = pack_bf16x2(d[2], d[3]);
            } else {
#pragma unroll
                for (int ph = 0; ph < 2; ph++) {
                    const float a = q[2 * ph], b = q[2 * ph + 1];
                    if constexpr (ZM == ZM_ASYM) out[2 * i + ph] = pack_bf16x2(cp.s * (a - cp.zq1), cp.s * (b - cp.zq1));
                    else out[2 * i + ph] = pack_bf16x2(__builtin_fmaf(a, cp.s, -cp.z), __builtin_fmaf(b, cp.s, -cp.z));
                }
            }
        }
    }
}

template <int DT>
__device__ __forceinline__ float dot2_acc(uint32_t w_pair, uint32_t x_pair, float acc) {
    if constexpr (DT == BIE_F16) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_t, w_pair), __builtin_bit_cast(half2_t, x_pair), acc, false);
    } else {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w_pair), __builtin_bit_cast(bf16x2_t, x_pair), acc, false);
    }
}

// scalar reference dequant of ONE value (used by the generic / dequant / pack kernels); same roundings
template <int DT>
__device__ __forceinline__ float dequant_scalar_sym(uint32_t q, float s, float z) {
    return dt_traits<DT>::round(dt_traits<DT>::round((float)q * s) - z);
}
template <int DT>
__device__ __forceinline__ float dequant_scalar_asym(uint32_t q, float s, int zq1) {
    return dt_traits<DT>::round(s * (float)((int)q - zq1));
}

}  // namespace bie
