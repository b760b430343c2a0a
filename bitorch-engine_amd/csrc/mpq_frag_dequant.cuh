// 8 consecutive-k weights of one column -> the 4-dword operand fragment of v_mfma_f32_32x32x16_{f16,bf16}, in the reference's roundings
// (mpq_dequant.cuh).  Shared by the fused GEMM (mpq_gemm.hip: dequantised in registers, per M tile) and the dequantise-once pass of the
// dense form (mpq_dense.hip).
#pragma once
#include "mpq_dequant.cuh"

#pragma clang fp contract(off)

namespace bie {

// ---- 8-value chunk dequant -> 4 dwords (MFMA operand order) ---------------------------------------
// element order e (0..7) of the produced fragment -> k offset inside the chunk
// Both dtypes produce the fragment in NATURAL k order, so the x tile needs no re-ordering on its way into LDS and can be
// staged by LDS-DMA (global_load_lds).  (An earlier fp16 variant used the cheaper (k, k+4) pairing of the 0x6400 trick and
// paid ~80 VALU per thread per K tile to permute x instead.)
template <int DT, int WBIT>
__host__ __device__ constexpr int frag_src_k(int e) {
    return e;
}

// raw bits of chunk c8 (8 consecutive k) of column n.  WBIT<=4: returned in .x (low 8*WBIT bits), WBIT==8: two words
template <int WBIT>
__device__ __forceinline__ uint2_t load_chunk(const uint32_t* __restrict__ qw, int c8, int n, int N) {
    uint2_t r;
    if constexpr (WBIT == 8) {
        r.x = qw[(long)(2 * c8) * N + n];
        r.y = qw[(long)(2 * c8 + 1) * N + n];
    } else {
        constexpr int CPW = 4 / WBIT;  // chunks per word: w4 -> 1, w2 -> 2, w1 -> 4
        r.x = qw[(long)(c8 / CPW) * N + n];
        r.y = 0;
    }
    return r;
}

template <int DT, int WBIT, int ZM>
__device__ __forceinline__ uint4_t dequant8(uint2_t raw, int c8, const ColParams<DT, ZM>& cp) {
    uint32_t o[4];
    if constexpr (DT == BIE_F16) {
        const half2_t k1024 = half2_t{(half_t)1024.0f, (half_t)1024.0f};
        // P[i] = fp16 pair (1024 + q[2i], 1024 + q[2i+1]): the field value sits in the mantissa of 0x6400
        uint32_t P[4];
        if constexpr (WBIT == 8) {  // byte pairs -> (b0, 0x64, b1, 0x64) with one v_perm_b32 each
            P[0] = __builtin_amdgcn_perm(0x64646464u, raw.x, 0x04010400u);
            P[1] = __builtin_amdgcn_perm(0x64646464u, raw.x, 0x04030402u);
            P[2] = __builtin_amdgcn_perm(0x64646464u, raw.y, 0x04010400u);
            P[3] = __builtin_amdgcn_perm(0x64646464u, raw.y, 0x04030402u);
        } else if constexpr (WBIT == 4) {  // nibbles -> bytes in natural order (2 masks + 2 perms), then as for 8 bit
            const uint32_t lo = raw.x & 0x0f0f0f0fu, hi = (raw.x >> 4) & 0x0f0f0f0fu;
            const uint32_t b03 = __builtin_amdgcn_perm(hi, lo, 0x05010400u), b47 = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
            P[0] = __builtin_amdgcn_perm(0x64646464u, b03, 0x04010400u);
            P[1] = __builtin_amdgcn_perm(0x64646464u, b03, 0x04030402u);
            P[2] = __builtin_amdgcn_perm(0x64646464u, b47, 0x04010400u);
            P[3] = __builtin_amdgcn_perm(0x64646464u, b47, 0x04030402u);
        } else {
            constexpr int CPW = 4 / WBIT;
            constexpr uint32_t CM = (1u << (8 * WBIT)) - 1u;
            const uint32_t sub = (raw.x >> ((c8 % CPW) * 8 * WBIT)) & CM;
            const uint32_t T = sub | (sub << (16 - WBIT));  // value 2i+1 lands 16 bits above value 2i (overlap bits are never read)
            constexpr uint32_t M1 = (1u << WBIT) - 1u;
#pragma unroll
            for (int i = 0; i < 4; i++) P[i] = ((T >> (2 * WBIT * i)) & (M1 | (M1 << 16))) | 0x64006400u;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const half2_t q = __builtin_bit_cast(half2_t, P[i]);
            half2_t r;
            if constexpr (ZM == ZM_ASYM) {
                r = (q - cp.zoff2) * cp.s2;
            } else if constexpr (ZM == ZM_FUSED) {
                r = __builtin_elementwise_fma(q - k1024, cp.s2, -cp.z2);
            } else {
                r = (q - k1024) * cp.s2;
                r = r - cp.z2;
            }
            o[i] = __builtin_bit_cast(uint32_t, r);
        }
    } else {
        float q[8];
        if constexpr (WBIT == 8) {
            q[0] = cvt_ubyte<0>(raw.x); q[1] = cvt_ubyte<1>(raw.x); q[2] = cvt_ubyte<2>(raw.x); q[3] = cvt_ubyte<3>(raw.x);
            q[4] = cvt_ubyte<0>(raw.y); q[5] = cvt_ubyte<1>(raw.y); q[6] = cvt_ubyte<2>(raw.y); q[7] = cvt_ubyte<3>(raw.y);
        } else if constexpr (WBIT == 4) {
            // nibbles -> bytes in natural k order (two v_perm_b32), then four v_cvt_pk_f32_fp8: bytes 0..15 read as fp8 e4m3
            // are q * 2^-9 exactly (mpq_dequant.cuh); 2^9 sits in cp.s / cp.zq1 (make_col_params<.., FP8READ = true>)
            const uint32_t lo = raw.x & 0x0f0f0f0fu, hi = (raw.x >> 4) & 0x0f0f0f0fu;
            const uint32_t p01 = __builtin_amdgcn_perm(hi, lo, 0x05010400u), p23 = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
            const float2_t f0 = __builtin_amdgcn_cvt_pk_f32_fp8(p01, false), f1 = __builtin_amdgcn_cvt_pk_f32_fp8(p01, true);
            const float2_t f2 = __builtin_amdgcn_cvt_pk_f32_fp8(p23, false), f3 = __builtin_amdgcn_cvt_pk_f32_fp8(p23, true);
            q[0] = f0.x; q[1] = f0.y; q[2] = f1.x; q[3] = f1.y; q[4] = f2.x; q[5] = f2.y; q[6] = f3.x; q[7] = f3.y;
        } else {
            constexpr int CPW = 4 / WBIT;
            constexpr uint32_t CM = (1u << (8 * WBIT)) - 1u;
            constexpr uint32_t M1 = (1u << WBIT) - 1u;
            const uint32_t sub = (raw.x >> ((c8 % CPW) * 8 * WBIT)) & CM;
#pragma unroll
            for (int e = 0; e < 8; e++) q[e] = (float)((sub >> (WBIT * e)) & M1);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float a = q[2 * i], b = q[2 * i + 1];
            if constexpr (ZM == ZM_ASYM) {
                o[i] = pack_bf16x2(cp.s * (a - cp.zq1), cp.s * (b - cp.zq1));
            } else if constexpr (ZM == ZM_FUSED) {
                o[i] = pack_bf16x2(__builtin_fmaf(a, cp.s, -cp.z), __builtin_fmaf(b, cp.s, -cp.z));
            } else {
                const uint32_t t1 = pack_bf16x2(a * cp.s, b * cp.s);  // fl16(q*s)
                const float ta = __uint_as_float(t1 << 16), tb = __uint_as_float(t1 & 0xffff0000u);
                o[i] = pack_bf16x2(ta - cp.z, tb - cp.z);
            }
        }
    }
    return uint4_t{o[0], o[1], o[2], o[3]};
}

}  // namespace bie
