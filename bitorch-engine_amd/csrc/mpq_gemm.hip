// W{1,2,4,8}A16 fused dequant + MFMA GEMM for gfx950 (M > 8) -- the compute-bound half of
// bie_mpq_forward.  Replaces the reference's "materialise the whole K x N fp16 weight, then cuBLAS"
// branch (layers/qlinear/nbit/cuda/mpq_layer.py:59-63, unpack_qweight utils.py:30-51).
//
// Key observation: for v_mfma_f32_32x32x16_{f16,bf16} the B fragment of lane l is the 8 consecutive-k
// values (k = 8*(l>>5) .. +7) of ONE output column (n = l&31) -- which in the reference's packed
// layout qweight[k/(32/w)][n] is 8*w contiguous bits of one int32 word.  So the packed weights go
//     HBM -> one dword per lane -> dequant in registers -> MFMA B operand
// with no LDS round trip and no transposition, and each wave dequantises ONLY its own 32 columns:
//   * block = 4 waves, block tile BM x 128 (each wave: all BM rows x 32 columns, BM/32 accumulators
//     of 32x32), so the VALU dequant cost per MFMA shrinks with BM (BM=256: 8 MFMAs per fragment);
//   * x (the A operand) is the operand shared by the 4 waves: staged global -> registers -> LDS in
//     [BM][64] tiles, double buffered, one barrier per K tile; the LDS image is XOR-swizzled so the
//     ds_read_b128 fragment reads are bank-conflict free; the staging pass also applies the 8-element
//     k permutation that the dequant's pair order implies (mpq_dequant.cuh) -- the contraction is
//     invariant under a consistent permutation of k on both operands;
//   * accumulation in fp32 (AGPR/VGPR unified file), one rounding to fp16/bf16 at the store;
//   * optional split-K (grid.z) for skinny M; partials reduced in fixed order by splitk_finalize.
#include "mpq_dequant.cuh"

#pragma clang fp contract(off)

namespace bie {

constexpr int GEMM_BN = 128;
constexpr int GEMM_BK = 64;

// ---- 8-value chunk dequant -> 4 dwords (MFMA operand order) ---------------------------------------
// element order e (0..7) of the produced fragment -> k offset inside the chunk
template <int DT, int WBIT>
__host__ __device__ constexpr int frag_src_k(int e) {
    if (DT == BIE_F16) {
        if (WBIT == 8) return ((e >> 1) & 1) + ((e & 1) << 1) + (e & 4);  // (0,2,1,3,4,6,5,7)
        return (e >> 1) + (e & 1) * 4;                                      // (0,4,1,5,2,6,3,7)
    }
    return e;  // bf16: natural
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
        uint32_t P[4];
        if constexpr (WBIT == 8) {
            P[0] = (raw.x & 0x00ff00ffu);
            P[1] = ((raw.x >> 8) & 0x00ff00ffu);
            P[2] = (raw.y & 0x00ff00ffu);
            P[3] = ((raw.y >> 8) & 0x00ff00ffu);
        } else if constexpr (WBIT == 4) {
#pragma unroll
            for (int i = 0; i < 4; i++) P[i] = (raw.x >> (4 * i)) & 0x000f000fu;
        } else {
            constexpr int CPW = 4 / WBIT;
            constexpr uint32_t CM = (1u << (8 * WBIT)) - 1u;
            const uint32_t sub = (raw.x >> ((c8 % CPW) * 8 * WBIT)) & CM;
            const uint32_t T = sub | (sub << (16 - 4 * WBIT));  // value i+4 lands 16 bits above value i
            constexpr uint32_t M1 = (1u << WBIT) - 1u;
#pragma unroll
            for (int i = 0; i < 4; i++) P[i] = (T >> (WBIT * i)) & (M1 | (M1 << 16));
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const half2_t q = __builtin_bit_cast(half2_t, P[i] | 0x64006400u);
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
            q[0] = (float)(raw.x & 0xffu); q[1] = (float)((raw.x >> 8) & 0xffu);
            q[2] = (float)((raw.x >> 16) & 0xffu); q[3] = (float)(raw.x >> 24);
            q[4] = (float)(raw.y & 0xffu); q[5] = (float)((raw.y >> 8) & 0xffu);
            q[6] = (float)((raw.y >> 16) & 0xffu); q[7] = (float)(raw.y >> 24);
        } else if constexpr (WBIT == 4) {
            const uint32_t lo = raw.x & 0x0f0f0f0fu, hi = (raw.x >> 4) & 0x0f0f0f0fu;
            q[0] = (float)(lo & 0xffu); q[2] = (float)((lo >> 8) & 0xffu);
            q[4] = (float)((lo >> 16) & 0xffu); q[6] = (float)(lo >> 24);
            q[1] = (float)(hi & 0xffu); q[3] = (float)((hi >> 8) & 0xffu);
            q[5] = (float)((hi >> 16) & 0xffu); q[7] = (float)(hi >> 24);
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
                const uint32_t t1 = pack_bf16x2(a * cp.s, b * cp.s);
                const float ta = __uint_as_float(t1 << 16), tb = __uint_as_float(t1 & 0xffff0000u);
                o[i] = pack_bf16x2(ta - cp.z, tb - cp.z);
            }
        }
    }
    return uint4_t{o[0], o[1], o[2], o[3]};
}

// 16-byte A chunk (8 consecutive k, natural order) -> fragment order
template <int DT, int WBIT>
__device__ __forceinline__ uint4_t permute_a_chunk(uint4_t v) {
    if constexpr (DT == BIE_BF16) {
        return v;
    } else {
        const uint32_t d[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int ka = frag_src_k<DT, WBIT>(2 * i), kb = frag_src_k<DT, WBIT>(2 * i + 1);
            const uint32_t a = (ka & 1) ? (d[ka >> 1] >> 16) : (d[ka >> 1] & 0xffffu);
            const uint32_t b = (kb & 1) ? (d[kb >> 1] & 0xffff0000u) : (d[kb >> 1] << 16);
            o[i] = a | b;
        }
        return uint4_t{o[0], o[1], o[2], o[3]};
    }
}

// swizzled LDS byte offset of the 16-byte slot (row, s) of a [BM][64] 16-bit tile: two rows share a
// 256-byte bank row, the slot index is XORed with (row>>1)&7 -> the four 16-lane groups of a
// ds_read_b128 fragment read (32 rows, fixed s) each touch 16 distinct slots.
__device__ __forceinline__ int a_lds_off(int row, int s) {
    return ((row >> 1) << 8) | ((((row & 1) << 3) | (s ^ ((row >> 1) & 7))) << 4);
}

template <int DT>
__device__ __forceinline__ float16_t mfma32(uint4_t a, uint4_t b, float16_t c) {
    if constexpr (DT == BIE_F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

template <int DT, int WBIT, int ZM>
struct WTile {  // one K tile (64) worth of this lane's weights + dequant constants
    uint2_t raw[4];
    uint32_t sb[4];
    uint32_t zb[4];  // sym: z bits ; asym: zq + 1
};

template <int DT, int WBIT, int ZM>
__device__ __forceinline__ void load_wtile(WTile<DT, WBIT, ZM>& t, const uint32_t* __restrict__ qw,
                                           const uint16_t* __restrict__ scales, const void* __restrict__ zeros, int k0,
                                           int h, int n, int N, int group_size) {
    constexpr int NB = 32 / WBIT;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const int k = k0 + kk * 16 + h * 8;
        const int c8 = k >> 3;
        t.raw[kk] = load_chunk<WBIT>(qw, c8, n, N);
        const int g = k / group_size;
        t.sb[kk] = scales[(long)g * N + n];
        if constexpr (ZM == ZM_ASYM) {
            constexpr uint32_t M1 = (WBIT == 32) ? 0xffffffffu : ((1u << WBIT) - 1u);
            const uint32_t zw = reinterpret_cast<const uint32_t*>(zeros)[(long)g * (N / NB) + n / NB];
            t.zb[kk] = ((zw >> ((n % NB) * WBIT)) & M1) + 1u;
        } else {
            t.zb[kk] = reinterpret_cast<const uint16_t*>(zeros)[(long)g * N + n];
        }
    }
}

template <int DT, int WBIT, int ZM, int BM>
__global__ __launch_bounds__(256) void mpq_gemm_kernel(const uint16_t* __restrict__ x, const uint32_t* __restrict__ qw,
                                                       const uint16_t* __restrict__ scales, const void* __restrict__ zeros,
                                                       const uint16_t* __restrict__ bias, const uint16_t* __restrict__ perm,
                                                       float* __restrict__ part, uint16_t* __restrict__ y, int M, int K, int N,
                                                       int group_size, int tiles_per_split, int S, int m_tiles, int n_tiles) {
    constexpr int TM = BM / 32;              // 32x32 accumulators per wave
    constexpr int A_CHUNKS = BM * 8 / 256;   // 16-byte chunks staged per thread per tile
    constexpr int A_BYTES = BM * GEMM_BK * 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    // XCD-aware tile order: the dispatcher places block b on XCD b % 8 (speed only, never correctness).  Give each XCD
    // a contiguous chunk of the m-major tile order so that the ~64 blocks co-resident on an XCD share 1-2 x row-tiles
    // (2 MB each at K=4096) in that XCD's 4 MiB L2 instead of every XCD cycling through all of x.
    int m_tile, n_tile;
    {
        const int T = m_tiles * n_tiles;
        const int b = blockIdx.x;
        const int xcd = b & 7, i = b >> 3;
        const int q = T >> 3, r = T & 7;
        const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int t = start + i;
        m_tile = t / n_tiles;
        n_tile = t - m_tile * n_tiles;
    }
    const int m0 = m_tile * BM;
    const int n_base = n_tile * GEMM_BN + wave * 32;
    const int n = n_base + j;
    const int n_ld = n < N ? n : N - 1;  // clamp loads of out-of-range columns (never stored)
    const int split = blockIdx.y;
    const int T_total = K / GEMM_BK;
    const int t_begin = split * tiles_per_split;
    int t_end = t_begin + tiles_per_split;
    if (t_end > T_total) t_end = T_total;

    float16_t acc[TM];
#pragma unroll
    for (int t = 0; t < TM; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[t][e] = 0.0f;

    // A staging assignment: chunk c = tid + i*256 -> row = c / 8, slot = c % 8
    uint4_t areg[A_CHUNKS];
    auto load_a = [&](int kt) {
#pragma unroll
        for (int i = 0; i < A_CHUNKS; i++) {
            const int c = tid + i * 256;
            const int row = c >> 3, s = c & 7;
            const int m = m0 + row;
            uint4_t v = uint4_t{0u, 0u, 0u, 0u};
            if (m < M) {
                if (perm == nullptr) {
                    v = *reinterpret_cast<const uint4_t*>(x + (long)m * K + (long)kt * GEMM_BK + s * 8);
                } else {  // MBWQ act-order: gather x[m][q_perm[k]]
                    const uint16_t* pp = perm + kt * GEMM_BK + s * 8;
                    const uint16_t* xr = x + (long)m * K;
                    v.x = (uint32_t)xr[pp[0]] | ((uint32_t)xr[pp[1]] << 16);
                    v.y = (uint32_t)xr[pp[2]] | ((uint32_t)xr[pp[3]] << 16);
                    v.z = (uint32_t)xr[pp[4]] | ((uint32_t)xr[pp[5]] << 16);
                    v.w = (uint32_t)xr[pp[6]] | ((uint32_t)xr[pp[7]] << 16);
                }
            }
            areg[i] = v;
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_CHUNKS; i++) {
            const int c = tid + i * 256;
            const int row = c >> 3, s = c & 7;
            *reinterpret_cast<uint4_t*>(lds + buf * A_BYTES + a_lds_off(row, s)) = permute_a_chunk<DT, WBIT>(areg[i]);
        }
    };

    WTile<DT, WBIT, ZM> wcur, wnext;
    if (t_begin < t_end) {
        load_a(t_begin);
        load_wtile<DT, WBIT, ZM>(wcur, qw, scales, zeros, t_begin * GEMM_BK, h, n_ld, N, group_size);
        store_a(0);
    }
    __syncthreads();

    int cur = 0;
    for (int kt = t_begin; kt < t_end; kt++) {
        const bool has_next = (kt + 1 < t_end);
        if (has_next) {
            load_a(kt + 1);
            load_wtile<DT, WBIT, ZM>(wnext, qw, scales, zeros, (kt + 1) * GEMM_BK, h, n_ld, N, group_size);
        }
        const unsigned char* abuf = lds + cur * A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const ColParams<DT, ZM> cp = make_col_params<DT, WBIT, ZM>(wcur.sb[kk], wcur.zb[kk]);
            const int c8 = (kt * GEMM_BK + kk * 16 + h * 8) >> 3;
            const uint4_t bfrag = dequant8<DT, WBIT, ZM>(wcur.raw[kk], c8, cp);
#pragma unroll
            for (int t = 0; t < TM; t++) {
                const uint4_t afrag = *reinterpret_cast<const uint4_t*>(abuf + a_lds_off(t * 32 + j, kk * 2 + h));
                acc[t] = mfma32<DT>(afrag, bfrag, acc[t]);
            }
        }
        if (has_next) {
            store_a(cur ^ 1);
            wcur = wnext;
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    if (n < N) {
        float bv = 0.0f;
        const bool use_bias = (S == 1) && (bias != nullptr);
        if (use_bias) bv = dt_traits<DT>::load(bias, n);
#pragma unroll
        for (int t = 0; t < TM; t++) {
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = m0 + t * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row < M) {
                    if (S == 1) {
                        float o = dt_traits<DT>::round(acc[t][e]);
                        if (use_bias) o = o + bv;
                        dt_traits<DT>::store(y, (long)row * N + n, o);
                    } else {
                        part[((long)split * M + row) * N + n] = acc[t][e];
                    }
                }
            }
        }
    }
}

// ---- launch plumbing ---------------------------------------------------------------------------------
struct GemmPlan {
    int BM, S, tiles_per_split;
};

static GemmPlan plan_gemm(int M, int K, int N) {
    GemmPlan p;
    p.BM = M <= 32 ? 32 : (M <= 64 ? 64 : (M <= 128 ? 128 : 256));
    const int tiles = cdiv(M, p.BM) * cdiv(N, GEMM_BN);
    const int T = K / GEMM_BK;
    int S = 1;
    if (tiles < 256) {
        S = cdiv(512, tiles);
        const int max_s = T / 4 > 0 ? T / 4 : 1;  // keep >= 4 K tiles per split
        if (S > max_s) S = max_s;
        if (S > 16) S = 16;
    }
    p.tiles_per_split = cdiv(T, S);
    p.S = cdiv(T, p.tiles_per_split);
    return p;
}

bool mpq_gemm_ok(int M, int K, int N, int w_bit, int group_size, int dtype, bool has_gidx) {
    if (dtype != BIE_F16 && dtype != BIE_BF16) return false;
    if (has_gidx) return false;
    const int gs = group_size > K ? K : group_size;
    if (K % GEMM_BK || gs % 8) return false;
    if (N % (32 / w_bit)) return false;
    return true;
}

size_t mpq_gemm_workspace_bytes(int M, int K, int N) {
    if (K % GEMM_BK) return 0;
    const GemmPlan p = plan_gemm(M, K, N);
    return p.S > 1 ? (size_t)p.S * M * N * sizeof(float) : 0;
}

template <int DT, int WBIT, int ZM>
static int gemm_launch_bm(const GemmPlan& p, const void* x, const int32_t* qw, const void* scales, const void* zeros,
                          const void* bias, const uint16_t* perm, float* part, void* y, int M, int K, int N, int group_size, hipStream_t st) {
    const int m_tiles = cdiv(M, p.BM), n_tiles = cdiv(N, GEMM_BN);
    dim3 grid(m_tiles * n_tiles, p.S);
    const size_t lds = (size_t)2 * p.BM * GEMM_BK * 2;
#define L(BMV)                                                                                                        \
    hipLaunchKernelGGL((mpq_gemm_kernel<DT, WBIT, ZM, BMV>), grid, dim3(256), lds, st, (const uint16_t*)x,           \
                       (const uint32_t*)qw, (const uint16_t*)scales, zeros, (const uint16_t*)bias, perm, part, (uint16_t*)y, \
                       M, K, N, group_size, p.tiles_per_split, p.S, m_tiles, n_tiles)
    switch (p.BM) {
        case 32: L(32); break;
        case 64: L(64); break;
        case 128: L(128); break;
        default: L(256); break;
    }
#undef L
    return check_launch("mpq_gemm_kernel");
}

template <int DT, int WBIT>
static int gemm_launch_a(const GemmPlan& p, int zm, const void* x, const int32_t* qw, const void* scales,
                         const void* zeros, const void* bias, const uint16_t* perm, float* part, void* y, int M, int K, int N, int group_size,
                         hipStream_t st) {
    if (zm == ZM_ASYM) return gemm_launch_bm<DT, WBIT, ZM_ASYM>(p, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
    if (zm == ZM_FUSED) return gemm_launch_bm<DT, WBIT, ZM_FUSED>(p, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
    return gemm_launch_bm<DT, WBIT, ZM_SYM>(p, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
}

template <int DT>
static int gemm_launch_w(const GemmPlan& p, int w_bit, int zm, const void* x, const int32_t* qw, const void* scales,
                         const void* zeros, const void* bias, const uint16_t* perm, float* part, void* y, int M, int K, int N, int group_size,
                         hipStream_t st) {
    switch (w_bit) {
        case 1: return gemm_launch_a<DT, 1>(p, zm, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
        case 2: return gemm_launch_a<DT, 2>(p, zm, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
        case 4: return gemm_launch_a<DT, 4>(p, zm, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
        default: return gemm_launch_a<DT, 8>(p, zm, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
    }
}

int mpq_gemm_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const void* bias, void* y,
                    float* part, int M, int K, int N, int w_bit, int group_size, int zm, int dtype, const uint16_t* perm,
                    hipStream_t st) {
    const GemmPlan p = plan_gemm(M, K, N);
    const int gs = group_size > K ? K : group_size;
    int rc;
    if (dtype == BIE_F16)
        rc = gemm_launch_w<BIE_F16>(p, w_bit, zm, x, qw, scales, zeros, bias, perm, part, y, M, K, N, gs, st);
    else
        rc = gemm_launch_w<BIE_BF16>(p, w_bit, zm, x, qw, scales, zeros, bias, perm, part, y, M, K, N, gs, st);
    if (rc) return rc;
    if (p.S > 1) return launch_splitk_finalize(part, bias, y, p.S, M, N, dtype, st);
    return BIE_OK;
}

}  // namespace bie
