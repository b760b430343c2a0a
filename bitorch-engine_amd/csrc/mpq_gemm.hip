// W{1,2,4,8}A16 fused dequant + MFMA GEMM for gfx950 (M >= 3) -- the compute-bound half of
// bie_mpq_forward.  Replaces the reference's "materialise the whole K x N fp16 weight, then cuBLAS"
// branch (layers/qlinear/nbit/cuda/mpq_layer.py:59-63, unpack_qweight utils.py:30-51).
//
// Key observation: for v_mfma_f32_32x32x16_{f16,bf16} the operand fragment of lane l is the 8 consecutive-k
// values (k = 8*(l>>5) .. +7) of ONE output column (n = l&31) -- which in the reference's packed
// layout qweight[k/(32/w)][n] is 8*w contiguous bits of one int32 word.  So the packed weights go
//     HBM -> one dword per lane -> dequant in registers -> MFMA operand
// with no LDS round trip and no transposition, and each wave dequantises ONLY its own 64 columns:
//   * block = 4 waves, block tile BM x 256 (each wave: all BM rows x 64 columns, 2*BM/32 accumulators of 32x32), so the
//     VALU dequant cost per MFMA shrinks with BM (BM = 256: one wave per SIMD with the 512-register budget);
//   * x is the operand shared by the 4 waves: global_load_lds_dwordx4 straight into an XOR-swizzled [BM][64] LDS image
//     (swizzle applied on the source address), double buffered, one barrier per K tile; ds_read_b128 fragment reads are
//     bank-conflict free; both dtypes produce the weight fragment in natural k order, so x needs no re-ordering (the MBWQ
//     q_perm gather of x is the one case that still goes global -> registers -> LDS);
//   * the K loop is software-pipelined across the tile boundary: every MFMA group has the next group's A fragments in
//     flight and the next group's B fragments being dequantised; loads of the next tile are issued under the first groups;
//   * MFMAs are issued as D = W_frag . x_frag: a lane owns 4 consecutive output columns -> 8-byte row-contiguous stores;
//   * accumulation in fp32, one rounding to fp16/bf16 at the store;
//   * tile height and split-K factor from a measured cost model (plan_gemm); partials reduced in fixed order by
//     splitk_finalize (an in-kernel ticketed reduction was measured 5-8x slower at GEMM partial volumes).
#include "mpq_frag_dequant.cuh"
#ifndef BIE_GEMM_LAB
#define BIE_GEMM_LAB 0  // compile-time ablation switch used by tools/ (0 = product code)
#endif

#pragma clang fp contract(off)

namespace bie {

constexpr int GEMM_BN = 256;  // 4 waves x 64 columns
constexpr int GEMM_BK = 64;

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

// ---- hand-issued LDS fragment reads ---------------------------------------------------------------------------
// hipcc schedules ds_read -> MFMA pairs just-in-time through one or two register sets (every MFMA then waits a full
// LDS round trip).  The A fragments of the NEXT k16 step are therefore issued explicitly (asm volatile keeps program
// order), left in flight under the current step's MFMAs + dequant VALU, and collected by ONE counted wait that names
// every destination register (so no consumer can be scheduled above it).
template <int TM>
__device__ __forceinline__ void lds_issue_frags(uint4_t (&af)[TM], uint32_t addr) {
#if BIE_GEMM_LAB == 4
#pragma unroll
    for (int t = 0; t < TM; t++) af[t] = uint4_t{addr, addr + t, 0x3c003c00u, 0x3c003c00u};
    return;
#endif
    asm volatile("ds_read_b128 %0, %1" : "=v"(af[0]) : "v"(addr));
    if constexpr (TM > 1) asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(af[1]) : "v"(addr));
    if constexpr (TM > 2) {
        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(af[2]) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(af[3]) : "v"(addr));
    }
    if constexpr (TM > 4) {
        asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(af[4]) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:20480" : "=v"(af[5]) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:24576" : "=v"(af[6]) : "v"(addr));
        asm volatile("ds_read_b128 %0, %1 offset:28672" : "=v"(af[7]) : "v"(addr));
    }
}
template <int TM>
__device__ __forceinline__ void lds_wait_frags(uint4_t (&af)[TM]) {
    if constexpr (TM == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0])::"memory");
    else if constexpr (TM == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1])::"memory");
    else if constexpr (TM == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3])::"memory");
    else asm volatile("s_waitcnt lgkmcnt(0)"
                      : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]), "+v"(af[6]), "+v"(af[7])::"memory");
}

constexpr int GEMM_NF = 2;  // 32-column B fragments per wave (wave tile = BM x 64)

template <int WBIT>
struct RawChunk {
    using type = uint32_t;
};
template <>
struct RawChunk<8> {
    using type = uint2_t;
};

template <int WBIT>
__device__ __forceinline__ uint2_t raw_as_u2(typename RawChunk<WBIT>::type r) {
    if constexpr (WBIT == 8) return r;
    else return uint2_t{r, 0u};
}

// GPT ("group per tile"): group_size % 64 == 0, one (scale, zero) pair per fragment per K tile; otherwise one per
// k16 step (group_size 16 / 32).  group_size is a power of two on this path (1 << gshift).
template <int DT, int WBIT, int ZM, bool GPT>
struct WTile {  // one K tile (64) worth of this lane's weights + dequant constants, per fragment
    static constexpr int NG = GPT ? 1 : 4;
    typename RawChunk<WBIT>::type raw[GEMM_NF][4];
    uint32_t sb[GEMM_NF][NG];
    uint32_t zb[GEMM_NF][NG];  // sym: z bits ; asym: packed qzeros word (field extracted at use)
};

// Per-lane base pointers (64-bit, computed once); everything added per tile / k16 step is wave-uniform, so the
// loop's address arithmetic stays on the scalar unit (global_load ... v[base], s[offset]).
template <int WBIT>
struct WPtrs {
    __amdgpu_buffer_rsrc_t wrsrc;  // qweight as a raw buffer: the loop's row offsets are scalar (soffset), the lane part is voffset
    uint32_t wvoff[GEMM_NF];       // byte offset of (lane row part, column)
    uint32_t row_bytes;            // N * 4
    const uint32_t* w[GEMM_NF];    // qweight + lane row part + column
    const uint16_t* s[GEMM_NF];    // scales + column
    const uint16_t* z[GEMM_NF];    // fp zeros + column            (sym)
    const uint32_t* zq[GEMM_NF];   // packed qzeros + column / NB  (asym)
    int zshift[GEMM_NF];           // bit offset of this column's field in the qzeros word
};

template <int DT, int WBIT, int ZM, bool GPT>
__device__ __forceinline__ void load_wtile(WTile<DT, WBIT, ZM, GPT>& t, const WPtrs<WBIT>& p, int k0, int N, int gshift) {
    constexpr int NB = 32 / WBIT;
    constexpr int NG = GPT ? 1 : 4;
#pragma unroll
    for (int f = 0; f < GEMM_NF; f++) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int c8u = (k0 >> 3) + kk * 2;  // uniform part of the chunk index (the lane adds h)
            if constexpr (WBIT == 8) {
                const uint32_t row = (uint32_t)(2 * c8u) * p.row_bytes;
                t.raw[f][kk] = uint2_t{__builtin_amdgcn_raw_buffer_load_b32(p.wrsrc, p.wvoff[f], row, 0),
                                       __builtin_amdgcn_raw_buffer_load_b32(p.wrsrc, p.wvoff[f], row + p.row_bytes, 0)};
            } else {
                constexpr int CPW = 4 / WBIT;  // chunks per word: w4 1, w2 2, w1 4
                t.raw[f][kk] = __builtin_amdgcn_raw_buffer_load_b32(p.wrsrc, p.wvoff[f], (uint32_t)(c8u / CPW) * p.row_bytes, 0);
            }
        }
#pragma unroll
        for (int gi = 0; gi < NG; gi++) {
            const long g = (k0 + gi * 16) >> gshift;  // group_size >= 16: both lane halves of a step share the group
            t.sb[f][gi] = p.s[f][g * N];
            if constexpr (ZM == ZM_ASYM) t.zb[f][gi] = p.zq[f][g * (N / NB)];
            else t.zb[f][gi] = p.z[f][g * N];
        }
    }
}

// One wave per SIMD (512 registers per lane: 2*TM 32x32 accumulators, double-buffered A fragments, prefetched weight
// tiles), so the MFMA stream of a wave is covered by ITS OWN independent LDS reads / dequant VALU issued in the
// 32-cycle MFMA shadows.  The wave tile is BM x 64: every A fragment read from LDS feeds two MFMAs, which halves the
// LDS read traffic per flop (with BM x 32 wave tiles the LDS pipe -- 4 waves re-reading the whole x tile plus the
// staging writes -- was as busy as the matrix pipe).
template <int DT, int WBIT, int ZM, int BM, bool PERM, bool GPT>
__global__ __launch_bounds__(256, (BM <= 128 ? 2 : 1)) void mpq_gemm_kernel(const uint16_t* __restrict__ x, const uint32_t* __restrict__ qw,
                                                          const uint16_t* __restrict__ scales, const void* __restrict__ zeros,
                                                          const uint16_t* __restrict__ bias, const uint16_t* __restrict__ perm,
                                                          float* __restrict__ part, uint16_t* __restrict__ y, int M, int K, int N,
                                                          int gshift, int tiles_per_split, int S, int m_tiles, int n_tiles, int ldy) {
    constexpr int TM = BM / 32;              // 32-row accumulator tiles per wave
    constexpr int NF = GEMM_NF;
    constexpr int A_CHUNKS = BM * 8 / 256;   // 16-byte chunks staged per thread per tile
    constexpr int A_BYTES = BM * GEMM_BK * 2;
    constexpr int VALU_PER_MFMA = ((DT == BIE_BF16 ? 48 : 28) + TM - 1) / TM;  // dequant VALU of NF fragments spread over TM*NF MFMAs
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, j = lane & 31;
    // XCD-aware tile order: the dispatcher places block b on XCD b % 8 (speed only, never correctness).  Give each XCD
    // a contiguous chunk of the m-major tile order so that the blocks co-resident on an XCD share few x row-tiles in
    // that XCD's 4 MiB L2 instead of every XCD cycling through all of x.
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
    int ncol[NF], ncol_ld[NF];
#pragma unroll
    for (int f = 0; f < NF; f++) {
        ncol[f] = n_tile * GEMM_BN + wave * (32 * NF) + f * 32 + j;
        ncol_ld[f] = ncol[f] < N ? ncol[f] : N - 1;  // clamp loads of out-of-range columns (never stored)
    }
    WPtrs<WBIT> wp;
    {
        constexpr int NB = 32 / WBIT;
        constexpr uint32_t dummy = 0;
        (void)dummy;
        const int hrow = (WBIT == 4) ? h : (WBIT == 8 ? 2 * h : 0);  // lane-dependent packed-row offset of chunk c8u + h
        {   // descriptor inputs through readfirstlane: the compiler must KNOW they are wave-uniform, or it wraps every buffer
            // load in a waterfall loop (cdna_hip_programming.md T20)
            const uint64_t qb = (uint64_t)(uintptr_t)qw;
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)qb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(qb >> 32));
            wp.wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)hi << 32) | lo), 0,
                                                         __builtin_amdgcn_readfirstlane((uint32_t)((long)(K / NB) * N * 4)), 0x00020000);
        }
        wp.row_bytes = __builtin_amdgcn_readfirstlane((uint32_t)N * 4u);
#pragma unroll
        for (int f = 0; f < NF; f++) {
            const int n = ncol_ld[f];
            wp.wvoff[f] = (uint32_t)(hrow * N + n) * 4u;
            wp.w[f] = qw + (long)hrow * N + n;
            wp.s[f] = scales + n;
            wp.z[f] = reinterpret_cast<const uint16_t*>(zeros) + n;
            wp.zq[f] = reinterpret_cast<const uint32_t*>(zeros) + n / NB;
            wp.zshift[f] = (n % NB) * WBIT;
        }
    }
    const int split = blockIdx.y;
    const int T_total = K / GEMM_BK;
    const int t_begin = split * tiles_per_split;
    int t_end = t_begin + tiles_per_split;
    if (t_end > T_total) t_end = T_total;

    float16_t acc[NF][TM];
#pragma unroll
    for (int f = 0; f < NF; f++)
#pragma unroll
        for (int t = 0; t < TM; t++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[f][t][e] = 0.0f;

    // A staging assignment: chunk c = tid + i*256 -> row = c / 8, slot = c % 8.  Rows past M are clamped to M-1: they
    // are computed but never stored (branch-free loads).
    uint4_t areg[A_CHUNKS];
    const uint16_t* arow[A_CHUNKS];
    int aoff[A_CHUNKS];
#pragma unroll
    for (int i = 0; i < A_CHUNKS; i++) {
        const int c = tid + i * 256;
        const int row = c >> 3, s = c & 7;
        int m = m0 + row;
        if (m > M - 1) m = M - 1;
        arow[i] = x + (long)m * K + (PERM ? 0 : s * 8);
        aoff[i] = a_lds_off(row, s);
    }
    auto load_a = [&](int kt) {
#pragma unroll
        for (int i = 0; i < A_CHUNKS; i++) {
            if constexpr (!PERM) {
                areg[i] = *reinterpret_cast<const uint4_t*>(arow[i] + kt * GEMM_BK);
            } else {  // MBWQ act-order: gather x[m][q_perm[k]]
                const int s = (tid + i * 256) & 7;
                const uint16_t* pp = perm + kt * GEMM_BK + s * 8;
                const uint16_t* xr = arow[i];
                uint4_t v;
                v.x = (uint32_t)xr[pp[0]] | ((uint32_t)xr[pp[1]] << 16);
                v.y = (uint32_t)xr[pp[2]] | ((uint32_t)xr[pp[3]] << 16);
                v.z = (uint32_t)xr[pp[4]] | ((uint32_t)xr[pp[5]] << 16);
                v.w = (uint32_t)xr[pp[6]] | ((uint32_t)xr[pp[7]] << 16);
                areg[i] = v;
            }
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_CHUNKS; i++)
            *reinterpret_cast<uint4_t*>(lds + buf * A_BYTES + aoff[i]) = permute_a_chunk<DT, WBIT>(areg[i]);
    };
    // Direct global -> LDS staging of the x tile (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass -- the
    // ds_write_b128 pass alone cost ~600-900 of ~4500 cycles per K tile, profiles/r01_f_gemm_phase_cycles.txt).  One
    // wave-instruction fills 1 KiB of LDS lane-linearly (lane l -> bytes 16*l), i.e. 8 tile rows of the swizzled image;
    // the XOR swizzle is realised on the SOURCE side: lane l fetches the global chunk that belongs in LDS slot l.
    // (The MBWQ q_perm gather of x stays on the register path.)
    constexpr bool GLDS = !PERM && BM >= 32;
    constexpr int A_PIECES = BM / 32;  // 1 KiB pieces per wave per tile
    const uint16_t* asrc[A_PIECES];
    if constexpr (GLDS) {
#pragma unroll
        for (int i = 0; i < A_PIECES; i++) {
            const int piece = wave * A_PIECES + i;
            const int rp = piece * 4 + (lane >> 4), slot16 = lane & 15;
            const int row = 2 * rp + (slot16 >> 3), sl = (slot16 & 7) ^ (rp & 7);
            int m = m0 + row;
            if (m > M - 1) m = M - 1;
            asrc[i] = x + (long)m * K + sl * 8;
        }
    }
    // the tile's pieces are issued in two portions, under the first two MFMA groups of the previous tile (the third group is left so that they land before the barrier)
    // (an LDS-DMA issue costs 60-180 cycles beside MFMAs, MI355X_MICROARCH.md; all eight in one group starved it)
    auto glds_a = [&](int kt, int buf, int part) {
        if constexpr (GLDS) {
            auto* dst = (__attribute__((address_space(3))) unsigned char*)lds + buf * A_BYTES + wave * (A_PIECES * 1024);
#if defined(__HIP_DEVICE_COMPILE__)  // device-only builtin: the host pass must still see a well-formed body to emit the launch stub
#pragma unroll
            for (int i = 0; i < A_PIECES; i++)
                if (part < 0 || (i * 2) / A_PIECES == part) __builtin_amdgcn_global_load_lds(asrc[i] + kt * GEMM_BK, dst + i * 1024, 16, 0, 0);
#else
            (void)dst;
#endif
        }
    };

    // A fragment LDS byte addresses of this lane: k16-step kk -> slot kk*2 + h of row j (+ t * 4096 bytes per 32 rows)
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
    uint32_t foff[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) foff[kk] = lds_base + (uint32_t)a_lds_off(j, kk * 2 + h);

    auto dequant_step = [&](const WTile<DT, WBIT, ZM, GPT>& w, int kt, int kk, uint4_t (&bf)[NF]) {
        constexpr uint32_t M1 = (1u << WBIT) - 1u;
        const int gi = GPT ? 0 : kk;
#pragma unroll
        for (int f = 0; f < NF; f++) {
            uint32_t zb = w.zb[f][gi];
            if constexpr (ZM == ZM_ASYM) zb = ((zb >> wp.zshift[f]) & M1) + 1u;
#if BIE_GEMM_LAB == 2
            bf[f] = uint4_t{raw_as_u2<WBIT>(w.raw[f][kk]).x, w.sb[f][gi], zb, 0x3c003c00u};
#else
            bf[f] = dequant8<DT, WBIT, ZM>(raw_as_u2<WBIT>(w.raw[f][kk]), (kt * GEMM_BK + kk * 16 + h * 8) >> 3,
                                           make_col_params<DT, WBIT, ZM, (WBIT == 4)>(w.sb[f][gi], zb));
#endif
        }
    };

    // MFMAs of k16 step kk (2*TM of them) with the scheduler asked to drop VALU_PER_MFMA of the independent VALU work
    // queued before them (the next step's dequant) into every 32-cycle MFMA shadow.  One wave per SIMD issues in order:
    // an MFMA issued while the matrix pipe is busy blocks everything behind it, a run of VALU leaves the pipe idle.
    // `vmem`: global loads / LDS-DMA pieces issued in the same scheduling region (the next tile's packed words, group constants and
    // x pieces): one per MFMA shadow.  Issued in a burst ahead of the group -- 24 memory instructions back to back in the first
    // group of a tile -- they kept the in-order wave off the matrix pipe for ~650 cycles per K tile (profiles/r03_h_gemm_stamps.txt).
    auto mfma_step = [&](const uint4_t (&af)[TM], const uint4_t (&bf)[NF], int vmem) {
#pragma unroll
        for (int t = 0; t < TM; t++) {
#pragma unroll
            for (int f = 0; f < NF; f++) {
#if BIE_GEMM_LAB == 3
                acc[f][t][0] += __uint_as_float(af[t].x ^ bf[f].x ^ bf[f].y ^ bf[f].z ^ bf[f].w);
#else
                acc[f][t] = mfma32<DT>(bf[f], af[t], acc[f][t]);
#endif
            }
        }
#pragma unroll
        for (int i = 0; i < TM * NF; i++) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);  // then VALU
            if (i < vmem) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // then one global load / LDS-DMA piece
        }
    };

    // Software pipeline, rotated across the K-tile boundary: EVERY MFMA group has the next group's A fragments in flight
    // from LDS and the next group's B fragments being dequantised -- including the last group of a tile, which covers
    // the staging of the next x tile (registers -> LDS, barrier) and the dequant of the next tile's first step.  The
    // global loads of tile kt+1 (x rows into registers, packed weights + group constants) are issued under the FIRST MFMA
    // group of tile kt and consumed under its last one, so nothing loaded is pending across the loop's back edge (the
    // compiler would otherwise have to wait for fresh loads before it can copy them into the loop-carried registers).
    WTile<DT, WBIT, ZM, GPT> wcur, wnext;
    const int t_last = t_end - 1;
    int cur = 0;
    uint4_t af0[TM], af1[TM];
    uint4_t bfrag[NF], bnext[NF];
    if (t_begin < t_end) {
        if constexpr (GLDS) glds_a(t_begin, 0, -1);
        else load_a(t_begin);
        load_wtile<DT, WBIT, ZM, GPT>(wcur, wp, t_begin * GEMM_BK, N, gshift);
        if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else store_a(0);
        __syncthreads();
        lds_issue_frags<TM>(af0, foff[0]);
        dequant_step(wcur, t_begin, 0, bfrag);
        lds_wait_frags<TM>(af0);
    }
#ifdef BIE_GEMM_STAMPS
    unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
    unsigned stamp_sum[6] = {0, 0, 0, 0, 0, 0};
#define BIE_STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamp_sum[i] += (unsigned)(now_ - stamp_prev); stamp_prev = now_; }
#else
#define BIE_STAMP(i)
#endif
    for (int kt = t_begin; kt < t_end; kt++) {
        // Branch-free body: the look-ahead tile index is clamped to the last tile (a harmless re-load at the end).
        const int ktn = (kt + 1 < t_end) ? kt + 1 : t_last;
        const uint32_t abase = (uint32_t)(cur * A_BYTES), abase_n = (uint32_t)((cur ^ 1) * A_BYTES);
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            if (kk & 1) lds_issue_frags<TM>(af0, foff[kk + 1] + abase);
            else lds_issue_frags<TM>(af1, foff[kk + 1] + abase);
            __builtin_amdgcn_sched_barrier(0);  // keep the MFMAs below the issue point of the fragment reads
            // the other x buffer: its last readers finished before the previous barrier.  The DMA pieces and the next tile's loads
            // share the scheduling region of this group's MFMAs (one memory instruction per MFMA shadow)
#if BIE_GEMM_LAB == 9     // timing experiment: no loads / DMA in the loop at all (stale operands)
            if (kt < 0) { glds_a(ktn, cur ^ 1, kk); load_wtile<DT, WBIT, ZM, GPT>(wnext, wp, ktn * GEMM_BK, N, gshift); }
#elif BIE_GEMM_LAB == 8   // timing experiment: everything the first group issues moves to the second and third
            if constexpr (GLDS) { if (kk >= 1) glds_a(ktn, cur ^ 1, kk - 1); }
            if (kk == 1) load_wtile<DT, WBIT, ZM, GPT>(wnext, wp, ktn * GEMM_BK, N, gshift);
#else
            if constexpr (GLDS) glds_a(ktn, cur ^ 1, kk);
            if (kk == 0) {
                if constexpr (!GLDS) load_a(ktn);
                load_wtile<DT, WBIT, ZM, GPT>(wnext, wp, ktn * GEMM_BK, N, gshift);
            }
#endif
            dequant_step(wcur, kt, kk + 1, bnext);
            constexpr int NLD = (WBIT == 8 ? 16 : 8) + 2 * NF * (GPT ? 1 : 4);  // loads of load_wtile
            const int nv = (GLDS && kk < 2 ? A_PIECES / 2 : 0) + (kk == 0 ? NLD : 0);
            if (kk & 1) mfma_step(af1, bfrag, nv);
            else mfma_step(af0, bfrag, nv);
            if (kk & 1) lds_wait_frags<TM>(af0);
            else lds_wait_frags<TM>(af1);
#pragma unroll
            for (int f = 0; f < NF; f++) bfrag[f] = bnext[f];
            BIE_STAMP(kk)
        }
        // last step of the tile (A fragments in af1): x tile kt+1 goes to the other LDS buffer -- its last readers finished
        // before the previous barrier -- and the pipeline is primed for it
        if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA pieces have landed
        else store_a(cur ^ 1);
        BIE_STAMP(3)
        __syncthreads();
        BIE_STAMP(4)
        lds_issue_frags<TM>(af0, foff[0] + abase_n);
        __builtin_amdgcn_sched_barrier(0);
        dequant_step(wnext, ktn, 0, bnext);
        mfma_step(af1, bfrag, 0);
        lds_wait_frags<TM>(af0);
#pragma unroll
        for (int f = 0; f < NF; f++) bfrag[f] = bnext[f];
        wcur = wnext;
        cur ^= 1;
        BIE_STAMP(5)
    }

    // ---- epilogue.  The MFMAs were issued as D = W_frag (A operand, rows = n) x x_frag (B operand, cols = m), so in the
    // 32x32 C/D layout (col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)) a lane holds ONE output row m = t*32 + j
    // and, per group of four registers, FOUR CONSECUTIVE columns n = 8*q + 4*h + (0..3): one 8-byte store per group
    // (4 per tile instead of 16 two-byte stores); the two lane halves complete 16 contiguous bytes per row.
#ifdef BIE_GEMM_STAMPS
    if (acc[0][0][0] == 123.456f)
#endif
#pragma unroll
    for (int f = 0; f < NF; f++) {
        const int nf0 = n_tile * GEMM_BN + wave * (32 * NF) + f * 32;  // first column of this fragment (wave-uniform)
        const bool use_bias = (S == 1) && (bias != nullptr);
#pragma unroll
        for (int t = 0; t < TM; t++) {
            const int row = m0 + t * 32 + j;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int n0 = nf0 + 8 * q + 4 * h;
                if (row < M && n0 < N) {  // N % 4 == 0 on this path: a group of 4 columns is all-or-nothing
                    if (S == 1) {
                        float o[4];
#pragma unroll
                        for (int c = 0; c < 4; c++) o[c] = acc[f][t][4 * q + c];
                        if (use_bias) {  // y = dt(dt(acc) + bias); without bias the pack below is the one rounding (an explicit round first costs 3 VALU per value)
#pragma unroll
                            for (int c = 0; c < 4; c++) o[c] = dt_traits<DT>::round(o[c]) + dt_traits<DT>::load(bias, n0 + c);
                        }
                        uint2_t pk;
                        if constexpr (DT == BIE_F16) {
                            pk.x = f32_to_f16_bits(o[0]) | (f32_to_f16_bits(o[1]) << 16);
                            pk.y = f32_to_f16_bits(o[2]) | (f32_to_f16_bits(o[3]) << 16);
                        } else {
                            pk.x = pack_bf16x2(o[0], o[1]);
                            pk.y = pack_bf16x2(o[2], o[3]);
                        }
                        *reinterpret_cast<uint2_t*>(y + (long)row * ldy + n0) = pk;  // ldy: the row pitch of y (N, or a wider destination's)
                    } else {
                        float4_t v = {acc[f][t][4 * q], acc[f][t][4 * q + 1], acc[f][t][4 * q + 2], acc[f][t][4 * q + 3]};
                        *reinterpret_cast<float4_t*>(part + ((long)split * M + row) * N + n0) = v;
                    }
                }
            }
        }
    }
#ifdef BIE_GEMM_STAMPS
    // timing build only: the LAST row of y receives, per wave of block 0, the six phase cycle sums (y is garbage there)
    __syncthreads();
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
        uint32_t* dbg = reinterpret_cast<uint32_t*>(y + (long)(M - 1) * N) + wave * 8;
        for (int i = 0; i < 6; i++) dbg[i] = stamp_sum[i];
        dbg[6] = (uint32_t)(t_end - t_begin);
    }
#endif
}

// ---- launch plumbing ---------------------------------------------------------------------------------
struct GemmPlan {
    int BM, S, tiles_per_split;
};

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// Tile height and split-K factor from a measured cost model (tools/gemm_plan_sweep2.py,
// profiles/r01_g_gemm_plan_sweep_after_glds.log).  A K tile (64 k) of a BM x 256 block tile costs c0(BM) us on an otherwise
// idle chip (the dequant of the 256 columns is paid per tile whatever BM is, hence the weak dependence on BM) and up to
// 35 % more with all CUs busy; BM = 256 runs one block per CU (512 registers per wave), BM <= 128 two (a co-resident pair
// takes ~1.7x one block).  A launch is a sequence of rounds of `cap` blocks; split-K adds S fp32 partial slabs written and
// re-read plus the finalize launch.  The plan depends on (M, K, N) only: bie_mpq_workspace_bytes has to reproduce it
// without knowing dtype or bit width.
#include "mpq_gemm_plan_table.inc"

// nearest grid point in log space; -1 when x is more than 20 % outside the grid
static int plan_grid_index(const int* g, int n, long x) {
    if ((double)x * 1.2 < (double)g[0] || (double)x > (double)g[n - 1] * 1.2) return -1;
    int i = 0;
    while (i + 1 < n && (double)x * (double)x > (double)g[i] * (double)g[i + 1]) i++;
    return i;
}

static GemmPlan plan_gemm(int M, int K, int N) {
    // tuning knobs: read ONCE per process -- unless BIE_TUNING is set (tests / sweep tools change them between calls), in which
    // case they are re-read on every launch.  No getenv on the product's launch path.
    static const bool tuning = getenv("BIE_TUNING") != nullptr;
    static const int bm_once = env_int("BIE_GEMM_BM", 0), s_once = env_int("BIE_GEMM_S", 0);
    const int force_bm = tuning ? env_int("BIE_GEMM_BM", 0) : bm_once, force_s = tuning ? env_int("BIE_GEMM_S", 0) : s_once;
    const int T = K / GEMM_BK;
    // Round 6: where a sweep of EVERY (BM, S) over 40 layer shapes x 11 row counts found a plan more than 2 % ahead of this model's choice
    // (114 of 440 cells, up to 24 %: profiles/r06_gemm_plan_table_sweep.txt -> mpq_gemm_plan_table.inc), the grid point's plan is taken for the grid's own (K, N) and the
    // row counts around its M that make the same number of row tiles.  BIE_GEMM_PLAN_TABLE=0: the model alone.
    static const int table_once = env_int("BIE_GEMM_PLAN_TABLE", 1);
    if (!force_bm && !force_s && (tuning ? env_int("BIE_GEMM_PLAN_TABLE", 1) : table_once)) {
        const int ki = plan_grid_index(kPlanK, (int)(sizeof(kPlanK) / sizeof(int)), K), ni = plan_grid_index(kPlanN, (int)(sizeof(kPlanN) / sizeof(int)), N),
                  mi = plan_grid_index(kPlanM, (int)(sizeof(kPlanM) / sizeof(int)), M);
        if (ki >= 0 && ni >= 0 && mi >= 0) {
            const unsigned e = kPlanTable[ki][ni][mi];
            const int BM = 32 << (e >> 5), S = (int)(e & 31u);
            // a plan is a statement about tile COUNTS: it is taken for the grid's own K and N only, and for row counts that make as many row tiles as the
            // grid point's (nearest-point lookup across shapes measured WORSE than the model off the grid: 1.015 against 1.005 over 132 held-out cells,
            // up to +28 % -- profiles/r06_gemm_plan_table_holdout_nearest.txt)
            if (e != 0 && K == kPlanK[ki] && N == kPlanN[ni] && cdiv(M, BM) == cdiv(kPlanM[mi], BM) && !(BM > 32 && BM >= 2 * M) && S >= 1 && (S == 1 || T / S >= 2)) {
                const int tps = cdiv(T, S);
                return GemmPlan{BM, cdiv(T, tps), tps};
            }
        }
    }
    const int bms[4] = {32, 64, 128, 256};
    const double c0[4] = {0.93, 1.00, 1.19, 1.53};
    double best = 1e30;
    GemmPlan p{256, 1, T};
    for (int i = 0; i < 4; i++) {
        const int BM = bms[i];
        if (force_bm ? (BM != force_bm) : (BM > 32 && BM >= 2 * M)) continue;
        const long tiles = (long)cdiv(M, BM) * cdiv(N, GEMM_BN);
        const long cap = BM <= 128 ? 512 : 256;
        for (int S = 1; S <= 16; S++) {
            if (force_s ? (S != (force_s > T ? T : force_s)) : (S > 1 && T / S < 2)) continue;
            const int tps = cdiv(T, S), Sr = cdiv(T, tps);
            if (Sr != S && !force_s) continue;  // an equivalent smaller S exists
            const long blocks = tiles * Sr;
            double units = 0.0;  // in K-tile times of one lone block
            for (long left = blocks; left > 0; left -= cap) {
                const long r = left < cap ? left : cap;
                const double busy = r < 256 ? (double)r / 256.0 : 1.0;
                const double pair = r > 256 ? 1.7 : 1.0;  // the round lasts as long as its busiest CU
                units += (1.0 + 0.35 * busy) * pair;
            }
            // + pipeline fill and epilogue of a block: about three K tiles, six for the 256-row tile (profiles/r03_plan_sweep3.txt: with 3 the
            // model tied (256, S = 16) with (128, S = 8) at 11008 -> 4096, M = 256 and took the former: 72 us against 46)
            double t = units * (tps + (BM == 256 ? 6 : 3)) * c0[i];
            if (Sr > 1) t += 5.0 + (double)Sr * M * N * 8.0 / 6.0e6;
            if (t < best) { best = t; p.BM = BM; p.S = Sr; p.tiles_per_split = tps; }
        }
    }
    return p;
}

static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

bool mpq_gemm_ok(int M, int K, int N, int w_bit, int group_size, int dtype, bool has_gidx) {
    if (dtype != BIE_F16 && dtype != BIE_BF16) return false;
    if (has_gidx) return false;
    if (K % GEMM_BK) return false;
    if (group_size < K && !(is_pow2(group_size) && group_size >= 16)) return false;
    if (N % (32 / w_bit) || (N & 3)) return false;
    return true;
}

// mpq_dense.hip: dequantise once + dense GEMM, for M large enough that the fused kernel would dequantise every tile several times
bool mpq_dense_ok(int M, int K, int N);
size_t mpq_dense_workspace_bytes(int K, int N);
int mpq_dense_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const void* bias, void* y, void* scratch, int M, int K, int N,
                     int w_bit, int gshift, int zm, int dtype, hipStream_t st, int ldy);

size_t mpq_gemm_workspace_bytes(int M, int K, int N) {
    if (K % GEMM_BK) return 0;
    const GemmPlan p = plan_gemm(M, K, N);
    const size_t split = p.S > 1 ? (size_t)p.S * M * N * sizeof(float) : 0;
    const size_t dense = mpq_dense_ok(M, K, N) ? mpq_dense_workspace_bytes(K, N) : 0;
    return split > dense ? split : dense;
}

struct GemmArgs {
    const void* x; const int32_t* qw; const void* scales; const void* zeros; const void* bias; const uint16_t* perm;
    float* part; void* y; int M, K, N, gshift;
    hipStream_t st;
    int ldy;
};

template <int DT, int WBIT, int ZM, bool PERM, bool GPT>
static int gemm_launch_bm(const GemmPlan& p, const GemmArgs& a) {
    const int m_tiles = cdiv(a.M, p.BM), n_tiles = cdiv(a.N, GEMM_BN);
    dim3 grid(m_tiles * n_tiles, p.S);
    const size_t lds = (size_t)2 * p.BM * GEMM_BK * 2;
#define L(BMV)                                                                                                          \
    hipLaunchKernelGGL((mpq_gemm_kernel<DT, WBIT, ZM, BMV, PERM, GPT>), grid, dim3(256), lds, a.st, (const uint16_t*)a.x, \
                       (const uint32_t*)a.qw, (const uint16_t*)a.scales, a.zeros, (const uint16_t*)a.bias, a.perm, a.part, \
                       (uint16_t*)a.y, a.M, a.K, a.N, a.gshift, p.tiles_per_split, p.S, m_tiles, n_tiles, a.ldy)
    switch (p.BM) {
        case 32: L(32); break;
        case 64: L(64); break;
        case 128: L(128); break;
        default: L(256); break;
    }
#undef L
    return check_launch("mpq_gemm_kernel");
}

template <int DT, int WBIT, int ZM, bool PERM>
static int gemm_launch_g(const GemmPlan& p, const GemmArgs& a, bool gpt) {
    return gpt ? gemm_launch_bm<DT, WBIT, ZM, PERM, true>(p, a) : gemm_launch_bm<DT, WBIT, ZM, PERM, false>(p, a);
}

// MPQ flavours: {f16, bf16} x {1,2,4,8} x {sym, asym}, no x gather
template <int DT, int ZM>
static int gemm_launch_mpq_w(const GemmPlan& p, const GemmArgs& a, int w_bit, bool gpt) {
    switch (w_bit) {
        case 1: return gemm_launch_g<DT, 1, ZM, false>(p, a, gpt);
        case 2: return gemm_launch_g<DT, 2, ZM, false>(p, a, gpt);
        case 4: return gemm_launch_g<DT, 4, ZM, false>(p, a, gpt);
        default: return gemm_launch_g<DT, 8, ZM, false>(p, a, gpt);
    }
}

// ldy: the row pitch of y in elements (N for a tight [M, N]; a wider destination lets a column shard's epilogue store straight into
// its column range of the full output -- SURVEY section 8e).  The split-K finalize pass writes tight rows only.
int mpq_gemm_launch_ld(const void* x, const int32_t* qw, const void* scales, const void* zeros, const void* bias, void* y,
                       float* part, int M, int K, int N, int w_bit, int group_size, int zm, int dtype, const uint16_t* perm,
                       hipStream_t st, int ldy);
int mpq_gemm_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const void* bias, void* y,
                    float* part, int M, int K, int N, int w_bit, int group_size, int zm, int dtype, const uint16_t* perm,
                    hipStream_t st) {
    return mpq_gemm_launch_ld(x, qw, scales, zeros, bias, y, part, M, K, N, w_bit, group_size, zm, dtype, perm, st, N);
}
bool mpq_gemm_pitch_ok(int M, int K, int N, int ldy) {  // a pitched destination: 8-byte stores need ldy % 4 == 0, and no split-K finalize pass
    return ldy >= N && (ldy & 3) == 0 && (ldy == N || plan_gemm(M, K, N).S == 1 || mpq_dense_ok(M, K, N));
}
int mpq_gemm_launch_ld(const void* x, const int32_t* qw, const void* scales, const void* zeros, const void* bias, void* y,
                       float* part, int M, int K, int N, int w_bit, int group_size, int zm, int dtype, const uint16_t* perm,
                       hipStream_t st, int ldy) {
    const GemmPlan p = plan_gemm(M, K, N);
    int gshift = 31;  // group_size >= K: a single group
    bool gpt = true;
    if (group_size < K) {
        gshift = 0;
        while ((1 << gshift) < group_size) gshift++;
        gpt = (group_size % GEMM_BK) == 0;
    }
    if (!perm && mpq_dense_ok(M, K, N) && (zm != ZM_FUSED || dtype == BIE_F16))  // large M: dequantise once into `part`, dense MFMA GEMM
        return mpq_dense_launch(x, qw, scales, zeros, bias, y, part, M, K, N, w_bit, gshift, zm, dtype, st, ldy);
    if (ldy != N && p.S > 1) {
        set_error("mpq_gemm_launch: a pitched destination (ldy=%d, N=%d) is not served by the split-K plan of this shape", ldy, N);
        return BIE_ERR_UNSUPPORTED;
    }
    const GemmArgs a{x, qw, scales, zeros, bias, perm, part, y, M, K, N, gshift, st, ldy};
    int rc;
    if (zm == ZM_FUSED) {  // MBWQ uniform: fp16, 2/4 bit, optional q_perm gather
        if (dtype != BIE_F16 || !(w_bit == 2 || w_bit == 4)) {
            set_error("mpq_gemm_launch: fused-rounding (MBWQ) mode needs fp16 and 2/4-bit weights");
            return BIE_ERR_UNSUPPORTED;
        }
        if (w_bit == 4) rc = perm ? gemm_launch_g<BIE_F16, 4, ZM_FUSED, true>(p, a, gpt) : gemm_launch_g<BIE_F16, 4, ZM_FUSED, false>(p, a, gpt);
        else rc = perm ? gemm_launch_g<BIE_F16, 2, ZM_FUSED, true>(p, a, gpt) : gemm_launch_g<BIE_F16, 2, ZM_FUSED, false>(p, a, gpt);
    } else {
        if (perm) {
            set_error("mpq_gemm_launch: q_perm is only supported in MBWQ mode");
            return BIE_ERR_UNSUPPORTED;
        }
        if (dtype == BIE_F16) rc = zm == ZM_ASYM ? gemm_launch_mpq_w<BIE_F16, ZM_ASYM>(p, a, w_bit, gpt) : gemm_launch_mpq_w<BIE_F16, ZM_SYM>(p, a, w_bit, gpt);
        else rc = zm == ZM_ASYM ? gemm_launch_mpq_w<BIE_BF16, ZM_ASYM>(p, a, w_bit, gpt) : gemm_launch_mpq_w<BIE_BF16, ZM_SYM>(p, a, w_bit, gpt);
    }
    if (rc) return rc;
    if (p.S > 1) return launch_splitk_finalize(part, bias, y, p.S, M, N, dtype, st);
    return BIE_OK;
}

}  // namespace bie
