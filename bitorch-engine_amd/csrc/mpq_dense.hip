// W{1,2,4,8}A16 GEMM for large M: dequantise ONCE per call, then a dense bf16 / fp16 MFMA GEMM on the hand-ordered pipeline of
// mfma_pipe.cuh.  Replaces the same branch of the reference as mpq_gemm.hip -- "materialise the fp16 weight, then cuBLAS"
// (layers/qlinear/nbit/cuda/mpq_layer.py:59-63, unpack_qweight utils.py:30-51) -- with the same arithmetic: the weight values are
// produced by dequant8 (mpq_frag_dequant.cuh: the reference's two roundings), accumulated in fp32, one rounding at the store.
//
// Why a second form: the fused kernel (mpq_gemm.hip) dequantises every weight tile once per 256-row M tile, in registers, beside the
// MFMAs -- 6 VALU per weight at the reference's roundings, and the VALU + LDS stream alone takes 60-67 % of its loop
// (profiles/r03_t_gemm_ablations.txt).  At M >= 1024 a tile is dequantised 4-16 times.  Here the weights are
// dequantised once into MFMA FRAGMENT ORDER (K*N*2 bytes of scratch, L2 / MALL resident while the GEMM runs): fragment (nb, ks) =
// columns 32*nb .. +31 x k 16*ks .. +15, lane l owns the 8 values of column 32*nb + (l & 31), k 16*ks + 8*(l >> 5) .. +7 -- what
// v_mfma_f32_32x32x16 takes from that lane -- so one global_load_lds_dwordx4 per wave moves a fragment into LDS, contiguous on both
// sides, no swizzle.  x stays row-major: its LDS image is [rows][32 k] with the 16-byte slot index XORed by (row >> 2) & 3, realised on
// the SOURCE side of the LDS-DMA (lane l fetches the chunk that belongs in slot l), so the fragment reads are conflict-free.
// The loop is binary_fp4.hip's: 4 waves as 2 x 2, wave tile 32*WM x 32*WN, K = 32 per stage (the same 32 KiB / 32 MFMAs per wave as an
// FP4 stage of K = 128), 3 stages, one barrier per stage between its two MFMA clusters, reads behind a cluster's first MFMAs, refill
// pieces behind the rest, MFMAs as ordered inline asm.
#include "mpq_frag_dequant.cuh"
#include "mfma_pipe.cuh"
#include <stdlib.h>
#include <type_traits>

#pragma clang fp contract(off)

#ifndef BIE_DENSE_LAB
#define BIE_DENSE_LAB 0  // compile-time ablation switch of tools/dense_lab.py (0 = product code): 1 no epilogue stores, 2 no main loop,
#endif                   // 3 the round-4 epilogue (64 rows x 16 bytes per store instruction); 8 the MFMA-pipe probe (round 6): the loop's own
                         // instruction stream -- fragment reads, barriers, MFMAs -- on operands that STAY in LDS after the prologue (no LDS-DMA,
                         // no vmcnt wait in the loop, no stores): what this tiling can reach on this chip with zero memory traffic; 9: 8 without the
                         // fragment reads (MFMAs and barriers only)

namespace bie {

// ---- pass 1: packed weights -> dequantised fragments -------------------------------------------------------------
// One wave takes FPW consecutive fragments of one 32-column block (FPW divides K / 16, so they share the block): FPW independent packed
// loads in flight, FPW KiB of the image written back to back.
template <int DT, int WBIT, int ZM, int FPW>
__global__ __launch_bounds__(256) void mpq_dequant_frag_kernel(const uint32_t* __restrict__ qw, const uint16_t* __restrict__ scales, const void* __restrict__ zeros,
                                                               uint4_t* __restrict__ img, int N, int gshift, long nfrag, int ks_per_col) {
    const long f0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * FPW;
    if (f0 >= nfrag) return;
    const int lane = threadIdx.x & 63;
    const long nb = f0 / ks_per_col;
    const int ks0 = (int)(f0 - nb * ks_per_col);
    int n = (int)nb * 32 + (lane & 31);
    if (n > N - 1) n = N - 1;  // columns past N: valid values nobody stores
    uint2_t raw[FPW];
    uint32_t sb[FPW], zb[FPW];
#pragma unroll
    for (int i = 0; i < FPW; i++) {
        const int c8 = (ks0 + i) * 2 + (lane >> 5);
        const long g = (long)(c8 * 8) >> gshift;
        raw[i] = load_chunk<WBIT>(qw, c8, n, N);
        sb[i] = scales[g * N + n];
        if constexpr (ZM == ZM_ASYM) {
            constexpr int NB = 32 / WBIT;
            constexpr uint32_t M1 = (1u << WBIT) - 1u;
            const uint32_t word = reinterpret_cast<const uint32_t*>(zeros)[g * (N / NB) + n / NB];
            zb[i] = ((word >> ((n % NB) * WBIT)) & M1) + 1u;
        } else {
            zb[i] = reinterpret_cast<const uint16_t*>(zeros)[g * N + n];
        }
    }
#pragma unroll
    for (int i = 0; i < FPW; i++) {
        const int c8 = (ks0 + i) * 2 + (lane >> 5);
        img[(f0 + i) * 64 + lane] = dequant8<DT, WBIT, ZM>(raw[i], c8, make_col_params<DT, WBIT, ZM, (WBIT == 4)>(sb[i], zb[i]));
    }
}

// The same image for an EXPLICIT g_idx that is not a permutation of k // group_size (unequal groups: the act-order forms that ARE such a
// permutation are re-ordered at load time, q_linear_cuda.act_order_sorted): every k carries its own group, so the 8 values of a lane are
// dequantised one by one with the scalar functions (same roundings).  Replaces the reference's dense weight of the M > 32 branch for such
// checkpoints (unpack_qweight with g_idx, layers/qlinear/nbit/cuda/utils.py:36-51) -- rare, not tuned, but no vendor GEMM behind it.
template <int DT, int WBIT, bool ASYM>
__global__ __launch_bounds__(256) void mpq_dequant_frag_gidx_kernel(const uint32_t* __restrict__ qw, const uint16_t* __restrict__ scales, const void* __restrict__ zeros,
                                                                    const int32_t* __restrict__ g_idx, uint4_t* __restrict__ img, int N, long nfrag, int ks_per_col) {
    const long f = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= nfrag) return;
    const int lane = threadIdx.x & 63;
    const long nb = f / ks_per_col;
    const int ks = (int)(f - nb * ks_per_col);
    int n = (int)nb * 32 + (lane & 31);
    if (n > N - 1) n = N - 1;
    constexpr int NBW = 32 / WBIT;
    constexpr uint32_t M1 = (1u << WBIT) - 1u;
    const int k0 = ks * 16 + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int k = k0 + e;
        const uint32_t q = (qw[(long)(k / NBW) * N + n] >> ((k % NBW) * WBIT)) & M1;
        const long g = g_idx[k];
        const float sc = dt_traits<DT>::load(scales, g * N + n);
        if constexpr (ASYM) {
            const uint32_t word = reinterpret_cast<const uint32_t*>(zeros)[g * (N / NBW) + n / NBW];
            v[e] = dequant_scalar_asym<DT>(q, sc, (int)((word >> ((n % NBW) * WBIT)) & M1) + 1);
        } else {
            v[e] = dequant_scalar_sym<DT>(q, sc, dt_traits<DT>::load(zeros, g * N + n));
        }
    }
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if constexpr (DT == BIE_BF16) o[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
        else o[i] = f32_to_f16_bits(v[2 * i]) | (f32_to_f16_bits(v[2 * i + 1]) << 16);
    }
    img[f * 64 + lane] = uint4_t{o[0], o[1], o[2], o[3]};
}

// ---- pass 2: dense GEMM ----------------------------------------------------------------------------------------------
template <int DT>
__device__ __forceinline__ void mfma16(float16_t& c, const v4i_t& a, const v4i_t& b) {
    if constexpr (DT == BIE_F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

template <int DT, int WM, int WN>
__global__ __launch_bounds__(256) void mpq_dense_gemm_kernel(const uint16_t* __restrict__ x, const uint8_t* __restrict__ wimg, const uint16_t* __restrict__ bias,
                                                             uint16_t* __restrict__ y, int M, int N, int K, int tiles_n, int NB32, int gm, int ldy,
                                                             float* __restrict__ part, int kt_per_split) {
    constexpr int AF = 2 * WM, BF = 2 * WN;  // 32-row / 32-column blocks per workgroup tile
    constexpr int NFR = (AF + BF) * 2;       // KiB per stage (k = 32): x rows 64 bytes each, weight fragments 1 KiB per k16 step
    constexpr int PW = NFR / 4;              // LDS-DMA pieces per wave and stage
    constexpr int STAGE = NFR * 1024;
    constexpr int NR = WM + WN, NM = WM * WN;
    constexpr int RPM = (2 * NR + NM - 1) / NM, M0 = (NR + RPM - 1) / RPM, DPM = (PW + (NM - M0) - 1) / (NM - M0);
    static_assert((AF * 2) % PW == 0, "a wave's pieces are all x pieces or all weight pieces");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * STAGE];

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wy = wave >> 1, wx = wave & 1;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    int tile_m, tile_n;
    pipe_tile(bid, nblk, tiles_n, gm, tile_m, tile_n);
    // K split over blockIdx.y (few tiles, long K: the mixed-bit layout's prefill at 49 <= M < 1024): split s takes the stages
    // [s * kt_per_split, ...) and leaves its fp32 sums in part[s][M][N] for launch_splitk_finalize; part == nullptr: the whole K, y directly
    const int KS = K >> 4;
    // (128 x 128 tiles only: the 256 x 256 instance has no register to spare for it and never needs it)
    constexpr bool SPLITK = WM == 2 && WN == 2;
    int kt0 = 0, KT = K >> 5;
    if constexpr (SPLITK) {
        if (part) {
            kt0 = (int)blockIdx.y * kt_per_split;
            KT = (K >> 5) - kt0 < kt_per_split ? (K >> 5) - kt0 : kt_per_split;
        }
    }

    // this wave's LDS-DMA sources.  Stage image: [x: AF*2 pieces of 16 rows x 64 bytes][weights: BF*2 fragments (column block, k16 step)]
    const bool x_wave = wave * PW < AF * 2;
    [[maybe_unused]] const long step = x_wave ? 64 : 2048;  // bytes per stage: 32 k of a row / two fragments
    const uint8_t* src[PW];
#pragma unroll
    for (int j = 0; j < PW; j++) {
        const int p = wave * PW + j;
        if (x_wave) {
            const int rt = p * 16 + (lane >> 2);  // row of the tile; LDS slot lane & 3 holds logical slot (lane & 3) ^ ((rt >> 2) & 3)
            long m = (long)tile_m * (AF * 32) + rt;
            if (m > M - 1) m = M - 1;
            src[j] = reinterpret_cast<const uint8_t*>(x) + (m * K) * 2 + (((lane & 3) ^ ((rt >> 2) & 3)) << 4) + (long)kt0 * 64;
        } else {
            const int q = p - AF * 2;
            long nb = (long)tile_n * BF + (q >> 1);
            if (nb > NB32 - 1) nb = NB32 - 1;
            src[j] = wimg + ((nb * KS + (q & 1)) * 64 + lane) * 16 + (long)kt0 * 2048;
        }
    }
    [[maybe_unused]] const int kt_last = KT - 1;
    auto issue_piece = [&](int kt, int j) {
#if defined(__HIP_DEVICE_COMPILE__)
#if BIE_DENSE_LAB == 8 || BIE_DENSE_LAB == 9
        if (kt >= 3) return;  // probe: only the prologue's three stages are ever fetched
#endif
        const int ks = kt < kt_last ? kt : kt_last;  // a look-ahead past the end re-fetches the last tile into a buffer nobody reads again
        auto* dst = (__attribute__((address_space(3))) unsigned char*)lds + (kt % 3) * STAGE + wave * (PW * 1024);
        __builtin_amdgcn_global_load_lds(src[j] + ks * step, dst + j * 1024, 16, 0, 0);
#endif
    };

    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
    uint32_t a_addr[2];  // x fragment of k16 step s: row rl of the wave's first block, logical slot 2*s + hh
    {
        const int rl = lane & 31, hh = lane >> 5, sw = (rl >> 2) & 3;
        a_addr[0] = lds_base + (uint32_t)((wy * WM * 32 + rl) * 64 + ((hh ^ sw) << 4));
        a_addr[1] = lds_base + (uint32_t)((wy * WM * 32 + rl) * 64 + (((2 + hh) ^ sw) << 4));
    }
    const uint32_t b_addr = lds_base + AF * 2048 + (wx * WN * 2) * 1024 + lane * 16;

    float16_t acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    v4i_t XA[WM], XB[WN], YA[WM], YB[WN], ZA[WM], ZB[WN];
    // fragment read R (0 .. NR-1) of k16 step H of the stage at byte offset so: the first WM are x row blocks (2 KiB apart)
    auto read_item = [&](auto ic, auto hc, uint32_t so, v4i_t (&TA)[WM], v4i_t (&TB)[WN]) {
        constexpr int R = decltype(ic)::value, H = decltype(hc)::value;
#if BIE_DENSE_LAB == 9
        if (so == 0xffffffffu)  // never: the fragment registers keep the prologue's values
#endif
        if constexpr (R < WM) TA[R] = lds_read16<R * 2048>(a_addr[H] + so);
        else TB[R - WM] = lds_read16<(R - WM) * 2048 + H * 1024>(b_addr + so);
    };

#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int j = 0; j < PW; j++) issue_piece(s, j);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NR>([&](auto rc) { read_item(rc, ic_t<0>{}, 0u, XA, XB); });

    auto stage = [&](int kt, v4i_t (&PA)[WM], v4i_t (&PB)[WN], v4i_t (&QA)[WM], v4i_t (&QB)[WN], v4i_t (&NA)[WM], v4i_t (&NB)[WN]) {
        const uint32_t so = (uint32_t)(kt % 3) * STAGE, sn = (uint32_t)((kt + 1) % 3) * STAGE;
        wait_frags<0>(PA, PB);
        static_for<0, NM>([&](auto mc) {
            constexpr int m = decltype(mc)::value, i = m / WN, j = m % WN;
            mfma16<DT>(acc[i][j], PB[j], PA[i]);  // D = w_frag (rows = output features) x x_frag (columns = rows of x)
            static_for<imin(m * RPM, NR), imin((m + 1) * RPM, NR)>([&](auto rc) { read_item(rc, ic_t<1>{}, so, QA, QB); });
        });
        wait_frags<0>(QA, QB);  // every LDS read of this stage has returned: its buffer may be refilled behind the barrier
#if BIE_DENSE_LAB != 5 && BIE_DENSE_LAB != 6 && BIE_DENSE_LAB != 8 && BIE_DENSE_LAB != 9
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");  // K tile kt+1 landed (kt+2 still in flight)
#endif
#if BIE_DENSE_LAB == 4 || BIE_DENSE_LAB == 6      // timing experiments (wrong results): no barrier in the loop (4), no counted wait (5), neither (6)
        if (kt & 1)
#endif
        __builtin_amdgcn_s_barrier();
        static_for<0, NM>([&](auto mc) {
            constexpr int m = decltype(mc)::value, i = m / WN, j = m % WN;
            mfma16<DT>(acc[i][j], QB[j], QA[i]);
            static_for<imin(m * RPM, NR), imin((m + 1) * RPM, NR)>([&](auto rc) { read_item(rc, ic_t<0>{}, sn, NA, NB); });
            if constexpr (m >= M0)
                static_for<imin((m - M0) * DPM, PW), imin((m - M0 + 1) * DPM, PW)>([&](auto pc) { issue_piece(kt + 3, decltype(pc)::value); });
        });
    };
    int kt = 0;
#if BIE_DENSE_LAB == 2
    kt = KT;
#endif
    for (; kt + 3 <= KT; kt += 3) {
        stage(kt, XA, XB, YA, YB, ZA, ZB);
        stage(kt + 1, ZA, ZB, XA, XB, YA, YB);
        stage(kt + 2, YA, YB, ZA, ZB, XA, XB);
    }
    if (kt < KT) {
        stage(kt, XA, XB, YA, YB, ZA, ZB);
        if (kt + 1 < KT) stage(kt + 1, ZA, ZB, XA, XB, YA, YB);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the clamped look-ahead pieces / reads must not outlive the workgroup's LDS
    mfma_drain();

    int le = threadIdx.x & 63;
    asm volatile("" : "+v"(le));  // everything the epilogue derives from the lane id is computed HERE, not carried through the loop (512 registers: no room)
    const int rl = le & 31, hh = le >> 5;
    if constexpr (SPLITK) if (part != nullptr) {  // K split: fp32 partial sums, four consecutive features per lane and register quad (16-byte stores)
        float* pr = part + (long)blockIdx.y * M * N;
#pragma unroll
        for (int i = 0; i < WM; i++) {
            const int m = (tile_m * AF + wy * WM + i) * 32 + rl;
#pragma unroll
            for (int j = 0; j < WN; j++)
#pragma unroll
                for (int rq = 0; rq < 4; rq++) {
                    const int n = (tile_n * BF + wx * WN + j) * 32 + 8 * rq + 4 * hh;
                    if (m < M && n < N) *reinterpret_cast<float4_t*>(pr + (long)m * N + n) = float4_t{acc[i][j][4 * rq], acc[i][j][4 * rq + 1], acc[i][j][4 * rq + 2], acc[i][j][4 * rq + 3]};
                }
        }
        return;
    }
    // C/D layout: column = lane & 31 = row m of x, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) = output feature inside the 32-block.
    // y = dt(dt(acc) + bias) as mpq_gemm.hip; the half-waves trade packed quads (v_permlane32_swap_b32) so that a lane holds 8
    // consecutive features of its row: 16 bytes.
    const bool vec_ok = (N & 7) == 0 && (ldy & 7) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;  // ldy: the row pitch of y in elements (N, or a wider destination's)
    auto pack2 = [&](float lo, float hi) -> uint32_t {
        if constexpr (DT == BIE_BF16) return pack_bf16x2(lo, hi);
        else return f32_to_f16_bits(lo) | (f32_to_f16_bits(hi) << 16);
    };
    // the eight consecutive features (column block j, quad pair qp, half hh) of row rl as one 16-byte word
    auto pack8 = [&](const float (&v)[8]) -> uint4_t {
        const uint32_t p0 = pack2(v[0], v[1]), p1 = pack2(v[2], v[3]), p2 = pack2(v[4], v[5]), p3 = pack2(v[6], v[7]);
        const auto s0 = __builtin_amdgcn_permlane32_swap(p0, p2, false, false);  // first operand's upper half <-> second's lower half
        const auto s1 = __builtin_amdgcn_permlane32_swap(p1, p3, false, false);
        return uint4_t{s0[0], s1[0], s0[1], s1[1]};
    };
#if BIE_DENSE_LAB == 1 || BIE_DENSE_LAB == 8 || BIE_DENSE_LAB == 9
    if (M == -12345)  // never: the accumulators stay live, nothing is stored
#endif
#if BIE_DENSE_LAB != 3
    if (vec_ok) {
        // Row-contiguous stores through the wave's own quarter of the (now idle) stage buffers.  A lane owns features of ONE row, so a
        // direct store instruction touches 64 rows x 16 bytes = 64 partial lines (the round-4 epilogue: store-issue-bound, ~7 B/clk/CU,
        // MI355X_MICROARCH.md "epilogue store tail"; ~10 us of the 4096^3 launch).  Here a block of 32 rows x (32 * WN) features goes to
        // LDS as [row][16-byte chunk ^ (row & 15)] (ds_write_b128, conflict-free: the 8 lanes of a store group hit 8 different chunks)
        // and comes back with 16 consecutive lanes on one row: a store instruction writes 64 / (4 * WN) rows x (64 * WN) contiguous bytes
        // -- whole 128-byte lines.  No barrier: the region is private to the wave (its own lgkmcnt orders write -> read).
        static_assert(WN == 4 || WN == 2, "row pitch of the staging block: 16 or 8 chunks");
        constexpr int CH = 4 * WN;        // 16-byte chunks per staged row (32 * WN features)
        constexpr int RPI = 64 / CH;      // rows one read / store instruction covers
        constexpr int BLK = 32 * CH * 16; // bytes of one staged block (32 rows)
        __builtin_amdgcn_s_barrier();     // every wave is out of the loop: nobody reads stage data any more
        const uint32_t stg = lds_base + (uint32_t)wave * (2 * BLK);  // two blocks per wave: block i + 1 is written while block i drains
        const int rrow = le / CH, rch = le % CH;
        auto staged = [&](auto has_bias) {
            constexpr bool HB = decltype(has_bias)::value;
#pragma unroll
            for (int i = 0; i < WM; i++) {
                const uint32_t blk = stg + (uint32_t)(i & 1) * BLK;
#pragma unroll
                for (int j = 0; j < WN; j++) {
#pragma unroll
                    for (int qp = 0; qp < 2; qp++) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            v[e] = acc[i][j][8 * qp + e];
                            if constexpr (HB) {  // dt(dt(acc) + bias), as mpq_gemm.hip
                                const int n = (tile_n * BF + wx * WN + j) * 32 + 8 * (2 * qp + (e >> 2)) + 4 * hh + (e & 3);
                                v[e] = dt_traits<DT>::round(v[e]) + dt_traits<DT>::load(bias, n < N ? n : N - 1);
                            }
                        }
                        const uint4_t w = pack8(v);
                        const int ch = 4 * j + 2 * qp + hh;
                        const uint32_t a = blk + (uint32_t)(rl * (CH * 16) + ((ch ^ (rl & (CH - 1))) << 4));
                        asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(w) : "memory");
                        __builtin_amdgcn_sched_barrier(0);  // one 8-value group at a time: the 256 accumulators leave the AGPRs as they are packed
                    }
                }
                uint4_t w[32 / RPI];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int t = 0; t < 32 / RPI; t++) {
                    const int r = t * RPI + rrow;
                    const uint32_t a = blk + (uint32_t)(r * (CH * 16) + ((rch ^ (r & (CH - 1))) << 4));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(w[t]) : "v"(a) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);  // the reads return by the wait, not by data dependence: nothing that uses w[] may move up
#pragma unroll
                for (int t = 0; t < 32 / RPI; t++) {
                    const int r = t * RPI + rrow;
                    const int m = (tile_m * AF + wy * WM + i) * 32 + r;
                    const int n = (tile_n * BF + wx * WN) * 32 + 8 * rch;
                    asm volatile("" : "+v"(w[t]));
#if BIE_DENSE_LAB == 7   // lab: streaming (non-temporal) stores -- y drains to HBM while the epilogue issues instead of at the kernel-end write-back
                    if (m < M && n < N) __builtin_nontemporal_store(w[t], reinterpret_cast<uint4_t*>(y + (long)m * ldy + n));
#else
                    if (m < M && n < N) *reinterpret_cast<uint4_t*>(y + (long)m * ldy + n) = w[t];
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (bias == nullptr) staged(std::false_type{});
        else staged(std::true_type{});
        return;
    }
#endif
    // direct stores (odd N / pitch / alignment; BIE_DENSE_LAB == 3: the round-4 epilogue)
    auto store8 = [&](int i, int j, int qp, const float (&v)[8]) {
        const int m = (tile_m * AF + wy * WM + i) * 32 + rl;
        const int nb = (tile_n * BF + wx * WN + j) * 32;
        uint16_t* yr = y + (long)(m < M ? m : 0) * ldy;
        if (vec_ok) {
            const uint4_t w = pack8(v);
            const int n = nb + 8 * (2 * qp + hh);
            if (m < M && n < N) *reinterpret_cast<uint4_t*>(yr + n) = w;
        } else if (m < M) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int n = nb + 8 * (2 * qp + (e >> 2)) + 4 * hh + (e & 3);
                if (n < N) dt_traits<DT>::store(yr, n, v[e]);
            }
        }
    };
#if BIE_DENSE_LAB == 1 || BIE_DENSE_LAB == 8 || BIE_DENSE_LAB == 9
    if (M == -12345)
#endif
    if (bias == nullptr) {
#pragma unroll
        for (int i = 0; i < WM; i++)
#pragma unroll
            for (int j = 0; j < WN; j++)
#pragma unroll
                for (int qp = 0; qp < 2; qp++) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] = acc[i][j][8 * qp + e];
                    store8(i, j, qp, v);
                }
    } else {
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
            for (int qp = 0; qp < 2; qp++) {
                float bv[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int n = (tile_n * BF + wx * WN + j) * 32 + 8 * (2 * qp + (e >> 2)) + 4 * hh + (e & 3);
                    bv[e] = dt_traits<DT>::load(bias, n < N ? n : N - 1);
                }
#pragma unroll
                for (int i = 0; i < WM; i++) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] = dt_traits<DT>::round(acc[i][j][8 * qp + e]) + bv[e];  // dt(dt(acc) + bias), as mpq_gemm.hip
                    store8(i, j, qp, v);
                }
            }
    }
}

// ---- launch plumbing ---------------------------------------------------------------------------------------------------
static int env_int_dense(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// When: measured against the fused kernel on one box (profiles/r03_dense_ab.txt, r03_dense_k_slope.txt), both forms run the same
// 0.93 us per 32 MFMAs per wave with every CU busy (0.62 is the bare bf16 MFMA stream at that load's clock: moving a stage's 32 KiB
// into LDS and 64 KiB out of it costs the same 0.3 us whether the weights arrive dequantised or are dequantised beside the MFMAs).
// What differs is per launch: this form pays a dequantise pass (K*N*(w/8 + 2) bytes) but only ~16 us of prologue + epilogue per
// round of tiles and no K split, the fused one ~30 us.  So it wins when K is short and the tile grid is whole rounds of the 256 CUs
// (4096x4096: 57 vs 61 us at M = 1024, 96 vs 102 at 2048, 124 vs 137 at 4096, 279 vs 301 at 8192) and loses otherwise (4096x11008
// M = 1024: 150 vs 120; K = 8192 .. 16384: equal or slower) -- the rule below.  BIE_GEMM_DENSE=0 switches it off, =2 forces it (tests).
// Depends on (M, K, N) and the process environment only: bie_mpq_workspace_bytes has to reproduce the choice.
#include "mpq_dense_table.inc"

// nearest grid point in log space (the geometric mean of two neighbours is the boundary); -1 when x is more than 20 % outside the grid
static int dense_grid_index(const int* g, int n, long x) {
    if ((double)x * 1.2 < (double)g[0] || (double)x > (double)g[n - 1] * 1.2) return -1;
    int i = 0;
    while (i + 1 < n && (double)x * (double)x > (double)g[i] * (double)g[i + 1]) i++;
    return i;
}

bool mpq_dense_ok(int M, int K, int N) {
    static const bool tuning = getenv("BIE_TUNING") != nullptr;
    static const int on_once = env_int_dense("BIE_GEMM_DENSE", 1), min_once = env_int_dense("BIE_GEMM_DENSE_MIN_M", 897);
    const int on = tuning ? env_int_dense("BIE_GEMM_DENSE", 1) : on_once, min_m = tuning ? env_int_dense("BIE_GEMM_DENSE_MIN_M", 897) : min_once;
    if (!on || (K & 31) || (N & 7)) return false;
    if (on == 2) return true;  // forced (tests: every shape the kernels can take)
    // profiles/r03_dense_ab4_gm4.txt (dense / fused time, 7 layer shapes x M = 512 .. 8192, tiles walked gm = 4 rows deep per XCD run): with
    // 256 x 256 tiles (>= 192 of them) the dense form is 0.89-0.98 of the fused time up to K = 5120 at every M >= 1024, and beyond that K from
    // M = 4096 on (0.89-1.0; at M = 2048 1.04-1.08: the dequantise pass is K*N work that only M amortises); with 128 x 128 tiles it wins
    // only on whole rounds of short-K square layers (0.82-0.93) and loses 1.05-1.6x elsewhere.
    // Round 6: inside the measured grid (70 layer shapes x 6 row counts, fused against dense on one box: profiles/r06_dense_rule_sweep.txt) the answer is the
    // measurement at the nearest grid point -- the analytic rule below left 1.9 % on the table on average (77 of 420 cells more than 4 % off, up to 28 %: it
    // knew Llama-7B's shapes only); BIE_GEMM_DENSE_TABLE=0 switches the table off.  Outside the grid (M > 4915, K or N beyond 20 % of it) the rule answers.
    static const int table_once = env_int_dense("BIE_GEMM_DENSE_TABLE", 1);
    if ((tuning ? env_int_dense("BIE_GEMM_DENSE_TABLE", 1) : table_once) && M >= min_m) {
        const int ki = dense_grid_index(kDenseK, 8, K), ni = dense_grid_index(kDenseN, 9, N), mi = dense_grid_index(kDenseM, 6, M);
        if (ki >= 0 && ni >= 0 && mi >= 0) return (kDenseTable[ki][ni] >> mi) & 1;
    }
    const long t256 = (long)cdiv(M, 256) * cdiv(N, 256);
    if (M < min_m) return false;  // 897 = the first row count with the tile grid of 1024 rows (eight 128-row / four 256-row tiles): 4096x4096 at 1023 rows 49.0 against 58.2 us fused (profiles/r06_dense_mid_m.txt); below, the fused kernel wins every cell
    // beyond K = 5120 from M = 4096 while the image (K*N*2 bytes) stays inside the 256 MiB MALL, from M = 8192 when it does not (8192 -> 28672, 470 MB:
    // 1.007 at M = 4096 in the A/B, 1.02 in the bench's two-layer rotation, 0.954 at M = 8192)
    if (t256 >= 192) return K <= 5120 || M >= ((long)K * N * 2 > (192l << 20) ? 8192 : 4096);
    const long g128 = (long)cdiv(M, 128) * cdiv(N, 128);
    return K <= 4096 && g128 >= 256 && g128 % 256 == 0;
}

size_t mpq_dense_workspace_bytes(int K, int N) { return (size_t)cdiv(N, 32) * 32 * K * 2; }

template <int DT, int ZM>
static void dequant_frag_launch(const int32_t* qw, const void* scales, const void* zeros, void* img, int K, int N, int w_bit, int gshift, hipStream_t st) {
    const int KS = K / 16;
    const long nfrag = (long)cdiv(N, 32) * KS;
    // fragments per wave: W4 takes 2 (four times the waves of the 8-fragment form -- two residency rounds, the store stream of the first under the
    // loads of the second: 0.5-0.8 % of the whole M = 4096 call on every shape, profiles/r06_dq_fpw_ab.txt); BIE_DQ_FPW = 8 | 4 | 1 for the A/B
    static const int fpw_knob = env_int_dense("BIE_DQ_FPW", 2);
    int fpw = (KS & 7) == 0 ? 8 : 1;
    if (fpw == 8 && w_bit == 4 && (fpw_knob == 1 || fpw_knob == 2 || fpw_knob == 4)) fpw = fpw_knob;
    const dim3 grid((unsigned)cdivl(nfrag / fpw, 4));
#define BIE_DQ2(WB, F) hipLaunchKernelGGL((mpq_dequant_frag_kernel<DT, WB, ZM, F>), grid, dim3(256), 0, st, (const uint32_t*)qw, (const uint16_t*)scales, zeros, (uint4_t*)img, N, gshift, nfrag, KS)
#define BIE_DQ(WB) do { if (fpw == 8) BIE_DQ2(WB, 8); else if (WB == 4 && fpw == 4) BIE_DQ2(4, 4); else if (WB == 4 && fpw == 2) BIE_DQ2(4, 2); else BIE_DQ2(WB, 1); } while (0)
    switch (w_bit) {
        case 1: BIE_DQ(1); break;
        case 2: BIE_DQ(2); break;
        case 4: BIE_DQ(4); break;
        default: BIE_DQ(8); break;
    }
#undef BIE_DQ2
#undef BIE_DQ
}

// K splits of the 128 x 128-tile grid when it is far from filling the chip: enough workgroups for two per CU, at least eight stages each
int mpq_dense_splits(int M, int K, int N) {
    const long t256 = (long)cdiv(M, 256) * cdiv(N, 256);
    if (t256 >= 192) return 1;
    const long g128 = (long)cdiv(M, 128) * cdiv(N, 128);
    if (g128 >= 256) return 1;
    long S = (512 + g128 - 1) / g128;
    const long max_s = (K >> 5) / 8 > 0 ? (K >> 5) / 8 : 1;
    if (S > max_s) S = max_s;
    if (S > 32) S = 32;
    return S < 1 ? 1 : (int)S;
}
size_t mpq_dense_part_bytes(int M, int K, int N) {
    const int S = mpq_dense_splits(M, K, N);
    return S > 1 ? (size_t)S * M * N * sizeof(float) : 0;
}

template <int DT>
static void dense_gemm_launch(const void* x, const void* img, const void* bias, void* y, int M, int K, int N, int ldy, hipStream_t st, float* part = nullptr) {
    const int NB32 = cdiv(N, 32);
    if (part != nullptr) {  // K split over gridDim.y (mpq_dense_splits), fp32 partial sums; the caller runs launch_splitk_finalize
        const int S = mpq_dense_splits(M, K, N), KTs = cdiv(K >> 5, S), tn = cdiv(N, 128);
        hipLaunchKernelGGL((mpq_dense_gemm_kernel<DT, 2, 2>), dim3((unsigned)(cdiv(M, 128) * tn), (unsigned)cdiv(K >> 5, KTs)), dim3(256), 0, st, (const uint16_t*)x, (const uint8_t*)img,
                           (const uint16_t*)nullptr, (uint16_t*)y, M, N, K, tn, NB32, 1, ldy, part, KTs);
        return;
    }
    const long t256 = (long)cdiv(M, 256) * cdiv(N, 256);
    static const int tile_once = env_int_dense("BIE_GEMM_DENSE_TILE", 0), gm_once = env_int_dense("BIE_GEMM_DENSE_GM", 4);
    const bool tuning = getenv("BIE_TUNING") != nullptr;
    const int tile = tuning ? env_int_dense("BIE_GEMM_DENSE_TILE", 0) : tile_once;
    int gm = tuning ? env_int_dense("BIE_GEMM_DENSE_GM", 4) : gm_once;  // tile rows an XCD's run walks down before moving one tile column on
    if (gm < 1) gm = 1;
    if (tile == 256 || (tile != 128 && t256 >= 192)) {
        const int tn = cdiv(N, 256);
        hipLaunchKernelGGL((mpq_dense_gemm_kernel<DT, 4, 4>), dim3((unsigned)t256), dim3(256), 0, st, (const uint16_t*)x, (const uint8_t*)img, (const uint16_t*)bias,
                           (uint16_t*)y, M, N, K, tn, NB32, gm, ldy, (float*)nullptr, 0);
    } else {
        const int tn = cdiv(N, 128);
        hipLaunchKernelGGL((mpq_dense_gemm_kernel<DT, 2, 2>), dim3((unsigned)(cdiv(M, 128) * tn)), dim3(256), 0, st, (const uint16_t*)x, (const uint8_t*)img,
                           (const uint16_t*)bias, (uint16_t*)y, M, N, K, tn, NB32, gm, ldy, (float*)nullptr, 0);
    }
}

// the GEMM alone on an image somebody else wrote (mbwq.hip: the mixed-bit layout's own dequantise pass)
bool mpq_dense_shape_ok(int K, int N) { return (K & 31) == 0 && (N & 7) == 0; }
// part: mpq_dense_part_bytes(M, K, N) bytes of fp32 scratch, or nullptr when that is 0 (or for a pitched y: one split, whatever the grid)
int mpq_dense_gemm_only_launch(const void* x, const void* img, const void* bias, void* y, int M, int K, int N, int dtype, hipStream_t st, int ldy, float* part) {
    if (part != nullptr && (ldy != N || mpq_dense_splits(M, K, N) <= 1)) part = nullptr;
    if (dtype == BIE_F16) dense_gemm_launch<BIE_F16>(x, img, bias, y, M, K, N, ldy, st, part);
    else dense_gemm_launch<BIE_BF16>(x, img, bias, y, M, K, N, ldy, st, part);
    int rc = check_launch("mpq_dense_gemm_kernel");
    if (rc || part == nullptr) return rc;
    const int S = mpq_dense_splits(M, K, N), KTs = cdiv(K >> 5, S);
    return launch_splitk_finalize(part, bias, y, cdiv(K >> 5, KTs), M, N, dtype, st);
}

// explicit, irregular g_idx: per-k groups (mpq_dequant_frag_gidx_kernel), then the same GEMM
int mpq_dense_gidx_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const int32_t* g_idx, const void* bias, void* y, void* scratch,
                          int M, int K, int N, int w_bit, int asym, int dtype, hipStream_t st) {
    const int KS = K / 16;
    const long nfrag = (long)cdiv(N, 32) * KS;
    const dim3 grid((unsigned)cdivl(nfrag, 4));
#define BIE_DQG3(DTV, WB, AS) hipLaunchKernelGGL((mpq_dequant_frag_gidx_kernel<DTV, WB, AS>), grid, dim3(256), 0, st, (const uint32_t*)qw, (const uint16_t*)scales, zeros, g_idx, (uint4_t*)scratch, N, nfrag, KS)
#define BIE_DQG2(DTV, WB) do { if (asym) BIE_DQG3(DTV, WB, true); else BIE_DQG3(DTV, WB, false); } while (0)
#define BIE_DQG(DTV) do { switch (w_bit) { case 1: BIE_DQG2(DTV, 1); break; case 2: BIE_DQG2(DTV, 2); break; case 4: BIE_DQG2(DTV, 4); break; default: BIE_DQG2(DTV, 8); break; } } while (0)
    if (dtype == BIE_F16) BIE_DQG(BIE_F16); else BIE_DQG(BIE_BF16);
#undef BIE_DQG
#undef BIE_DQG2
#undef BIE_DQG3
    int rc = check_launch("mpq_dequant_frag_gidx_kernel");
    if (rc) return rc;
    return mpq_dense_gemm_only_launch(x, scratch, bias, y, M, K, N, dtype, st, N, nullptr);
}

// scratch: mpq_dense_workspace_bytes(K, N) bytes, 16-byte aligned
int mpq_dense_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const void* bias, void* y, void* scratch, int M, int K, int N,
                     int w_bit, int gshift, int zm, int dtype, hipStream_t st, int ldy) {
    if (dtype == BIE_F16) {
        if (zm == ZM_ASYM) dequant_frag_launch<BIE_F16, ZM_ASYM>(qw, scales, zeros, scratch, K, N, w_bit, gshift, st);
        else if (zm == ZM_FUSED) dequant_frag_launch<BIE_F16, ZM_FUSED>(qw, scales, zeros, scratch, K, N, w_bit, gshift, st);
        else dequant_frag_launch<BIE_F16, ZM_SYM>(qw, scales, zeros, scratch, K, N, w_bit, gshift, st);
    } else {
        if (zm == ZM_ASYM) dequant_frag_launch<BIE_BF16, ZM_ASYM>(qw, scales, zeros, scratch, K, N, w_bit, gshift, st);
        else dequant_frag_launch<BIE_BF16, ZM_SYM>(qw, scales, zeros, scratch, K, N, w_bit, gshift, st);
    }
    int rc = check_launch("mpq_dequant_frag_kernel");
    if (rc) return rc;
    if (dtype == BIE_F16) dense_gemm_launch<BIE_F16>(x, scratch, bias, y, M, K, N, ldy, st);
    else dense_gemm_launch<BIE_BF16>(x, scratch, bias, y, M, K, N, ldy, st);
    return check_launch("mpq_dense_gemm_kernel");
}

}  // namespace bie
