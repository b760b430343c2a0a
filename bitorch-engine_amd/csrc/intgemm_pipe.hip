// W8A8 integer GEMM on the ordered-asm MFMA pipeline of mfma_pipe.cuh (v_mfma_i32_32x32x32_i8) -- the large-grid form of
// intgemm.hip's int_gemm_kernel modes 1 (fp32 out, two scales) and 3 (raw int32), replacing the same reference code:
// q8_linear_cutlass_kernel.cu:44-230 (CUTLASS int8 tensor-op GEMM + the scale epilogue of q8_linear_cutlass_forward).
//   y[m][n] = epilogue( sum_k a[m][k] * w[n][k] )      a: [M, K] int8, w: [N, K] int8, both K-contiguous
// Both operands are row-major with k contiguous, i.e. both are what x is to mpq_dense.hip: LDS image [rows][64 bytes] per stage
// (K = 64 = two MFMA k steps), 16-byte slots XORed by (row >> 2) & 3 on the SOURCE side of the LDS-DMA, conflict-free fragment reads.
// Loop, tiles and schedule are binary_fp4.hip's / mpq_dense.hip's (the bytes per stage and MFMAs per stage are the same by construction).
// Exact int32 accumulation; the epilogues are int_gemm_kernel's, so the results are bit-identical to it.
#include "mfma_pipe.cuh"
#include <stdlib.h>

namespace bie {

typedef int int16v_t __attribute__((ext_vector_type(16)));
typedef int int4v_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma_i8(int16v_t& c, const v4i_t& a, const v4i_t& b) {
    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// OUT_I32: y int32 = acc; else y fp32 = ((float)acc * scale_a) * scale_w
template <bool OUT_I32, int WM, int WN>
__global__ __launch_bounds__(256) void i8_pipe_gemm_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ W, void* __restrict__ yv, int M, int N, int K,
                                                           int tiles_n, float scale_a, float scale_w) {
    constexpr int AF = 2 * WM, BF = 2 * WN;
    constexpr int NFR = (AF + BF) * 2;  // KiB per stage (64 bytes per row)
    constexpr int PW = NFR / 4;
    constexpr int STAGE = NFR * 1024;
    constexpr int NR = WM + WN, NM = WM * WN;
    constexpr int RPM = (2 * NR + NM - 1) / NM, M0 = (NR + RPM - 1) / RPM, DPM = (PW + (NM - M0) - 1) / (NM - M0);
    static_assert((AF * 2) % PW == 0, "a wave's pieces belong to one operand");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * STAGE];

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wy = wave >> 1, wx = wave & 1;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    int tile_m, tile_n;
    pipe_tile(bid, nblk, tiles_n, BIE_PIPE_GM, tile_m, tile_n);  // one contiguous run of tiles per XCD, BIE_PIPE_GM tile rows deep
    const int KT = K >> 6;

    // LDS-DMA sources: pieces of 16 rows x 64 bytes; [a rows of the tile][w rows of the tile]
    const bool a_wave = wave * PW < AF * 2;
    const uint8_t* src[PW];
#pragma unroll
    for (int j = 0; j < PW; j++) {
        const int p = wave * PW + j;
        const int rt = (a_wave ? p : p - AF * 2) * 16 + (lane >> 2);  // row of the operand's tile
        long r = a_wave ? (long)tile_m * (AF * 32) + rt : (long)tile_n * (BF * 32) + rt;
        const long rmax = (a_wave ? M : N) - 1;
        if (r > rmax) r = rmax;
        src[j] = (a_wave ? A : W) + r * K + (((lane & 3) ^ ((rt >> 2) & 3)) << 4);
    }
    [[maybe_unused]] const int kt_last = KT - 1;
    auto issue_piece = [&](int kt, int j) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int ks = kt < kt_last ? kt : kt_last;
        auto* dst = (__attribute__((address_space(3))) unsigned char*)lds + (kt % 3) * STAGE + wave * (PW * 1024);
        __builtin_amdgcn_global_load_lds(src[j] + (long)ks * 64, dst + j * 1024, 16, 0, 0);
#endif
    };

    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
    const int rl = lane & 31, hh = lane >> 5, sw = (rl >> 2) & 3;
    uint32_t a_addr[2], b_addr[2];  // k step s: logical slot 2*s + hh of row rl of the wave's first block
#pragma unroll
    for (int s = 0; s < 2; s++) {
        a_addr[s] = lds_base + (uint32_t)((wy * WM * 32 + rl) * 64 + (((2 * s + hh) ^ sw) << 4));
        b_addr[s] = lds_base + AF * 2048 + (uint32_t)((wx * WN * 32 + rl) * 64 + (((2 * s + hh) ^ sw) << 4));
    }

    int16v_t acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0;

    v4i_t XA[WM], XB[WN], YA[WM], YB[WN], ZA[WM], ZB[WN];
    auto read_item = [&](auto ic, auto hc, uint32_t so, v4i_t (&TA)[WM], v4i_t (&TB)[WN]) {
        constexpr int R = decltype(ic)::value, H = decltype(hc)::value;
        if constexpr (R < WM) TA[R] = lds_read16<R * 2048>(a_addr[H] + so);
        else TB[R - WM] = lds_read16<(R - WM) * 2048>(b_addr[H] + so);
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
            mfma_i8(acc[i][j], PB[j], PA[i]);  // D = w_frag (rows = output features) x a_frag (columns = rows of a)
            static_for<imin(m * RPM, NR), imin((m + 1) * RPM, NR)>([&](auto rc) { read_item(rc, ic_t<1>{}, so, QA, QB); });
        });
        wait_frags<0>(QA, QB);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
        __builtin_amdgcn_s_barrier();
        static_for<0, NM>([&](auto mc) {
            constexpr int m = decltype(mc)::value, i = m / WN, j = m % WN;
            mfma_i8(acc[i][j], QB[j], QA[i]);
            static_for<imin(m * RPM, NR), imin((m + 1) * RPM, NR)>([&](auto rc) { read_item(rc, ic_t<0>{}, sn, NA, NB); });
            if constexpr (m >= M0)
                static_for<imin((m - M0) * DPM, PW), imin((m - M0 + 1) * DPM, PW)>([&](auto pc) { issue_piece(kt + 3, decltype(pc)::value); });
        });
    };
    int kt = 0;
    for (; kt + 3 <= KT; kt += 3) {
        stage(kt, XA, XB, YA, YB, ZA, ZB);
        stage(kt + 1, ZA, ZB, XA, XB, YA, YB);
        stage(kt + 2, YA, YB, ZA, ZB, XA, XB);
    }
    if (kt < KT) {
        stage(kt, XA, XB, YA, YB, ZA, ZB);
        if (kt + 1 < KT) stage(kt + 1, ZA, ZB, XA, XB, YA, YB);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    mfma_drain();

    // C/D: column = lane & 31 = row m of a, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) = feature inside the 32-block: four consecutive n per
    // register group -> 16-byte stores (N % 4 == 0 on this path)
    const int n_l = 4 * hh;
#pragma unroll
    for (int i = 0; i < WM; i++) {
        const int m = (tile_m * AF + wy * WM + i) * 32 + rl;
        if (m < M) {
#pragma unroll
            for (int j = 0; j < WN; j++) {
                const int n0 = (tile_n * BF + wx * WN + j) * 32 + n_l;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int n = n0 + 8 * q;
                    if (n < N) {
                        if constexpr (OUT_I32) {
                            *reinterpret_cast<int4v_t*>((int*)yv + (long)m * N + n) = int4v_t{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        } else {
                            float4_t v;
#pragma unroll
                            for (int e = 0; e < 4; e++) v[e] = ((float)acc[i][j][4 * q + e] * scale_a) * scale_w;
                            *reinterpret_cast<float4_t*>((float*)yv + (long)m * N + n) = v;
                        }
                    }
                }
            }
        }
    }
}

// ---- W4A4: packed nibbles, two values per byte, expanded to i8 in registers beside the MFMAs ---------------------------------------
// A byte holds (v_even << 4 | v_odd), values in [-8, 7]; masked with 0xF0 it IS the signed i8 value 16 * v (intgemm.hip), so both operands
// carry a factor 16 and the exact int32 sum is shifted right by 8 in the epilogue.  A stage is 64 bytes = 128 values per row: a lane's
// 16-byte fragment read yields TWO MFMA operands (its first and second 8 bytes, 6 VALU each), i.e. four 16-MFMA clusters per stage on
// half the LDS read traffic per MFMA of the int8 form.  Which 16 values of k a lane contributes to which MFMA is the same function on
// both operands, which is all the contraction needs.  Replaces q4_linear_cutlass_kernel.cu:203-280,520-560 (CUTLASS int4b_t GEMM).
__device__ __forceinline__ v4i_t expand_q4_lo(const v4i_t& r) {
    return v4i_t{(int)((uint32_t)r.x & 0xF0F0F0F0u), (int)(((uint32_t)r.x << 4) & 0xF0F0F0F0u), (int)((uint32_t)r.y & 0xF0F0F0F0u), (int)(((uint32_t)r.y << 4) & 0xF0F0F0F0u)};
}
__device__ __forceinline__ v4i_t expand_q4_hi(const v4i_t& r) {
    return v4i_t{(int)((uint32_t)r.z & 0xF0F0F0F0u), (int)(((uint32_t)r.z << 4) & 0xF0F0F0F0u), (int)((uint32_t)r.w & 0xF0F0F0F0u), (int)(((uint32_t)r.w << 4) & 0xF0F0F0F0u)};
}

// MODE 0: y (DT) = fl(fl((float)acc) * fl(scale_a * scale_w)); MODE 2: y int32 = acc  (int_gemm_kernel's modes 0 / 2)
template <int MODE, int DT, int WM, int WN>
__global__ __launch_bounds__(256) void i4_pipe_gemm_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ W, void* __restrict__ yv, int M, int N, int K,
                                                           int tiles_n, float scale_a, float scale_w) {
    constexpr int AF = 2 * WM, BF = 2 * WN;
    constexpr int NFR = (AF + BF) * 2;
    constexpr int PW = NFR / 4;
    constexpr int STAGE = NFR * 1024;
    constexpr int NR = WM + WN, NM = WM * WN;
    constexpr int RPM = (2 * NR + NM - 1) / NM, M0 = (NR + RPM - 1) / RPM, DPM = (PW + (NM - M0) - 1) / (NM - M0), EPM = (NR + (NM - M0) - 1) / (NM - M0);
    __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * STAGE];

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wy = wave >> 1, wx = wave & 1;
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    int tile_m, tile_n;
    pipe_tile(bid, nblk, tiles_n, BIE_PIPE_GM, tile_m, tile_n);
    const int KT = K >> 7;             // 128 values = 64 bytes per stage
    const long rowbytes = K >> 1;

    const bool a_wave = wave * PW < AF * 2;
    const uint8_t* src[PW];
#pragma unroll
    for (int j = 0; j < PW; j++) {
        const int p = wave * PW + j;
        const int rt = (a_wave ? p : p - AF * 2) * 16 + (lane >> 2);
        long r = a_wave ? (long)tile_m * (AF * 32) + rt : (long)tile_n * (BF * 32) + rt;
        const long rmax = (a_wave ? M : N) - 1;
        if (r > rmax) r = rmax;
        src[j] = (a_wave ? A : W) + r * rowbytes + (((lane & 3) ^ ((rt >> 2) & 3)) << 4);
    }
    [[maybe_unused]] const int kt_last = KT - 1;
    auto issue_piece = [&](int kt, int j) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int ks = kt < kt_last ? kt : kt_last;
        auto* dst = (__attribute__((address_space(3))) unsigned char*)lds + (kt % 3) * STAGE + wave * (PW * 1024);
        __builtin_amdgcn_global_load_lds(src[j] + (long)ks * 64, dst + j * 1024, 16, 0, 0);
#endif
    };

    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
    const int rl = lane & 31, hh = lane >> 5, sw = (rl >> 2) & 3;
    uint32_t a_addr[2], b_addr[2];  // read phase H: the 16-byte slot 2*H + hh of row rl
#pragma unroll
    for (int s = 0; s < 2; s++) {
        a_addr[s] = lds_base + (uint32_t)((wy * WM * 32 + rl) * 64 + (((2 * s + hh) ^ sw) << 4));
        b_addr[s] = lds_base + AF * 2048 + (uint32_t)((wx * WN * 32 + rl) * 64 + (((2 * s + hh) ^ sw) << 4));
    }

    int16v_t acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0;

    v4i_t XA[WM], XB[WN], YA[WM], YB[WN], ZA[WM], ZB[WN];  // raw fragments (three rotating sets, as the other pipeline kernels)
    v4i_t EA[2][WM], EB[2][WN];                            // expanded operands: [0] first 8 bytes of a fragment, [1] second
    auto read_item = [&](auto ic, auto hc, uint32_t so, v4i_t (&TA)[WM], v4i_t (&TB)[WN]) {
        constexpr int R = decltype(ic)::value, H = decltype(hc)::value;
        if constexpr (R < WM) TA[R] = lds_read16<R * 2048>(a_addr[H] + so);
        else TB[R - WM] = lds_read16<(R - WM) * 2048>(b_addr[H] + so);
    };
    // expansion item R of half E (0: first 8 bytes, 1: second) from the raw set (TA, TB)
    auto expand_item = [&](auto ic, auto ec, const v4i_t (&TA)[WM], const v4i_t (&TB)[WN]) {
        constexpr int R = decltype(ic)::value, E = decltype(ec)::value;
        // The zero-instruction asm pins the six VALU of an expansion to THIS slot of the stream.  Left to itself hipcc sinks them to just
        // in front of the MFMA that consumes the operand -- and a VALU result read by the very next MFMA is a hazard the compiler pads
        // for its own MFMAs but cannot see into an asm one: the MFMA took the register's previous content ((x << 4) before its mask),
        // results off by a few units for K >= 384 (tools/dbg_q4.py).  Pinned here, every expansion sits >= 3 MFMAs ahead of its use.
        if constexpr (R < WM) {
            EA[E][R] = E ? expand_q4_hi(TA[R]) : expand_q4_lo(TA[R]);
            asm volatile("" : "+v"(EA[E][R]));
        } else {
            EB[E][R - WM] = E ? expand_q4_hi(TB[R - WM]) : expand_q4_lo(TB[R - WM]);
            asm volatile("" : "+v"(EB[E][R - WM]));
        }
    };

#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int j = 0; j < PW; j++) issue_piece(s, j);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NR>([&](auto rc) { read_item(rc, ic_t<0>{}, 0u, XA, XB); });
    wait_frags<0>(XA, XB);
    static_for<0, NR>([&](auto rc) { expand_item(rc, ic_t<0>{}, XA, XB); });

    // P: this stage's first read phase (landed, its first halves expanded into E[0]); Q: second phase; NX: next stage's first phase
    auto stage = [&](int kt, v4i_t (&PA)[WM], v4i_t (&PB)[WN], v4i_t (&QA)[WM], v4i_t (&QB)[WN], v4i_t (&NA)[WM], v4i_t (&NB)[WN]) {
        const uint32_t so = (uint32_t)(kt % 3) * STAGE, sn = (uint32_t)((kt + 1) % 3) * STAGE;
        // cluster 0: E[0] = lo(P); reads of Q behind the first MFMAs, hi(P) -> E[1] behind the rest
        static_for<0, NM>([&](auto mc) {
            constexpr int m = decltype(mc)::value, i = m / WN, j = m % WN;
            mfma_i8(acc[i][j], EB[0][j], EA[0][i]);
            static_for<imin(m * RPM, NR), imin((m + 1) * RPM, NR)>([&](auto rc) { read_item(rc, ic_t<1>{}, so, QA, QB); });
            if constexpr (m >= M0) static_for<imin((m - M0) * EPM, NR), imin((m - M0 + 1) * EPM, NR)>([&](auto rc) { expand_item(rc, ic_t<1>{}, PA, PB); });
        });
        // cluster 1: E[1] = hi(P); lo(Q) -> E[0]
        wait_frags<0>(QA, QB);  // every LDS read of this stage has returned
        static_for<0, NM>([&](auto mc) {
            constexpr int m = decltype(mc)::value, i = m / WN, j = m % WN;
            mfma_i8(acc[i][j], EB[1][j], EA[1][i]);
            static_for<imin(m * RPM, NR), imin((m + 1) * RPM, NR)>([&](auto rc) { expand_item(rc, ic_t<0>{}, QA, QB); });
        });
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");  // K tile kt+1 landed (kt+2 still in flight)
        __builtin_amdgcn_s_barrier();
        // cluster 2: E[0] = lo(Q); reads of the next stage's first phase + hi(Q) -> E[1] behind the first MFMAs, refill pieces behind the rest
        static_for<0, NM>([&](auto mc) {
            constexpr int m = decltype(mc)::value, i = m / WN, j = m % WN;
            mfma_i8(acc[i][j], EB[0][j], EA[0][i]);
            static_for<imin(m * RPM, NR), imin((m + 1) * RPM, NR)>([&](auto rc) {
                read_item(rc, ic_t<0>{}, sn, NA, NB);
                expand_item(rc, ic_t<1>{}, QA, QB);
            });
            if constexpr (m >= M0)
                static_for<imin((m - M0) * DPM, PW), imin((m - M0 + 1) * DPM, PW)>([&](auto pc) { issue_piece(kt + 3, decltype(pc)::value); });
        });
        // cluster 3: E[1] = hi(Q); lo(NX) -> E[0] for the next stage
        wait_frags<0>(NA, NB);
        static_for<0, NM>([&](auto mc) {
            constexpr int m = decltype(mc)::value, i = m / WN, j = m % WN;
            mfma_i8(acc[i][j], EB[1][j], EA[1][i]);
            static_for<imin(m * RPM, NR), imin((m + 1) * RPM, NR)>([&](auto rc) { expand_item(rc, ic_t<0>{}, NA, NB); });
        });
    };
    int kt = 0;
    for (; kt + 3 <= KT; kt += 3) {
        stage(kt, XA, XB, YA, YB, ZA, ZB);
        stage(kt + 1, ZA, ZB, XA, XB, YA, YB);
        stage(kt + 2, YA, YB, ZA, ZB, XA, XB);
    }
    if (kt < KT) {
        stage(kt, XA, XB, YA, YB, ZA, ZB);
        if (kt + 1 < KT) stage(kt + 1, ZA, ZB, XA, XB, YA, YB);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    mfma_drain();

    const int n_l = 4 * hh;
    const float sc = dt_traits<(MODE == 0 ? DT : BIE_F32)>::round(scale_a * scale_w);
#pragma unroll
    for (int i = 0; i < WM; i++) {
        const int m = (tile_m * AF + wy * WM + i) * 32 + rl;
        if (m < M) {
#pragma unroll
            for (int j = 0; j < WN; j++) {
                const int n0 = (tile_n * BF + wx * WN + j) * 32 + n_l;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int n = n0 + 8 * q;
                    if (n < N) {
                        int v[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) v[e] = acc[i][j][4 * q + e] >> 8;  // both operands carried a factor 16
                        if constexpr (MODE == 2) {
                            *reinterpret_cast<int4v_t*>((int*)yv + (long)m * N + n) = int4v_t{v[0], v[1], v[2], v[3]};
                        } else {
                            float o[4];
#pragma unroll
                            for (int e = 0; e < 4; e++) o[e] = dt_traits<DT>::round(dt_traits<DT>::round((float)v[e]) * sc);
                            if constexpr (DT == BIE_F32) *reinterpret_cast<float4_t*>((float*)yv + (long)m * N + n) = float4_t{o[0], o[1], o[2], o[3]};
                            else if constexpr (DT == BIE_F16)
                                *reinterpret_cast<uint2_t*>((uint16_t*)yv + (long)m * N + n) = uint2_t{f32_to_f16_bits(o[0]) | (f32_to_f16_bits(o[1]) << 16), f32_to_f16_bits(o[2]) | (f32_to_f16_bits(o[3]) << 16)};
                            else *reinterpret_cast<uint2_t*>((uint16_t*)yv + (long)m * N + n) = uint2_t{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                        }
                    }
                }
            }
        }
    }
}

// Shapes this form takes (K % 64 == 0 and N % 4 == 0 are the entry points' own preconditions): grids that give every wave work.
bool i8_pipe_ok(int M, int N, int K, const void* A, const void* W, const void* y) {
    static const int on = [] { const char* e = getenv("BIE_I8_PIPE"); return e ? atoi(e) : 1; }();
    if (!on || (K & 63) || (N & 3) || M < 128 || N < 128) return false;
    return ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
}

int i8_pipe_launch(bool out_i32, const void* A, const void* W, void* y, int M, int N, int K, float sa, float sw, hipStream_t st) {
    const long t256 = (long)cdiv(M, 256) * cdiv(N, 256);
#define BIE_I8(OI, WMV, GRID, TN) \
    hipLaunchKernelGGL((i8_pipe_gemm_kernel<OI, WMV, WMV>), dim3((unsigned)(GRID)), dim3(256), 0, st, (const uint8_t*)A, (const uint8_t*)W, y, M, N, K, TN, sa, sw)
    if (t256 >= 192) {
        const int tn = cdiv(N, 256);
        if (out_i32) BIE_I8(true, 4, t256, tn);
        else BIE_I8(false, 4, t256, tn);
    } else {
        const int tn = cdiv(N, 128);
        const long g = (long)cdiv(M, 128) * tn;
        if (out_i32) BIE_I8(true, 2, g, tn);
        else BIE_I8(false, 2, g, tn);
    }
#undef BIE_I8
    return check_launch("i8_pipe_gemm_kernel");
}

bool i4_pipe_ok(int M, int N, int K, const void* A, const void* W, const void* y) { return (K & 127) == 0 && i8_pipe_ok(M, N, K, A, W, y); }

// mode 0: typed output (dtype), mode 2: raw int32
int i4_pipe_launch(int mode, const void* A, const void* W, void* y, int M, int N, int K, float sa, float sw, int dtype, hipStream_t st) {
    const long t256 = (long)cdiv(M, 256) * cdiv(N, 256);
    const bool big = t256 >= 192;
    const int tn = cdiv(N, big ? 256 : 128);
    const dim3 grid((unsigned)(big ? t256 : (long)cdiv(M, 128) * tn));
#define BIE_I4(MODEV, DTV) \
    do { \
        if (big) hipLaunchKernelGGL((i4_pipe_gemm_kernel<MODEV, DTV, 4, 4>), grid, dim3(256), 0, st, (const uint8_t*)A, (const uint8_t*)W, y, M, N, K, tn, sa, sw); \
        else hipLaunchKernelGGL((i4_pipe_gemm_kernel<MODEV, DTV, 2, 2>), grid, dim3(256), 0, st, (const uint8_t*)A, (const uint8_t*)W, y, M, N, K, tn, sa, sw); \
    } while (0)
    if (mode == 2) BIE_I4(2, BIE_F32);
    else if (dtype == BIE_F16) BIE_I4(0, BIE_F16);
    else if (dtype == BIE_BF16) BIE_I4(0, BIE_BF16);
    else BIE_I4(0, BIE_F32);
#undef BIE_I4
    return check_launch("i4_pipe_gemm_kernel");
}

}  // namespace bie
