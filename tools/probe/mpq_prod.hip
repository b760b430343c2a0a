// MEASURED AND REJECTED (round 5; profiles/r05_prod_ab_*.txt, profiles/DEAD_ENDS.md): bit-identical to the dense form (same association order)
// and 3-30 % SLOWER on every (shape, M) cell at bf16 and fp16 -- 4096^3: 137.9 against 128.7 us on one box.  Not built into libbie_hip.so;
// to try it again: copy to bitorch-engine_amd/csrc/, add the dispatch in mpq_gemm_launch_ld (mpq_prod_ok / mpq_prod_launch below) and
// `-fno-slp-vectorize` for the file in the Makefile.
//
// W{1,2,4,8}A16 GEMM for large M, "producer" form: ONE launch, no dequantised image in HBM.
// Replaces the same branch of the reference as mpq_gemm.hip / mpq_dense.hip -- "materialise the fp16 weight, then cuBLAS"
// (layers/qlinear/nbit/cuda/mpq_layer.py:59-63, unpack_qweight utils.py:30-51) -- with the same arithmetic: weight values from
// dequant8 (mpq_frag_dequant.cuh: the reference's two roundings), fp32 accumulation, one rounding at the store.
//
// Where it sits between the two older forms:
//   * mpq_gemm.hip (fused): every wave dequantises the weight operand of its own 256 x 64 wave tile in registers -- 8 chunks of eight
//     weights per lane per 32 k = 5 VALU per MFMA at the reference's bf16 roundings, more than the 5 issue slots an MFMA gap hides;
//   * mpq_dense.hip: dequantise ONCE per call into a K*N*2-byte fragment image (a second launch, 42 MB of traffic at 4096^2: ~10 us of
//     the 118), then a dense GEMM that stages both operands through LDS;
//   * here the workgroup tile is 512 rows x 128 columns, four waves STACKED in M (each 128 x 128, the dense kernel's wave tile).  The
//     weight operand of a stage (128 columns x 32 k = 8 fragments of 1 KiB) is the same for all four waves, so each wave dequantises
//     TWO of them -- two packed dwords per lane, 2.5 VALU per MFMA -- and writes them to the stage buffer (ds_write_b128); everybody reads
//     them back as in the dense kernel.  A weight is dequantised once per 512 rows of x instead of once per 256, by one wave instead of
//     by each.  x: every wave moves only its own 128 rows (LDS-DMA, the dense kernel's source-side swizzle).
// Loop = mpq_dense_gemm_kernel's: K = 32 per stage, 3 stage buffers (40 KiB each), one barrier per stage between its two MFMA clusters,
// fragment reads behind a cluster's first MFMAs, refills behind the rest, MFMAs as ordered inline asm.  The packed words of stage L are
// requested four stages ahead (top of stage L - 4), covered by the counted vmcnt of stage L - 3's barrier, dequantised in pieces of <= 5
// VALU, one per MFMA gap -- fragment (wave, 0) in cluster 2 of stage L - 3, fragment (wave, 1) in cluster 1 of stage L - 2 (buffer L % 3
// is free from stage L - 3's barrier on) -- and read from stage L - 1's cluster 2 on.
#include "mpq_frag_dequant.cuh"
#include "mfma_pipe.cuh"
#include <stdlib.h>
#include <type_traits>

#pragma clang fp contract(off)

namespace bie {

template <int DT>
__device__ __forceinline__ void prod_mfma16(float16_t& c, const v4i_t& a, const v4i_t& b) {
    if constexpr (DT == BIE_F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// the packed words and group parameters of ONE stage for this lane: fragments (column block = wave, k16 step 0 / 1)
struct ProdWords {
    uint2_t raw[2];
    uint32_t sb, zb;
};

// ---- the dequantisation of one fragment word, cut into pieces of <= 5 VALU: one piece per MFMA gap -------------------------------------
// A 32-cycle MFMA gap hides about five single-issue instructions (MI355X_MICROARCH.md); dequant8 in one lump (40 VALU at bf16 sym) stalls
// the matrix pipe for a whole MFMA.  For the W4 symmetric forms (the bench's and GreenBit's) the same arithmetic -- value for value the
// sequence of dequant8 / mpq_dequant.cuh, the reference's two roundings -- is issued as PROD_PIECES steps on a small state; every
// other (dtype, width, zero mode) runs dequant8 whole in step 0.
constexpr int PROD_PIECES = 12;
template <int DT, int WBIT, int ZM>
struct ProdDq {
    static constexpr bool FINE = WBIT == 4 && ZM == ZM_SYM;
    float q[8];
    uint32_t p01, p23, o[4];
    float s, z;
    half2_t s2, z2;

    template <int P>
    __device__ __forceinline__ void step(uint2_t raw, int c8, uint32_t sb, uint32_t zb, uint32_t lds_addr) {
        if constexpr (!FINE) {
            if constexpr (P == 0) {
                const uint4_t f = dequant8<DT, WBIT, ZM>(raw, c8, make_col_params<DT, WBIT, ZM, (WBIT == 4)>(sb, zb));
                asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "v"(f) : "memory");
            }
        } else if constexpr (DT == BIE_BF16) {
            if constexpr (P == 0) {  // nibbles -> bytes in natural k order
                const uint32_t lo = raw.x & 0x0f0f0f0fu, hi = (raw.x >> 4) & 0x0f0f0f0fu;
                p01 = __builtin_amdgcn_perm(hi, lo, 0x05010400u);
                p23 = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
            } else if constexpr (P == 1) {  // group constants; the fields are read as fp8 (q * 2^-9): 2^9 goes into s
                s = bf16_bits_to_f32(sb) * 512.0f;
                z = bf16_bits_to_f32(zb);
            } else if constexpr (P == 2) {
                const float2_t f0 = __builtin_amdgcn_cvt_pk_f32_fp8(p01, false), f1 = __builtin_amdgcn_cvt_pk_f32_fp8(p01, true);
                const float2_t f2 = __builtin_amdgcn_cvt_pk_f32_fp8(p23, false), f3 = __builtin_amdgcn_cvt_pk_f32_fp8(p23, true);
                q[0] = f0.x; q[1] = f0.y; q[2] = f1.x; q[3] = f1.y; q[4] = f2.x; q[5] = f2.y; q[6] = f3.x; q[7] = f3.y;
            } else if constexpr (P >= 3 && P <= 10) {
                constexpr int i = (P - 3) >> 1;
                if constexpr (((P - 3) & 1) == 0) {
                    o[i] = pack_bf16x2(q[2 * i] * s, q[2 * i + 1] * s);  // fl16(q * s)
                } else {
                    const float ta = __uint_as_float(o[i] << 16), tb = __uint_as_float(o[i] & 0xffff0000u);
                    o[i] = pack_bf16x2(ta - z, tb - z);               // fl16(. - z)
                }
            } else {
                const uint4_t f = uint4_t{o[0], o[1], o[2], o[3]};
                asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "v"(f) : "memory");
            }
        } else {  // fp16 W4 sym: (1024 + q) pairs in natural k order, exact subtraction, fl16(q * s), fl16(. - z): packed fp16 ALU
            if constexpr (P == 0) {
                const uint32_t lo = raw.x & 0x0f0f0f0fu, hi = (raw.x >> 4) & 0x0f0f0f0fu;
                p01 = __builtin_amdgcn_perm(hi, lo, 0x05010400u);
                p23 = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
            } else if constexpr (P == 1) {
                const half_t sh = __builtin_bit_cast(half_t, (uint16_t)sb), zh = __builtin_bit_cast(half_t, (uint16_t)zb);
                s2 = half2_t{sh, sh};
                z2 = half2_t{zh, zh};
            } else if constexpr (P == 2) {
                o[0] = __builtin_amdgcn_perm(0x64646464u, p01, 0x04010400u);
                o[1] = __builtin_amdgcn_perm(0x64646464u, p01, 0x04030402u);
                o[2] = __builtin_amdgcn_perm(0x64646464u, p23, 0x04010400u);
                o[3] = __builtin_amdgcn_perm(0x64646464u, p23, 0x04030402u);
            } else if constexpr (P >= 3 && P <= 6) {
                constexpr int i = P - 3;
                const half2_t k1024 = half2_t{(half_t)1024.0f, (half_t)1024.0f};
                half2_t r = (__builtin_bit_cast(half2_t, o[i]) - k1024) * s2;
                r = r - z2;
                o[i] = __builtin_bit_cast(uint32_t, r);
            } else if constexpr (P == 7) {
                const uint4_t f = uint4_t{o[0], o[1], o[2], o[3]};
                asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr), "v"(f) : "memory");
            }
        }
    }
};

template <int DT, int WBIT, int ZM>
__global__ __launch_bounds__(256) void mpq_prod_gemm_kernel(const uint16_t* __restrict__ x, const uint32_t* __restrict__ qw, const uint16_t* __restrict__ scales,
                                                            const void* __restrict__ zeros, const uint16_t* __restrict__ bias, uint16_t* __restrict__ y, int M, int N,
                                                            int K, int tiles_n, int gm, int ldy, int gshift) {
    constexpr int WM = 4, WN = 4;            // 32-row blocks per wave, 32-column blocks per workgroup (= per wave: the waves share the columns)
    constexpr int AF = 16, BF = 4;           // 32-row / 32-column blocks per workgroup tile: 512 x 128
    constexpr int PW = 8;                    // LDS-DMA pieces per wave and stage: its own 128 rows x 64 bytes
    constexpr int ASZ = AF * 2048;           // x part of a stage
    constexpr int STAGE = ASZ + BF * 2048;   // + 8 weight fragments
    constexpr int NR = WM + WN, NM = WM * WN;
    constexpr int RPM = (2 * NR + NM - 1) / NM, M0 = (NR + RPM - 1) / RPM, DPM = (PW + (NM - M0) - 1) / (NM - M0);
    constexpr int LW = (WBIT == 8 ? 4 : 2) + 2;  // vector-memory operations of load_words: the packed loads (a dword per chunk, two at 8 bit), scale, zero
    constexpr int VMS = PW + LW;                 // ... and of a stage: + its 8 LDS-DMA pieces
    __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * STAGE];

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile_m, tile_n;
    pipe_tile(blockIdx.x, gridDim.x, tiles_n, gm, tile_m, tile_n);
    const int KT = K >> 5;
    // A ragged last tile is SHIFTED to end at M (at N): it recomputes rows (columns) of its neighbour and stores the same values again,
    // and nothing in the loop needs a clamp -- every address is tile base (scalar) + a lane part that never changes.  M >= 512, N >= 128.
    const int m0 = __builtin_amdgcn_readfirstlane(tile_m * 512 + 512 <= M ? tile_m * 512 : M - 512);
    const int n0 = __builtin_amdgcn_readfirstlane(tile_n * 128 + 128 <= N ? tile_n * 128 : N - 128);

    // Buffer descriptors (wave-uniform through readfirstlane, or every load is wrapped in a waterfall loop): scalar offset = row / group /
    // stage part, vector offset = the lane's part.  The host admits only tensors below 2 GB (mpq_prod_ok).
    const auto rsrc_of = [](const void* p, uint32_t bytes) {
        const uint64_t b = (uint64_t)(uintptr_t)p;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    constexpr int NBW = 32 / WBIT;
    const int G = (int)(((long)(K - 1) >> gshift) + 1);
    [[maybe_unused]] const auto rx = rsrc_of(x, (uint32_t)((long)M * K * 2));
    const auto rq = rsrc_of(qw, (uint32_t)((long)(K / NBW) * N * 4));
    const auto rs = rsrc_of(scales, (uint32_t)((long)G * N * 2));
    const auto rz = rsrc_of(zeros, (uint32_t)(ZM == ZM_ASYM ? (long)G * (N / NBW) * 4 : (long)G * N * 2));

    // ---- x: LDS-DMA pieces of 16 rows x 64 bytes; piece j of this wave = rows m0 + wave*128 + 16 j + (lane >> 2), LDS slot lane & 3 holds
    // logical slot (lane & 3) ^ ((row >> 2) & 3) = (lane & 3) ^ ((lane >> 4) & 3) for every piece (16 j and 128 wave are multiples of 16)
    [[maybe_unused]] const uint32_t xvoff = (uint32_t)(lane >> 2) * (uint32_t)K * 2u + (uint32_t)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    [[maybe_unused]] const uint32_t xrow0 = (uint32_t)(m0 + wave * 128) * (uint32_t)K * 2u;  // scalar
    [[maybe_unused]] const uint32_t xpiece = 16u * (uint32_t)K * 2u;
    [[maybe_unused]] const int kt_last = KT - 1;
    auto issue_piece = [&](int kt, int j) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int ks = kt < kt_last ? kt : kt_last;  // a look-ahead past the end re-fetches the last tile into a buffer nobody reads again
        auto* dst = (__attribute__((address_space(3))) unsigned char*)lds + (kt % 3) * STAGE + wave * (PW * 1024);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, dst + j * 1024, 16, xvoff, xrow0 + (uint32_t)j * xpiece + (uint32_t)ks * 64u, 0, 0);
#endif
    };

    // ---- weights: column of this lane, packed words of a stage (scalar offsets: packed row / group row)
    const int ncol = n0 + wave * 32 + (lane & 31);
    const int hh0 = lane >> 5;
    const uint32_t qvoff = (uint32_t)ncol * 4u + (uint32_t)hh0 * (uint32_t)N * 4u * (WBIT == 8 ? 2u : (WBIT == 4 ? 1u : 0u));  // k-octet hh0 of a k16 step: its packed row
    const uint32_t svoff = (uint32_t)ncol * 2u;
    const uint32_t zvoff = ZM == ZM_ASYM ? (uint32_t)(ncol / NBW) * 4u : (uint32_t)ncol * 2u;
    const uint32_t qrow_bytes = (uint32_t)N * 4u;
    auto load_words = [&](int L, ProdWords& w) {
        const int Lc = L < kt_last ? L : kt_last;
        const uint32_t g = (uint32_t)((long)(Lc * 32) >> gshift);
#pragma unroll
        for (int H = 0; H < 2; H++) {
            const uint32_t c8u = (uint32_t)(Lc * 4 + 2 * H);  // chunk of lane half 0; half 1 = the next chunk (in qvoff where that is another packed row)
            if constexpr (WBIT == 8) {
                w.raw[H] = uint2_t{__builtin_amdgcn_raw_buffer_load_b32(rq, qvoff, (2 * c8u) * qrow_bytes, 0),
                                   __builtin_amdgcn_raw_buffer_load_b32(rq, qvoff, (2 * c8u + 1) * qrow_bytes, 0)};
            } else {
                constexpr uint32_t CPW = 4 / WBIT;  // chunks per word: w4 1, w2 2, w1 4 (both lane halves then read the same word)
                w.raw[H] = uint2_t{__builtin_amdgcn_raw_buffer_load_b32(rq, qvoff, (c8u / CPW) * qrow_bytes, 0), 0u};
            }
        }
        w.sb = __builtin_amdgcn_raw_buffer_load_b16(rs, svoff, g * (uint32_t)N * 2u, 0);
        if constexpr (ZM == ZM_ASYM) w.zb = __builtin_amdgcn_raw_buffer_load_b32(rz, zvoff, g * (uint32_t)(N / NBW) * 4u, 0);
        else w.zb = __builtin_amdgcn_raw_buffer_load_b16(rz, zvoff, g * (uint32_t)N * 2u, 0);
    };
    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
    const uint32_t bw_addr = lds_base + ASZ + (uint32_t)wave * 2048 + (uint32_t)lane * 16;  // fragment (wave, H) of a stage: + H * 1024
    // piece P of fragment (wave, H) of stage L (written to buffer L % 3); the ASYM zero code is taken out of its packed word in piece 0
    auto dq_step = [&](auto pc, int L, auto hc, const ProdWords& w, ProdDq<DT, WBIT, ZM>& st) {
        constexpr int P = decltype(pc)::value, H = decltype(hc)::value;
        const int Lc = L < kt_last ? L : kt_last;
        uint32_t zb = w.zb;
        if constexpr (ZM == ZM_ASYM) {
            constexpr uint32_t M1 = (1u << WBIT) - 1u;
            zb = ((zb >> ((ncol % NBW) * WBIT)) & M1) + 1u;
        }
        st.template step<P>(w.raw[H], Lc * 4 + 2 * H + hh0, w.sb, zb, bw_addr + (uint32_t)(L % 3) * STAGE + H * 1024);
    };
    auto produce = [&](int L, auto hc, const ProdWords& w) {  // the whole fragment at once (prologue)
        ProdDq<DT, WBIT, ZM> st;
        static_for<0, PROD_PIECES>([&](auto pc) { dq_step(pc, L, hc, w, st); });
    };

    uint32_t a_addr[2];  // x fragment of k16 step s: row rl of the wave's first block, logical slot 2*s + hh
    {
        const int rl = lane & 31, hh = lane >> 5, sw = (rl >> 2) & 3;
        a_addr[0] = lds_base + (uint32_t)((wave * 128 + rl) * 64 + ((hh ^ sw) << 4));
        a_addr[1] = lds_base + (uint32_t)((wave * 128 + rl) * 64 + (((2 + hh) ^ sw) << 4));
    }
    const uint32_t b_addr = lds_base + ASZ + lane * 16;

    float16_t acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    v4i_t XA[WM], XB[WN], YA[WM], YB[WN], ZA[WM], ZB[WN];
    auto read_item = [&](auto ic, auto hc, uint32_t so, v4i_t (&TA)[WM], v4i_t (&TB)[WN]) {
        constexpr int R = decltype(ic)::value, H = decltype(hc)::value;
        if constexpr (R < WM) TA[R] = lds_read16<R * 2048>(a_addr[H] + so);
        else TB[R - WM] = lds_read16<(R - WM) * 2048 + H * 1024>(b_addr + so);
    };

    // ---- prologue.  Vector-memory order (the loop's counted waits rely on it): words 0..2 | x stage 0 | x stage 1, words 3 | x stage 2, words 4
    ProdWords P0, P1;       // stages 0 and 1: produced right here
    ProdWords W0, W1, W2;   // the loop's ring: stage L lives in slot L % 3 (W2 = stage 2, whose SECOND fragment the first cluster of stage 0
                            // produces, like every later stage's; W0 = stage 3; W1 receives stage 4 at the top of stage 0)
    load_words(0, P0);
    load_words(1, P1);
    load_words(2, W2);
#pragma unroll
    for (int j = 0; j < PW; j++) issue_piece(0, j);
#pragma unroll
    for (int j = 0; j < PW; j++) issue_piece(1, j);
    load_words(3, W0);
#pragma unroll
    for (int j = 0; j < PW; j++) issue_piece(2, j);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PW + LW) : "memory");  // words 0..2 are here (behind them: x stages 0, 1, words 3, x stage 2)
    __builtin_amdgcn_sched_barrier(0);
    produce(0, ic_t<0>{}, P0); produce(0, ic_t<1>{}, P0);
    produce(1, ic_t<0>{}, P1); produce(1, ic_t<1>{}, P1);
    produce(2, ic_t<0>{}, W2);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PW + LW) : "memory");  // x stage 0 landed; the fragments of stages 0, 1 and (2, 0) are written
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NR>([&](auto rc) { read_item(rc, ic_t<0>{}, 0u, XA, XB); });

    // stage kt: PA/PB = its k16 step 0 fragments (already read), QA/QB receive step 1, NA/NB step 0 of stage kt + 1.
    // WC = the words of stage kt + 3 (slot kt % 3), consumed here; WL = the slot the words of stage kt + 5 go to ((kt + 2) % 3)
    // Fragment (wave, 0) of stage kt + 3 is dequantised in this stage's second cluster, fragment (wave, 1) in the NEXT stage's first
    // cluster (WP = the words the previous stage consumed; its buffer (kt + 2) % 3 has been free since the previous stage's barrier):
    // one piece per MFMA gap, fenced so that the compiler keeps it there.
    ProdDq<DT, WBIT, ZM> dq0, dq1;
    auto stage = [&](int kt, v4i_t (&PA)[WM], v4i_t (&PB)[WN], v4i_t (&QA)[WM], v4i_t (&QB)[WN], v4i_t (&NA)[WM], v4i_t (&NB)[WN], ProdWords& WC, ProdWords& WL, ProdWords& WP) {
        const uint32_t so = (uint32_t)(kt % 3) * STAGE, sn = (uint32_t)((kt + 1) % 3) * STAGE;
        wait_frags<0>(PA, PB);
        static_for<0, NM>([&](auto mc) {
            constexpr int m = decltype(mc)::value, i = m / WN, j = m % WN;
            prod_mfma16<DT>(acc[i][j], PB[j], PA[i]);  // D = w_frag (rows = output features) x x_frag (columns = rows of x)
            static_for<imin(m * RPM, NR), imin((m + 1) * RPM, NR)>([&](auto rc) { read_item(rc, ic_t<1>{}, so, QA, QB); });
            if constexpr (m == 0) { __builtin_amdgcn_sched_barrier(0); load_words(kt + 4, WL); __builtin_amdgcn_sched_barrier(0); }
            if constexpr (m >= 3 && m < 3 + PROD_PIECES) {
                __builtin_amdgcn_sched_barrier(0);
                dq_step(ic_t<m - 3>{}, kt + 2, ic_t<1>{}, WP, dq1);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        wait_frags<0>(QA, QB);  // every LDS read (and fragment write) of this wave has completed: buffer kt % 3 may be refilled behind the barrier
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMS) : "memory");  // everything issued up to stage kt - 2 is here: x of stage kt + 1, the words of stage kt + 3
        __builtin_amdgcn_s_barrier();
        static_for<0, NM>([&](auto mc) {
            constexpr int m = decltype(mc)::value, i = m / WN, j = m % WN;
            prod_mfma16<DT>(acc[i][j], QB[j], QA[i]);
            static_for<imin(m * RPM, NR), imin((m + 1) * RPM, NR)>([&](auto rc) { read_item(rc, ic_t<0>{}, sn, NA, NB); });
            if constexpr (m >= M0)
                static_for<imin((m - M0) * DPM, PW), imin((m - M0 + 1) * DPM, PW)>([&](auto pc) { issue_piece(kt + 3, decltype(pc)::value); });
            if constexpr (m < PROD_PIECES) {
                __builtin_amdgcn_sched_barrier(0);
                dq_step(ic_t<m>{}, kt + 3, ic_t<0>{}, WC, dq0);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    int kt = 0;
    for (; kt + 3 <= KT; kt += 3) {  // slots: consumed here (stage kt + 3) | loaded at the top (stage kt + 4) | the previous stage's (kt + 2: its second fragment)
        stage(kt, XA, XB, YA, YB, ZA, ZB, W0, W1, W2);
        stage(kt + 1, ZA, ZB, XA, XB, YA, YB, W1, W2, W0);
        stage(kt + 2, YA, YB, ZA, ZB, XA, XB, W2, W0, W1);
    }
    if (kt < KT) {
        stage(kt, XA, XB, YA, YB, ZA, ZB, W0, W1, W2);
        if (kt + 1 < KT) stage(kt + 1, ZA, ZB, XA, XB, YA, YB, W1, W2, W0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the clamped look-ahead pieces / reads / writes must not outlive the workgroup's LDS
    mfma_drain();

    // ---- epilogue: mpq_dense_gemm_kernel's (whole-line stores through the wave's own staging blocks); wy = wave, wx = 0
    int le = threadIdx.x & 63;
    asm volatile("" : "+v"(le));  // lane-derived epilogue values are computed here, not carried through the loop
    const int rl = le & 31, hh = le >> 5;
    auto pack2 = [&](float lo, float hi) -> uint32_t {
        if constexpr (DT == BIE_BF16) return pack_bf16x2(lo, hi);
        else return f32_to_f16_bits(lo) | (f32_to_f16_bits(hi) << 16);
    };
    auto pack8 = [&](const float (&v)[8]) -> uint4_t {
        const uint32_t p0 = pack2(v[0], v[1]), p1 = pack2(v[2], v[3]), p2 = pack2(v[4], v[5]), p3 = pack2(v[6], v[7]);
        const auto s0 = __builtin_amdgcn_permlane32_swap(p0, p2, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(p1, p3, false, false);
        return uint4_t{s0[0], s1[0], s0[1], s1[1]};
    };
    {   // the host admits only N % 8 == 0, ldy % 8 == 0 and a 16-byte aligned y (mpq_prod_ok / mpq_prod_launch): 16-byte stores throughout
        constexpr int CH = 4 * WN, RPI = 64 / CH, BLK = 32 * CH * 16;
        __builtin_amdgcn_s_barrier();  // every wave is out of the loop: nobody reads stage data any more
        const uint32_t stg = lds_base + (uint32_t)wave * (2 * BLK);
        const int rrow = le / CH, rch = le % CH;
        auto staged = [&](auto has_bias) {
            constexpr bool HB = decltype(has_bias)::value;
            // bias: added on the way BACK from the staging block, where a lane holds eight consecutive features of a row -- the same eight
            // for every row it stores: one 16-byte load.  y = dt(dt(acc) + bias) as mpq_gemm.hip (the block holds dt(acc)).
            uint4_t b8 = uint4_t{0u, 0u, 0u, 0u};
            if constexpr (HB) b8 = *reinterpret_cast<const uint4_t*>(bias + n0 + 8 * rch);
            auto unpack_lo = [&](uint32_t p) -> float { if constexpr (DT == BIE_BF16) return bf16_bits_to_f32(p & 0xffffu); else return f16_bits_to_f32(p & 0xffffu); };
            auto unpack_hi = [&](uint32_t p) -> float { if constexpr (DT == BIE_BF16) return bf16_bits_to_f32(p >> 16); else return f16_bits_to_f32(p >> 16); };
            auto add2 = [&](uint32_t a, uint32_t b) -> uint32_t { return pack2(unpack_lo(a) + unpack_lo(b), unpack_hi(a) + unpack_hi(b)); };
#pragma unroll
            for (int i = 0; i < WM; i++) {
                const uint32_t blk = stg + (uint32_t)(i & 1) * BLK;
#pragma unroll
                for (int j = 0; j < WN; j++)
#pragma unroll
                    for (int qp = 0; qp < 2; qp++) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; e++) v[e] = acc[i][j][8 * qp + e];
                        const uint4_t w = pack8(v);
                        const int ch = 4 * j + 2 * qp + hh;
                        const uint32_t a = blk + (uint32_t)(rl * (CH * 16) + ((ch ^ (rl & (CH - 1))) << 4));
                        asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(w) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
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
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 32 / RPI; t++) {
                    const int r = t * RPI + rrow;
                    const int m = m0 + (wave * WM + i) * 32 + r;
                    const int n = n0 + 8 * rch;
                    asm volatile("" : "+v"(w[t]));
                    if constexpr (HB) w[t] = uint4_t{add2(w[t].x, b8.x), add2(w[t].y, b8.y), add2(w[t].z, b8.z), add2(w[t].w, b8.w)};
                    *reinterpret_cast<uint4_t*>(y + (long)m * ldy + n) = w[t];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (bias == nullptr) staged(std::false_type{});
        else staged(std::true_type{});
    }
}

// ---- launch plumbing ---------------------------------------------------------------------------------------------------
static int env_int_prod(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// BIE_GEMM_PROD: 0 off, 1 where measured faster than the dense form (default), 2 forced wherever the kernel can run (tests)
bool mpq_prod_ok(int M, int K, int N, int group_size) {
    static const bool tuning = getenv("BIE_TUNING") != nullptr;
    static const int on_once = env_int_prod("BIE_GEMM_PROD", 1), min_once = env_int_prod("BIE_GEMM_PROD_MIN_M", 2048);
    const int on = tuning ? env_int_prod("BIE_GEMM_PROD", 1) : on_once, min_m = tuning ? env_int_prod("BIE_GEMM_PROD_MIN_M", 2048) : min_once;
    if (!on || (K & 31) || K < 160 || group_size < 32 || (group_size & (group_size - 1))) return false;  // a stage (32 k) lies in one group; five stages of look-ahead
    if (M < 512 || N < 128 || (N & 7) || (long)M * K * 2 >= (1l << 31) || (long)K * N >= (1l << 31)) return false;   // shifted ragged tiles; 16-byte stores; 32-bit buffer offsets
    if (on == 2) return true;
    return M >= min_m && (long)cdiv(M, 512) * cdiv(N, 128) >= 192;
}

template <int DT, int ZM>
static void prod_launch_w(const void* x, const int32_t* qw, const void* scales, const void* zeros, const void* bias, void* y, int M, int K, int N, int w_bit, int gshift,
                          int ldy, hipStream_t st) {
    const int tn = cdiv(N, 128), tm = cdiv(M, 512);
    static const int gm_once = env_int_prod("BIE_GEMM_PROD_GM", 2);
    int gm = getenv("BIE_TUNING") ? env_int_prod("BIE_GEMM_PROD_GM", 2) : gm_once;  // tile rows an XCD's run walks down before moving one tile column on
    if (gm < 1) gm = 1;
#define BIE_PL(WB) hipLaunchKernelGGL((mpq_prod_gemm_kernel<DT, WB, ZM>), dim3((unsigned)(tm * tn)), dim3(256), 0, st, (const uint16_t*)x, (const uint32_t*)qw, \
                                      (const uint16_t*)scales, zeros, (const uint16_t*)bias, (uint16_t*)y, M, N, K, tn, gm, ldy, gshift)
    switch (w_bit) {
        case 1: BIE_PL(1); break;
        case 2: BIE_PL(2); break;
        case 4: BIE_PL(4); break;
        default: BIE_PL(8); break;
    }
#undef BIE_PL
}

bool mpq_prod_dest_ok(const void* y, int ldy, const void* bias) {
    return (ldy & 7) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0;  // 16-byte stores, one 16-byte bias load per lane
}

int mpq_prod_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const void* bias, void* y, int M, int K, int N, int w_bit, int gshift, int zm,
                    int dtype, hipStream_t st, int ldy) {
    if (dtype == BIE_F16) {
        if (zm == ZM_ASYM) prod_launch_w<BIE_F16, ZM_ASYM>(x, qw, scales, zeros, bias, y, M, K, N, w_bit, gshift, ldy, st);
        else if (zm == ZM_FUSED) prod_launch_w<BIE_F16, ZM_FUSED>(x, qw, scales, zeros, bias, y, M, K, N, w_bit, gshift, ldy, st);
        else prod_launch_w<BIE_F16, ZM_SYM>(x, qw, scales, zeros, bias, y, M, K, N, w_bit, gshift, ldy, st);
    } else {
        if (zm == ZM_ASYM) prod_launch_w<BIE_BF16, ZM_ASYM>(x, qw, scales, zeros, bias, y, M, K, N, w_bit, gshift, ldy, st);
        else prod_launch_w<BIE_BF16, ZM_SYM>(x, qw, scales, zeros, bias, y, M, K, N, w_bit, gshift, ldy, st);
    }
    return check_launch("mpq_prod_gemm_kernel");
}

}  // namespace bie
