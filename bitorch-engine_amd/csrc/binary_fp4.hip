// 1-bit W / 1-bit A GEMM on the MATRIX pipe of gfx950: the +-1 operands as FP4 (E2M1) fragments of
// v_mfma_scale_f32_32x32x64_f8f6f4.  CDNA4 has no 1-bit MFMA; binary.hip therefore contracts on the VALU (v_xor + v_bcnt:
// 1.26 POP/s issue peak, xnor_gemm128_kernel sits at 0.8 of it).  +1 = 0x2 and -1 = 0xA are exact E2M1 values, their products
// are +-1, the fp32 accumulator holds K - 2*popcount(x ^ w) exactly (|sum| <= K < 2^24), so the matrix pipe (~10 PFLOP/s dense
// for FP4, four times the bf16 rate) computes the SAME integers as the XNOR kernels -- bit-exact, zero-padding contributes 0.0.
// Replaces, for large M, the same reference functions as binary.hip's GEMMs: binary_linear_cuda_kernel.cu (BMMAS_new :155-181,
// BMM32_Arow_Brow_UD :308-393), binary_linear_cutlass_kernel.cu (:293-332, CUTLASS uint1b_t XOR-popcount GEMM),
// binary_linear.cpp (_xnor_gemm_unrolled :249-295).
//
// Operand images ("fragment order"): a [rows, K] sign matrix is stored as 1 KiB fragments, fragment (rb, kb) = rows 32*rb ..
// +31 x k 64*kb .. +63, lane l of the wave that will feed the MFMA owns bytes 16*l .. +15 = the 32 nibbles of row
// 32*rb + (l & 31), k 64*kb + 32*(l >> 5) .. +31.  Fragments of one row block are contiguous along k; K is padded to 128
// and rows to 32 with 0.0 nibbles.  A wave-level global_load_lds_dwordx4 moves one fragment HBM/L2 -> LDS as 1 KiB contiguous on
// both sides, the LDS image needs no swizzle (ds_read_b128 at 16*lane is conflict-free), and because A and B use the same
// (register, nibble) -> k mapping the contraction is right whatever order the pipe walks k in.
//
// GEMM kernel: 4 waves as 2 x 2, wave tile 32*WM x 32*WN (WM = WN = 4: workgroup tile 256 x 256, 256 accumulator registers, one
// wave per SIMD), K step 128 per LDS stage, 3 stages (loads run two stages ahead), ONE barrier per stage placed between the two
// k64 halves of a stage so that the first fragment reads of the next stage are issued under the second half's 16 MFMAs
// (three rotating fragment register sets).  LDS traffic: 8 ds_read_b128 per 16 MFMAs (512 cycles) per wave = 64 B/clk/CU.
#include "bie_common.h"
#include <stdlib.h>

namespace bie {

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef int v4i_t __attribute__((ext_vector_type(4)));

static inline long fp4_row_blocks(long rows) { return (rows + 31) / 32; }
static inline long fp4_k_tiles(long K) { return (K + 127) / 128; }
size_t binary_fp4_image_bytes(long rows, long K) { return (size_t)(fp4_row_blocks(rows) * fp4_k_tiles(K) * 2) * 1024; }

// 8 sign bits -> 8 E2M1 nibbles: bit 1 (value >= 0) -> 0x2 (+1.0), bit 0 -> 0xA (-1.0)
__device__ __forceinline__ uint32_t fp4_from_bits8(uint32_t b) {
    uint32_t x = b & 0xffu;
    x = (x | (x << 12)) & 0x000f000fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;  // bit j of b -> bit 4j
    return 0xaaaaaaaau ^ (x << 3);
}

// row-packed sign bits [rows, K/8] (LSB first) -> fragment-ordered FP4 image.  One lane per 16 output bytes.
__global__ __launch_bounds__(256) void fp4_image_kernel(const uint8_t* __restrict__ bits, uint4_t* __restrict__ img, long rows, long row_bytes,
                                                        long nfrag, int kb_per_row, int words) {
    const long f = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= nfrag) return;
    const int lane = threadIdx.x & 63;
    const long rb = f / kb_per_row;
    const int kb = (int)(f - rb * kb_per_row);
    const long row = rb * 32 + (lane & 31);
    const long b0 = (long)kb * 8 + (lane >> 5) * 4;  // first byte of this lane's 32 sign bits
    uint32_t o[4] = {0u, 0u, 0u, 0u};
    if (row < rows) {
        const uint8_t* p = bits + row * row_bytes;
        if (words && b0 + 4 <= row_bytes) {
            const uint32_t w = *reinterpret_cast<const uint32_t*>(p + b0);
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = fp4_from_bits8(w >> (8 * i));
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++)
                if (b0 + i < row_bytes) o[i] = fp4_from_bits8(p[b0 + i]);
        }
    }
    img[f * 64 + lane] = uint4_t{o[0], o[1], o[2], o[3]};
}

// values [rows, K] (dtype; + bias[K] when given) -> the same image, sign taken as (v >= 0): pack_rows + the kernel above in one pass
template <int DT>
__global__ __launch_bounds__(256) void fp4_image_values_kernel(const void* __restrict__ v, const void* __restrict__ bias, uint4_t* __restrict__ img, long rows,
                                                               long K, long nfrag, int kb_per_row) {
    const long f = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= nfrag) return;
    const int lane = threadIdx.x & 63;
    const long rb = f / kb_per_row;
    const int kb = (int)(f - rb * kb_per_row);
    const long row = rb * 32 + (lane & 31);
    const long k0 = (long)kb * 64 + (lane >> 5) * 32;
    uint32_t o[4] = {0u, 0u, 0u, 0u};
    if (row < rows) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t d = 0;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const long k = k0 + 8 * i + e;
                if (k < K) {
                    bool pos;
                    if constexpr (DT == 3) pos = ((const int8_t*)v)[row * K + k] >= 0;
                    else {
                        float a = dt_traits<DT>::load(v, row * K + k);
                        if (bias) a = dt_traits<DT>::round(a + dt_traits<DT>::load(bias, k));  // x + bias_a rounded in the tensor dtype, as torch does
                        pos = a >= 0.0f;
                    }
                    d |= (pos ? 0x2u : 0xau) << (4 * e);
                }
            }
            o[i] = d;
        }
    }
    img[f * 64 + lane] = uint4_t{o[0], o[1], o[2], o[3]};
}

// ---- LDS fragment reads (hand-issued: the compiler must not order them against the LDS-DMA by its own alias rules) ----
template <int OFF>
__device__ __forceinline__ v4i_t lds_read16(uint32_t addr) {
    v4i_t r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int N, int BASE>
__device__ __forceinline__ void read_frags(v4i_t (&f)[N], uint32_t addr) {
    static_assert(N == 1 || N == 2 || N == 4, "1, 2 or 4 fragments");
    f[0] = lds_read16<BASE>(addr);
    if constexpr (N >= 2) f[1] = lds_read16<BASE + 2048>(addr);
    if constexpr (N >= 4) {
        f[2] = lds_read16<BASE + 4096>(addr);
        f[3] = lds_read16<BASE + 6144>(addr);
    }
}
// s_waitcnt lgkmcnt(CNT) tied to the fragment registers it makes valid
template <int CNT, int NA, int NB>
__device__ __forceinline__ void wait_frags(v4i_t (&a)[NA], v4i_t (&b)[NB]) {
    if constexpr (NA == 4 && NB == 4)
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(CNT) : "memory");
    else if constexpr (NA == 2 && NB == 2)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : "n"(CNT) : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(b[0]) : "n"(CNT) : "memory");
}

// The MFMA as inline asm, on purpose.  Issued through the builtin the MFMAs are pure values: hipcc first sank all 96 of a three-stage
// loop body below the hand-issued reads, and a zero-instruction asm fence on the accumulator ("+a") that pins them makes the hazard
// recogniser put an s_nop between consecutive MFMAs -- one extra issue state costs a lone in-order wave ~28 cycles per MFMA here
// (26 ns per MFMA against 13.8 ns for the bare stream, profiles/r03_fp4_b_probe_mfma.txt / r03_fp4_b_ablations.txt).  As asm volatile
// the whole loop body (reads, waits, barrier, MFMAs) is one ordered stream and nothing is padded.  Hazards that are now ours:
// operands come from ds_read (covered by the lgkmcnt waits); an accumulator is re-used 16 MFMAs later; the epilogue's reads of the
// accumulators sit behind mfma_drain().  cbsz = blgp = 4: both operands FP4 (E2M1); block scales E8M0 0x7f = 2^0.
__device__ __forceinline__ void mfma_fp4(float16_t& c, const v4i_t& a, const v4i_t& b, int scale) {
    asm volatile("v_mfma_scale_f32_32x32x64_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+a"(c) : "v"(a), "v"(b), "v"(scale));
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }  // >= 18 wait states: XDL write -> VALU read

template <int I> struct ic_t { static constexpr int value = I; };
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(ic_t<I>{});
        static_for<I + 1, N>(f);
    }
}

// VAR 0: fragment reads and LDS-DMA pieces BETWEEN the MFMAs (one item per MFMA shadow: a lone in-order wave per SIMD overlaps
//        nothing it does not interleave), branch-free stage body (look-ahead clamped to the last K tile);
// VAR 1: the same pipeline with reads / DMA issued in bursts between the two 16-MFMA clusters of a stage (kept for the A/B).
template <int WM, int WN, int VAR>
__global__ __launch_bounds__(256) void xnor_fp4_gemm_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B, float* __restrict__ y, int M, int N,
                                                            int KT, int RBA, int RBB, int tiles_n, float scale) {
    constexpr int AF = 2 * WM, BF = 2 * WN;  // 32-row blocks per workgroup tile
    constexpr int NFR = (AF + BF) * 2;       // 1 KiB fragments per stage (k = 128)
    constexpr int PW = NFR / 4;              // LDS-DMA pieces per wave and stage
    constexpr int STAGE = NFR * 1024;
    constexpr int NR = WM + WN;              // fragment reads per k64 half
    constexpr int NM = WM * WN;              // MFMAs per k64 half
    // timing ablations (lab build only, results are wrong): VAR 2: no LDS-DMA in the loop, 3: no fragment reads, 4: neither, 5: and no barrier
    constexpr bool DMA = VAR < 2 || VAR == 3, READS = VAR < 3, BARRIER = VAR < 5;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * STAGE];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wy = wave >> 1, wx = wave & 1;
    // workgroups are dealt round-robin over the 8 XCDs: give every XCD one contiguous run of tiles (shared x rows stay in its L2)
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    if ((nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;

    // this wave's LDS-DMA sources: pieces wave*PW .. +PW-1 of the stage image [A fragments (row block, k half)] [B fragments]
    const uint8_t* src[PW];
#pragma unroll
    for (int j = 0; j < PW; j++) {
        const int p = wave * PW + j, fr = p >> 1, kbl = p & 1;
        const uint8_t* base;
        long rb;
        if (fr < AF) {
            rb = (long)tile_m * AF + fr;
            if (rb > RBA - 1) rb = RBA - 1;
            base = A;
        } else {
            rb = (long)tile_n * BF + (fr - AF);
            if (rb > RBB - 1) rb = RBB - 1;
            base = B;
        }
        src[j] = base + ((rb * (2 * KT) + kbl) * 64 + lane) * 16;
    }
    [[maybe_unused]] const int kt_last = KT - 1;
    // piece j of K tile kt (clamped: a look-ahead past the end re-fetches the last tile into a buffer nobody reads again)
    auto issue_piece = [&](int kt, int j) {
#if defined(__HIP_DEVICE_COMPILE__)
        const int ks = kt < kt_last ? kt : kt_last;
        auto* dst = (__attribute__((address_space(3))) unsigned char*)lds + (kt % 3) * STAGE + wave * (PW * 1024);
        __builtin_amdgcn_global_load_lds(src[j] + (long)ks * 2048, dst + j * 1024, 16, 0, 0);
#endif
    };

    const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
    const uint32_t a_addr = lds_base + (wy * WM * 2) * 1024 + lane * 16;
    const uint32_t b_addr = lds_base + ((AF + wx * WN) * 2) * 1024 + lane * 16;

    float16_t acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    v4i_t XA[WM], XB[WN], YA[WM], YB[WN], ZA[WM], ZB[WN];
    int e8m0_one = 0x7f7f7f7f;
    asm volatile("" : "+v"(e8m0_one));  // one VGPR for the whole kernel (not re-materialised per MFMA)

    // prologue: K tiles 0..2 requested, tile 0 landed on every wave, its first k64 half on the way to registers
#pragma unroll
    for (int s = 0; s < 3; s++)
#pragma unroll
        for (int j = 0; j < PW; j++) issue_piece(s, j);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_frags<WM, 0>(XA, a_addr);
    read_frags<WN, 0>(XB, b_addr);

    // fragment read r (0 .. NR-1) of k64 half H of the stage at byte offset so: the first WM are x (A operand) row blocks
    auto read_item = [&](auto ic, auto hc, uint32_t so, v4i_t (&TA)[WM], v4i_t (&TB)[WN]) {
        constexpr int R = decltype(ic)::value, H = decltype(hc)::value;
        if constexpr (R < WM) TA[R] = lds_read16<H * 1024 + R * 2048>(a_addr + so);
        else TB[R - WM] = lds_read16<H * 1024 + (R - WM) * 2048>(b_addr + so);
    };

    // one stage (K tile kt): P = its first k64 half (reads issued during the previous stage), Q takes the second half, NX the next
    // stage's first half.  The buffer a stage leaves is refilled with K tile kt+3 behind the barrier.
    auto stage = [&](int kt, v4i_t (&PA)[WM], v4i_t (&PB)[WN], v4i_t (&QA)[WM], v4i_t (&QB)[WN], v4i_t (&NA)[WM], v4i_t (&NB)[WN]) {
        const uint32_t so = (uint32_t)(kt % 3) * STAGE, sn = (uint32_t)((kt + 1) % 3) * STAGE;
        if constexpr (VAR != 1) {
            wait_frags<0>(PA, PB);
            static_for<0, NM>([&](auto mc) {
                constexpr int m = decltype(mc)::value, i = m / WN, j = m % WN;
                mfma_fp4(acc[i][j], PB[j], PA[i], e8m0_one);
                if constexpr (READS) static_for<m * NR / NM, (m + 1) * NR / NM>([&](auto rc) { read_item(rc, ic_t<1>{}, so, QA, QB); });
            });
            wait_frags<0>(QA, QB);  // every LDS read of this stage has returned: its buffer may be refilled behind the barrier
            if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");  // K tile kt+1 landed (kt+2 still in flight)
            if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
            constexpr int NI = NR + PW;  // items of the second half: next stage's first reads, then the refill pieces
            static_for<0, NM>([&](auto mc) {
                constexpr int m = decltype(mc)::value, i = m / WN, j = m % WN;
                mfma_fp4(acc[i][j], QB[j], QA[i], e8m0_one);
                static_for<m * NI / NM, (m + 1) * NI / NM>([&](auto xc) {
                    constexpr int x = decltype(xc)::value;
                    if constexpr (x < NR) { if constexpr (READS) read_item(xc, ic_t<0>{}, sn, NA, NB); }
                    else if constexpr (DMA) issue_piece(kt + 3, x - NR);
                });
            });
        } else {
            read_frags<WM, 1024>(QA, a_addr + so);
            read_frags<WN, 1024>(QB, b_addr + so);
            wait_frags<NR>(PA, PB);
            static_for<0, NM>([&](auto mc) {
                constexpr int m = decltype(mc)::value, i = m / WN, j = m % WN;
                mfma_fp4(acc[i][j], PB[j], PA[i], e8m0_one);
            });
            wait_frags<0>(QA, QB);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int j = 0; j < PW; j++) issue_piece(kt + 3, j);
            read_frags<WM, 0>(NA, a_addr + sn);
            read_frags<WN, 0>(NB, b_addr + sn);
            static_for<0, NM>([&](auto mc) {
                constexpr int m = decltype(mc)::value, i = m / WN, j = m % WN;
                mfma_fp4(acc[i][j], QB[j], QA[i], e8m0_one);
            });
        }
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
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the clamped look-ahead pieces / reads must not outlive the workgroup's LDS
    mfma_drain();

    // The MFMAs were issued as D = w_frag (A operand: D rows = output features n) x x_frag (B operand: D columns = rows m of x), so in the
    // 32 x 32 C/D layout (column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) a lane holds ONE row m and, per group of
    // four registers, FOUR CONSECUTIVE n: one 16-byte store per group (64 per lane instead of 256 dword stores -- the dword form was
    // store-issue bound: 13 us of fixed cost per launch, profiles/r03_fp4_c_k_slope.txt).
    const int m_l = lane & 31, n_l = 4 * (lane >> 5);
    const bool vec_ok = (N & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
#pragma unroll
    for (int i = 0; i < WM; i++) {
        const int m = (tile_m * AF + wy * WM + i) * 32 + m_l;
        if (m < M) {
            float* yr = y + (long)m * N;
#pragma unroll
            for (int j = 0; j < WN; j++) {
                const int n0 = (tile_n * BF + wx * WN + j) * 32 + n_l;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int n = n0 + 8 * q;
                    const float4_t v = {acc[i][j][4 * q] * scale, acc[i][j][4 * q + 1] * scale, acc[i][j][4 * q + 2] * scale, acc[i][j][4 * q + 3] * scale};
                    if (vec_ok && n + 3 < N) *reinterpret_cast<float4_t*>(yr + n) = v;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; e++)
                            if (n + e < N) yr[n + e] = v[e];
                    }
                }
            }
        }
    }
}

// ---- launchers ------------------------------------------------------------------------------------------------
int binary_fp4_image_launch(const uint8_t* rowpacked, uint8_t* image, long rows, long K, hipStream_t st) {
    const long kb_per_row = 2 * fp4_k_tiles(K), nfrag = fp4_row_blocks(rows) * kb_per_row, row_bytes = K / 8;
    const int words = (row_bytes % 4 == 0) && ((uintptr_t)rowpacked % 4 == 0);
    hipLaunchKernelGGL(fp4_image_kernel, dim3((unsigned)cdivl(nfrag, 4)), dim3(256), 0, st, rowpacked, (uint4_t*)image, rows, row_bytes, nfrag, (int)kb_per_row, words);
    return check_launch("fp4_image_kernel");
}

int binary_fp4_image_values_launch(const void* v, const void* bias, uint8_t* image, long rows, long K, int dtype, hipStream_t st) {
    const long kb_per_row = 2 * fp4_k_tiles(K), nfrag = fp4_row_blocks(rows) * kb_per_row;
    const dim3 grid((unsigned)cdivl(nfrag, 4));
    switch (dtype) {
        case BIE_F16: hipLaunchKernelGGL(fp4_image_values_kernel<BIE_F16>, grid, dim3(256), 0, st, v, bias, (uint4_t*)image, rows, K, nfrag, (int)kb_per_row); break;
        case BIE_BF16: hipLaunchKernelGGL(fp4_image_values_kernel<BIE_BF16>, grid, dim3(256), 0, st, v, bias, (uint4_t*)image, rows, K, nfrag, (int)kb_per_row); break;
        case BIE_F32: hipLaunchKernelGGL(fp4_image_values_kernel<BIE_F32>, grid, dim3(256), 0, st, v, bias, (uint4_t*)image, rows, K, nfrag, (int)kb_per_row); break;
        default: hipLaunchKernelGGL(fp4_image_values_kernel<3>, grid, dim3(256), 0, st, v, nullptr, (uint4_t*)image, rows, K, nfrag, (int)kb_per_row); break;
    }
    return check_launch("fp4_image_values_kernel");
}

int binary_fp4_gemm_launch(const uint8_t* ximg, const uint8_t* wimg, float* y, long M, long N, long K, float scale, int tile, hipStream_t st) {
    const int KT = (int)fp4_k_tiles(K), RBA = (int)fp4_row_blocks(M), RBB = (int)fp4_row_blocks(N);
    const long t256 = cdivl(M, 256) * cdivl(N, 256);
    const char* ev = getenv("BIE_FP4_VAR");  // 1: burst form (A/B only)
    const int var = ev ? atoi(ev) : 0;
    // 256 x 256 tiles (one wave per SIMD, LDS reads at half the array's rate) once they fill most of the chip, else 128 x 128
    const bool big = tile == 256 || (tile != 128 && t256 >= 192);
    if (big) {
        const int tn = (int)cdivl(N, 256);
        if (var == 1) hipLaunchKernelGGL((xnor_fp4_gemm_kernel<4, 4, 1>), dim3((unsigned)t256), dim3(256), 0, st, ximg, wimg, y, (int)M, (int)N, KT, RBA, RBB, tn, scale);
#ifdef BIE_FP4_LAB
        else if (var == 2) hipLaunchKernelGGL((xnor_fp4_gemm_kernel<4, 4, 2>), dim3((unsigned)t256), dim3(256), 0, st, ximg, wimg, y, (int)M, (int)N, KT, RBA, RBB, tn, scale);
        else if (var == 3) hipLaunchKernelGGL((xnor_fp4_gemm_kernel<4, 4, 3>), dim3((unsigned)t256), dim3(256), 0, st, ximg, wimg, y, (int)M, (int)N, KT, RBA, RBB, tn, scale);
        else if (var == 4) hipLaunchKernelGGL((xnor_fp4_gemm_kernel<4, 4, 4>), dim3((unsigned)t256), dim3(256), 0, st, ximg, wimg, y, (int)M, (int)N, KT, RBA, RBB, tn, scale);
        else if (var == 5) hipLaunchKernelGGL((xnor_fp4_gemm_kernel<4, 4, 5>), dim3((unsigned)t256), dim3(256), 0, st, ximg, wimg, y, (int)M, (int)N, KT, RBA, RBB, tn, scale);
#endif
        else hipLaunchKernelGGL((xnor_fp4_gemm_kernel<4, 4, 0>), dim3((unsigned)t256), dim3(256), 0, st, ximg, wimg, y, (int)M, (int)N, KT, RBA, RBB, tn, scale);
    } else {
        const int tn = (int)cdivl(N, 128);
        const dim3 grid((unsigned)(cdivl(M, 128) * tn));
        if (var == 1) hipLaunchKernelGGL((xnor_fp4_gemm_kernel<2, 2, 1>), grid, dim3(256), 0, st, ximg, wimg, y, (int)M, (int)N, KT, RBA, RBB, tn, scale);
        else hipLaunchKernelGGL((xnor_fp4_gemm_kernel<2, 2, 0>), grid, dim3(256), 0, st, ximg, wimg, y, (int)M, (int)N, KT, RBA, RBB, tn, scale);
    }
    return check_launch("xnor_fp4_gemm_kernel");
}

}  // namespace bie
