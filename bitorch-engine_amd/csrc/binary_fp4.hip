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
#include "mfma_pipe.cuh"
#include <stdlib.h>

namespace bie {

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

// values [rows, K] (dtype; + bias[K] when given) -> the same image, sign taken as (v >= 0): bie_binary_pack_rows_u8 + the kernel above in
// one pass.  One wave per fragment (32 rows x 64 k): lane (row l >> 3 of 8, chunk l & 7) loads 8 consecutive values (8 lanes = one
// row's 64 values, contiguous), four passes cover the 32 rows; the nibble words go through a 1 KiB LDS image of the fragment so that
// the store is the fragment's 1 KiB, contiguous.
template <int DT>
__device__ __forceinline__ void load8(const void* p, long idx, float (&v)[8]) {
    if constexpr (DT == BIE_F32) {
        const float4_t a = *reinterpret_cast<const float4_t*>((const float*)p + idx), b = *reinterpret_cast<const float4_t*>((const float*)p + idx + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else if constexpr (DT == 3) {
        const uint2_t a = *reinterpret_cast<const uint2_t*>((const int8_t*)p + idx);
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = (float)(int8_t)(((e < 4 ? a.x : a.y) >> (8 * (e & 3))) & 0xff);
    } else {
        const uint4_t a = *reinterpret_cast<const uint4_t*>((const uint16_t*)p + idx);
        const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const uint32_t h = (w[e >> 1] >> (16 * (e & 1))) & 0xffffu;
            v[e] = DT == BIE_F16 ? f16_bits_to_f32(h) : bf16_bits_to_f32(h);
        }
    }
}

template <int DT>
__global__ __launch_bounds__(256) void fp4_image_values_kernel(const void* __restrict__ v, const void* __restrict__ bias, uint4_t* __restrict__ img, long rows,
                                                               long K, long nfrag, int kb_per_row, int vec) {
    __shared__ uint32_t tile[4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long f = (long)blockIdx.x * 4 + wave;
    if (f < nfrag) {
        const long rb = f / kb_per_row;
        const int kb = (int)(f - rb * kb_per_row);
        const int c = lane & 7;
        const long k0 = (long)kb * 64 + 8 * c;
        float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (bias) {
            if (vec && k0 + 8 <= K) load8<DT>(bias, k0, bv);
            else
                for (int e = 0; e < 8; e++)
                    if (k0 + e < K) {
                        if constexpr (DT != 3) bv[e] = dt_traits<DT>::load(bias, k0 + e);
                    }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int r = t * 8 + (lane >> 3);
            const long row = rb * 32 + r;
            uint32_t d = 0;
            if (row < rows && k0 < K) {
                float a[8];
                int valid = 8;
                if (vec && k0 + 8 <= K) load8<DT>(v, row * K + k0, a);
                else {
                    valid = (int)(K - k0 < 8 ? K - k0 : 8);
                    for (int e = 0; e < 8; e++) {
                        a[e] = 0.f;
                        if (e < valid) {
                            if constexpr (DT == 3) a[e] = (float)((const int8_t*)v)[row * K + k0 + e];
                            else a[e] = dt_traits<DT>::load(v, row * K + k0 + e);
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    float x = a[e];
                    if constexpr (DT != 3) {
                        if (bias) x = dt_traits<DT>::round(x + bv[e]);  // x + bias_a rounded in the tensor dtype, as torch does
                    }
                    if (e < valid) d |= (x >= 0.0f ? 0x2u : 0xau) << (4 * e);
                }
            }
            tile[wave][((c >> 2) * 32 + r) * 4 + (c & 3)] = d;  // fragment lane (k half, row), dword c & 3
        }
    }
    __syncthreads();
    if (f < nfrag) img[f * 64 + lane] = *reinterpret_cast<const uint4_t*>(&tile[wave][lane * 4]);
}

// Binary conv2d as the same GEMM: rows = output pixels (b, oy, ox), k' = (tap, channel) -- any k order works as long as both operands
// use it, and tap-major is the order the channel-minor activation bits (pack_nhwc_bits_kernel: xbits[b][h][w][C/32]) and the tap-major
// weight words (conv_weight_taps_kernel: wtaps[oc][tap][C/32]) already have.  A lane's 32 nibbles are ONE word of xbits (C % 32 == 0)
// or, outside the picture, 32 x -1.0: the reference counts padding as -1 (binary_conv.cpp:319-365), which is bit 0 = 0xA here.
// Replaces the bit-im2col image of binary_conv.cpp:319-365 for large batches (im2binary_col + the GEMM of :464-530).
__global__ __launch_bounds__(256) void conv_fp4_image_kernel(const uint32_t* __restrict__ xbits, uint4_t* __restrict__ img, int H, int W, int CW, int OH, int OW,
                                                             int ks, int stride, int pad, int dil, long rows, long nfrag, int kb_per_row,
                                                             int words_per_row) {
    const long f = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (f >= nfrag) return;
    const int lane = threadIdx.x & 63;
    const long rb = f / kb_per_row;
    const int kb = (int)(f - rb * kb_per_row);
    const long row = rb * 32 + (lane & 31);
    const int wi = kb * 2 + (lane >> 5);
    uint4_t o = {0u, 0u, 0u, 0u};
    if (row < rows && wi < words_per_row) {
        const int tap = wi / CW, cw = wi - tap * CW;
        const int ti = tap / ks, tj = tap - ti * ks;
        const int P = OH * OW;
        const long b = row / P;
        const int p = (int)(row - b * P);
        const int oy = p / OW, ox = p - oy * OW;
        const int iy = oy * stride - pad + ti * dil, ix = ox * stride - pad + tj * dil;
        uint32_t w = 0u;  // outside the picture: every channel -1
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) w = xbits[((b * H + iy) * W + ix) * CW + cw];
        o = uint4_t{fp4_from_bits8(w), fp4_from_bits8(w >> 8), fp4_from_bits8(w >> 16), fp4_from_bits8(w >> 24)};
    }
    img[f * 64 + lane] = o;
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
// VAR 0: fragment reads and LDS-DMA pieces BETWEEN the MFMAs (one item per MFMA shadow: a lone in-order wave per SIMD overlaps
//        nothing it does not interleave), branch-free stage body (look-ahead clamped to the last K tile);
// VAR 1: the same pipeline with reads / DMA issued in bursts between the two 16-MFMA clusters of a stage (kept for the A/B).
// ODT = -1: y fp32 [M, N] = (K - 2*popc) * scale.  ODT = -2: the conv2d output, rows m = (image b, pixel p) of conv_p pixels, columns =
// output channels: y fp32 [B, N, conv_p] (NCHW), y[(b * N + n) * conv_p + p].  ODT = BIE_F16 / BF16 / F32: the BinaryLinearCuda layer epilogue, y (ODT) =
// dt(dt(dt(K - 2*popc) * scale_a) * scale_w) with scale_a / scale_w device scalars of that dtype (NULL = 1): the roundings of
// `forward(...).to(input.dtype) * scale_a * scale_w` (layers/qlinear/binary/cuda/layer.py:58-63), as xnor_fused_kernel.
template <int WM, int WN, int VAR, int ODT>
__global__ __launch_bounds__(256) void xnor_fp4_gemm_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B, void* __restrict__ yv, int M, int N,
                                                            int KT, int RBA, int RBB, int tiles_n, float scale, const void* __restrict__ scale_a,
                                                            const void* __restrict__ scale_w, int conv_p) {
    constexpr int AF = 2 * WM, BF = 2 * WN;  // 32-row blocks per workgroup tile
    constexpr int NFR = (AF + BF) * 2;       // 1 KiB fragments per stage (k = 128)
    constexpr int PW = NFR / 4;              // LDS-DMA pieces per wave and stage
    constexpr int STAGE = NFR * 1024;
    constexpr int NR = WM + WN;              // fragment reads per k64 half
    constexpr int NM = WM * WN;              // MFMAs per k64 half
    // Items between the MFMAs: a cluster's fragment reads go behind its FIRST MFMAs (RPM per MFMA: all issued by the middle of the
    // cluster, so the lgkmcnt(0) in front of the next cluster finds them landed -- spread over the whole cluster the last read sat
    // behind the last MFMA and its latency was exposed twice per stage), the refill pieces behind the remaining ones (DPM per MFMA).
    constexpr int RPM = (2 * NR + NM - 1) / NM, M0 = (NR + RPM - 1) / RPM, DPM = (PW + (NM - M0) - 1) / (NM - M0);
    // timing ablations (lab build only, results are wrong): VAR 2: no LDS-DMA in the loop, 3: no fragment reads, 4: neither, 5: and no barrier
    constexpr bool DMA = VAR < 2 || VAR == 3, READS = VAR < 3, BARRIER = VAR < 5;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * STAGE];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wy = wave >> 1, wx = wave & 1;
    // workgroups are dealt round-robin over the 8 XCDs: give every XCD one contiguous run of tiles (shared x rows stay in its L2)
    int bid = blockIdx.x;
    const int nblk = gridDim.x;
    // tile rows per XCD run (profiles/r03_pipe_gm_ab.txt): 4 everywhere (128 x 128 tiles 41 vs 51 us at 4096^3, the bf16 layer epilogue 105 vs 112 us
    // at M = 4096 8192x8192) except the 256 x 256 tile with fp32 output, whose 1 KiB output rows go out faster one tile row at a time (8192^3: 246 vs 282 us)
    constexpr int GM = (ODT == -1 && WM == 4) ? 1 : BIE_PIPE_GM;
    int tile_m, tile_n;
    pipe_tile(bid, nblk, tiles_n, GM, tile_m, tile_n);

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
                if constexpr (READS) static_for<imin(m * RPM, NR), imin((m + 1) * RPM, NR)>([&](auto rc) { read_item(rc, ic_t<1>{}, so, QA, QB); });
            });
            wait_frags<0>(QA, QB);  // every LDS read of this stage has returned: its buffer may be refilled behind the barrier
            if constexpr (DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW) : "memory");  // K tile kt+1 landed (kt+2 still in flight)
            if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
            static_for<0, NM>([&](auto mc) {
                constexpr int m = decltype(mc)::value, i = m / WN, j = m % WN;
                mfma_fp4(acc[i][j], QB[j], QA[i], e8m0_one);
                // the next stage's first reads behind the first MFMAs, the refill pieces behind the rest
                if constexpr (READS) static_for<imin(m * RPM, NR), imin((m + 1) * RPM, NR)>([&](auto rc) { read_item(rc, ic_t<0>{}, sn, NA, NB); });
                if constexpr (DMA && m >= M0)
                    static_for<imin((m - M0) * DPM, PW), imin((m - M0 + 1) * DPM, PW)>([&](auto pc) { issue_piece(kt + 3, decltype(pc)::value); });
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
    const int m_l = lane & 31;
    if constexpr (ODT == -2) {
        // NCHW scatter: for one register the 32 lanes of a half-wave hold 32 consecutive pixels of one channel -> dword stores, contiguous
        // across lanes inside an image
        float* y = (float*)yv;
        const int n_l = 4 * (lane >> 5);
#pragma unroll
        for (int i = 0; i < WM; i++) {
            const int m = (tile_m * AF + wy * WM + i) * 32 + m_l;
            if (m < M) {
                const int b = m / conv_p, p = m - b * conv_p;
                float* yb = y + (long)b * N * conv_p + p;
#pragma unroll
                for (int j = 0; j < WN; j++) {
                    const int n0 = (tile_n * BF + wx * WN + j) * 32 + n_l;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int n = n0 + 8 * (r >> 2) + (r & 3);
                        if (n < N) yb[(long)n * conv_p] = acc[i][j][r] * scale;
                    }
                }
            }
        }
    } else if constexpr (ODT < 0 || ODT == BIE_F32) {
        float* y = (float*)yv;
        float sa = 1.0f, sw = 1.0f;
        if constexpr (ODT == BIE_F32) {
            if (scale_a) sa = *(const float*)scale_a;
            if (scale_w) sw = *(const float*)scale_w;
        }
        const int n_l = 4 * (lane >> 5);
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
                        float4_t v;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            if constexpr (ODT < 0) v[e] = acc[i][j][4 * q + e] * scale;
                            else v[e] = (acc[i][j][4 * q + e] * sa) * sw;  // fp32 layer: two rounded multiplies
                        }
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
    } else {
        // 16-bit outputs: the two half-waves hold the n + 0..3 / n + 4..7 quads of the same row; v_permlane32_swap_b32 trades the packed
        // quads of register groups q and q + 1 between them, after which a lane owns 8 consecutive n of group 2*(q/2) + (lane >> 5):
        // one 16-byte store (32 per lane)
        uint16_t* y = (uint16_t*)yv;
        const float sa = scale_a ? dt_traits<ODT>::load(scale_a, 0) : 1.0f, sw = scale_w ? dt_traits<ODT>::load(scale_w, 0) : 1.0f;
        // The three roundings on PAIRS (the generic per-value form -- round, multiply, round, multiply, round, pack -- was 11 VALU per value,
        // 2900 per wave): bf16: v_cvt_pk_bf16_f32 is rounding and pack in one; the two halves are widened (shift / mask), multiplied in
        // fp32 (exact products of two bf16 values) and re-packed: 5.5 per value.  fp16: v_pk_mul_f16 IS dt(v * s) for fp16 operands
        // (the fp32 product of two fp16 values is exact, so one rounding either way): 2.5 per value.
        auto fin2 = [&](float c0, float c1) -> uint32_t {
            if constexpr (ODT == BIE_BF16) {
                uint32_t p = pack_bf16x2(c0, c1);
                if (scale_a) p = pack_bf16x2(__uint_as_float(p << 16) * sa, __uint_as_float(p & 0xffff0000u) * sa);
                if (scale_w) p = pack_bf16x2(__uint_as_float(p << 16) * sw, __uint_as_float(p & 0xffff0000u) * sw);
                return p;
            } else {
                half2_t p = half2_t{(half_t)c0, (half_t)c1};
                if (scale_a) p = p * half2_t{(half_t)sa, (half_t)sa};
                if (scale_w) p = p * half2_t{(half_t)sw, (half_t)sw};
                return __builtin_bit_cast(uint32_t, p);
            }
        };
        auto lo_f32 = [&](uint32_t p) { return ODT == BIE_BF16 ? __uint_as_float(p << 16) : f16_bits_to_f32(p & 0xffffu); };
        auto hi_f32 = [&](uint32_t p) { return ODT == BIE_BF16 ? __uint_as_float(p & 0xffff0000u) : f16_bits_to_f32(p >> 16); };
        const bool vec_ok = (N & 7) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
        const int half = lane >> 5;
#pragma unroll
        for (int i = 0; i < WM; i++) {
            const int m = (tile_m * AF + wy * WM + i) * 32 + m_l;
            uint16_t* yr = y + (long)(m < M ? m : 0) * N;
#pragma unroll
            for (int j = 0; j < WN; j++) {
                const int nb = (tile_n * BF + wx * WN + j) * 32;
#pragma unroll
                for (int qp = 0; qp < 2; qp++) {
                    // registers 8qp .. 8qp+3: group 2qp (n + 4*half + e), 8qp+4 .. 8qp+7: group 2qp + 1
                    const uint32_t p0 = fin2(acc[i][j][8 * qp], acc[i][j][8 * qp + 1]), p1 = fin2(acc[i][j][8 * qp + 2], acc[i][j][8 * qp + 3]);
                    const uint32_t p2 = fin2(acc[i][j][8 * qp + 4], acc[i][j][8 * qp + 5]), p3 = fin2(acc[i][j][8 * qp + 6], acc[i][j][8 * qp + 7]);
                    if (vec_ok) {
                        const auto s0 = __builtin_amdgcn_permlane32_swap(p0, p2, false, false);  // first operand's upper half <-> second's lower half
                        const auto s1 = __builtin_amdgcn_permlane32_swap(p1, p3, false, false);
                        const int n = nb + 8 * (2 * qp + half);
                        if (m < M && n < N) *reinterpret_cast<uint4_t*>(yr + n) = uint4_t{s0[0], s1[0], s0[1], s1[1]};
                    } else if (m < M) {
#pragma unroll
                        for (int e = 0; e < 8; e++) {
                            const int n = nb + 8 * (2 * qp + (e >> 2)) + 4 * half + (e & 3);
                            const uint32_t pe = e < 2 ? p0 : (e < 4 ? p1 : (e < 6 ? p2 : p3));
                            if (n < N) dt_traits<ODT>::store(yr, n, (e & 1) ? hi_f32(pe) : lo_f32(pe));
                        }
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
    const int vec = (K % 8 == 0) && ((reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0;  // 8-value chunks stay aligned in every row
    switch (dtype) {
        case BIE_F16: hipLaunchKernelGGL(fp4_image_values_kernel<BIE_F16>, grid, dim3(256), 0, st, v, bias, (uint4_t*)image, rows, K, nfrag, (int)kb_per_row, vec); break;
        case BIE_BF16: hipLaunchKernelGGL(fp4_image_values_kernel<BIE_BF16>, grid, dim3(256), 0, st, v, bias, (uint4_t*)image, rows, K, nfrag, (int)kb_per_row, vec); break;
        case BIE_F32: hipLaunchKernelGGL(fp4_image_values_kernel<BIE_F32>, grid, dim3(256), 0, st, v, bias, (uint4_t*)image, rows, K, nfrag, (int)kb_per_row, vec); break;
        default: hipLaunchKernelGGL(fp4_image_values_kernel<3>, grid, dim3(256), 0, st, v, nullptr, (uint4_t*)image, rows, K, nfrag, (int)kb_per_row, vec); break;
    }
    return check_launch("fp4_image_values_kernel");
}

template <int ODT>
static void fp4_gemm_launch_dt(const uint8_t* ximg, const uint8_t* wimg, void* y, long M, long N, long K, float scale, const void* sa, const void* sw,
                               int tile, hipStream_t st, int conv_p = 0) {
    const int KT = (int)fp4_k_tiles(K), RBA = (int)fp4_row_blocks(M), RBB = (int)fp4_row_blocks(N);
    const long t256 = cdivl(M, 256) * cdivl(N, 256);
    const char* ev = getenv("BIE_FP4_VAR");  // 1: burst form (A/B only); 2..5: timing ablations of the lab build
    const int var = ev ? atoi(ev) : 0;
    // 256 x 256 tiles (one wave per SIMD, LDS reads at half the array's rate) once they fill most of the chip, else 128 x 128
    const bool big = tile == 256 || (tile != 128 && t256 >= 192);
#define BIE_FP4_GO(WM_, VAR_, GRID_, TN_) \
    hipLaunchKernelGGL((xnor_fp4_gemm_kernel<WM_, WM_, VAR_, ODT>), GRID_, dim3(256), 0, st, ximg, wimg, y, (int)M, (int)N, KT, RBA, RBB, TN_, scale, sa, sw, conv_p)
    if (big) {
        const int tn = (int)cdivl(N, 256);
        const dim3 grid((unsigned)t256);
        if constexpr (ODT == -1) {
            if (var == 1) { BIE_FP4_GO(4, 1, grid, tn); return; }
#ifdef BIE_FP4_LAB
            if (var == 2) { BIE_FP4_GO(4, 2, grid, tn); return; }
            if (var == 3) { BIE_FP4_GO(4, 3, grid, tn); return; }
            if (var == 4) { BIE_FP4_GO(4, 4, grid, tn); return; }
            if (var == 5) { BIE_FP4_GO(4, 5, grid, tn); return; }
#endif
        }
        BIE_FP4_GO(4, 0, grid, tn);
    } else if (tile == 64 || (tile != 128 && cdivl(M, 128) * cdivl(N, 128) < 192)) {
        // mid M: 128 x 64 tiles (wave tile 64 x 32) double the workgroups when the 128 x 128 grid would leave CUs idle
        const int tn = (int)cdivl(N, 64);
        hipLaunchKernelGGL((xnor_fp4_gemm_kernel<2, 1, 0, ODT>), dim3((unsigned)(cdivl(M, 128) * tn)), dim3(256), 0, st, ximg, wimg, y, (int)M, (int)N, KT, RBA, RBB, tn, scale, sa, sw, conv_p);
    } else {
        const int tn = (int)cdivl(N, 128);
        const dim3 grid((unsigned)(cdivl(M, 128) * tn));
        if constexpr (ODT == -1) {
            if (var == 1) { BIE_FP4_GO(2, 1, grid, tn); return; }
        }
        BIE_FP4_GO(2, 0, grid, tn);
    }
#undef BIE_FP4_GO
}

// dtype < 0: y float = (K - 2*popc) * scale; else the layer epilogue in that dtype with device scalars sa / sw
int binary_fp4_gemm_launch(const uint8_t* ximg, const uint8_t* wimg, void* y, long M, long N, long K, float scale, const void* sa, const void* sw, int dtype,
                           int tile, hipStream_t st) {
    if (dtype < 0) fp4_gemm_launch_dt<-1>(ximg, wimg, y, M, N, K, scale, nullptr, nullptr, tile, st);
    else if (dtype == BIE_F16) fp4_gemm_launch_dt<BIE_F16>(ximg, wimg, y, M, N, K, 1.0f, sa, sw, tile, st);
    else if (dtype == BIE_BF16) fp4_gemm_launch_dt<BIE_BF16>(ximg, wimg, y, M, N, K, 1.0f, sa, sw, tile, st);
    else fp4_gemm_launch_dt<BIE_F32>(ximg, wimg, y, M, N, K, 1.0f, sa, sw, tile, st);
    return check_launch("xnor_fp4_gemm_kernel");
}

// conv2d forward on the matrix pipe: x [B, C, H, W] -> channel-minor sign bits (ws head, binary.hip's pack_nhwc_bits) -> FP4 image of the
// (pixel, tap, channel) matrix -> GEMM against the weights' image with the NCHW epilogue.  C % 32 == 0.
int binary_pack_nhwc_bits_launch(const void* x, uint32_t* xbits, int B, int C, int HW, int dtype, hipStream_t st);  // binary.hip
size_t binary_conv_fp4_workspace_bytes(int B, int C, int H, int W, int ks, int stride, int pad, int dil) {
    const int OH = (H + 2 * pad - dil * (ks - 1) - 1) / stride + 1, OW = (W + 2 * pad - dil * (ks - 1) - 1) / stride + 1;
    if (OH <= 0 || OW <= 0) return 0;
    const size_t xb = ((size_t)B * H * W * (C / 32) * 4 + 1023) & ~(size_t)1023;
    return xb + binary_fp4_image_bytes((long)B * OH * OW, (long)ks * ks * C);
}
int binary_conv_fp4_launch(const void* x, const uint8_t* wimg, float* y, void* ws, int B, int C, int H, int W, int OC, int ks, int stride, int pad, int dil,
                           float scale, int dtype, int tile, hipStream_t st) {
    const int OH = (H + 2 * pad - dil * (ks - 1) - 1) / stride + 1, OW = (W + 2 * pad - dil * (ks - 1) - 1) / stride + 1;
    const int CW = C / 32;
    uint32_t* xbits = reinterpret_cast<uint32_t*>(ws);
    uint8_t* ximg = reinterpret_cast<uint8_t*>(ws) + (((size_t)B * H * W * CW * 4 + 1023) & ~(size_t)1023);
    int rc = binary_pack_nhwc_bits_launch(x, xbits, B, C, H * W, dtype, st);
    if (rc) return rc;
    const long rows = (long)B * OH * OW, Kp = (long)ks * ks * C;
    const long kb_per_row = 2 * fp4_k_tiles(Kp), nfrag = fp4_row_blocks(rows) * kb_per_row;
    hipLaunchKernelGGL(conv_fp4_image_kernel, dim3((unsigned)cdivl(nfrag, 4)), dim3(256), 0, st, xbits, (uint4_t*)ximg, H, W, CW, OH, OW, ks, stride, pad, dil, rows,
                       nfrag, (int)kb_per_row, ks * ks * CW);
    rc = check_launch("conv_fp4_image_kernel");
    if (rc) return rc;
    fp4_gemm_launch_dt<-2>(ximg, wimg, y, rows, OC, Kp, scale, nullptr, nullptr, tile, st, OH * OW);
    return check_launch("xnor_fp4_gemm_kernel<conv>");
}

}  // namespace bie
