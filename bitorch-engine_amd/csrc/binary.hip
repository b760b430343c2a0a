// 1-bit W / 1-bit A linear + conv2d (XNOR-popcount) and the functions/cuda pack helpers for gfx950.
// CDNA4 has no 1-bit MFMA, so the contraction runs on the VALU: v_xor_b32 + v_bcnt_u32_b32 (popcount
// with accumulate) = 2 instructions per 32 binary MACs per lane.
//   y = (K - 2 * popcount(xbits ^ wbits)) * scale,  bit = (value >= 0)
// Replaces binary_linear.cpp (_get_binary_row :43-54, _xnor_gemm_unrolled :249-295, forward :494-512),
// binary_linear_cuda_kernel.cu (BMMAS_new :155-181, BMM32_Arow_Brow_UD :308-393),
// binary_linear_cutlass_kernel.cu (:44-113), binary_conv.cpp (im2binary_col :319-365, forward :464-530)
// and functions_cuda_kernel.cu (:73-207).
#include "bie_common.h"
#include <stdlib.h>

namespace bie {

template <int DT>
__device__ __forceinline__ bool sign_bit(const void* p, long i) {
    if constexpr (DT == 3) return ((const int8_t*)p)[i] >= 0;
    else return dt_traits<DT>::load(p, i) >= 0.0f;
}

// Wave-wide integer sum on the DPP network (no LDS traffic): quad swaps, half-row and row mirrors, then the two row broadcasts;
// the total lands in lane 63 and is read back as a scalar.
__device__ __forceinline__ int wave_sum_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);  // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);  // row_mirror: every lane of a row holds the row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast31 into rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

// ---- packing ---------------------------------------------------------------------------------------------
// rows x K values -> rows x K/8 bytes, LSB first.  One thread per output byte.
template <int DT>
__global__ __launch_bounds__(256) void pack_rows_kernel(const void* __restrict__ a, uint8_t* __restrict__ out, long n_bytes) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_bytes) return;
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) v |= (uint32_t)sign_bit<DT>(a, i * 8 + j) << j;
    out[i] = (uint8_t)v;
}

// w [N][K] -> column bit-planes [K/8][N]
template <int DT>
__global__ __launch_bounds__(256) void pack_cols_kernel(const void* __restrict__ w, uint8_t* __restrict__ out, long N, long K) {
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    const long kb = blockIdx.y;
    if (n >= N) return;
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) v |= (uint32_t)sign_bit<DT>(w, n * K + kb * 8 + j) << j;
    out[kb * N + n] = (uint8_t)v;
}

// ---- XNOR GEMM on 32-bit words: y[b][m][n] = (Kbits - 2*popc(A[m] ^ B[b][n])) * scale -------------------------
// 64 x 64 output tile per block, 16 x 16 threads, 4 x 4 outputs per thread, KC words per LDS stage.
constexpr int XT = 64, XKC = 32;

template <int RM>  // rows per thread: the block tile is 16*RM rows x 64 columns (RM = 1 for skinny M: 4x the blocks)
__global__ __launch_bounds__(256) void xnor_gemm_kernel(const uint32_t* __restrict__ A, const uint32_t* __restrict__ B,
                                                        float* __restrict__ y, int M, int N, int KW, int Kbits, float scale,
                                                        long strideA, long strideB, long strideY) {
    constexpr int XTM = 16 * RM;
    __shared__ uint32_t As[XTM][XKC + 1];
    __shared__ uint32_t Bs[XT][XKC + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * XTM, n0 = blockIdx.x * XT;
    A += (long)blockIdx.z * strideA;
    B += (long)blockIdx.z * strideB;
    y += (long)blockIdx.z * strideY;
    int acc[RM][4];
#pragma unroll
    for (int i = 0; i < RM; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = 0;
    for (int k0 = 0; k0 < KW; k0 += XKC) {
        for (int idx = threadIdx.x; idx < XT * XKC; idx += 256) {
            const int r = idx / XKC, c = idx % XKC;
            const int k = k0 + c;
            if (r < XTM) As[r][c] = (m0 + r < M && k < KW) ? A[(long)(m0 + r) * KW + k] : 0u;
            Bs[r][c] = (n0 + r < N && k < KW) ? B[(long)(n0 + r) * KW + k] : 0u;
        }
        __syncthreads();
#pragma unroll 8
        for (int c = 0; c < XKC; c++) {
            uint32_t a[RM], b[4];
#pragma unroll
            for (int i = 0; i < RM; i++) a[i] = As[ty + 16 * i][c];
#pragma unroll
            for (int j = 0; j < 4; j++) b[j] = Bs[tx + 16 * j][c];
#pragma unroll
            for (int i = 0; i < RM; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[i][j] += __builtin_popcount(a[i] ^ b[j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < RM; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int m = m0 + ty + 16 * i, n = n0 + tx + 16 * j;
            if (m < M && n < N) y[(long)m * N + n] = (float)(Kbits - 2 * acc[i][j]) * scale;
        }
}

// Large-M variant: 128 x 128 block tile, 8 x 8 outputs per thread, K in chunks of 16 words.  Per 4 k-words a thread reads 8 + 8
// 16-byte LDS vectors for 256 xor + popcount pairs (v_bcnt_u32_b32 accumulates by itself: exactly 2 VALU per 32 binary MACs), i.e.
// 1 LDS instruction per 32 VALU where the 64 x 64 kernel above needs 1 per 4.  Row stride 20 words: the 16 rows a wave's B read
// touches fall on 16 disjoint 4-bank groups, 16-byte aligned.  The next chunk's global loads are in flight during the popcounts.
constexpr int XB = 128, XBK = 16, XBS = 20;
__global__ __launch_bounds__(256) void xnor_gemm128_kernel(const uint32_t* __restrict__ A, const uint32_t* __restrict__ B,
                                                           float* __restrict__ y, int M, int N, int KW, int Kbits, float scale) {
    __shared__ __attribute__((aligned(16))) uint32_t As[XB * XBS];
    __shared__ __attribute__((aligned(16))) uint32_t Bs[XB * XBS];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * XB, n0 = blockIdx.x * XB;
    // staging: thread t moves 16-byte pieces (row = t / 4 + 64 * h, words 4 * (t % 4) ..) of both operands
    const int sr = threadIdx.x >> 2, sc = (threadIdx.x & 3) * 4;
    const uint32_t* ap[2];
    const uint32_t* bp[2];
    bool av[2], bv[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int ra = m0 + sr + 64 * h, rb = n0 + sr + 64 * h;
        av[h] = ra < M;
        bv[h] = rb < N;
        ap[h] = A + (long)(av[h] ? ra : 0) * KW + sc;
        bp[h] = B + (long)(bv[h] ? rb : 0) * KW + sc;
    }
    uint4_t ra4[2], rb4[2];
    auto gload = [&](int k0) {  // KW % 4 == 0 on this path: a 16-byte piece is all in range or all out
        const bool kin = k0 + sc < KW;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            ra4[h] = (av[h] && kin) ? *reinterpret_cast<const uint4_t*>(ap[h] + k0) : uint4_t{0u, 0u, 0u, 0u};
            rb4[h] = (bv[h] && kin) ? *reinterpret_cast<const uint4_t*>(bp[h] + k0) : uint4_t{0u, 0u, 0u, 0u};
        }
    };
    int acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = 0;
    gload(0);
    for (int k0 = 0; k0 < KW; k0 += XBK) {
        __syncthreads();  // the previous chunk's readers are done
#pragma unroll
        for (int h = 0; h < 2; h++) {
            *reinterpret_cast<uint4_t*>(As + (sr + 64 * h) * XBS + sc) = ra4[h];
            *reinterpret_cast<uint4_t*>(Bs + (sr + 64 * h) * XBS + sc) = rb4[h];
        }
        __syncthreads();
        if (k0 + XBK < KW) gload(k0 + XBK);
#pragma unroll
        for (int c = 0; c < XBK; c += 4) {
            uint4_t a[8], b[8];
#pragma unroll
            for (int i = 0; i < 8; i++) a[i] = *reinterpret_cast<const uint4_t*>(As + (ty + 16 * i) * XBS + c);
#pragma unroll
            for (int j = 0; j < 8; j++) b[j] = *reinterpret_cast<const uint4_t*>(Bs + (tx + 16 * j) * XBS + c);
#pragma unroll
            for (int i = 0; i < 8; i++)
#pragma unroll
                for (int j = 0; j < 8; j++)
                {   // v_bcnt_u32_b32 D = popcount(S0) + S1: written out, or the compiler sums the four popcounts first (+25 % v_add)
                    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i][j]) : "v"(a[i].x ^ b[j].x));
                    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i][j]) : "v"(a[i].y ^ b[j].y));
                    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i][j]) : "v"(a[i].z ^ b[j].z));
                    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i][j]) : "v"(a[i].w ^ b[j].w));
                }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int m = m0 + ty + 16 * i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int n = n0 + tx + 16 * j;
            if (n < N) y[(long)m * N + n] = (float)(Kbits - 2 * acc[i][j]) * scale;
        }
    }
}

// Skinny-M variant (decode): one wave per output column n, lanes stride over the K words (coalesced),
// x rows cached in registers, wave reduction.  M <= 4 per launch.
template <int MT>
__global__ __launch_bounds__(256) void xnor_gemv_kernel(const uint32_t* __restrict__ X, const uint32_t* __restrict__ Wt,
                                                        float* __restrict__ y, int M, int N, int KW, int Kbits, float scale) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    int acc[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m] = 0;
    for (int k = lane; k < KW; k += 64) {
        const uint32_t w = Wt[(long)n * KW + k];
#pragma unroll
        for (int m = 0; m < MT; m++)
            if (m < M) acc[m] += __builtin_popcount(w ^ X[(long)m * KW + k]);
    }
#pragma unroll
    for (int m = 0; m < MT; m++) {
        const int v = wave_sum_dpp(acc[m]);
        if (lane == 0 && m < M) y[(long)m * N + n] = (float)(Kbits - 2 * v) * scale;
    }
}

// ---- one launch per layer forward (M <= 64) ---------------------------------------------------------------------
// y[m][n] = dt( dt( dt(K - 2*popc) * scale_a ) * scale_w ),  bits of x taken from (x + bias_a) >= 0:
// BinaryLinearCuda.forward = set_activation (x + bias_a) -> binary_linear_cuda.forward(...).to(dtype) -> * scale_a * scale_w
// (reference layers/qlinear/binary/cuda/layer.py:58-63, 283-284), i.e. what the reference spends an add, a pack, a GEMM, a cast
// and two multiplies on.  grid = (column blocks, row blocks); a workgroup sign-packs ITS x rows itself -- a thread loads 8
// consecutive values (16 bytes) and stores one byte of bits to LDS, no cross-lane traffic -- and then sweeps its columns:
//   ROWS == 4 (M <= 4, one row block): a wave takes 4 output columns at a time, lanes stride the K words (coalesced), DPP wave
//     reduction;
//   ROWS == 8 (row blocks of 8): lane = (row, column sub-index): 8 columns per wave step, every lane walks the K words of its
//     column (the 8 lanes of a column share the address) against its own packed x row in LDS (row stride KW + 1 words:
//     conflict-free), no reduction at all.
// Every workgroup of a row block repeats that block's packing (L2 reads), so the column blocks are kept wide: the packing is
// 32 / columns-per-workgroup of the XNOR work.
// The sign of fl(x + b) in the layer dtype equals the sign of the fp32 sum: a floating-point sum of two finite numbers never
// underflows to zero, so only exact cancellation gives 0 (>= 0 either way).
template <int DT>
__device__ __forceinline__ void load8(const void* p, long i, float (&v)[8]) {
    if constexpr (DT == BIE_F32) {
        const float4_t a = *reinterpret_cast<const float4_t*>((const float*)p + i);
        const float4_t b = *reinterpret_cast<const float4_t*>((const float*)p + i + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        const uint4_t a = *reinterpret_cast<const uint4_t*>((const uint16_t*)p + i);
        const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if constexpr (DT == BIE_BF16) {
                v[2 * q] = __uint_as_float(w[q] << 16);
                v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u);
            } else {
                v[2 * q] = f16_bits_to_f32(w[q] & 0xffffu);
                v[2 * q + 1] = f16_bits_to_f32(w[q] >> 16);
            }
        }
    }
}

template <int DT, int ROWS>
__global__ __launch_bounds__(256) void xnor_fused_kernel(const void* __restrict__ x, const void* __restrict__ bias_a,
                                                         const uint32_t* __restrict__ Wt, const void* __restrict__ scale_a,
                                                         const void* __restrict__ scale_w, void* __restrict__ y, int M, int N,
                                                         int K, int cols_per_wg, int y_f32) {
    extern __shared__ uint32_t xb[];  // [ROWS][KW + 1]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KW = K >> 5, XS = KW + 1, KB = K >> 3;
    const int m0 = blockIdx.y * ROWS;
    uint8_t* xbytes = reinterpret_cast<uint8_t*>(xb);
#pragma unroll 4
    for (int t = threadIdx.x; t < ROWS * KB; t += 256) {
        const int r = t / KB, kb = t - r * KB;
        uint32_t bits = 0;
        if (m0 + r < M) {
            float v[8];
            load8<DT>(x, (long)(m0 + r) * K + kb * 8, v);
            if (bias_a) {
                float b[8];
                load8<DT>(bias_a, kb * 8, b);
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] += b[q];
            }
#pragma unroll
            for (int q = 0; q < 8; q++) bits |= (uint32_t)(v[q] >= 0.0f) << q;
        }
        xbytes[r * XS * 4 + kb] = (uint8_t)bits;
    }
    __syncthreads();
    const float sa = scale_a ? dt_traits<DT>::load(scale_a, 0) : 1.0f;
    const float sw = scale_w ? dt_traits<DT>::load(scale_w, 0) : 1.0f;
    auto put = [&](int m, int n, int pc) {
        float v = (float)(K - 2 * pc);
        if (y_f32) {  // the extension-level forward: raw fp32 counts (binary_linear_cuda.forward returns float)
            ((float*)y)[(long)m * N + n] = v;
            return;
        }
        v = dt_traits<DT>::round(v);
        if (scale_a) v = dt_traits<DT>::round(v * sa);
        if (scale_w) v = dt_traits<DT>::round(v * sw);
        dt_traits<DT>::store(y, (long)m * N + n, v);
    };
    const int col0 = blockIdx.x * cols_per_wg;
    const int col1 = min(N, col0 + cols_per_wg);
    if constexpr (ROWS == 4) {
        for (int nb = col0 + wave * 4; nb < col1; nb += 16) {  // 4 columns at a time: their loads are in flight together
            int acc[4][4] = {};
            const uint32_t* wr[4];
#pragma unroll
            for (int c = 0; c < 4; c++) wr[c] = Wt + (long)min(nb + c, N - 1) * KW;
            for (int k = lane; k < KW; k += 64) {
                uint32_t w[4];
#pragma unroll
                for (int c = 0; c < 4; c++) w[c] = wr[c][k];
#pragma unroll
                for (int m = 0; m < 4; m++) {  // rows >= M hold zeros: harmless
                    const uint32_t xv = xb[m * XS + k];
#pragma unroll
                    for (int c = 0; c < 4; c++) acc[c][m] += __builtin_popcount(w[c] ^ xv);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; c++)
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    if (m >= M) break;
                    const int v = wave_sum_dpp(acc[c][m]);
                    if (lane == 0 && nb + c < col1) put(m, nb + c, v);
                }
        }
    } else {
        const int r = lane & 7, cs = lane >> 3;
        const uint32_t* xr = xb + r * XS;
        for (int nb = col0 + wave * 8; nb < col1; nb += 32) {
            const int n = nb + cs;
            const bool live = n < col1 && m0 + r < M;
            const uint32_t* wr = Wt + (long)(n < N ? n : N - 1) * KW;
            int pc = 0;
            int k = 0;
            if ((KW & 3) == 0)
                for (; k < KW; k += 4) {
                    const uint4_t w = *reinterpret_cast<const uint4_t*>(wr + k);
                    pc += __builtin_popcount(w.x ^ xr[k]) + __builtin_popcount(w.y ^ xr[k + 1]) +
                          __builtin_popcount(w.z ^ xr[k + 2]) + __builtin_popcount(w.w ^ xr[k + 3]);
                }
            for (; k < KW; k++) pc += __builtin_popcount(wr[k] ^ xr[k]);
            if (live) put(m0 + r, n, pc);
        }
    }
}

// ---- mid M (5 <= M, K % 512 == 0): RT rows x (4 waves x 16*G columns) per workgroup, full K per wave --------------------------
// The M <= 64 kernel above re-packs its 8 x rows in every one of its 128 column blocks (2.25x the XNOR work at 4096 x 4096) and the
// 64 x 64 LDS tile kernel reads one LDS word per 4 popcounts; both sit near 0.1 of the v_xor + v_bcnt rate at M = 64.  Here a lane is
// (column c = lane >> 2, k quarter kq = lane & 3): a 16-byte load per lane covers 64 contiguous bytes of each of 16 weight rows (the
// lane-per-column form would touch 64 cache lines per instruction), a chunk = 16 k-words, PD chunks in flight per wave ahead of the
// one being counted.  x bits (packed by the workgroup itself from the layer-dtype x, or copied when they arrive packed) are read
// from LDS with 4 distinct addresses per instruction (one per kq): 1 LDS read per 8 * G popcount pairs.  The four kq partial counts
// are summed on the DPP quad network; lane kq then stores rows kq (and kq + 4).  No reduction across waves or workgroups.
template <int DT, int RT, int G>  // DT = -1: x is already sign-packed ([M, K/8] bytes)
__global__ __launch_bounds__(256) void xnor_mid_kernel(const void* __restrict__ x, const void* __restrict__ bias_a,
                                                       const uint32_t* __restrict__ Wt, const void* __restrict__ scale_a,
                                                       const void* __restrict__ scale_w, void* __restrict__ y, int M, int N, int K,
                                                       float scale, int y_f32) {
    constexpr int PD = G >= 4 ? 4 : 8;  // chunks in flight: 16 (G = 4, 2) or 8 loads per lane
    extern __shared__ uint32_t xb[];  // [RT][KW]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KW = K >> 5, KB = K >> 3, T = KW >> 4;
    const int m0 = blockIdx.y * RT;
    const int c = lane >> 2, kq = lane & 3;
    const int nw0 = (blockIdx.x * 4 + wave) * (16 * G);
    const uint32_t* wrow[G];
#pragma unroll
    for (int g = 0; g < G; g++) wrow[g] = Wt + (long)min(nw0 + g * 16 + c, N - 1) * KW + 4 * kq;
    uint4_t wr[PD][G];
#pragma unroll
    for (int p = 0; p < PD; p++)
#pragma unroll
        for (int g = 0; g < G; g++) wr[p][g] = __builtin_nontemporal_load(reinterpret_cast<const uint4_t*>(wrow[g] + 16 * min(p, T - 1)));
    if constexpr (DT >= 0) {
        uint8_t* xbytes = reinterpret_cast<uint8_t*>(xb);
        for (int kb = threadIdx.x; kb < KB; kb += 256) {  // a thread packs bytes kb of all RT rows: the bias is loaded once
            float b[8];
            if (bias_a) load8<DT>(bias_a, kb * 8, b);
            float v[RT][8];
#pragma unroll
            for (int r = 0; r < RT; r++) load8<DT>(x, (long)min(m0 + r, M - 1) * K + kb * 8, v[r]);
#pragma unroll
            for (int r = 0; r < RT; r++) {
                uint32_t bits = 0;
#pragma unroll
                for (int q = 0; q < 8; q++) bits |= (uint32_t)((bias_a ? v[r][q] + b[q] : v[r][q]) >= 0.0f) << q;
                xbytes[r * KB + kb] = (uint8_t)(m0 + r < M ? bits : 0u);
            }
        }
    } else {
        const uint32_t* X = reinterpret_cast<const uint32_t*>(x);
        for (int t = threadIdx.x; t < RT * KW; t += 256) {
            const int r = t / KW, kw = t - r * KW;
            xb[t] = (m0 + r < M) ? X[(long)(m0 + r) * KW + kw] : 0u;
        }
    }
    __syncthreads();
    int acc[G][RT];
#pragma unroll
    for (int g = 0; g < G; g++)
#pragma unroll
        for (int m = 0; m < RT; m++) acc[g][m] = 0;
    const uint32_t* xq = xb + 4 * kq;
    for (int t0 = 0; t0 < T; t0 += PD) {
#pragma unroll
        for (int p = 0; p < PD; p++) {
            const int t = t0 + p;
            if (t >= T) break;
            uint4_t xv[RT];
#pragma unroll
            for (int m = 0; m < RT; m++) xv[m] = *reinterpret_cast<const uint4_t*>(xq + m * KW + 16 * t);
            uint4_t w[G];
#pragma unroll
            for (int g = 0; g < G; g++) w[g] = wr[p][g];
#pragma unroll
            for (int g = 0; g < G; g++) wr[p][g] = __builtin_nontemporal_load(reinterpret_cast<const uint4_t*>(wrow[g] + 16 * min(t + PD, T - 1)));
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int m = 0; m < RT; m++)
                    acc[g][m] += __builtin_popcount(w[g].x ^ xv[m].x) + __builtin_popcount(w[g].y ^ xv[m].y) +
                                 __builtin_popcount(w[g].z ^ xv[m].z) + __builtin_popcount(w[g].w ^ xv[m].w);
        }
    }
    float sa = 1.0f, sw = 1.0f;
    if constexpr (DT >= 0) {
        if (scale_a) sa = dt_traits<DT>::load(scale_a, 0);
        if (scale_w) sw = dt_traits<DT>::load(scale_w, 0);
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int n = nw0 + g * 16 + c;
        const bool b0 = (kq & 1) != 0, b1 = (kq & 2) != 0;
#pragma unroll
        for (int h = 0; h < RT / 4; h++) {
            // 4 rows x 4 kq lanes: transpose-reduce on the quad network -- lane kq ends with the full count of row 4h + kq
            const int a0 = acc[g][4 * h], a1 = acc[g][4 * h + 1], a2 = acc[g][4 * h + 2], a3 = acc[g][4 * h + 3];
            const int r01 = (b0 ? a1 : a0) + __builtin_amdgcn_update_dpp(0, b0 ? a0 : a1, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
            const int r23 = (b0 ? a3 : a2) + __builtin_amdgcn_update_dpp(0, b0 ? a2 : a3, 0xB1, 0xf, 0xf, false);
            const int pc = (b1 ? r23 : r01) + __builtin_amdgcn_update_dpp(0, b1 ? r01 : r23, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
            const int m = m0 + 4 * h + kq;
            if (m >= M || n >= N) continue;
            float v = (float)(K - 2 * pc);
            if constexpr (DT < 0) {
                ((float*)y)[(long)m * N + n] = v * scale;
            } else {
                if (y_f32) {
                    ((float*)y)[(long)m * N + n] = v;
                } else {
                    v = dt_traits<DT>::round(v);
                    if (scale_a) v = dt_traits<DT>::round(v * sa);
                    if (scale_w) v = dt_traits<DT>::round(v * sw);
                    dt_traits<DT>::store(y, (long)m * N + n, v);
                }
            }
        }
    }
}

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}
struct MidPlan { int rt, g; };
// widest tile whose grid still gives every CU a workgroup; BIE_BINARY_MID=<rt><g> (e.g. 84) pins one for tools/
static MidPlan mid_plan(long M, long N) {
    static const int forced = env_int("BIE_BINARY_MID", 0);
    if (forced == 84 || forced == 44 || forced == 42 || forced == 41) return MidPlan{forced / 10, forced % 10};
    // (8 rows with 2 or 1 column groups measured slower than 4 x 4 at equal grid size: profiles/r03_u_binary_mid_ab.txt)
    const int cand[4][2] = {{8, 4}, {4, 4}, {4, 2}, {4, 1}};
    for (int i = 0; i < 4; i++) {
        const long wgs = cdivl(N, 64 * cand[i][1]) * cdivl(M, cand[i][0]);
        if (wgs >= (i == 0 ? 1024 : 256)) return MidPlan{cand[i][0], cand[i][1]};
    }
    return MidPlan{4, 1};
}
// upper M: beyond it the 128 x 128 register-tile kernel (weights re-read once per 128 rows, not once per RT) wins -- measured crossover
static long mid_max_rows() {
    static const int v = env_int("BIE_BINARY_MID_MAX", 512);
    return v;
}
bool binary_mid_ok(long M, long N, long K) {
    return M >= 5 && M <= mid_max_rows() && K % 512 == 0 && K <= (1L << 18) && N >= 1 && N < (1L << 31);
}

template <int DT>
static void mid_launch_dt(const void* x, const void* bias_a, const uint32_t* wp, const void* sa, const void* sw, void* y, int M, int N, int K,
                          float scale, int y_f32, hipStream_t st) {
    const MidPlan p = mid_plan(M, N);
    const size_t lds = (size_t)p.rt * (K / 32) * 4;
    dim3 grid((unsigned)cdivl(N, 64 * p.g), (unsigned)cdivl(M, p.rt));
#define BIE_MID(RT_, G_) hipLaunchKernelGGL((xnor_mid_kernel<DT, RT_, G_>), grid, dim3(256), lds, st, x, bias_a, wp, sa, sw, y, M, N, K, scale, y_f32)
    if (p.rt == 8) BIE_MID(8, 4);
    else if (p.g == 4) BIE_MID(4, 4);
    else if (p.g == 2) BIE_MID(4, 2);
    else BIE_MID(4, 1);
#undef BIE_MID
}

// byte-granular compatibility kernel: any K % 8 == 0, either weight layout.  One thread per output.
__global__ __launch_bounds__(256) void xnor_bytes_kernel(const uint8_t* __restrict__ X, const uint8_t* __restrict__ W,
                                                         float* __restrict__ y, long M, long N, long KB, int w_layout,
                                                         float scale) {
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    const long m = blockIdx.y;
    if (n >= N) return;
    int pc = 0;
    for (long b = 0; b < KB; b++) {
        const uint8_t wv = w_layout ? W[b * N + n] : W[n * KB + b];
        pc += __builtin_popcount((unsigned)(X[m * KB + b] ^ wv));
    }
    y[m * N + n] = (float)(KB * 8 - 2 * pc) * scale;
}

// column bit-plane weights (the binary_linear_cpp layout: byte [b*N + n] holds k = 8b..8b+7 of output n): a lane owns 4
// consecutive columns (one dword per plane), the 16 waves of a block interleave the K/8 planes, x bytes are wave-uniform
// (scalar loads); one block row per x row.
__global__ __launch_bounds__(1024) void xnor_planes_kernel(const uint8_t* __restrict__ X, const uint8_t* __restrict__ W,
                                                           float* __restrict__ y, long M, long N, long KB, float scale) {
    __shared__ int red[16][64][4];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long m = blockIdx.y;
    const long n4 = ((long)blockIdx.x * 64 + lane) * 4;
    const long nl = n4 < N ? n4 : N - 4;
    int acc[4] = {0, 0, 0, 0};
    const uint8_t* xr = X + m * KB;
    for (long b = wave; b < KB; b += 16) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(W + b * N + nl);
        const uint32_t v = w ^ ((uint32_t)xr[b] * 0x01010101u);
        acc[0] += __builtin_popcount(v & 0xffu);
        acc[1] += __builtin_popcount(v & 0xff00u);
        acc[2] += __builtin_popcount(v & 0xff0000u);
        acc[3] += __builtin_popcount(v & 0xff000000u);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) red[wave][lane][j] = acc[j];
    __syncthreads();
    if (wave == 0 && n4 < N) {
        float4_t o;
        int t[4] = {0, 0, 0, 0};
        for (int wv = 0; wv < 16; wv++)
#pragma unroll
            for (int j = 0; j < 4; j++) t[j] += red[wv][lane][j];
        o.x = (float)(KB * 8 - 2 * t[0]) * scale;
        o.y = (float)(KB * 8 - 2 * t[1]) * scale;
        o.z = (float)(KB * 8 - 2 * t[2]) * scale;
        o.w = (float)(KB * 8 - 2 * t[3]) * scale;
        *reinterpret_cast<float4_t*>(y + m * N + n4) = o;
    }
}

// ---- conv2d: bit-im2col (pad -> bit 0 == -1) into 32-bit words, zero-padded to a word multiple -----------------
// cols[b][p][KW] with p = oh*OW + ow and k = c*ks*ks + i*ks + j
template <int DT>
__global__ __launch_bounds__(256) void im2col_bits_kernel(const void* __restrict__ x, uint32_t* __restrict__ cols, int B, int C,
                                                          int H, int W, int OH, int OW, int ks, int stride, int pad, int dil,
                                                          int Kc, int KW) {
    const long total = (long)B * OH * OW * KW;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int kw = (int)(idx % KW);
    const long bp = idx / KW;
    const int p = (int)(bp % (OH * OW));
    const int b = (int)(bp / (OH * OW));
    const int oh = p / OW, ow = p % OW;
    uint32_t v = 0;
    for (int j = 0; j < 32; j++) {
        const int k = kw * 32 + j;
        if (k >= Kc) break;
        const int wj = k % ks, hi_ = (k / ks) % ks, c = k / (ks * ks);
        const int h_im = oh * stride - pad + hi_ * dil;
        const int w_im = ow * stride - pad + wj * dil;
        if (h_im >= 0 && w_im >= 0 && h_im < H && w_im < W)
            v |= (uint32_t)sign_bit<DT>(x, (((long)b * C + c) * H + h_im) * W + w_im) << j;
    }
    cols[idx] = v;
}

// row-packed bytes [rows][KB] -> words [rows][KW] zero-padded
__global__ __launch_bounds__(256) void bytes_to_words_kernel(const uint8_t* __restrict__ in, uint32_t* __restrict__ out, long rows,
                                                             int KB, int KW) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * KW) return;
    const long r = idx / KW;
    const int kw = (int)(idx % KW);
    uint32_t v = 0;
    for (int j = 0; j < 4; j++) {
        const int bidx = kw * 4 + j;
        if (bidx < KB) v |= (uint32_t)in[r * KB + bidx] << (8 * j);
    }
    out[idx] = v;
}

// ---- functions/cuda helpers -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unpack_u8_scaled_kernel(const uint8_t* __restrict__ in, const float* __restrict__ scale,
                                                               float* __restrict__ out, long n, long packed_dim) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = in[i];
    const float sc = scale[i / packed_dim];
    float4_t lo, hi;
    lo.x = (w & 1u) ? sc : -sc; lo.y = (w & 2u) ? sc : -sc; lo.z = (w & 4u) ? sc : -sc; lo.w = (w & 8u) ? sc : -sc;
    hi.x = (w & 16u) ? sc : -sc; hi.y = (w & 32u) ? sc : -sc; hi.z = (w & 64u) ? sc : -sc; hi.w = (w & 128u) ? sc : -sc;
    float4_t* o = reinterpret_cast<float4_t*>(out + i * 8);
    o[0] = lo;
    o[1] = hi;
}

__global__ __launch_bounds__(256) void q4_pack_kernel(const int32_t* __restrict__ in, int8_t* __restrict__ out, long n_out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    out[i] = (int8_t)(((in[2 * i] & 0xF) << 4) | (in[2 * i + 1] & 0xF));
}

__global__ __launch_bounds__(256) void q4_unpack_kernel(const int8_t* __restrict__ in, int32_t* __restrict__ out, long n_in) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_in) return;
    const uint32_t v = (uint8_t)in[i];
    out[2 * i] = (int32_t)(v >> 4);
    out[2 * i + 1] = (int32_t)(v & 0xF);
}

__global__ __launch_bounds__(256) void q4_unpack_scale_kernel(const int8_t* __restrict__ in, float* __restrict__ out, long n_in,
                                                              float scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_in) return;
    const int v = (uint8_t)in[i];
    int hi = v >> 4, lo = v & 0xF;
    if (hi > 7) hi -= 16;
    if (lo > 7) lo -= 16;
    out[2 * i] = (float)hi * scale;
    out[2 * i + 1] = (float)lo * scale;
}

// ---- the reference CUDA layer's packed-weight IMAGES (checkpoint compatibility) ----------------------------------
// binary_linear_cuda.w_pack (reference binary_linear_cuda_kernel.cu:830-882) stores, for the transposed weight A[k][n], 32-bit
// words holding 32 consecutive k of one column n MSB-first (__brev(__ballot) / Bval << 1), serialised big-endian
// (uint32_to_uint8 :33-41): byte (k % 32) / 8 of the word, bit 7 - k % 8.  Only the ORDER OF THE WORDS differs:
//   BTC32  (BMMA_toBit32Col_new :59-152, grid (K/128, N/8), block (32, 4, 8)): word ((n/8 * K/128 + k/128) * 8 + n%8) * 4 + (k%128)/32
//   BSTC32 (ToBit32RowUd :186-300, grid (K/32, N/32)):                          word (k/32) * N + n
// Against our row-packed LSB-first bytes [N][K/8] an image byte is therefore one source byte with its bits reversed.
__device__ __forceinline__ long image_byte_index(int layout, long n, long kb, long N, long K) {
    const long k = kb * 8;
    long word;
    if (layout == 0) word = ((n / 8 * (K / 128) + k / 128) * 8 + n % 8) * 4 + (k % 128) / 32;
    else word = (k / 32) * N + n;
    return word * 4 + (k % 32) / 8;
}

// TO_IMAGE: src = values [N][K] (DT) -> dst = image bytes;  otherwise src = image bytes -> dst = row-packed [N][K/8]
template <int DT, bool TO_IMAGE>
__global__ __launch_bounds__(256) void binary_image_kernel(const void* __restrict__ src, uint8_t* __restrict__ dst, long N, long K, int layout) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long KB = K / 8;
    if (idx >= N * KB) return;
    const long n = idx / KB, kb = idx - n * KB;
    const long img = image_byte_index(layout, n, kb, N, K);
    if constexpr (TO_IMAGE) {
        uint32_t v = 0;
#pragma unroll
        for (int t = 0; t < 8; t++) v |= (uint32_t)(sign_bit<DT>(src, n * K + kb * 8 + t)) << (7 - t);
        dst[img] = (uint8_t)v;
    } else {
        dst[idx] = (uint8_t)(__brev((uint32_t)reinterpret_cast<const uint8_t*>(src)[img]) >> 24);
    }
}

// ---- launchers -------------------------------------------------------------------------------------------------
#define BIE_DT_SWITCH(dtype, CALL)                 \
    switch (dtype) {                               \
        case BIE_F16: { constexpr int DT = BIE_F16; CALL; } break;   \
        case BIE_BF16: { constexpr int DT = BIE_BF16; CALL; } break; \
        case BIE_F32: { constexpr int DT = BIE_F32; CALL; } break;   \
        default: { constexpr int DT = 3; CALL; } break;              \
    }

int pack_rows_launch(const void* a, uint8_t* out, long n_bytes, int dtype, hipStream_t st) {
    dim3 grid((unsigned)cdivl(n_bytes, 256));
    BIE_DT_SWITCH(dtype, hipLaunchKernelGGL(pack_rows_kernel<DT>, grid, dim3(256), 0, st, a, out, n_bytes));
    return check_launch("pack_rows_kernel");
}

int binary_image_pack_launch(const void* w, uint8_t* image, long N, long K, int layout, int dtype, hipStream_t st) {
    dim3 grid((unsigned)cdivl(N * (K / 8), 256));
    BIE_DT_SWITCH(dtype, hipLaunchKernelGGL((binary_image_kernel<DT, true>), grid, dim3(256), 0, st, w, image, N, K, layout));
    return check_launch("binary_image_kernel<pack>");
}

int binary_image_unpack_launch(const uint8_t* image, uint8_t* rowpacked, long N, long K, int layout, hipStream_t st) {
    dim3 grid((unsigned)cdivl(N * (K / 8), 256));
    hipLaunchKernelGGL((binary_image_kernel<3, false>), grid, dim3(256), 0, st, image, rowpacked, N, K, layout);
    return check_launch("binary_image_kernel<unpack>");
}

int pack_cols_launch(const void* w, uint8_t* out, long N, long K, int dtype, hipStream_t st) {
    dim3 grid((unsigned)cdivl(N, 256), (unsigned)(K / 8));
    BIE_DT_SWITCH(dtype, hipLaunchKernelGGL(pack_cols_kernel<DT>, grid, dim3(256), 0, st, w, out, N, K));
    return check_launch("pack_cols_kernel");
}

int binary_linear_launch(const uint8_t* xp, const uint8_t* wp, float* y, long M, long N, long K, int w_layout, float scale,
                         hipStream_t st) {
    const bool words_ok = (K % 32 == 0) && w_layout == 0 && (((uintptr_t)xp | (uintptr_t)wp) & 3) == 0;
    if (w_layout == 1 && N % 4 == 0 && N >= 4 && ((uintptr_t)wp & 3) == 0 && ((uintptr_t)y & 15) == 0 && M <= 65535) {
        dim3 grid((unsigned)cdivl(N, 256), (unsigned)M);
        hipLaunchKernelGGL(xnor_planes_kernel, grid, dim3(1024), 0, st, xp, wp, y, M, N, K / 8, scale);
        return check_launch("xnor_planes_kernel");
    }
    if (!words_ok) {
        dim3 grid((unsigned)cdivl(N, 256), (unsigned)M);
        hipLaunchKernelGGL(xnor_bytes_kernel, grid, dim3(256), 0, st, xp, wp, y, M, N, K / 8, w_layout, scale);
        return check_launch("xnor_bytes_kernel");
    }
    const int KW = (int)(K / 32);
    if (M <= 4) {
        dim3 grid((unsigned)cdivl(N, 4));
        if (M == 1) hipLaunchKernelGGL(xnor_gemv_kernel<1>, grid, dim3(256), 0, st, (const uint32_t*)xp, (const uint32_t*)wp, y, (int)M, (int)N, KW, (int)K, scale);
        else if (M == 2) hipLaunchKernelGGL(xnor_gemv_kernel<2>, grid, dim3(256), 0, st, (const uint32_t*)xp, (const uint32_t*)wp, y, (int)M, (int)N, KW, (int)K, scale);
        else hipLaunchKernelGGL(xnor_gemv_kernel<4>, grid, dim3(256), 0, st, (const uint32_t*)xp, (const uint32_t*)wp, y, (int)M, (int)N, KW, (int)K, scale);
        return check_launch("xnor_gemv_kernel");
    }
    if (binary_mid_ok(M, N, K) && ((uintptr_t)wp & 15) == 0) {  // mid M: register tiles of RT rows x 16*G columns per wave, x bits in LDS
        mid_launch_dt<-1>(xp, nullptr, (const uint32_t*)wp, nullptr, nullptr, y, (int)M, (int)N, (int)K, scale, 1, st);
        return check_launch("xnor_mid_kernel");
    }
    if ((KW & 3) == 0 && cdivl(N, XB) * cdivl(M, XB) >= 256) {  // large M: 128 x 128 tiles, 8 x 8 outputs per thread
        dim3 grid((unsigned)cdivl(N, XB), (unsigned)cdivl(M, XB));
        hipLaunchKernelGGL(xnor_gemm128_kernel, grid, dim3(256), 0, st, (const uint32_t*)xp, (const uint32_t*)wp, y, (int)M, (int)N, KW, (int)K, scale);
        return check_launch("xnor_gemm128_kernel");
    }
    if (cdivl(N, XT) * cdivl(M, XT) < 192) {  // skinny M: 16-row tiles, 4x the blocks
        dim3 grid((unsigned)cdivl(N, XT), (unsigned)cdivl(M, 16), 1);
        hipLaunchKernelGGL(xnor_gemm_kernel<1>, grid, dim3(256), 0, st, (const uint32_t*)xp, (const uint32_t*)wp, y, (int)M, (int)N, KW,
                           (int)K, scale, 0L, 0L, 0L);
    } else {
        dim3 grid((unsigned)cdivl(N, XT), (unsigned)cdivl(M, XT), 1);
        hipLaunchKernelGGL(xnor_gemm_kernel<4>, grid, dim3(256), 0, st, (const uint32_t*)xp, (const uint32_t*)wp, y, (int)M, (int)N, KW,
                           (int)K, scale, 0L, 0L, 0L);
    }
    return check_launch("xnor_gemm_kernel");
}

// batch of independent XNOR GEMMs (BinaryMatMul: heads x batch): ONE launch, the batch index is blockIdx.z of the tiled kernel
int binary_matmul_batched_launch(const uint8_t* xp, const uint8_t* wp, float* y, long batch, long M, long N, long K, long stride_x,
                                 long stride_w, long stride_y, float scale, hipStream_t st) {
    const bool words_ok = (K % 32 == 0) && (((uintptr_t)xp | (uintptr_t)wp) & 3) == 0 && (stride_x % 4 == 0) && (stride_w % 4 == 0) && batch <= 65535;
    if (!words_ok || batch == 1) {
        for (long b = 0; b < batch; b++) {
            const int rc = binary_linear_launch(xp + b * stride_x, wp + b * stride_w, y + b * stride_y, M, N, K, 0, scale, st);
            if (rc) return rc;
        }
        return BIE_OK;
    }
    const int KW = (int)(K / 32);
    if (cdivl(N, XT) * cdivl(M, XT) * batch < 192) {
        dim3 grid((unsigned)cdivl(N, XT), (unsigned)cdivl(M, 16), (unsigned)batch);
        hipLaunchKernelGGL(xnor_gemm_kernel<1>, grid, dim3(256), 0, st, (const uint32_t*)xp, (const uint32_t*)wp, y, (int)M, (int)N, KW, (int)K,
                           scale, stride_x / 4, stride_w / 4, stride_y);
    } else {
        dim3 grid((unsigned)cdivl(N, XT), (unsigned)cdivl(M, XT), (unsigned)batch);
        hipLaunchKernelGGL(xnor_gemm_kernel<4>, grid, dim3(256), 0, st, (const uint32_t*)xp, (const uint32_t*)wp, y, (int)M, (int)N, KW, (int)K,
                           scale, stride_x / 4, stride_w / 4, stride_y);
    }
    return check_launch("xnor_gemm_kernel<batched>");
}

// 1 <= M <= 64 (beyond that packing x once and running the tiled XNOR GEMM is the better split), K % 32 == 0
bool binary_linear_fused_ok(long M, long N, long K) {
    // the packed x rows of one workgroup (4 or 8 rows of K/32 + 1 words) live in LDS: 64 KiB
    if (binary_mid_ok(M, N, K)) return true;  // 5 <= M <= 512, K % 512 == 0: xnor_mid_kernel packs its rows itself
    return M >= 1 && M <= 64 && N >= 1 && K >= 32 && K % 32 == 0 && (size_t)(M <= 4 ? 4 : 8) * (K / 32 + 1) * 4 <= 65536 && N < (1L << 31);
}

template <int DT>
static void fused_launch_dt(const void* x, const void* bias_a, const uint8_t* wp, const void* sa, const void* sw, void* y, int M, int N, int K, int y_f32, hipStream_t st) {
    if (binary_mid_ok(M, N, K)) {
        mid_launch_dt<DT>(x, bias_a, (const uint32_t*)wp, sa, sw, y, M, N, K, 1.0f, y_f32, st);
        return;
    }
    const int rows = M <= 4 ? 4 : 8;
    const size_t lds = (size_t)rows * (K / 32 + 1) * 4;
    const int row_blocks = M <= 4 ? 1 : (int)cdivl(M, 8);
    const int step = M <= 4 ? 16 : 32;                                  // columns one workgroup covers per sweep
    const int want = M <= 4 ? 256 : (row_blocks >= 4 ? 64 : 128);       // column blocks: wide enough that the repeated packing stays small
    int cols = (int)cdivl(cdivl(N, want), step) * step;
    if (cols < step) cols = step;
    dim3 grid((unsigned)cdivl(N, cols), (unsigned)row_blocks);
    if (M <= 4) hipLaunchKernelGGL((xnor_fused_kernel<DT, 4>), grid, dim3(256), lds, st, x, bias_a, (const uint32_t*)wp, sa, sw, y, M, N, K, cols, y_f32);
    else hipLaunchKernelGGL((xnor_fused_kernel<DT, 8>), grid, dim3(256), lds, st, x, bias_a, (const uint32_t*)wp, sa, sw, y, M, N, K, cols, y_f32);
}

int binary_linear_fused_launch(const void* x, const void* bias_a, const uint8_t* wp, const void* sa, const void* sw, void* y, long M,
                               long N, long K, int dtype, int y_f32, hipStream_t st) {
    if (dtype == BIE_F16) fused_launch_dt<BIE_F16>(x, bias_a, wp, sa, sw, y, (int)M, (int)N, (int)K, y_f32, st);
    else if (dtype == BIE_BF16) fused_launch_dt<BIE_BF16>(x, bias_a, wp, sa, sw, y, (int)M, (int)N, (int)K, y_f32, st);
    else fused_launch_dt<BIE_F32>(x, bias_a, wp, sa, sw, y, (int)M, (int)N, (int)K, y_f32, st);
    return check_launch("xnor_fused_kernel");
}

// ---- conv2d without the im2col image -------------------------------------------------------------------------------------------
// The flattened (c, i, j) bit order of the reference's packed weights interleaves the taps inside a word; with the weights
// re-laid ONCE as wtaps[oc][tap][C/32 words] (channel bits of one tap contiguous) and the activations sign-packed channel-minor,
// xbits[b][h][w][C/32 words], a tap of an output pixel is a plain XNOR-popcount over C/32 words and zero padding is a zero x word
// (bit 0 = -1, as the reference counts it; the channel padding bits are 0 on both sides and cancel).
//   conv_weight_taps_kernel : one-time re-layout of the packed weights (cached by the caller)
//   pack_nhwc_bits_kernel   : x [B, C, H, W] -> xbits (1/16 .. 1/32 of x's bytes; replaces the 9x larger bit-im2col image)
//   xnor_conv_taps_kernel   : a wave = (image, output row, 64 output channels): it copies the ks input rows it needs into a private
//                             LDS slab (+ one zero pixel for out-of-range columns), lane = output channel streams that channel's
//                             tap words with 16-byte loads, the x words are wave-uniform LDS reads (broadcast), 8 output pixels
//                             are accumulated at a time.  No workgroup barrier, no reduction.
__global__ __launch_bounds__(256) void conv_weight_taps_kernel(const uint8_t* __restrict__ wpacked, uint32_t* __restrict__ wtaps,
                                                               int OC, int C, int T, int CW) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)OC * T * CW) return;
    const int cw = (int)(idx % CW);
    const int t = (int)((idx / CW) % T);
    const long oc = idx / ((long)CW * T);
    const uint8_t* row = wpacked + oc * ((long)C * T / 8);
    uint32_t v = 0;
    for (int cc = 0; cc < 32; cc++) {
        const int c = cw * 32 + cc;
        if (c >= C) break;
        const int k = c * T + t;
        v |= (uint32_t)((row[k >> 3] >> (k & 7)) & 1u) << cc;
    }
    wtaps[idx] = v;
}

template <int DT>
__global__ __launch_bounds__(256) void pack_nhwc_bits_kernel(const void* __restrict__ x, uint32_t* __restrict__ xbits, int B, int C,
                                                             int HW, int CW) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // (b, cw, hw) with hw fastest: the 32 channel loads are coalesced along hw
    if (idx >= (long)B * CW * HW) return;
    const int hw = (int)(idx % HW);
    const int cw = (int)((idx / HW) % CW);
    const long b = idx / ((long)HW * CW);
    uint32_t v = 0;
    if (cw * 32 + 32 <= C) {  // whole word: the 32 channel loads are independent and all in flight
#pragma unroll
        for (int cc = 0; cc < 32; cc++) v |= (uint32_t)sign_bit<DT>(x, (b * C + cw * 32 + cc) * HW + hw) << cc;
    } else {
        for (int cc = 0; cw * 32 + cc < C; cc++) v |= (uint32_t)sign_bit<DT>(x, (b * C + cw * 32 + cc) * HW + hw) << cc;
    }
    xbits[(b * HW + hw) * CW + cw] = v;
}

constexpr int CONV_PX = 8;  // output pixels accumulated per pass
// CW4 > 0: C / 128 at compile time (16-byte weight / x words, no bounds branches); 0: any C, word by word.
// The tap loop is a RUNTIME loop on purpose: unrolled (compile-time kernel size), hipcc hoists the 36 weight loads out of the pixel
// loop and all 288 LDS reads above the first popcount -- 700-1500 spilled registers in every variant tried (register-resident
// weights, laundered pointers, volatile reads, a 128-register launch bound), 22 us per output row.  Inside one tap the 8 x CW4 LDS
// reads are independent and batch up; the next tap's weight words are requested before the current tap's popcounts.
template <int CW4>
__global__ __launch_bounds__(256) void xnor_conv_taps_kernel(const uint32_t* __restrict__ xbits, const uint32_t* __restrict__ wtaps,
                                                             float* __restrict__ y, int B, int C, int H, int W, int OC, int OH,
                                                             int OW, int ks, int stride, int pad, int dil, int cw_rt, float scale,
                                                             long items, int slab_words) {
    const int CW = CW4 > 0 ? 4 * CW4 : cw_rt;
    constexpr int NQ = CW4 > 0 ? CW4 : 1;
    extern __shared__ __attribute__((aligned(16))) uint32_t conv_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long item = (long)blockIdx.x * 4 + wave;  // (b, oh, oc block): oc block fastest, so neighbours share the same input rows in L2
    if (item >= items) return;
    const int ocbs = (OC + 63) >> 6;
    const int ocb = (int)(item % ocbs);
    const int oh = (int)((item / ocbs) % OH);
    const long b = item / ((long)ocbs * OH);
    const int T = ks * ks;
    const int oc = ocb * 64 + lane;
    const uint32_t* wl = wtaps + (long)(oc < OC ? oc : OC - 1) * T * CW;
    auto load_tap = [&](int t, uint4_t (&w)[NQ]) {
#pragma unroll
        for (int q = 0; q < NQ; q++) w[q] = *reinterpret_cast<const uint4_t*>(wl + t * CW + 4 * q);
    };
    uint4_t wc[NQ], wn[NQ];
    if constexpr (CW4 > 0) load_tap(0, wc);
    uint32_t* slab = conv_lds + wave * slab_words;  // [ks][W + 1][CW]; pixel W of every row is the zero pixel
    const int rowstride = (W + 1) * CW;
    for (int i = 0; i < ks; i++) {
        const int ih = oh * stride - pad + i * dil;
        const bool valid = ih >= 0 && ih < H;
        const uint32_t* src = xbits + ((b * H + (valid ? ih : 0)) * W) * CW;
#pragma unroll 2
        for (int t = lane; t < rowstride; t += 64) slab[i * rowstride + t] = (valid && t < W * CW) ? src[t] : 0u;
    }
    __builtin_amdgcn_wave_barrier();  // wave-private slab: a wave's LDS operations complete in order
    const int Kc = C * T;
    for (int ow0 = 0; ow0 < OW; ow0 += CONV_PX) {
        int acc[CONV_PX];
#pragma unroll
        for (int px = 0; px < CONV_PX; px++) acc[px] = 0;
        int i = 0, j = 0;
#pragma unroll 1
        for (int t = 0; t < T; t++) {
            int pix[CONV_PX];  // wave-uniform slab offsets of the 8 pixels' x words for this tap
#pragma unroll
            for (int px = 0; px < CONV_PX; px++) {
                const int iw = (ow0 + px) * stride - pad + j * dil;
                pix[px] = i * rowstride + ((iw >= 0 && iw < W && ow0 + px < OW) ? iw : W) * CW;
            }
            if constexpr (CW4 > 0) {
                load_tap(t + 1 < T ? t + 1 : 0, wn);  // the last one is tap 0 of the next pixel group
#pragma unroll
                for (int px = 0; px < CONV_PX; px++)
#pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        const uint4_t x4 = *reinterpret_cast<const uint4_t*>(slab + pix[px] + 4 * q);
                        acc[px] += __builtin_popcount(wc[q].x ^ x4.x) + __builtin_popcount(wc[q].y ^ x4.y) +
                                   __builtin_popcount(wc[q].z ^ x4.z) + __builtin_popcount(wc[q].w ^ x4.w);
                    }
#pragma unroll
                for (int q = 0; q < NQ; q++) wc[q] = wn[q];
            } else {
                const uint32_t* wr = wl + t * CW;
                for (int cw = 0; cw < CW; cw++) {
                    const uint32_t w1 = wr[cw];
#pragma unroll
                    for (int px = 0; px < CONV_PX; px++) acc[px] += __builtin_popcount(w1 ^ slab[pix[px] + cw]);
                }
            }
            if (++j == ks) {
                j = 0;
                i++;
            }
        }
        if (oc < OC) {
            float* yr = y + ((b * OC + oc) * OH + oh) * OW + ow0;
#pragma unroll
            for (int px = 0; px < CONV_PX; px++)
                if (ow0 + px < OW) yr[px] = (float)(Kc - 2 * acc[px]) * scale;
        }
    }
}

// The same pass with the lane's tap words staged in LDS by DMA: the 1-tap-ahead prefetch above still pays one global-load latency per
// tap (a lane's tap rows are private 16-byte pieces, 9 x ~1.5 us per output row on a 7x7 map).  Here a one-wave workgroup issues ALL
// T * CW4 loads as global_load_lds_dwordx4 up front (no registers; piece p lands at p * 1024 + lane * 16, conflict-free to read
// back), waits once, and runs the tap loop out of LDS.  T * CW4 * 1 KiB + the row slab per workgroup (37.5 KiB for 3x3x512).
template <int CW4, int PXC>  // PXC output pixels per workgroup: 2 when the problem is small (more, shorter waves), else 8
__global__ __launch_bounds__(64) void xnor_conv_taps_dma_kernel(const uint32_t* __restrict__ xbits, const uint32_t* __restrict__ wtaps,
                                                                float* __restrict__ y, int B, int C, int H, int W, int OC, int OH, int OW,
                                                                int ks, int stride, int pad, int dil, float scale) {
    constexpr int CW = 4 * CW4;
    extern __shared__ __attribute__((aligned(16))) uint32_t conv_lds[];
    const int lane = threadIdx.x;
    const int pchunks = (OW + PXC - 1) / PXC;
    const long item = blockIdx.x / pchunks;  // (b, oh, oc block), oc block fastest; pixel chunk innermost
    const int ow0 = (int)(blockIdx.x % pchunks) * PXC;
    const int ocbs = (OC + 63) >> 6;
    const int ocb = (int)(item % ocbs);
    const int oh = (int)((item / ocbs) % OH);
    const long b = item / ((long)ocbs * OH);
    const int T = ks * ks;
    const int oc = ocb * 64 + lane;
    const uint32_t* wl = wtaps + (long)(oc < OC ? oc : OC - 1) * T * CW;
    uint32_t* wlds = conv_lds;                       // [T * CW4][64 lanes][4 words]
    uint32_t* slab = conv_lds + T * CW4 * 256;       // [ks][W + 1][CW]
#if defined(__HIP_DEVICE_COMPILE__)
    {
        auto* dst = (__attribute__((address_space(3))) unsigned char*)wlds;
        for (int p = 0; p < T * CW4; p++) __builtin_amdgcn_global_load_lds(wl + 4 * p, dst + p * 1024, 16, 0, 0);
    }
#endif
    const int rowstride = (W + 1) * CW;
    for (int i = 0; i < ks; i++) {
        const int ih = oh * stride - pad + i * dil;
        const bool valid = ih >= 0 && ih < H;
        const uint32_t* src = xbits + ((b * H + (valid ? ih : 0)) * W) * CW;
#pragma unroll 2
        for (int t = lane; t < rowstride; t += 64) slab[i * rowstride + t] = (valid && t < W * CW) ? src[t] : 0u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA pieces have landed
    __builtin_amdgcn_wave_barrier();
    const int Kc = C * T;
    {
        int acc[PXC];
#pragma unroll
        for (int px = 0; px < PXC; px++) acc[px] = 0;
        int i = 0, j = 0;
#pragma unroll 1
        for (int t = 0; t < T; t++) {
            int pix[PXC];
#pragma unroll
            for (int px = 0; px < PXC; px++) {
                const int iw = (ow0 + px) * stride - pad + j * dil;
                pix[px] = i * rowstride + ((iw >= 0 && iw < W && ow0 + px < OW) ? iw : W) * CW;
            }
            uint4_t wq[CW4];
#pragma unroll
            for (int q = 0; q < CW4; q++) wq[q] = *reinterpret_cast<const uint4_t*>(wlds + (t * CW4 + q) * 256 + lane * 4);
#pragma unroll
            for (int px = 0; px < PXC; px++)
#pragma unroll
                for (int q = 0; q < CW4; q++) {
                    const uint4_t x4 = *reinterpret_cast<const uint4_t*>(slab + pix[px] + 4 * q);
                    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[px]) : "v"(wq[q].x ^ x4.x));
                    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[px]) : "v"(wq[q].y ^ x4.y));
                    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[px]) : "v"(wq[q].z ^ x4.z));
                    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[px]) : "v"(wq[q].w ^ x4.w));
                }
            if (++j == ks) {
                j = 0;
                i++;
            }
        }
        if (oc < OC) {
            float* yr = y + ((b * OC + oc) * OH + oh) * OW + ow0;
#pragma unroll
            for (int px = 0; px < PXC; px++)
                if (ow0 + px < OW) yr[px] = (float)(Kc - 2 * acc[px]) * scale;
        }
    }
}

// The tap pass in the lane mapping of xnor_mid_kernel: lane = (output channel c = lane >> 2 of 16, k quarter kq = lane & 3 of a
// 16-word chunk of the oc's [T][CW] tap words), G channel groups per wave.  A 16-byte load per lane covers 64 contiguous bytes of 16
// weight rows and feeds PXN pixels x 4 words; x words are LDS reads with 4 distinct addresses (one per kq): 1 LDS read per 8 * G
// popcounts where the lane-per-channel kernel above needs 1 per 8.  The PXN pixel counts of the four kq lanes are summed on the DPP
// quad network, lane kq stores pixels kq and 4 + kq.  CW = 4 * CW4; a chunk holds 4 / CW4 taps (the tap is wave-uniform when
// CW4 == 4, per lane otherwise); the tail chunk's unused words count zero (w = 0 against the zero pixel).  PXN = 7 for the 7 / 14 / 28
// wide maps of a ResNet (no eighth idle pixel), else 8.
template <int CW4, int G, int PXN>
__global__ __launch_bounds__(256) void xnor_conv_quad_kernel(const uint32_t* __restrict__ xbits, const uint32_t* __restrict__ wtaps,
                                                             float* __restrict__ y, int B, int C, int H, int W, int OC, int OH,
                                                             int OW, int ks, int stride, int pad, int dil, float scale,
                                                             long items, int slab_words) {
    constexpr int CW = 4 * CW4, TPC = 4 / CW4;  // taps per 16-word chunk
    extern __shared__ __attribute__((aligned(16))) uint32_t conv_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long item = (long)blockIdx.x * 4 + wave;  // (b, oh, oc block): oc block fastest
    if (item >= items) return;
    const int ocbs = (OC + 16 * G - 1) / (16 * G);
    const int ocb = (int)(item % ocbs);
    const int oh = (int)((item / ocbs) % OH);
    const long b = item / ((long)ocbs * OH);
    const int T = ks * ks, TW = T * CW, NCH = (TW + 15) >> 4;
    const int c = lane >> 2, kq = lane & 3;
    const int tq = kq / CW4, cwq = 4 * (kq % CW4);  // the lane's tap inside a chunk and its first channel word
    const uint32_t* wl[G];
#pragma unroll
    for (int g = 0; g < G; g++) wl[g] = wtaps + (long)min(ocb * 16 * G + g * 16 + c, OC - 1) * TW;
    (void)wl;
    auto load_chunk = [&](int ch, uint4_t (&w)[G]) {
        const int wi = min(16 * ch + 4 * kq, TW - 4);
#pragma unroll
        for (int g = 0; g < G; g++) w[g] = *reinterpret_cast<const uint4_t*>(wl[g] + wi);
    };
    uint4_t wc[G], wn[G];
    load_chunk(0, wc);
    uint32_t* slab = conv_lds + wave * slab_words;  // [ks][W + 1][CW]; pixel W of every row is the zero pixel
    const int rowstride = (W + 1) * CW;
    for (int i = 0; i < ks; i++) {
        const int ih = oh * stride - pad + i * dil;
        const bool valid = ih >= 0 && ih < H;
        const uint32_t* src = xbits + ((b * H + (valid ? ih : 0)) * W) * CW;
#pragma unroll 2
        for (int t = lane; t < rowstride; t += 64) slab[i * rowstride + t] = (valid && t < W * CW) ? src[t] : 0u;
    }
    __builtin_amdgcn_wave_barrier();  // wave-private slab: a wave's LDS operations complete in order
    const int Kc = C * T;
    const int zero_px = W * CW + cwq;
    for (int ow0 = 0; ow0 < OW; ow0 += PXN) {
        int acc[G][8];
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int px = 0; px < 8; px++) acc[g][px] = 0;
        int ti = 0, tj = tq;  // (kernel row, kernel column) of the lane's tap; tq < TPC <= 4 may already exceed ks - 1
        while (tj >= ks) { tj -= ks; ti++; }
#pragma unroll 1
        for (int ch = 0; ch < NCH; ch++) {
            load_chunk(ch + 1 < NCH ? ch + 1 : 0, wn);  // the last one is chunk 0 of the next pixel group
            const bool live = ch * TPC + tq < T;
            if (!live) {
#pragma unroll
                for (int g = 0; g < G; g++) wc[g] = uint4_t{0u, 0u, 0u, 0u};
            }
            const int base = ti * rowstride + cwq, jd = tj * dil - pad + ow0 * stride;
            uint4_t xv[PXN];
#pragma unroll
            for (int px = 0; px < PXN; px++) {
                const int iw = px * stride + jd;
                const bool in = (unsigned)iw < (unsigned)W && ow0 + px < OW && live;
                xv[px] = *reinterpret_cast<const uint4_t*>(slab + (in ? base + iw * CW : zero_px));
            }
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int px = 0; px < PXN; px++) {
                    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[g][px]) : "v"(wc[g].x ^ xv[px].x));
                    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[g][px]) : "v"(wc[g].y ^ xv[px].y));
                    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[g][px]) : "v"(wc[g].z ^ xv[px].z));
                    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[g][px]) : "v"(wc[g].w ^ xv[px].w));
                }
#pragma unroll
            for (int g = 0; g < G; g++) wc[g] = wn[g];
            tj += TPC;
            while (tj >= ks) { tj -= ks; ti++; }
        }
        const bool b0 = (kq & 1) != 0, b1 = (kq & 2) != 0;
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int oc = ocb * 16 * G + g * 16 + c;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int a0 = acc[g][4 * h], a1 = acc[g][4 * h + 1], a2 = acc[g][4 * h + 2], a3 = acc[g][4 * h + 3];
                const int r01 = (b0 ? a1 : a0) + __builtin_amdgcn_update_dpp(0, b0 ? a0 : a1, 0xB1, 0xf, 0xf, false);
                const int r23 = (b0 ? a3 : a2) + __builtin_amdgcn_update_dpp(0, b0 ? a2 : a3, 0xB1, 0xf, 0xf, false);
                const int pc = (b1 ? r23 : r01) + __builtin_amdgcn_update_dpp(0, b1 ? r01 : r23, 0x4E, 0xf, 0xf, false);
                const int pxi = 4 * h + kq;
                if (oc < OC && pxi < PXN && ow0 + pxi < OW) y[((b * OC + oc) * OH + oh) * OW + ow0 + pxi] = (float)(Kc - 2 * pc) * scale;
            }
        }
    }
}

// LDS of one workgroup (4 waves) of the implicit conv; 0 = geometry outside its range (rows too long for a 16 KiB slab)
size_t binary_conv_taps_lds_bytes(int C, int W, int ks) {
    const size_t slab = (size_t)ks * (W + 1) * cdiv(C, 32);
    const size_t words = (slab + 3) / 4 * 4;
    return words * 4 <= 16384 ? words * 4 * 4 : 0;
}

int binary_pack_nhwc_bits_launch(const void* x, uint32_t* xbits, int B, int C, int HW, int dtype, hipStream_t st) {
    const int CW = cdiv(C, 32);
    dim3 grid((unsigned)cdivl((long)B * CW * HW, 256));
    BIE_DT_SWITCH(dtype, hipLaunchKernelGGL(pack_nhwc_bits_kernel<DT>, grid, dim3(256), 0, st, x, xbits, B, C, HW, CW));
    return check_launch("pack_nhwc_bits_kernel");
}

size_t binary_conv_taps_workspace_bytes(int B, int C, int H, int W) { return (size_t)B * H * W * cdiv(C, 32) * 4; }

int binary_conv_weight_taps_launch(const uint8_t* wpacked, uint32_t* wtaps, int OC, int C, int ks, hipStream_t st) {
    const int T = ks * ks, CW = cdiv(C, 32);
    const long total = (long)OC * T * CW;
    hipLaunchKernelGGL(conv_weight_taps_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, wpacked, wtaps, OC, C, T, CW);
    return check_launch("conv_weight_taps_kernel");
}

int binary_conv_taps_launch(const void* x, const uint32_t* wtaps, float* y, void* ws, int B, int C, int H, int W, int OC, int ks,
                            int stride, int pad, int dil, float scale, int dtype, hipStream_t st) {
    const int OH = (H + 2 * pad - dil * (ks - 1) - 1) / stride + 1;
    const int OW = (W + 2 * pad - dil * (ks - 1) - 1) / stride + 1;
    const int CW = cdiv(C, 32);
    uint32_t* xbits = reinterpret_cast<uint32_t*>(ws);
    {
        const long total = (long)B * CW * H * W;
        dim3 grid((unsigned)cdivl(total, 256));
        BIE_DT_SWITCH(dtype, hipLaunchKernelGGL(pack_nhwc_bits_kernel<DT>, grid, dim3(256), 0, st, x, xbits, B, C, H * W, CW));
        int rc = check_launch("pack_nhwc_bits_kernel");
        if (rc) return rc;
    }
    const size_t lds = binary_conv_taps_lds_bytes(C, W, ks);
    const long items = (long)B * OH * cdiv(OC, 64);
    {   // one-wave workgroups with the tap words DMA-staged in LDS, when they fit next to the row slab in 64 KiB
        const size_t wbytes = (size_t)ks * ks * CW * 256, sbytes = (size_t)ks * (W + 1) * CW * 4;
        static const bool dma_on = [] { const char* e = getenv("BIE_CONV_DMA"); return !e || atoi(e) != 0; }();
        if (dma_on && (CW == 16 || CW == 8 || CW == 4) && wbytes + sbytes <= 65536 && items < 1024) {
            // small problems only (measured 7x7x512 3x3: B = 1 15.3 -> see profiles; at B = 32 the 4-wave kernel below is faster, 24.9 against 33 us)
            const size_t l2 = wbytes + ((sbytes + 15) & ~(size_t)15);
            const long grid = items * cdiv(OW, 2);
#define LD(C4V) hipLaunchKernelGGL((xnor_conv_taps_dma_kernel<C4V, 2>), dim3((unsigned)grid), dim3(64), l2, st, xbits, wtaps, y, B, C, H, W, OC, OH, OW, ks, stride, pad, dil, scale)
            if (CW == 16) LD(4);
            else if (CW == 8) LD(2);
            else LD(1);
#undef LD
            return check_launch("xnor_conv_taps_dma_kernel");
        }
    }
    static const bool quad_on = [] { const char* e = getenv("BIE_CONV_QUAD"); return !e || atoi(e) != 0; }();
    if (quad_on && (CW == 16 || CW == 8 || CW == 4) && ks * ks * CW >= 16) {  // lane = (channel, k quarter): 64-byte pieces of 16 weight rows per load
        const int g = (long)B * OH * cdiv(OC, 64) >= 1024 ? 4 : ((long)B * OH * cdiv(OC, 32) >= 1024 ? 2 : 1);
        const long qitems = (long)B * OH * cdiv(OC, 16 * g);
        dim3 grid((unsigned)cdivl(qitems, 4));
#define LQ3(CV, GV, PV) hipLaunchKernelGGL((xnor_conv_quad_kernel<CV, GV, PV>), grid, dim3(256), lds, st, xbits, wtaps, y, B, C, H, W, OC, OH, OW, ks, \
                                           stride, pad, dil, scale, qitems, (int)(lds / 16))
#define LQ2(CV, GV) do { if (OW % 7 == 0) LQ3(CV, GV, 7); else LQ3(CV, GV, 8); } while (0)
#define LQ(CV) do { if (g == 4) LQ2(CV, 4); else if (g == 2) LQ2(CV, 2); else LQ2(CV, 1); } while (0)
        if (CW == 16) LQ(4);
        else if (CW == 8) LQ(2);
        else LQ(1);
#undef LQ
#undef LQ2
#undef LQ3
        return check_launch("xnor_conv_quad_kernel");
    }
#define LC(C4V) hipLaunchKernelGGL(xnor_conv_taps_kernel<C4V>, dim3((unsigned)cdivl(items, 4)), dim3(256), lds, st, xbits, wtaps, y, B, C, H, W, OC, OH, OW, \
                                   ks, stride, pad, dil, CW, scale, items, (int)(lds / 16))
    if (CW == 16) LC(4);  // 512 / 256 / 128 channels: the deep layers of a binary ResNet
    else if (CW == 8) LC(2);
    else if (CW == 4) LC(1);
    else LC(0);
#undef LC
    return check_launch("xnor_conv_taps_kernel");
}

size_t binary_conv_workspace_bytes(int B, int C, int H, int W, int OC, int ks, int stride, int pad, int dil) {
    const int OH = (H + 2 * pad - dil * (ks - 1) - 1) / stride + 1;
    const int OW = (W + 2 * pad - dil * (ks - 1) - 1) / stride + 1;
    const int Kc = C * ks * ks;
    const int KW = cdiv(Kc, 32);
    return ((size_t)B * OH * OW * KW + (size_t)OC * KW) * 4;
}

int binary_conv_launch(const void* x, const uint8_t* wpacked, float* y, void* ws, int B, int C, int H, int W, int OC, int ks,
                       int stride, int pad, int dil, float scale, int dtype, hipStream_t st) {
    const int OH = (H + 2 * pad - dil * (ks - 1) - 1) / stride + 1;
    const int OW = (W + 2 * pad - dil * (ks - 1) - 1) / stride + 1;
    const int P = OH * OW;
    const int Kc = C * ks * ks;
    const int KW = cdiv(Kc, 32);
    uint32_t* wwords = reinterpret_cast<uint32_t*>(ws);
    uint32_t* cols = wwords + (size_t)OC * KW;
    {
        const long total = (long)OC * KW;
        hipLaunchKernelGGL(bytes_to_words_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, wpacked, wwords, (long)OC,
                           Kc / 8, KW);
        int rc = check_launch("bytes_to_words_kernel");
        if (rc) return rc;
    }
    {
        const long total = (long)B * P * KW;
        dim3 grid((unsigned)cdivl(total, 256));
        BIE_DT_SWITCH(dtype, hipLaunchKernelGGL(im2col_bits_kernel<DT>, grid, dim3(256), 0, st, x, cols, B, C, H, W, OH, OW, ks, stride,
                                                pad, dil, Kc, KW));
        int rc = check_launch("im2col_bits_kernel");
        if (rc) return rc;
    }
    // y[b][oc][p] = (Kc - 2 popc(w[oc] ^ cols[b][p])) * scale : A = weights (shared), B = cols of image b
    if ((long)cdiv(P, XT) * cdiv(OC, XT) * B < 192) {  // few tiles (7x7 maps, small batch): 16-row tiles give 4x the blocks
        dim3 grid((unsigned)cdiv(P, XT), (unsigned)cdiv(OC, 16), (unsigned)B);
        hipLaunchKernelGGL(xnor_gemm_kernel<1>, grid, dim3(256), 0, st, wwords, cols, y, OC, P, KW, Kc, scale, 0L, (long)P * KW,
                           (long)OC * P);
    } else {
        dim3 grid((unsigned)cdiv(P, XT), (unsigned)cdiv(OC, XT), (unsigned)B);
        hipLaunchKernelGGL(xnor_gemm_kernel<4>, grid, dim3(256), 0, st, wwords, cols, y, OC, P, KW, Kc, scale, 0L, (long)P * KW,
                           (long)OC * P);
    }
    return check_launch("xnor_gemm_kernel(conv)");
}

int pack_sign_launch(const void* a, uint8_t* out, long n_bytes, int dtype, hipStream_t st) {
    return pack_rows_launch(a, out, n_bytes, dtype, st);
}

int unpack_u8_scaled_launch(const uint8_t* in, const float* scale, float* out, long n, long packed_dim, hipStream_t st) {
    hipLaunchKernelGGL(unpack_u8_scaled_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, st, in, scale, out, n, packed_dim);
    return check_launch("unpack_u8_scaled_kernel");
}
int q4_pack_launch(const int32_t* in, int8_t* out, long n_out, hipStream_t st) {
    hipLaunchKernelGGL(q4_pack_kernel, dim3((unsigned)cdivl(n_out, 256)), dim3(256), 0, st, in, out, n_out);
    return check_launch("q4_pack_kernel");
}
int q4_unpack_launch(const int8_t* in, int32_t* out, long n_in, hipStream_t st) {
    hipLaunchKernelGGL(q4_unpack_kernel, dim3((unsigned)cdivl(n_in, 256)), dim3(256), 0, st, in, out, n_in);
    return check_launch("q4_unpack_kernel");
}
int q4_unpack_scale_launch(const int8_t* in, float* out, long n_in, float scale, hipStream_t st) {
    hipLaunchKernelGGL(q4_unpack_scale_kernel, dim3((unsigned)cdivl(n_in, 256)), dim3(256), 0, st, in, out, n_in, scale);
    return check_launch("q4_unpack_scale_kernel");
}

}  // namespace bie
