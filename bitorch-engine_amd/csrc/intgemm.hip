// W4A4 / W8A8 integer GEMM on the i8 matrix cores (v_mfma_i32_32x32x32_i8) -- replaces the reference's CUTLASS
// int4b_t / int8 tensor-op GEMMs (layers/qlinear/nbit/cutlass/q4_linear_cutlass_kernel.cu:203-280,520-560,
// q8_linear_cutlass_kernel.cu:44-215) and the quantise+nibble-pack kernel (q4_linear_cutlass_kernel.cu:74-170).
//   y[m][n] = epilogue( sum_k a[m][k] * w[n][k] )      a: [M, K], w: [N, K], both K-contiguous ("NT")
// gfx950 has no int4 MFMA: packed nibbles are expanded to i8 in registers.  A byte holds (v_even << 4 | v_odd),
// values in [-8, 7]; masking with 0xF0 keeps v * 16 as a *signed* i8 -- no per-nibble sign extension needed; both
// operands carry the factor 16, the exact int32 sum is shifted right by 8 in the epilogue.  The even/odd
// de-interleave permutes k identically on both operands, which the contraction does not see.
#include "bie_common.h"

namespace bie {

typedef int int4_t __attribute__((ext_vector_type(4)));
typedef int int16v_t __attribute__((ext_vector_type(16)));

constexpr int IG_BM = 128, IG_BN = 128, IG_BK = 64;  // BK in values

// 16-byte slot (row, s) of a [rows][64-byte] tile; slots XOR-swizzled so that 16 rows at a fixed s are conflict free
__device__ __forceinline__ int ig_off64(int row, int s) { return row * 64 + ((s ^ ((row >> 2) & 3)) << 4); }
// 8-byte slot (row, s) of a [rows][32-byte] tile (packed int4)
__device__ __forceinline__ int ig_off32(int row, int s) { return row * 32 + ((s ^ ((row >> 3) & 3)) << 3); }

__device__ __forceinline__ int4_t expand_q4(uint2_t p) {
    int4_t r;
    r.x = (int)(p.x & 0xF0F0F0F0u);
    r.y = (int)((p.x << 4) & 0xF0F0F0F0u);
    r.z = (int)(p.y & 0xF0F0F0F0u);
    r.w = (int)((p.y << 4) & 0xF0F0F0F0u);
    return r;
}

typedef int int4v_t __attribute__((ext_vector_type(4)));

// MODE 0: W4A4, out dtype DT: o = fl(fl((float)acc) * fl(scale))            (q4_linear_cutlass_forward :640-680)
// MODE 1: W8A8, out fp32     : o = ((float)acc * scale_a) * scale_w         (q8_linear_cutlass_forward :215-230)
// MODE 2 / 3: W4A4 / W8A8 with the raw int32 accumulators as output (q4_gemm :526-555 / q8_gemm: what the reference's
//             backward entry points return or scale on the torch side)
template <int MODE, int DT>
__global__ __launch_bounds__(256) void int_gemm_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ W,
                                                       void* __restrict__ y, int M, int N, int K, float scale_a, float scale_w,
                                                       long strideA, long strideW, long strideY) {
    constexpr bool Q4 = (MODE == 0 || MODE == 2);
    constexpr int ROWB = Q4 ? 32 : 64;  // bytes per tile row
    __shared__ __attribute__((aligned(16))) unsigned char As[IG_BM * ROWB];
    __shared__ __attribute__((aligned(16))) unsigned char Ws[IG_BN * ROWB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // 2 x 2 waves, wave tile 64 x 64
    const int h = lane >> 5, j = lane & 31;
    const int m0 = blockIdx.y * IG_BM, n0 = blockIdx.x * IG_BN;
    A += (long)blockIdx.z * strideA;
    W += (long)blockIdx.z * strideW;
    const long rowbytes = Q4 ? K / 2 : K;

    int16v_t acc[2][2];  // [n fragment][m fragment]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[a][b][e] = 0;

    for (int k0 = 0; k0 < K; k0 += IG_BK) {
        const long kb = Q4 ? k0 / 2 : k0;
        if constexpr (Q4) {  // 128 rows x 32 B = 512 x 8 B chunks per operand: 2 per thread
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int c = tid + i * 256, row = c >> 2, s = c & 3;
                int ra = m0 + row; if (ra > M - 1) ra = M - 1;
                int rw = n0 + row; if (rw > N - 1) rw = N - 1;
                *reinterpret_cast<uint2_t*>(As + ig_off32(row, s)) = *reinterpret_cast<const uint2_t*>(A + ra * rowbytes + kb + s * 8);
                *reinterpret_cast<uint2_t*>(Ws + ig_off32(row, s)) = *reinterpret_cast<const uint2_t*>(W + rw * rowbytes + kb + s * 8);
            }
        } else {  // 128 rows x 64 B = 512 x 16 B chunks per operand: 2 per thread
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int c = tid + i * 256, row = c >> 2, s = c & 3;
                int ra = m0 + row; if (ra > M - 1) ra = M - 1;
                int rw = n0 + row; if (rw > N - 1) rw = N - 1;
                *reinterpret_cast<uint4_t*>(As + ig_off64(row, s)) = *reinterpret_cast<const uint4_t*>(A + ra * rowbytes + kb + s * 16);
                *reinterpret_cast<uint4_t*>(Ws + ig_off64(row, s)) = *reinterpret_cast<const uint4_t*>(W + rw * rowbytes + kb + s * 16);
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {  // two k32 steps per tile; lane half h takes 16 of the 32 values
            int4_t af[2], wf[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int ra = wm * 64 + t * 32 + j, rw = wn * 64 + t * 32 + j;
                if constexpr (Q4) {
                    af[t] = expand_q4(*reinterpret_cast<const uint2_t*>(As + ig_off32(ra, kk * 2 + h)));
                    wf[t] = expand_q4(*reinterpret_cast<const uint2_t*>(Ws + ig_off32(rw, kk * 2 + h)));
                } else {
                    af[t] = *reinterpret_cast<const int4_t*>(As + ig_off64(ra, kk * 2 + h));
                    wf[t] = *reinterpret_cast<const int4_t*>(Ws + ig_off64(rw, kk * 2 + h));
                }
            }
            // D = W_frag (A operand, rows = n) x a_frag (B operand, cols = m): a lane ends up with one output row m and
            // 4 consecutive columns n per register group (vector stores in the epilogue)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[a], af[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }

    char* yb = reinterpret_cast<char*>(y) + (long)blockIdx.z * strideY * (MODE >= 1 ? 4 : dt_traits<DT>::bytes);
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int row = m0 + wm * 64 + b * 32 + j;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int nq = n0 + wn * 64 + a * 32 + 8 * q + 4 * h;
                if (row < M && nq < N) {
                    if constexpr (MODE >= 2) {
                        int4v_t iv;
#pragma unroll
                        for (int c = 0; c < 4; c++) iv[c] = Q4 ? (acc[a][b][4 * q + c] >> 8) : acc[a][b][4 * q + c];
                        *reinterpret_cast<int4v_t*>(reinterpret_cast<int*>(yb) + (long)row * N + nq) = iv;
                        continue;
                    }
                    float o[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        int v = acc[a][b][4 * q + c];
                        if constexpr (Q4) v >>= 8;  // both operands carried a factor 16
                        if constexpr (MODE == 0) o[c] = dt_traits<DT>::round(dt_traits<DT>::round((float)v) * dt_traits<DT>::round(scale_a * scale_w));
                        else o[c] = ((float)v * scale_a) * scale_w;
                    }
                    if constexpr (MODE == 1 || DT == BIE_F32) {
                        *reinterpret_cast<float4_t*>(reinterpret_cast<float*>(yb) + (long)row * N + nq) = float4_t{o[0], o[1], o[2], o[3]};
                    } else {
                        uint2_t pk;
                        if constexpr (DT == BIE_F16) {
                            pk.x = f32_to_f16_bits(o[0]) | (f32_to_f16_bits(o[1]) << 16);
                            pk.y = f32_to_f16_bits(o[2]) | (f32_to_f16_bits(o[3]) << 16);
                        } else {
                            pk.x = pack_bf16x2(o[0], o[1]);
                            pk.y = pack_bf16x2(o[2], o[3]);
                        }
                        *reinterpret_cast<uint2_t*>(reinterpret_cast<uint16_t*>(yb) + (long)row * N + nq) = pk;
                    }
                }
            }
        }
}

// q = clamp(roundf(x / max(scale, 1e-5)), -8, 7) (half away from zero, like the CUDA kernels); two values per byte,
// FIRST value in the high nibble.  For fp16/bf16 the division is rounded to the storage type first (__hdiv).
template <int DT>
__global__ __launch_bounds__(256) void q4_quantize_pack_kernel(const void* __restrict__ x, int8_t* __restrict__ out, long n_out,
                                                               float scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_out) return;
    const float s = fmaxf(dt_traits<DT>::round(scale), dt_traits<DT>::round(0.00001f));
    int q[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const float v = dt_traits<DT>::round(dt_traits<DT>::load(x, 2 * i + c) / s);
        q[c] = (int)fminf(fmaxf(roundf(v), -8.0f), 7.0f) & 0xF;
    }
    out[i] = (int8_t)((q[0] << 4) | q[1]);
}

// NHWC nibble-packed activations [B, H, W, C/2] -> im2col rows [B*OH*OW, Kp/2], k = (kh*KS + kw)*C + c (c fastest, the order
// of the [OC, KS, KS, C/2] packed filter), zero nibbles for padding taps and for the tail up to Kp (multiple of 64 values).
// One thread per 4 bytes (8 values); C % 8 == 0.
__global__ __launch_bounds__(256) void q4_im2col_kernel(const uint32_t* __restrict__ a, uint32_t* __restrict__ col, int B, int H, int W,
                                                        int C, int OH, int OW, int KS, int stride, int pad, int dil, int Kp) {
    const int wpr = Kp / 8;  // dwords per im2col row
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * OH * OW * wpr;
    if (idx >= total) return;
    const int kw8 = (int)(idx % wpr);
    const long row = idx / wpr;
    const int ow = (int)(row % OW), oh = (int)((row / OW) % OH), b = (int)(row / ((long)OW * OH));
    const int k = kw8 * 8;
    uint32_t v = 0;
    if (k < KS * KS * C) {
        const int tap = k / C, c = k % C;
        const int ih = oh * stride - pad + (tap / KS) * dil, iw = ow * stride - pad + (tap % KS) * dil;
        if (ih >= 0 && ih < H && iw >= 0 && iw < W) v = a[(((long)b * H + ih) * W + iw) * (C / 8) + c / 8];
    }
    col[idx] = v;
}

int q4_conv2d_launch(const int8_t* a_packed, const int8_t* w_packed, void* y, void* workspace, int B, int H, int W, int C, int OC,
                     int KS, int stride, int pad, int dil, float sa, float sw, int dtype, hipStream_t st);

// intgemm_pipe.hip: W8A8 on the ordered-asm pipeline (256 x 256 / 128 x 128 tiles, LDS-DMA of both operands), bit-identical results
bool i8_pipe_ok(int M, int N, int K, const void* A, const void* W, const void* y);
int i8_pipe_launch(bool out_i32, const void* A, const void* W, void* y, int M, int N, int K, float sa, float sw, hipStream_t st);
bool i4_pipe_ok(int M, int N, int K, const void* A, const void* W, const void* y);
int i4_pipe_launch(int mode, const void* A, const void* W, void* y, int M, int N, int K, float sa, float sw, int dtype, hipStream_t st);

int int_gemm_launch(int mode, const void* A, const void* W, void* y, int M, int N, int K, float sa, float sw, int dtype, int batch,
                    long strideA, long strideW, long strideY, hipStream_t st) {
    if ((mode == 1 || mode == 3) && batch == 1 && i8_pipe_ok(M, N, K, A, W, y)) return i8_pipe_launch(mode == 3, A, W, y, M, N, K, sa, sw, st);
    if ((mode == 0 || mode == 2) && batch == 1 && i4_pipe_ok(M, N, K, A, W, y)) return i4_pipe_launch(mode, A, W, y, M, N, K, sa, sw, dtype, st);
    dim3 grid(cdiv(N, IG_BN), cdiv(M, IG_BM), batch);
#define L(MODE, DT) hipLaunchKernelGGL((int_gemm_kernel<MODE, DT>), grid, dim3(256), 0, st, (const uint8_t*)A, (const uint8_t*)W, y, M, N, K, sa, sw, strideA, strideW, strideY)
    if (mode == 1) L(1, BIE_F32);
    else if (mode == 2) L(2, BIE_F32);
    else if (mode == 3) L(3, BIE_F32);
    else if (dtype == BIE_F16) L(0, BIE_F16);
    else if (dtype == BIE_BF16) L(0, BIE_BF16);
    else L(0, BIE_F32);
#undef L
    return check_launch("int_gemm_kernel");
}

size_t q4_conv2d_workspace_bytes(int B, int H, int W, int C, int OC, int KS, int stride, int pad, int dil) {
    const int OH = (H + 2 * pad - dil * (KS - 1) - 1) / stride + 1, OW = (W + 2 * pad - dil * (KS - 1) - 1) / stride + 1;
    if (OH <= 0 || OW <= 0) return 0;
    const int K = KS * KS * C, Kp = cdiv(K, 64) * 64;
    size_t bytes = (size_t)B * OH * OW * (Kp / 2);
    if (Kp != K) bytes += (size_t)OC * (Kp / 2);  // zero-padded copy of the filter
    return (bytes + 255) / 256 * 256;
}

__global__ __launch_bounds__(256) void q4_pad_rows_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, long rows, int wi, int wo) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * wo) return;
    const int c = (int)(idx % wo);
    out[idx] = c < wi ? in[(idx / wo) * wi + c] : 0u;
}

int q4_conv2d_launch(const int8_t* a_packed, const int8_t* w_packed, void* y, void* workspace, int B, int H, int W, int C, int OC,
                     int KS, int stride, int pad, int dil, float sa, float sw, int dtype, hipStream_t st) {
    const int OH = (H + 2 * pad - dil * (KS - 1) - 1) / stride + 1, OW = (W + 2 * pad - dil * (KS - 1) - 1) / stride + 1;
    const int K = KS * KS * C, Kp = cdiv(K, 64) * 64;
    const long rows = (long)B * OH * OW;
    uint32_t* col = reinterpret_cast<uint32_t*>(workspace);
    const long total = rows * (Kp / 8);
    hipLaunchKernelGGL(q4_im2col_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, (const uint32_t*)a_packed, col, B, H, W, C, OH, OW,
                       KS, stride, pad, dil, Kp);
    int rc = check_launch("q4_im2col_kernel");
    if (rc) return rc;
    const int8_t* wp = w_packed;
    if (Kp != K) {
        uint32_t* wpad = col + total;
        hipLaunchKernelGGL(q4_pad_rows_kernel, dim3((unsigned)cdivl((long)OC * (Kp / 8), 256)), dim3(256), 0, st, (const uint32_t*)w_packed, wpad,
                           (long)OC, K / 8, Kp / 8);
        rc = check_launch("q4_pad_rows_kernel");
        if (rc) return rc;
        wp = reinterpret_cast<const int8_t*>(wpad);
    }
    return int_gemm_launch(0, col, wp, y, (int)rows, OC, Kp, sa, sw, dtype, 1, 0, 0, 0, st);
}

int q4_quantize_pack_launch(const void* x, int8_t* out, long n_out, float scale, int dtype, hipStream_t st) {
    dim3 grid((unsigned)cdivl(n_out, 256));
    if (dtype == BIE_F16) hipLaunchKernelGGL(q4_quantize_pack_kernel<BIE_F16>, grid, dim3(256), 0, st, x, out, n_out, scale);
    else if (dtype == BIE_BF16) hipLaunchKernelGGL(q4_quantize_pack_kernel<BIE_BF16>, grid, dim3(256), 0, st, x, out, n_out, scale);
    else hipLaunchKernelGGL(q4_quantize_pack_kernel<BIE_F32>, grid, dim3(256), 0, st, x, out, n_out, scale);
    return check_launch("q4_quantize_pack_kernel");
}

}  // namespace bie
