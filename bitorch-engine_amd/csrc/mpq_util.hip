// Elementwise companions of the MPQ path (bandwidth-bound, coalesced along N):
//   mpq_dequant_kernel : packed int32 -> dense [K, N] weights, bit-exact twin of unpack_qweight
//                        (reference layers/qlinear/nbit/cuda/utils.py:30-51)
//   mpq_pack_kernel    : dense [K, N] -> packed int32, bit-exact twin of pack_fp_weight (utils.py:72-147)
//   mpq_grad_input     : grad_x = grad_y . W^T (back_quant_mm_kernel, mpq_linear_cuda_kernel.cu:635-1049)
//   mpq_sort_rows      : act-order preparation: packed rows re-ordered so that every group's k are consecutive
//   gather_cols        : x[:, perm] (the activation side of the same re-ordering)
#include "mpq_dequant.cuh"

#pragma clang fp contract(off)

namespace bie {

// Each thread handles one packed word = NB consecutive k of one column; lanes run along N so both the
// word loads and the NB row stores are coalesced.
template <int DT>
__global__ __launch_bounds__(256) void mpq_dequant_kernel(const uint32_t* __restrict__ qw, const void* __restrict__ scales,
                                                          const void* __restrict__ zeros, const int32_t* __restrict__ g_idx,
                                                          void* __restrict__ out, int K, int N, int w_bit, int group_size,
                                                          int asym) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (n >= N) return;
    const int nb = 32 / w_bit;
    const uint32_t mask = (w_bit == 32) ? 0xffffffffu : ((1u << w_bit) - 1u);
    const int zero_width = N / nb;
    const uint32_t word = qw[(long)r * N + n];
    int g_prev = -1;
    float s = 0.f, z = 0.f;
    int zq1 = 0;
    for (int j = 0; j < nb; j++) {
        const int k = r * nb + j;
        if (k >= K) break;
        const int g = g_idx ? g_idx[k] : (k / group_size);
        if (g != g_prev) {
            g_prev = g;
            s = dt_traits<DT>::load(scales, (long)g * N + n);
            if (asym) {
                const uint32_t zw = reinterpret_cast<const uint32_t*>(zeros)[(long)g * zero_width + n / nb];
                zq1 = (int)((zw >> ((n % nb) * w_bit)) & mask) + 1;
            } else {
                z = dt_traits<DT>::load(zeros, (long)g * N + n);
            }
        }
        const uint32_t q = (word >> (j * w_bit)) & mask;
        float w;
        if (asym) {
            w = s * (float)((int)q - zq1);  // rounded once by the store
        } else {
            w = dt_traits<DT>::round((float)q * s) - z;  // second rounding by the store
        }
        dt_traits<DT>::store(out, (long)k * N + n, w);
    }
}

template <int DT>
__global__ __launch_bounds__(256) void mpq_pack_kernel(const void* __restrict__ weight, const void* __restrict__ scales,
                                                       const void* __restrict__ zeros, const int32_t* __restrict__ g_idx,
                                                       uint32_t* __restrict__ out, int K, int N, int w_bit, int group_size,
                                                       int asym) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (n >= N) return;
    const int nb = 32 / w_bit;
    const uint32_t mask = (1u << w_bit) - 1u;
    const int zero_width = N / nb;
    uint32_t word = 0;
    for (int j = 0; j < nb; j++) {
        const int k = r * nb + j;
        if (k >= K) break;
        const int g = g_idx ? g_idx[k] : (k / group_size);
        const float w = dt_traits<DT>::load(weight, (long)k * N + n);
        const float s = dt_traits<DT>::load(scales, (long)g * N + n);
        float v;
        if (asym) {
            const uint32_t zw = reinterpret_cast<const uint32_t*>(zeros)[(long)g * zero_width + n / nb];
            const int zq1 = (int)((zw >> ((n % nb) * w_bit)) & mask) + 1;
            v = dt_traits<DT>::round(dt_traits<DT>::round(w / s) + (float)zq1);
        } else {
            const float z = dt_traits<DT>::load(zeros, (long)g * N + n);
            v = dt_traits<DT>::round(dt_traits<DT>::round(w + z) / s);
        }
        v = dt_traits<DT>::round(rintf(v));  // torch.round: half to even
        // NaN / inf -> clamp like torch's .to(int32).clamp: keep it defined
        float c = fminf(fmaxf(v, 0.0f), (float)mask);
        if (!(v == v)) c = 0.0f;
        word |= ((uint32_t)c) << (j * w_bit);
    }
    out[(long)r * N + n] = word;
}

// grad_x[m][k] = sum_n gy[m][n] * W[k][n].  One wave per (m, packed row): lanes stride over n (coalesced
// packed-word loads), NB accumulators per lane, wave reduction with DPP shuffles.  Backward is not on the
// inference hot path; this is the straightforward bandwidth-friendly form.
template <int DT, int NB>
__global__ __launch_bounds__(256) void mpq_grad_input_kernel(const void* __restrict__ gy, const uint32_t* __restrict__ qw,
                                                             const void* __restrict__ scales, const void* __restrict__ zeros,
                                                             const int32_t* __restrict__ g_idx, void* __restrict__ gx, int M,
                                                             int K, int N, int group_size, int asym) {
    constexpr int w_bit = 32 / NB;
    constexpr uint32_t mask = (w_bit == 32) ? 0xffffffffu : ((1u << w_bit) - 1u);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    const int m = blockIdx.y;
    if (r * NB >= K) return;
    const int zero_width = N / NB;
    float acc[NB];
    int gk[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) {
        acc[j] = 0.f;
        const int k = r * NB + j;
        gk[j] = g_idx ? g_idx[k < K ? k : K - 1] : (k / group_size);
    }
    for (int n = lane; n < N; n += 64) {
        const uint32_t word = qw[(long)r * N + n];
        const float g = dt_traits<DT>::load(gy, (long)m * N + n);
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const uint32_t q = (word >> (j * w_bit)) & mask;
            const float s = dt_traits<DT>::load(scales, (long)gk[j] * N + n);
            float w;
            if (asym) {
                const uint32_t zw = reinterpret_cast<const uint32_t*>(zeros)[(long)gk[j] * zero_width + n / NB];
                w = dequant_scalar_asym<DT>(q, s, (int)((zw >> ((n % NB) * w_bit)) & mask) + 1);
            } else {
                w = dequant_scalar_sym<DT>(q, s, dt_traits<DT>::load(zeros, (long)gk[j] * N + n));
            }
            acc[j] = __builtin_fmaf(w, g, acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < NB; j++) {
        float v = acc[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        const int k = r * NB + j;
        if (lane == 0 && k < K) dt_traits<DT>::store(gx, (long)m * K + k, v);
    }
}

// Act-order (explicit g_idx) layers: the reference looks the group of every k up per weight (mpq_linear_cuda_kernel.cu:300-317).
// W·x is invariant under a common permutation of k, so the packed matrix is re-ordered ONCE (perm = stable argsort of g_idx: group
// g's members become rows g·gs … g·gs+gs-1, i.e. an implicit-group matrix) and the decode / prefill kernels run unchanged on
// x[:, perm].  Pure bit movement: out field k' = in field perm[k'].  One thread per output word; lanes along N (coalesced).
template <int NB>
__global__ __launch_bounds__(256) void mpq_sort_rows_kernel(const uint32_t* __restrict__ qw, const int32_t* __restrict__ perm,
                                                            uint32_t* __restrict__ out, int K, int N) {
    constexpr int W = 32 / NB;
    constexpr uint32_t MASK = W == 32 ? 0xffffffffu : ((1u << W) - 1u);
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (n >= N) return;
    uint32_t word = 0;
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const int kp = r * NB + i;
        if (kp < K) {
            const int k = perm[kp];  // wave-uniform
            const uint32_t src = qw[(long)(k / NB) * N + n];
            word |= ((src >> ((k % NB) * W)) & MASK) << (i * W);
        }
    }
    out[(long)r * N + n] = word;
}

template <typename T>
__global__ __launch_bounds__(256) void gather_cols_kernel(const T* __restrict__ x, const int32_t* __restrict__ perm,
                                                          T* __restrict__ out, int M, int K) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const int src = perm[k];
    for (int m = blockIdx.y; m < M; m += gridDim.y) out[(long)m * K + k] = x[(long)m * K + src];
}

// Many rows (prefill): stage a whole row in LDS with 16-byte loads, gather from LDS, write 16 bytes per lane -- x and out are each
// streamed once (the per-element form above pulls a 64-byte line per two-byte read: +41 us at M = 4096, K = 4096 in round 4's bench row
// c2_act_order; this form moves the same 64 MB at the streaming rate).  Two-byte elements, K % 8 == 0, K <= 32768.
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint16_t* __restrict__ x, const int32_t* __restrict__ perm, uint16_t* __restrict__ out, int M, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gr[];
    uint16_t* row = reinterpret_cast<uint16_t*>(smem_gr);
    const int K8 = K >> 3;
    for (int m = blockIdx.x; m < M; m += gridDim.x) {
        const uint4_t* src = reinterpret_cast<const uint4_t*>(x + (long)m * K);
        for (int i = threadIdx.x; i < K8; i += 256) reinterpret_cast<uint4_t*>(row)[i] = src[i];
        __syncthreads();
        uint4_t* dst = reinterpret_cast<uint4_t*>(out + (long)m * K);
        for (int i = threadIdx.x; i < K8; i += 256) {
            const uint4_t p0 = reinterpret_cast<const uint4_t*>(perm)[2 * i], p1 = reinterpret_cast<const uint4_t*>(perm)[2 * i + 1];  // eight source columns
            uint4_t o;
            o.x = (uint32_t)row[p0.x] | ((uint32_t)row[p0.y] << 16);
            o.y = (uint32_t)row[p0.z] | ((uint32_t)row[p0.w] << 16);
            o.z = (uint32_t)row[p1.x] | ((uint32_t)row[p1.y] << 16);
            o.w = (uint32_t)row[p1.z] | ((uint32_t)row[p1.w] << 16);
            dst[i] = o;
        }
        __syncthreads();
    }
}

int mpq_sort_rows_launch(const int32_t* qw, const int32_t* perm, int32_t* out, int K, int N, int w_bit, hipStream_t st) {
    const int nb = 32 / w_bit;
    dim3 grid(cdiv(N, 256), cdiv(K, nb));
#define L(NBV) hipLaunchKernelGGL(mpq_sort_rows_kernel<NBV>, grid, dim3(256), 0, st, (const uint32_t*)qw, perm, (uint32_t*)out, K, N)
    switch (w_bit) {
        case 1: L(32); break;
        case 2: L(16); break;
        case 4: L(8); break;
        default: L(4); break;
    }
#undef L
    return check_launch("mpq_sort_rows_kernel");
}

int gather_cols_launch(const void* x, const int32_t* perm, void* out, int M, int K, int elem_bytes, hipStream_t st) {
    if (elem_bytes == 2 && M >= 16 && (K & 7) == 0 && K <= 32768 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(perm)) & 15) == 0) {
        hipLaunchKernelGGL(gather_rows_kernel, dim3(M < 2048 ? M : 2048), dim3(256), (size_t)K * 2, st, (const uint16_t*)x, perm, (uint16_t*)out, M, K);
        return check_launch("gather_rows_kernel");
    }
    dim3 grid(cdiv(K, 256), M < 2048 ? M : 2048);
    if (elem_bytes == 2) hipLaunchKernelGGL(gather_cols_kernel<uint16_t>, grid, dim3(256), 0, st, (const uint16_t*)x, perm, (uint16_t*)out, M, K);
    else hipLaunchKernelGGL(gather_cols_kernel<uint32_t>, grid, dim3(256), 0, st, (const uint32_t*)x, perm, (uint32_t*)out, M, K);
    return check_launch("gather_cols_kernel");
}

int mpq_dequant_launch(const int32_t* qw, const void* scales, const void* zeros, const int32_t* g_idx, void* out, int K,
                       int N, int w_bit, int group_size, int asym, int dtype, hipStream_t st) {
    const int nb = 32 / w_bit;
    dim3 grid(cdiv(N, 256), cdiv(K, nb));
#define L(DT) hipLaunchKernelGGL(mpq_dequant_kernel<DT>, grid, dim3(256), 0, st, (const uint32_t*)qw, scales, zeros, g_idx, out, K, N, w_bit, group_size, asym)
    if (dtype == BIE_F16) L(BIE_F16);
    else if (dtype == BIE_BF16) L(BIE_BF16);
    else L(BIE_F32);
#undef L
    return check_launch("mpq_dequant_kernel");
}

int mpq_pack_launch(const void* weight, const void* scales, const void* zeros, const int32_t* g_idx, int32_t* out, int K,
                    int N, int w_bit, int group_size, int asym, int dtype, hipStream_t st) {
    const int nb = 32 / w_bit;
    dim3 grid(cdiv(N, 256), cdiv(K, nb));
#define L(DT) hipLaunchKernelGGL(mpq_pack_kernel<DT>, grid, dim3(256), 0, st, weight, scales, zeros, g_idx, (uint32_t*)out, K, N, w_bit, group_size, asym)
    if (dtype == BIE_F16) L(BIE_F16);
    else if (dtype == BIE_BF16) L(BIE_BF16);
    else L(BIE_F32);
#undef L
    return check_launch("mpq_pack_kernel");
}

template <int DT>
static void grad_launch_nb(int w_bit, dim3 grid, hipStream_t st, const void* gy, const int32_t* qw, const void* scales,
                           const void* zeros, const int32_t* g_idx, void* gx, int M, int K, int N, int group_size, int asym) {
#define L(NBV) hipLaunchKernelGGL((mpq_grad_input_kernel<DT, NBV>), grid, dim3(256), 0, st, gy, (const uint32_t*)qw, scales, zeros, g_idx, gx, M, K, N, group_size, asym)
    switch (w_bit) {
        case 1: L(32); break;
        case 2: L(16); break;
        case 4: L(8); break;
        default: L(4); break;
    }
#undef L
}

int mpq_grad_input_launch(const void* gy, const int32_t* qw, const void* scales, const void* zeros, const int32_t* g_idx,
                          void* gx, int M, int K, int N, int w_bit, int group_size, int asym, int dtype, hipStream_t st) {
    const int nb = 32 / w_bit;
    dim3 grid(cdiv(cdiv(K, nb), 4), M);
    if (dtype == BIE_F16) grad_launch_nb<BIE_F16>(w_bit, grid, st, gy, qw, scales, zeros, g_idx, gx, M, K, N, group_size, asym);
    else if (dtype == BIE_BF16) grad_launch_nb<BIE_BF16>(w_bit, grid, st, gy, qw, scales, zeros, g_idx, gx, M, K, N, group_size, asym);
    else grad_launch_nb<BIE_F32>(w_bit, grid, st, gy, qw, scales, zeros, g_idx, gx, M, K, N, group_size, asym);
    return check_launch("mpq_grad_input_kernel");
}

}  // namespace bie
