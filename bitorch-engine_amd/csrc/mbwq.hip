// MBWQ (fp16 only): uniform 4/2-bit GPTQ-like weights and the mixed 8/6/5/4/3/2-bit "exl2" layout.
//   * uniform: same packed layout as MPQ, W = fma(s, q, -z) with ONE rounding and an optional
//     q_perm gather of x -> served by the MPQ GEMV / MFMA GEMM kernels in ZM_FUSED mode
//     (replaces gemm_half_q4/q2_half_gptq_kernel, exl2/q_gemm_kernel_gptq.cuh:35-328, and
//     reconstruct_q4/q2_gptq_kernel, mbwq_linear_cuda_kernel.cu:314-501);
//   * exl2: K is ordered in bands 8,6,5,4,3,2 bit; inside a band 32 consecutive k of one column are a
//     `bits`-word LSB-first bitstream down `bits` consecutive packed rows (QMODE=0 dequant primitives,
//     exl2/quant/qdq_{2,3,4,5,6,8}.cuh).  Replaces reconstruct_exl2_kernel
//     (mbwq_linear_cuda_kernel.cu:92-308) and gemm_half_q_half_kernel (exl2/q_gemm_kernel.cuh:90-549).
//     One lane owns one column and walks 32-k chunks: `bits` coalesced dword loads per chunk, bit
//     extraction with compile-time shifts (v_alignbit for the straddling fields), fp16 fma dequant
//     (exactly the reference's single-rounding __hfma2), fp32 accumulation.  Decode (M <= 2) runs
//     exl2_gemv2_kernel (packed fp16 pairs, v_dot2_f32_f16, loads issued chunks ahead); 3 <= M <= 32
//     the scalar exl2_gemv_kernel in passes of 8 rows; larger M is routed by the Python front-end to
//     reconstruct + library GEMM like the reference.
#include "bie_common.h"

#pragma clang fp contract(off)

namespace bie {
unsigned* device_status_word();                            // splitk.hip
void test_forge_get(unsigned* tag_skew, int* spin_limit);  // splitk.hip

int mpq_gemv_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const void* bias, void* y,
                    float* part, int M, int K, int N, int w_bit, int group_size, int zm, int dtype, const uint16_t* perm,
                    hipStream_t st);
bool mpq_gemv_fast_ok(int M, int K, int N, int w_bit, int group_size, int dtype, bool has_gidx);
size_t mpq_gemv_workspace_bytes(int M, int K, int N, int w_bit);
int mpq_gemm_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const void* bias, void* y,
                    float* part, int M, int K, int N, int w_bit, int group_size, int zm, int dtype, const uint16_t* perm,
                    hipStream_t st);
bool mpq_gemm_ok(int M, int K, int N, int w_bit, int group_size, int dtype, bool has_gidx);
size_t mpq_gemm_workspace_bytes(int M, int K, int N);

struct Exl2Rows {
    int r[6];  // cumulative k boundaries of the 8,6,5,4,3,2-bit bands
};

__host__ __device__ __forceinline__ int exl2_bits_of_band(int b) {
    return b == 0 ? 8 : (b == 1 ? 6 : (b == 2 ? 5 : (b == 3 ? 4 : (b == 4 ? 3 : 2))));
}

// band index and first packed row of the 32-chunk starting at k (k % 32 == 0)
__device__ __forceinline__ void exl2_locate(const Exl2Rows& rows, int k, int& bits, int& prow) {
    int row = 0, prev = 0, band = 5;
    bool found = false;
#pragma unroll
    for (int b = 0; b < 6; b++) {
        const int hi = rows.r[b];
        const int bb = exl2_bits_of_band(b);
        if (!found) {
            if (k < hi) {
                band = b;
                row += ((k - prev) >> 5) * bb;
                found = true;
            } else {
                row += ((hi - prev) >> 5) * bb;
                prev = hi;
            }
        }
    }
    bits = exl2_bits_of_band(band);
    prow = row;
}

// 32 values of one column from BITS consecutive words (LSB-first stream)
template <int BITS>
__device__ __forceinline__ void exl2_extract32(const uint32_t (&w)[8], uint32_t (&q)[32]) {
    constexpr uint32_t mask = (1u << BITS) - 1u;
#pragma unroll
    for (int j = 0; j < 32; j++) {
        constexpr int dummy = 0;
        (void)dummy;
        const int bitpos = j * BITS;
        const int wi = bitpos >> 5, sh = bitpos & 31;
        uint32_t v = w[wi] >> sh;
        if (sh + BITS > 32) v |= w[wi + 1] << (32 - sh);
        q[j] = v & mask;
    }
}

// 16 fp16 pairs (1024 + q[2i], 1024 + q[2i+1]) of a chunk straight from the bitstream: a 32-bit window holding both
// fields (funnel shift when it straddles two words), AND to the 2*BITS field bits, (t << (16-BITS)) | t puts the second
// field at bit 16 (OR, not ADD: the overlap garbage lies outside the masks and cannot carry), AND-OR applies the field masks
// and the 0x6400 exponent.  3-4 VALU per pair.
template <int BITS>
__device__ __forceinline__ void exl2_pairs16(const uint32_t (&w)[8], uint32_t (&P)[16]) {
    constexpr uint32_t mask = (1u << BITS) - 1u;
    constexpr uint32_t mask2 = (BITS == 16) ? 0xffffffffu : ((1u << (2 * BITS)) - 1u);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int bitpos = 2 * i * BITS;
        const int wi = bitpos >> 5, sh = bitpos & 31;
        uint32_t v;
        if (sh == 0) v = w[wi];
        else if (sh + 2 * BITS <= 32) v = w[wi] >> sh;
        else v = __builtin_amdgcn_alignbit(w[wi + 1], w[wi], sh);
        const uint32_t t = v & mask2;
        const uint32_t u = (t << (16 - BITS)) | t;
        P[i] = (u & (mask | (mask << 16))) | 0x64006400u;
    }
}

template <int BITS>
__device__ __forceinline__ void exl2_load_chunk(const uint32_t* __restrict__ qw, long N, int prow, int n, uint32_t (&w)[8]) {
#pragma unroll
    for (int i = 0; i < BITS; i++) w[i] = qw[(long)(prow + i) * N + n];
}

__device__ __forceinline__ uint16_t exl2_dq(uint32_t q, half_t s, half_t z) {
    const half_t r = __builtin_fmaf16((half_t)(float)q, s, -z);  // v_fma_f16, one rounding == __hfma2
    return __builtin_bit_cast(uint16_t, r);
}

// ---- dense reconstruction: out[q_perm[k]][n] ------------------------------------------------------------
__global__ __launch_bounds__(256) void exl2_dequant_kernel(const uint32_t* __restrict__ qw, const uint16_t* __restrict__ scales,
                                                           const uint16_t* __restrict__ zeros, const uint16_t* __restrict__ perm,
                                                           const uint16_t* __restrict__ gmap, uint16_t* __restrict__ out,
                                                           Exl2Rows rows, int K, int N) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int k0 = blockIdx.y * 32;
    if (n >= N || k0 >= K) return;
    int bits, prow;
    exl2_locate(rows, k0, bits, prow);
    uint32_t w[8], q[32];
    switch (bits) {
        case 8: exl2_load_chunk<8>(qw, N, prow, n, w); exl2_extract32<8>(w, q); break;
        case 6: exl2_load_chunk<6>(qw, N, prow, n, w); exl2_extract32<6>(w, q); break;
        case 5: exl2_load_chunk<5>(qw, N, prow, n, w); exl2_extract32<5>(w, q); break;
        case 4: exl2_load_chunk<4>(qw, N, prow, n, w); exl2_extract32<4>(w, q); break;
        case 3: exl2_load_chunk<3>(qw, N, prow, n, w); exl2_extract32<3>(w, q); break;
        default: exl2_load_chunk<2>(qw, N, prow, n, w); exl2_extract32<2>(w, q); break;
    }
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int g = gmap[2 * (k0 + 16 * half)];
        const half_t s = __builtin_bit_cast(half_t, scales[(long)g * N + n]);
        const half_t z = __builtin_bit_cast(half_t, zeros[(long)g * N + n]);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int k = k0 + 16 * half + j;
            const int orow = perm ? (int)perm[k] : k;
            out[(long)orow * N + n] = exl2_dq(q[16 * half + j], s, z);
        }
    }
}

// uniform q4/q2: out[q_perm ? q_perm[k] : k][n] = fma(s, q, -z)
__global__ __launch_bounds__(256) void mbwq_q4_dequant_kernel(const uint32_t* __restrict__ qw, const uint16_t* __restrict__ scales,
                                                              const uint16_t* __restrict__ zeros, const uint16_t* __restrict__ perm,
                                                              uint16_t* __restrict__ out, int K, int N, int bits, int group_size) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (n >= N) return;
    const int nb = 32 / bits;
    const uint32_t mask = (1u << bits) - 1u;
    const uint32_t word = qw[(long)r * N + n];
    for (int j = 0; j < nb; j++) {
        const int k = r * nb + j;
        if (k >= K) break;
        const int g = k / group_size;
        const half_t s = __builtin_bit_cast(half_t, scales[(long)g * N + n]);
        const half_t z = __builtin_bit_cast(half_t, zeros[(long)g * N + n]);
        const int orow = perm ? (int)perm[k] : k;
        out[(long)orow * N + n] = exl2_dq((word >> (j * bits)) & mask, s, z);
    }
}

// ---- exl2 GEMV (M <= 8 per launch): partial[slab][m][n] ---------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256) void exl2_gemv_kernel(const uint16_t* __restrict__ x, const uint32_t* __restrict__ qw,
                                                        const uint16_t* __restrict__ scales, const uint16_t* __restrict__ zeros,
                                                        const uint16_t* __restrict__ perm, const uint16_t* __restrict__ gmap,
                                                        float* __restrict__ part, Exl2Rows rows, int M, int K, int N,
                                                        int chunks_per_slab) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [MT][chunks_per_slab*32] fp32
    const int tid = threadIdx.x;
    const int n = blockIdx.x * 256 + tid;
    const int c_begin = blockIdx.y * chunks_per_slab;
    int c_end = c_begin + chunks_per_slab;
    const int C = K >> 5;
    if (c_end > C) c_end = C;
    const int slab_k = chunks_per_slab * 32;
    for (int idx = tid; idx < MT * slab_k; idx += 256) {
        const int m = idx / slab_k, kk = idx - m * slab_k;
        const int k = c_begin * 32 + kk;
        float v = 0.f;
        if (m < M && k < K) v = f16_bits_to_f32(x[(long)m * K + (perm ? (int)perm[k] : k)]);
        xs[idx] = v;
    }
    __syncthreads();
    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m] = 0.f;
    if (n < N) {
        for (int c = c_begin; c < c_end; c++) {
            const int k0 = c * 32;
            int bits, prow;
            exl2_locate(rows, k0, bits, prow);
            uint32_t w[8], q[32];
            switch (bits) {
                case 8: exl2_load_chunk<8>(qw, N, prow, n, w); exl2_extract32<8>(w, q); break;
                case 6: exl2_load_chunk<6>(qw, N, prow, n, w); exl2_extract32<6>(w, q); break;
                case 5: exl2_load_chunk<5>(qw, N, prow, n, w); exl2_extract32<5>(w, q); break;
                case 4: exl2_load_chunk<4>(qw, N, prow, n, w); exl2_extract32<4>(w, q); break;
                case 3: exl2_load_chunk<3>(qw, N, prow, n, w); exl2_extract32<3>(w, q); break;
                default: exl2_load_chunk<2>(qw, N, prow, n, w); exl2_extract32<2>(w, q); break;
            }
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int g = gmap[2 * (k0 + 16 * half)];
                const half_t s = __builtin_bit_cast(half_t, scales[(long)g * N + n]);
                const half_t z = __builtin_bit_cast(half_t, zeros[(long)g * N + n]);
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const float wv = f16_bits_to_f32(exl2_dq(q[16 * half + j], s, z));
                    const int kk = (c - c_begin) * 32 + 16 * half + j;
#pragma unroll
                    for (int m = 0; m < MT; m++) acc[m] = __builtin_fmaf(wv, xs[m * slab_k + kk], acc[m]);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MT; m++)
            if (m < M) part[((long)blockIdx.y * M + m) * N + n] = acc[m];
    }
}

// ---- exl2 decode GEMV (M <= 2): one column per lane, 16 waves per block interleave the 32-k chunks of the block's K slab ----
// Per chunk: `bits` coalesced dword loads, pair extraction (exl2_pairs16), then the dequant and the dot product run on PACKED
// fp16: the pairs are the halves 1024+q, v_pk_add_f16 / v_pk_fma_f16 reproduce the
// reference's single-rounding __hfma2(q, s, -z) exactly, v_dot2_f32_f16 accumulates in fp32 against x pairs broadcast from
// LDS (x is gathered through q_perm once per block, as fp16).  ~3.5 VALU per weight instead of ~6.5 in the scalar-fp32
// kernel below, and no finalize launch when one slab covers K.
// EX2_NW waves per workgroup: 16 (one workgroup per CU) when the column blocks alone fill most of the chip (measured 17.8 us
// against 19.4 us at 4096x11008), 8 (two per CU) + K slabs for narrower layers (4096x4096: 7.9 us against 9.1 us)
template <int MT, int EX2_NW>
__global__ __launch_bounds__(EX2_NW * 64, (EX2_NW == 8 ? 2 : 1)) void exl2_gemv2_kernel(const uint16_t* __restrict__ x, const uint32_t* __restrict__ qw,
                                                                    const uint16_t* __restrict__ scales, const uint16_t* __restrict__ zeros,
                                                                    const uint16_t* __restrict__ perm, const uint16_t* __restrict__ gmap,
                                                                    unsigned long long* __restrict__ gran, unsigned* __restrict__ gen,
                                                                    uint16_t* __restrict__ y, Exl2Rows rows, int M, int K, int N,
                                                                    int chunks_per_slab, int S, unsigned epoch, unsigned* status,
                                                                    unsigned tag_skew, int spin_limit) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem2[];
    const int tid = threadIdx.x, lane = tid & 63;
    // the tile's generation is read ONCE, at kernel entry, by every wave: the reducer advances the word as soon as it is done, and a
    // wave that read it only in the epilogue could see the NEXT generation, tag with it and never be matched
    unsigned gen_entry = 0;
    if (S > 1) gen_entry = __hip_atomic_load(gen + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave-private x chunk buffers [wave][set 0..3][MT][32] fp16 (q_perm applied): each wave gathers the 32 activations of
    // its own chunk together with the chunk's loads -- no block-wide x slab, no barrier before the weight stream
    uint16_t* xs = reinterpret_cast<uint16_t*>(smem2) + wave * (4 * MT * 32);
    const int n = blockIdx.x * 64 + lane;
    const int nl = n < N ? n : N - 1;
    const int C = K >> 5;
    const int c_begin = blockIdx.y * chunks_per_slab;
    int c_end = c_begin + chunks_per_slab;
    if (c_end > C) c_end = C;
    // The slab's permutation indices and group-map entries are staged in LDS once per workgroup: a chunk's group constants and
    // gathered activations can then be requested TOGETHER with its packed words.  (Fetched per chunk they were two dependent
    // global loads behind the words -- PMC: 58 % of the wave time waiting, profiles/r02_pmc_exl2.txt.)
    uint16_t* perm_s = reinterpret_cast<uint16_t*>(smem2) + EX2_NW * (4 * MT * 32);  // [chunks_per_slab * 32]
    uint16_t* gmap_s = perm_s + chunks_per_slab * 32;                               // [chunks_per_slab * 2]
    {
        const int nk = (c_end - c_begin) * 32;
        for (int i = tid; i < nk; i += EX2_NW * 64) perm_s[i] = perm ? perm[c_begin * 32 + i] : (uint16_t)(c_begin * 32 + i);
        for (int i = tid; i < (c_end - c_begin) * 2; i += EX2_NW * 64) gmap_s[i] = gmap[2 * ((c_begin + (i >> 1)) * 32 + 16 * (i & 1))];
    }
    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m] = 0.f;
    const half2_t k1024 = half2_t{(half_t)1024.0f, (half_t)1024.0f};
    // The wave's chunks are taken BAND BY BAND: inside a band the bit width -- hence the number of loads per chunk -- is a compile-time
    // constant, every issue is unconditional (the look-ahead index is clamped to the band's last chunk: a harmless re-load), so the
    // compiler can wait with an exact vmcnt for the OLDEST chunk only.  (With the width switched at run time and `if (c < c_end)`
    // around the issues every wait came out as vmcnt(0..3): the 4-deep prefetch was really 1-deep and each of a wave's 8 chunks paid
    // a full memory latency -- PMC: 58 % of the wave time waiting.)
    __syncthreads();  // the metadata slab is in LDS
    auto band = [&](auto bits_tag, int cb0, int cb1, int prow0) {  // chunks [cb0, cb1) of this slab lie in one band starting at row prow0
        constexpr int BITS = decltype(bits_tag)::value;
        struct Chunk {
            uint32_t w[BITS];
            uint32_t s[2], z[2];
            uint32_t xraw[MT];
        };
        auto issue = [&](int c, Chunk& ch) {
            const int cl = c - c_begin;
            const int prow = prow0 + (c - cb0) * BITS;
#pragma unroll
            for (int i = 0; i < BITS; i++) ch.w[i] = __builtin_nontemporal_load(qw + (long)(prow + i) * N + nl);
            const int g0 = __builtin_amdgcn_readfirstlane((int)gmap_s[2 * cl]), g1 = __builtin_amdgcn_readfirstlane((int)gmap_s[2 * cl + 1]);
            const int pidx = (int)perm_s[cl * 32 + (lane & 31)];
            ch.s[0] = scales[(long)g0 * N + nl];
            ch.z[0] = zeros[(long)g0 * N + nl];
            ch.s[1] = scales[(long)g1 * N + nl];  // unconditional (same line as [0] when the halves share a group): static load count
            ch.z[1] = zeros[(long)g1 * N + nl];
#pragma unroll
            for (int m = 0; m < MT; m++) ch.xraw[m] = x[(long)(m < M ? m : 0) * K + pidx];
        };
        auto compute = [&](int set, const Chunk& ch) {
            uint16_t* xw = xs + set * (MT * 32);
            if (lane < 32) {
#pragma unroll
                for (int m = 0; m < MT; m++) xw[m * 32 + lane] = (uint16_t)ch.xraw[m];
            }
            // same wave writes then reads: the LDS pipe keeps a wave's operations in order, the compiler must too (the 2-byte
            // stores and the 16-byte loads below have different types)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            uint32_t w8[8];
#pragma unroll
            for (int i = 0; i < 8; i++) w8[i] = i < BITS ? ch.w[i] : 0u;
            uint32_t P[16];
            exl2_pairs16<BITS>(w8, P);
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const half_t sh = __builtin_bit_cast(half_t, (uint16_t)ch.s[half]);
                const half_t zh = __builtin_bit_cast(half_t, (uint16_t)ch.z[half]);
                const half2_t s2 = half2_t{sh, sh}, nz2 = half2_t{(half_t)-zh, (half_t)-zh};
                uint4_t xv[MT][2];
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    const uint4_t* xp = reinterpret_cast<const uint4_t*>(xw + m * 32 + 16 * half);
                    xv[m][0] = xp[0];
                    xv[m][1] = xp[1];
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const half2_t qh = __builtin_bit_cast(half2_t, P[8 * half + i]) - k1024;  // exact
                    const half2_t r = __builtin_elementwise_fma(qh, s2, nz2);           // one rounding == __hfma2(q, s, -z)
#pragma unroll
                    for (int m = 0; m < MT; m++) {
                        if (m >= M) continue;
                        const uint4_t xq = xv[m][i >> 2];
                        const uint32_t xpair = (i & 3) == 0 ? xq.x : ((i & 3) == 1 ? xq.y : ((i & 3) == 2 ? xq.z : xq.w));
                        acc[m] = __builtin_amdgcn_fdot2(r, __builtin_bit_cast(half2_t, xpair), acc[m], false);
                    }
                }
            }
        };
        // this wave's chunks in the band: c = first, first + NW, ... < cb1
        int first = c_begin + wave;
        if (first < cb0) first += ((cb0 - first + EX2_NW - 1) / EX2_NW) * EX2_NW;
        if (first >= cb1) return;
        const int cnt = (cb1 - first + EX2_NW - 1) / EX2_NW;
        const int last = first + (cnt - 1) * EX2_NW;
        auto at = [&](int jj) { const int c = first + jj * EX2_NW; return c < last ? c : last; };  // clamped look-ahead
        Chunk c0, c1, c2, c3;
        issue(at(0), c0);
        issue(at(1), c1);
        issue(at(2), c2);
        issue(at(3), c3);
        int jj = 0;
        for (; jj + 4 < cnt; jj += 4) {  // a further group follows: four full steps, each re-filling the set it has just consumed
            compute(0, c0);
            issue(at(jj + 4), c0);
            compute(1, c1);
            issue(at(jj + 5), c1);
            compute(2, c2);
            issue(at(jj + 6), c2);
            compute(3, c3);
            issue(at(jj + 7), c3);
        }
        compute(0, c0);  // last group: nothing left to request
        if (jj + 1 < cnt) compute(1, c1);
        if (jj + 2 < cnt) compute(2, c2);
        if (jj + 3 < cnt) compute(3, c3);
    };
    {
        int kprev = 0, prow = 0;
#pragma unroll
        for (int b = 0; b < 6; b++) {
            const int khi = rows.r[b];
            const int bits = exl2_bits_of_band(b);
            int cb0 = kprev >> 5, cb1 = khi >> 5;  // the band's chunk range
            const int prow_band = prow;
            prow += (cb1 - cb0) * bits;
            kprev = khi;
            const int skip = cb0 < c_begin ? c_begin - cb0 : 0;  // clip to the slab
            cb0 += skip;
            if (cb1 > c_end) cb1 = c_end;
            if (cb0 < cb1) {
                const int p0 = prow_band + skip * bits;
                switch (b) {
                    case 0: band(std::integral_constant<int, 8>{}, cb0, cb1, p0); break;
                    case 1: band(std::integral_constant<int, 6>{}, cb0, cb1, p0); break;
                    case 2: band(std::integral_constant<int, 5>{}, cb0, cb1, p0); break;
                    case 3: band(std::integral_constant<int, 4>{}, cb0, cb1, p0); break;
                    case 4: band(std::integral_constant<int, 3>{}, cb0, cb1, p0); break;
                    default: band(std::integral_constant<int, 2>{}, cb0, cb1, p0); break;
                }
            }
        }
    }
    // block reduction over the waves in wave order (deterministic)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem2);
#pragma unroll
    for (int m = 0; m < MT; m++) red[(wave * MT + m) * 64 + lane] = acc[m];
    __syncthreads();
    if (tid < 64 * MT) {
        const int om = tid >> 6, ol = tid & 63, on = blockIdx.x * 64 + ol;
        float tot = 0.f;
#pragma unroll
        for (int wv = 0; wv < EX2_NW; wv++) tot += red[(wv * MT + om) * 64 + ol];
        if (S > 1) {
            // K slabs: the tagged-granule reduction of the lookup GEMV (mpq_gemv_lut.hip) -- slabs 0..S-2 publish {fp32, tag} with
            // one write-through store per column and retire; the last slab's workgroup (highest block ids: dispatched after every
            // publisher, which never waits) polls them and adds in slab order.  No finalize launch, no atomics, deterministic.
            const unsigned gen_next = gen_entry + 1u;
            const unsigned tag = epoch | (gen_next & 0xffu);
            const long ncat = (long)gridDim.x * 64, col = (long)blockIdx.x * 64 + ol;
            const int slab = blockIdx.y;
            if (slab != S - 1) {
                const unsigned long long gval = ((unsigned long long)tag << 32) | __float_as_uint(tot);
                __hip_atomic_store(gran + ((long)slab * MT + om) * ncat + col, gval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            float v = 0.0f;
            for (int s0 = 0; s0 < S - 1; s0 += 4) {
                unsigned long long gv[4];
                bool ready;
                int spins = 0;
                do {
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) {
                        const int sidx = (s0 + jj < S - 1) ? s0 + jj : S - 2;
                        gv[jj] = __hip_atomic_load(gran + ((long)sidx * MT + om) * ncat + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    ready = true;
#pragma unroll
                    for (int jj = 0; jj < 4; jj++) ready = ready && ((unsigned)(gv[jj] >> 32) == (tag ^ tag_skew));
                    ready = __builtin_amdgcn_ballot_w64(!ready) == 0;
                    if (!ready) __builtin_amdgcn_s_sleep(2);
                } while (!ready && ++spins < spin_limit);
#pragma unroll
                for (int jj = 0; jj < 4; jj++)
                    if (s0 + jj < S - 1) v += __uint_as_float((unsigned)gv[jj]);
                if (!ready) {  // wave-uniform: never a silent number -- NaN in y and a bit in the status page (bie_device_status)
                    v = __uint_as_float(0x7fc00000u);
                    if (ol == 0 && status) __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            tot = v + tot;
            if (tid == 0) gen[blockIdx.x] = gen_next;  // the next launch (or a replay of this one) tags differently
        }
        if (om < M && on < N) y[(long)om * N + on] = f32_to_f16_bits(tot);
    }
}

// decode (M <= 2): column blocks x K slabs ~ 512 workgroups of 8 waves (two per CU -> one round), slabs in whole multiples of
// 8 chunks so that the 8 waves of a workgroup get equal shares; at most BIE_WS_COUNTERS column blocks use the slab reduction
static void exl2_decode_plan(int M, int K, int N, int& cps, int& S, int& nw) {
    const int C = K / 32, colblocks = cdiv(N, 64);
    constexpr int CPS_MAX = 768;  // the slab's q_perm / group-map copy in LDS: 68 bytes per chunk (52 KiB) beside the x buffers
    if (colblocks >= 160) {  // wide layers: 16-wave workgroups, one K slab (more only when K is too long for the LDS copy)
        nw = 16;
        S = cdiv(C, CPS_MAX);
        cps = cdiv(cdiv(C, S), nw) * nw;
        if (cps > C) cps = C;
        S = cdiv(C, cps);
        return;
    }
    nw = 8;
    int want = (512 + colblocks / 2) / colblocks;
    if (want < 1) want = 1;
    cps = cdiv(cdiv(C, want), nw) * nw;
    if (cps < 4 * nw) cps = 4 * nw;  // at least four chunks per wave: the depth of the kernel's prefetch
    if (cps > CPS_MAX) cps = CPS_MAX;
    if (cps > C) cps = C;
    S = cdiv(C, cps);
}

static int exl2_slabs(int K, int N) {
    const int C = K / 32;
    int S = cdiv(1024, cdiv(N, 256));
    if (S > C) S = C;
    if (S < 1) S = 1;
    int cps = cdiv(C, S);
    if (cps > 64) cps = 64;  // LDS: 8 rows x 64 chunks x 32 x 4 B = 64 KiB
    return cps;
}

size_t mbwq_workspace_bytes(int M, int K, int N) {
    size_t a = 0;
    for (int w : {2, 4}) {
        size_t t = M <= 8 ? mpq_gemv_workspace_bytes(M, K, N, w) : 0;
        if (t > a) a = t;
    }
    size_t b = mpq_gemm_workspace_bytes(M, K, N);
    const int cps = exl2_slabs(K, N);
    const int S = cdiv(K / 32, cps);
    const int mc = M < 8 ? M : 8;
    size_t c = (size_t)S * mc * N * sizeof(float);
    int cps2, S2, nw2;
    exl2_decode_plan(M, K, N, cps2, S2, nw2);
    const size_t d = M <= 2 && S2 > 1 ? (size_t)(S2 - 1) * M * cdiv(N, 64) * 64 * 8 : 0;  // decode granules (unused when the fp32 kernel takes over)
    if (d > c) c = d;
    size_t r = a > b ? a : b;
    return r > c ? r : c;
}

int mbwq_q4_dequant_launch(const int32_t* qw, const void* scales, const void* zeros, const int16_t* perm, void* out, int K,
                           int N, int bits, int group_size, hipStream_t st) {
    dim3 grid(cdiv(N, 256), cdiv(K, 32 / bits));
    hipLaunchKernelGGL(mbwq_q4_dequant_kernel, grid, dim3(256), 0, st, (const uint32_t*)qw, (const uint16_t*)scales,
                       (const uint16_t*)zeros, (const uint16_t*)perm, (uint16_t*)out, K, N, bits, group_size);
    return check_launch("mbwq_q4_dequant_kernel");
}

int mbwq_exl2_dequant_launch(const int32_t* qw, const void* scales, const void* zeros, const int16_t* perm,
                             const int16_t* gmap, const int* rows7, void* out, int K, int N, hipStream_t st) {
    Exl2Rows rows;
    for (int i = 0; i < 6; i++) rows.r[i] = rows7[i];
    dim3 grid(cdiv(N, 256), K / 32);
    hipLaunchKernelGGL(exl2_dequant_kernel, grid, dim3(256), 0, st, (const uint32_t*)qw, (const uint16_t*)scales,
                       (const uint16_t*)zeros, (const uint16_t*)perm, (const uint16_t*)gmap, (uint16_t*)out, rows, K, N);
    return check_launch("exl2_dequant_kernel");
}

int mbwq_q4_forward_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const int16_t* perm,
                           void* y, float* part, int M, int K, int N, int bits, int group_size, hipStream_t st) {
    const uint16_t* p = (const uint16_t*)perm;
    const bool gemm_ok = mpq_gemm_ok(M, K, N, bits, group_size, BIE_F16, false);
    if (M <= 8 && (M <= 2 || !gemm_ok) && mpq_gemv_fast_ok(M, K, N, bits, group_size, BIE_F16, false))
        return mpq_gemv_launch(x, qw, scales, zeros, nullptr, y, part, M, K, N, bits, group_size, 2, BIE_F16, p, st);
    if (gemm_ok)
        return mpq_gemm_launch(x, qw, scales, zeros, nullptr, y, part + BIE_WS_HEAD_BYTES / sizeof(float), M, K, N, bits, group_size, 2, BIE_F16, p, st);
    set_error("bie_mbwq_q4_forward: unsupported shape M=%d K=%d N=%d bits=%d group_size=%d (need K %% 64 == 0, N %% 4 == 0)", M,
              K, N, bits, group_size);
    return BIE_ERR_UNSUPPORTED;
}

int mbwq_exl2_forward_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const int16_t* perm,
                             const int16_t* gmap, const int* rows7, void* y, float* head, float* part, int M, int K, int N,
                             hipStream_t st) {
    Exl2Rows rows;
    for (int i = 0; i < 6; i++) rows.r[i] = rows7[i];
    const bool slab_ok = !(cdiv(N, 64) > BIE_WS_COUNTERS && K / 32 > 768);  // K slabs need one generation word per column block
    if (M <= 2 && slab_ok) {  // decode path.  (The packed-fp16 kernel instantiated for 4 / 8 rows measured SLOWER than the fp32 kernel below:
                   //  33.3 / 46.2 us against 28.0 / 39.9 us at 4096x11008 M = 3 / 8 -- 178-256 registers, one wave per SIMD.)
        const int colblocks = cdiv(N, 64);
        int cps2, S, nw;
        exl2_decode_plan(M, K, N, cps2, S, nw);
        const int MT = M;
        size_t lds2 = (size_t)nw * 4 * MT * 32 * sizeof(uint16_t)   // wave-private x chunk buffers
                      + (size_t)cps2 * 34 * sizeof(uint16_t);      // + the slab's q_perm indices and group-map entries
        const size_t red = (size_t)nw * MT * 64 * sizeof(float);
        if (lds2 < red) lds2 = red;
        dim3 grid2(colblocks, S);
        const unsigned epoch = next_launch_epoch();
        unsigned* gen = reinterpret_cast<unsigned*>(head) + BIE_WS_GEN_OFFSET;
        unsigned long long* gran = reinterpret_cast<unsigned long long*>(part);
        unsigned skew;
        int spin;
        test_forge_get(&skew, &spin);
#define L2(MTV, NWV)                                                                                                       \
    hipLaunchKernelGGL((exl2_gemv2_kernel<MTV, NWV>), grid2, dim3(NWV * 64), lds2, st, (const uint16_t*)x, (const uint32_t*)qw, \
                       (const uint16_t*)scales, (const uint16_t*)zeros, (const uint16_t*)perm, (const uint16_t*)gmap, gran, gen, \
                       (uint16_t*)y, rows, M, K, N, cps2, S, epoch, device_status_word(), skew, spin)
        if (nw == 16) {
            if (MT == 1) L2(1, 16); else L2(2, 16);
        } else {
            if (MT == 1) L2(1, 8); else L2(2, 8);
        }
#undef L2
        return check_launch("exl2_gemv2_kernel");
    }
    const int cps = exl2_slabs(K, N);
    const int S = cdiv(K / 32, cps);
    dim3 grid(cdiv(N, 256), S);
    for (int m0 = 0; m0 < M; m0 += 8) {
        const int mc = (M - m0) < 8 ? (M - m0) : 8;
        const uint16_t* xm = (const uint16_t*)x + (size_t)m0 * K;
        const int MT = mc <= 1 ? 1 : (mc <= 2 ? 2 : (mc <= 4 ? 4 : 8));
        const size_t lds = (size_t)MT * cps * 32 * sizeof(float);
#define L(MTV)                                                                                                     \
    hipLaunchKernelGGL(exl2_gemv_kernel<MTV>, grid, dim3(256), lds, st, xm, (const uint32_t*)qw, (const uint16_t*)scales, \
                       (const uint16_t*)zeros, (const uint16_t*)perm, (const uint16_t*)gmap, part, rows, mc, K, N, cps)
        switch (MT) {
            case 1: L(1); break;
            case 2: L(2); break;
            case 4: L(4); break;
            default: L(8); break;
        }
#undef L
        int rc = check_launch("exl2_gemv_kernel");
        if (rc) return rc;
        rc = launch_splitk_finalize(part, nullptr, (uint16_t*)y + (size_t)m0 * N, S, mc, N, BIE_F16, st);
        if (rc) return rc;
    }
    return BIE_OK;
}

}  // namespace bie
