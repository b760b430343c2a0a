// MBWQ (fp16 only): uniform 4/2-bit GPTQ-like weights and the mixed 8/6/5/4/3/2-bit "exl2" layout.
//   * uniform: same packed layout as MPQ, W = fma(s, q, -z) with ONE rounding and an optional
//     q_perm gather of x -> served by the MPQ GEMV / MFMA GEMM kernels in ZM_FUSED mode
//     (replaces gemm_half_q4/q2_half_gptq_kernel, exl2/q_gemm_kernel_gptq.cuh:35-328, and
//     reconstruct_q4/q2_gptq_kernel, mbwq_linear_cuda_kernel.cu:314-501);
//   * exl2: K is ordered in bands 8,6,5,4,3,2 bit; in the checkpoint 32 consecutive k of one column are a
//     `bits`-word LSB-first bitstream down `bits` consecutive packed rows (QMODE=0 dequant primitives,
//     exl2/quant/qdq_{2,3,4,5,6,8}.cuh); the load-time step (exl2_shuffle_kernel: the reference's shuffle_kernel
//     hook, mbwq_linear_cuda_kernel.cu:63-86) re-arranges every chunk into the half-pair layout below, which is
//     what all kernels of this file read.  Replaces reconstruct_exl2_kernel (mbwq_linear_cuda_kernel.cu:92-308)
//     and gemm_half_q_half_kernel (exl2/q_gemm_kernel.cuh:90-549).  One lane owns one column and walks 32-k
//     chunks: `bits` coalesced dword loads per chunk, one v_and_or_b32 per pair of values.
//       - decode (M <= 2; lists and sibling groups up to four rows): exl2_gemv2_body -- the pairs go to the matrix
//         pipe as they are (v_mfma_f32_4x4x4_16B_f16, lane = column), scale and zero applied once per chunk in fp32;
//       - 3 <= M <= 64: exl2_mfma_kernel (per-weight fp16 fma dequant == the reference's __hfma2, LDS transpose into
//         v_mfma_f32_16x16x32_f16); beyond: the Python front-end reconstructs + library GEMM like the reference;
//       - reconstruction (exl2_dequant_kernel): the reference's single-rounding __hfma2, bit-exact.
#include "bie_common.h"

#pragma clang fp contract(off)

#include <vector>
#include <cstring>

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
int mpq_gemv_generic_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const int32_t* g_idx, const void* bias, void* y, float* part,
                            int M, int K, int N, int w_bit, int group_size, int asym, int dtype, hipStream_t st, const uint16_t* perm);
constexpr int MBWQ_GENERIC_M_CHUNK = 32;  // rows per launch of the any-shape kernel (its partial sums: cdiv(K, 512) x rows x N floats)
size_t mpq_gemm_workspace_bytes(int M, int K, int N);

struct Exl2Rows {
    int r[6];  // cumulative k boundaries of the 8,6,5,4,3,2-bit bands
};

// rows of x the pre-permuted decode form is instantiated for (and the kernel in front writes: chunk-major, rows beyond M repeating row M - 1)
__host__ __device__ __forceinline__ int exl2_xp_rows_d(int M) { return M <= 1 ? 1 : (M <= 2 ? 2 : (M <= 4 ? 4 : (M <= 8 ? 8 : 16))); }
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

// ---- the chunk layout in HBM ("half-pair" layout, written once by exl2_shuffle_kernel below) -------------------------------------
// The checkpoint stores the 32 values of a chunk as one LSB-first stream over BITS words (exl2/quant/qdq_*.cuh, QMODE 0).  The
// reference re-arranges that stream at load time for its own extraction (shuffle_kernel, mbwq_linear_cuda_kernel.cu:63-86; its
// build configures the hook as a no-op); this engine uses the same hook for a layout made for v_and_or_b32 + packed fp16:
//   * pair j = (q[2j], q[2j+1]) occupies the SAME bit range [p, p+BITS) of the low and of the high 16-bit half of one word, so ONE
//     v_and_or_b32 (mask both halves, OR the exponents) yields the fp16 pair (2^(10-p') + q[2j], 2^(10-p') + q[2j+1]) -- the field
//     stays where it is, the exponent is chosen so that the field's lowest bit weighs 1 (p' = p - window base, p' + BITS <= 10);
//   * a half holds F = 16 / BITS whole fields, pairs 0 .. BITS*F-1 are (word j / F, field j % F);
//   * the R = 16 - F*BITS spare top bits of the halves (3-, 5-, 6-bit bands) form, word after word, a spare stream per half that
//     holds the remaining 16 - BITS*F pairs.
// Per pair: 1 VALU (+ one shift per 10-bit window of a word) instead of the 3-4 of window / mask / spread / mask on the stream.
template <int I, int N, class Fn>
__device__ __forceinline__ void exl2_static_for(Fn&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        exl2_static_for<I + 1, N>(f);
    }
}

struct Exl2Piece {  // `len` bits at bit `src` of a half of word `d` are bits [dst, dst + len) of a spare-stream field
    int d, src, len, dst;
};

template <int B>
struct Exl2Lay {
    static constexpr int F = 16 / B;
    static constexpr int R = 16 - F * B;
    static constexpr int MAIN = B * F;  // pairs in whole fields
    // base of the 10-bit mantissa window the field at bit p of a half is read through
    static constexpr int win(int p) {
        int b0 = 0;
        for (int q = 0; q <= p; q += B)
            if (q + B - b0 > 10) b0 = q;
        return b0;
    }
    // piece t of spare-stream field r (len == 0: the field has fewer pieces)
    static constexpr Exl2Piece piece(int r, int t) {
        int bit = r * B, need = B, dst = 0;
        for (int i = 0;; i++) {
            if (need <= 0 || R == 0) return Exl2Piece{0, 0, 0, 0};
            const int d = bit / R, off = bit % R;
            const int len = (R - off) < need ? (R - off) : need;
            if (i == t) return Exl2Piece{d, F * B + off, len, dst};
            bit += len; dst += len; need -= len;
        }
    }
};

// raw fields of pair J of a shuffled chunk: q[2J] | q[2J+1] << 16
template <int BITS, int J>
__device__ __forceinline__ uint32_t exl2_raw_pair(const uint32_t (&w)[8]) {
    using L = Exl2Lay<BITS>;
    constexpr uint32_t mask = (1u << BITS) - 1u;
    if constexpr (J < L::MAIN) {
        return (w[J / L::F] >> ((J % L::F) * BITS)) & (mask | (mask << 16));
    } else {
        uint32_t acc = 0;
        exl2_static_for<0, BITS>([&](auto t) {
            constexpr Exl2Piece pc = L::piece(J - L::MAIN, decltype(t)::value);
            if constexpr (pc.len > 0) acc |= ((w[pc.d] >> pc.src) & (((1u << pc.len) - 1u) * 0x00010001u)) << pc.dst;
        });
        return acc;
    }
}

// 32 values of one column from the BITS words of a shuffled chunk
template <int BITS>
__device__ __forceinline__ void exl2_extract32(const uint32_t (&w)[8], uint32_t (&q)[32]) {
    exl2_static_for<0, 16>([&](auto j) {
        constexpr int J = decltype(j)::value;
        const uint32_t u = exl2_raw_pair<BITS, J>(w);
        q[2 * J] = u & 0xffffu;
        q[2 * J + 1] = u >> 16;
    });
}

// the inverse: 32 values -> BITS words in the half-pair layout
template <int BITS>
__device__ __forceinline__ void exl2_pack32(const uint32_t (&q)[32], uint32_t (&w)[8]) {
    using L = Exl2Lay<BITS>;
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = 0u;
    exl2_static_for<0, 16>([&](auto j) {
        constexpr int J = decltype(j)::value;
        const uint32_t u = q[2 * J] | (q[2 * J + 1] << 16);
        if constexpr (J < L::MAIN) {
            w[J / L::F] |= u << ((J % L::F) * BITS);
        } else {
            exl2_static_for<0, BITS>([&](auto t) {
                constexpr Exl2Piece pc = L::piece(J - L::MAIN, decltype(t)::value);
                if constexpr (pc.len > 0) w[pc.d] |= ((u >> pc.dst) & (((1u << pc.len) - 1u) * 0x00010001u)) << pc.src;
            });
        }
    });
}

// the checkpoint's form of a chunk: one LSB-first stream over BITS words
template <int BITS>
__device__ __forceinline__ void exl2_extract32_stream(const uint32_t (&w)[8], uint32_t (&q)[32]) {
    constexpr uint32_t mask = (1u << BITS) - 1u;
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const int bitpos = j * BITS;
        const int wi = bitpos >> 5, sh = bitpos & 31;
        uint32_t v = w[wi] >> sh;
        if (sh + BITS > 32) v |= w[wi + 1] << (32 - sh);
        q[j] = v & mask;
    }
}

// The exponent words 2^(10-lp) | 2^(10-lp) << 16 for the window positions lp = 0 .. 8, kept in registers behind an opaque move:
// v_and_or_b32 and v_pk_add_f16 take no literal on gfx950 and only one scalar operand, and left to itself the compiler splits
// the and-or into v_and_b32 + v_or_b32 with two literals.  Entries a kernel never uses are never materialised.
struct Exl2Magic {
    uint32_t m[9];
    __device__ __forceinline__ Exl2Magic() {
#pragma unroll
        for (int lp = 0; lp < 9; lp++) {
            const uint32_t c = (0x6400u - ((uint32_t)lp << 10)) * 0x00010001u;
            asm("v_mov_b32 %0, %1" : "=v"(m[lp]) : "s"(c));
        }
    }
};

__device__ __forceinline__ uint32_t exl2_and_or(uint32_t v, uint32_t mask, uint32_t orv) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "s"(mask), "v"(orv));
    return r;
}

// 16 EXACT fp16 pairs (q[2i], q[2i+1]) of a shuffled chunk: v_and_or_b32 in place (exponent = 10 - field position inside the word's
// current 10-bit window), then the exact packed subtraction of that power of two.  The spare-stream pairs are gathered piece by piece.
template <int BITS>
__device__ __forceinline__ void exl2_qpairs16(const uint32_t (&w)[8], const Exl2Magic& mg, half2_t (&Q)[16]) {
    using L = Exl2Lay<BITS>;
    constexpr uint32_t mask = (1u << BITS) - 1u;
    exl2_static_for<0, 16>([&](auto j) {
        constexpr int J = decltype(j)::value;
        if constexpr (J < L::MAIN) {
            constexpr int d = J / L::F, p = (J % L::F) * BITS;
            constexpr int b0 = L::win(p), lp = p - b0;
            const uint32_t v = b0 ? (w[d] >> b0) : w[d];
            const uint32_t t = exl2_and_or(v, (mask << lp) * 0x00010001u, mg.m[lp]);
            Q[J] = __builtin_bit_cast(half2_t, t) - __builtin_bit_cast(half2_t, mg.m[lp]);  // exact
        } else {
            uint32_t acc = mg.m[0];
            exl2_static_for<0, BITS>([&](auto t) {
                constexpr Exl2Piece pc = L::piece(J - L::MAIN, decltype(t)::value);
                if constexpr (pc.len > 0) acc = exl2_and_or(w[pc.d] >> (pc.src - pc.dst), (((1u << pc.len) - 1u) << pc.dst) * 0x00010001u, acc);
            });
            Q[J] = __builtin_bit_cast(half2_t, acc) - __builtin_bit_cast(half2_t, mg.m[0]);  // exact
        }
    });
}

// (v & mask) | offset for an operand of the MATRIX pipe: written so that the compiler selects v_bfi_b32 ITSELF (the offset word has no bit
// inside the mask, so (mask & v) | (~mask & offset) is the same value).  The inline-asm v_and_or_b32 above must not feed an MFMA: a
// vector write needs 2 wait states before the matrix instruction reads the register (tools/probe/probe_mfma4.hip: raw_src_a / _b), the
// hazard recognizer inserts them for instructions it can see and cannot look into an asm block -- wrong sums, now and then, by schedule.
__device__ __forceinline__ uint32_t exl2_bfi(uint32_t v, uint32_t mask, uint32_t ofs) { return (v & mask) | (ofs & ~mask); }

// The decode kernels stop one step earlier: T[j] = the fp16 pair (2^(10-lp) + q[2j], 2^(10-lp) + q[2j+1]) as v_and_or_b32 leaves it, and
// ofs[j] = the offset pair (2^(10-lp), 2^(10-lp)) it carries (the same for every chunk of a band).  They take the dot products of T
// and of ofs with x on the matrix pipe and subtract there (exl2_gemv2_body).
template <int BITS, int J0 = 0, int J1 = 16>
__device__ __forceinline__ void exl2_tpairs16(const uint32_t (&w)[8], const Exl2Magic& mg, uint32_t (&T)[16]) {
    using L = Exl2Lay<BITS>;
    constexpr uint32_t mask = (1u << BITS) - 1u;
    exl2_static_for<J0, J1>([&](auto j) {
        constexpr int J = decltype(j)::value;
        if constexpr (J < L::MAIN) {
            constexpr int d = J / L::F, p = (J % L::F) * BITS;
            constexpr int b0 = L::win(p), lp = p - b0;
            const uint32_t v = b0 ? (w[d] >> b0) : w[d];
            T[J] = exl2_bfi(v, (mask << lp) * 0x00010001u, mg.m[lp]);
        } else {
            uint32_t acc = mg.m[0];
            exl2_static_for<0, BITS>([&](auto t) {
                constexpr Exl2Piece pc = L::piece(J - L::MAIN, decltype(t)::value);
                if constexpr (pc.len > 0) acc = exl2_bfi(w[pc.d] >> (pc.src - pc.dst), (((1u << pc.len) - 1u) << pc.dst) * 0x00010001u, acc);
            });
            T[J] = acc;
        }
    });
}
template <int BITS>
__device__ __forceinline__ void exl2_offset_pairs16(const Exl2Magic& mg, uint32_t (&ofs)[16]) {
    using L = Exl2Lay<BITS>;
    exl2_static_for<0, 16>([&](auto j) {
        constexpr int J = decltype(j)::value;
        if constexpr (J < L::MAIN) {
            constexpr int p = (J % L::F) * BITS;
            ofs[J] = mg.m[p - L::win(p)];
        } else {
            ofs[J] = mg.m[0];
        }
    });
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

// ---- the load-time re-arrangement (in place): LSB-first stream -> half-pair layout, one thread per (column, chunk) -----------------
// Replaces shuffle_kernel (mbwq_linear_cuda_kernel.cu:63-86), run once by q_linear_cuda.mbwq_trans_qweight (:620).  A thread reads
// all words of its chunk before it writes any, and chunks are disjoint: in place is safe.
template <int BITS>
__device__ __forceinline__ void exl2_shuffle_chunk(uint32_t* __restrict__ qw, long N, int prow, int n) {
    uint32_t w[8], q[32];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = i < BITS ? qw[(long)(prow + i) * N + n] : 0u;
    exl2_extract32_stream<BITS>(w, q);
    exl2_pack32<BITS>(q, w);
#pragma unroll
    for (int i = 0; i < BITS; i++) qw[(long)(prow + i) * N + n] = w[i];
}
// the way back (for writing a checkpoint in the reference's format from a prepared layer): half-pair layout -> LSB-first stream
template <int BITS>
__device__ __forceinline__ void exl2_pack32_stream(const uint32_t (&q)[32], uint32_t (&w)[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = 0u;
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const int bitpos = j * BITS;
        const int wi = bitpos >> 5, sh = bitpos & 31;
        w[wi] |= q[j] << sh;
        if (sh + BITS > 32) w[wi + 1] |= q[j] >> (32 - sh);
    }
}
template <int BITS, bool INVERSE>
__device__ __forceinline__ void exl2_reshuffle_chunk(uint32_t* __restrict__ qw, long N, int prow, int n) {
    if constexpr (!INVERSE) {
        exl2_shuffle_chunk<BITS>(qw, N, prow, n);
    } else {
        uint32_t w[8], q[32];
#pragma unroll
        for (int i = 0; i < 8; i++) w[i] = i < BITS ? qw[(long)(prow + i) * N + n] : 0u;
        exl2_extract32<BITS>(w, q);
        exl2_pack32_stream<BITS>(q, w);
#pragma unroll
        for (int i = 0; i < BITS; i++) qw[(long)(prow + i) * N + n] = w[i];
    }
}
template <bool INVERSE>
__global__ __launch_bounds__(256) void exl2_shuffle_kernel(uint32_t* __restrict__ qw, Exl2Rows rows, int K, int N) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int k0 = blockIdx.y * 32;
    if (n >= N || k0 >= K) return;
    int bits, prow;
    exl2_locate(rows, k0, bits, prow);
    switch (bits) {
        case 8: exl2_reshuffle_chunk<8, INVERSE>(qw, N, prow, n); break;
        case 6: exl2_reshuffle_chunk<6, INVERSE>(qw, N, prow, n); break;
        case 5: exl2_reshuffle_chunk<5, INVERSE>(qw, N, prow, n); break;
        case 4: exl2_reshuffle_chunk<4, INVERSE>(qw, N, prow, n); break;
        case 3: exl2_reshuffle_chunk<3, INVERSE>(qw, N, prow, n); break;
        default: exl2_reshuffle_chunk<2, INVERSE>(qw, N, prow, n); break;
    }
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

// ---- prefill: the same values straight into the MFMA fragment image of mpq_dense.hip (packed k order; x is gathered by q_perm) -----
// Fragment (nb, ks) = columns 32*nb .. +31 x k 16*ks .. +15; lane l of it owns the 8 values of column 32*nb + (l & 31), k 16*ks +
// 8*(l >> 5) .. +7 as four fp16 pairs in k order.  A thread takes one (column, 32-chunk): four 16-byte stores, each 512 contiguous bytes
// per 32 lanes.  Replaces reconstruct_exl2_kernel + at::matmul of the reference's prefill branch (mbwq_linear_cuda_kernel.cu:849-897,
// 968-1002) with the reference's rounding per weight (exl2_dq: one v_fma_f16).
__global__ __launch_bounds__(256) void exl2_dequant_frag_kernel(const uint32_t* __restrict__ qw, const uint16_t* __restrict__ scales,
                                                                const uint16_t* __restrict__ zeros, const uint16_t* __restrict__ gmap,
                                                                uint4_t* __restrict__ img, Exl2Rows rows, int K, int N) {
    const int nraw = blockIdx.x * 256 + threadIdx.x;  // up to the padded column count: columns past N repeat column N - 1 (nobody stores them)
    const int k0 = blockIdx.y * 32;
    if (nraw >= ((N + 31) & ~31) || k0 >= K) return;
    const int n = nraw < N ? nraw : N - 1;
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
    const long KS = K >> 4;
    const long nb = nraw >> 5;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int g = gmap[2 * (k0 + 16 * half)];
        const half_t s = __builtin_bit_cast(half_t, scales[(long)g * N + n]);
        const half_t z = __builtin_bit_cast(half_t, zeros[(long)g * N + n]);
#pragma unroll
        for (int h8 = 0; h8 < 2; h8++) {
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int j = 16 * half + 8 * h8 + 2 * i;
                o[i] = (uint32_t)exl2_dq(q[j], s, z) | ((uint32_t)exl2_dq(q[j + 1], s, z) << 16);
            }
            img[(nb * KS + (k0 >> 4) + half) * 64 + h8 * 32 + (nraw & 31)] = uint4_t{o[0], o[1], o[2], o[3]};
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

// ---- exl2 decode GEMV (M <= 2; four rows in the list / group forms): one column per lane, the waves of a block interleave the 32-k chunks of its K slab ----
// Per chunk: `bits` coalesced dword loads, one v_and_or_b32 per pair (exl2_tpairs16), the sums on the matrix pipe and the group's
// scale / zero once per chunk in fp32 (exl2_gemv2_body has the derivation and the three ways x reaches the waves); K slabs are
// reduced in the kernel by tagged granules: no finalize launch.
// EX2_NW waves per workgroup: 16 (one workgroup per CU) when the column blocks alone fill most of the chip (measured 17.8 us
// against 19.4 us at 4096x11008), 8 (two per CU) + K slabs for narrower layers (4096x4096: 7.9 us against 9.1 us)
struct Exl2Call {  // everything one workgroup of the decode kernel needs (kernel arguments, or an entry of a device-resident list)
    const uint16_t* x; const uint32_t* qw; const uint16_t* scales; const uint16_t* zeros; const uint16_t* perm; const uint16_t* gmap;
    unsigned long long* gran; unsigned* gen; uint16_t* y;
    Exl2Rows rows;
    int M, K, N, chunks_per_slab, S, colblocks;
    int gfirst[6], glog[6];  // DIRECT form: first group of each band and log2(chunks per group) (bie_mbwq_exl2_shuffle: regular groups)
    const uint16_t* xp;      // list form, DMODE 2: x[q_perm] (written by exl2_list_permute_kernel in front)
    const float2_t* cs;      // list form, DMODE 2: per chunk {sum offset_k x_k, sum x_k} (same kernel)
};

// STAGED (two x rows): the slab's activations are gathered through q_perm ONCE per workgroup into LDS.  !STAGED (one row): every wave
// gathers the 32 activations of its own chunk together with the chunk's loads (wave-private LDS buffers) and the weight stream
// starts one round trip earlier -- measured: one row 8.0 / 16.5 us per layer against 8.9 / 18.3 staged (4096x4096 / 4096x11008),
// two rows 24.5 against 19.1 (profiles/r03_l_exl2_staged_x.txt).  Both forms are exact.
// NARROW: the tensor has no 8 / 6 / 5-bit rows (host: rows7[2] == 0): those bands' prefetch sets (4 x up to 8 words) are what sets the
// kernel's register count -- 151 with them (8-wave workgroups: ONE per CU), <= 128 without (two per CU).
// the power of two the field of pair j carries in the pairs of exl2_tpairs16 (chunk of `bits`-bit values, Exl2Lay<bits>)
__device__ __forceinline__ float exl2_offset_of(int bits, int j) {  // pair j of a chunk of `bits`-bit values (Exl2Lay<bits>)
    const int F = 16 / bits;
    if (j >= bits * F) return 1024.0f;
    const int p = (j % F) * bits;
    int b0 = 0;
    for (int q = 0; q <= p; q += bits)
        if (q + bits - b0 > 10) b0 = q;
    return (float)(1 << (10 - (p - b0)));
}

// DIRECT (one or two x rows, regular groups -- every band's groups hold the same power-of-two number of whole chunks, bie_mbwq_exl2_shuffle
// says so): NOTHING is staged.  A chunk's group is arithmetic on the band table, its 32 permutation indices are loaded by the wave
// itself one round of the prefetch ahead of the gather that needs them, and the first packed words are requested at kernel entry
// instead of behind a metadata round trip and a barrier.  Measured on the 32-layer list (4096x4096, 3/2-bit g32): the staging
// prologue alone cost 0.6 of 2.3 us per layer (profiles/r04_exl2_ablation.txt).
struct Exl2Groups {
    int gfirst[6], glog[6];
};
// DMODE 2 (the list form): x arrives ALREADY permuted (xp = x[q_perm], one small launch in front for all entries of the list --
// every column block of a layer needs the same 2 K bytes, gathering them per workgroup is 64-172 times redundant and a 32-lane
// gather touches up to 32 cache lines), chunk-major, together with the two column-independent sums of every chunk.  A wave requests its chunk's
// block of xp with the chunk's words -- one 8-byte load per lane, the matrix instruction's A operand as it is (see XW in the body): no index
// loads, no gather and no LDS in the loop.
typedef float exl2_acc_t __attribute__((ext_vector_type(4)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ exl2_acc_t exl2_mfma4(half4_t a, half4_t b, exl2_acc_t c) { return __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 0, 0, 0); }
// the same instruction with ONE block's A operand broadcast to all sixteen (cbsz = 4): lanes 4 ABID .. 4 ABID + 3 hold rows 0..3 of the 4 x 4 A block,
// every block multiplies it with its own B (tools/probe/probe_mfma_bcast.hip: semantics checked on gfx950)
template <int ABID>
__device__ __forceinline__ exl2_acc_t exl2_mfma4_bcast(half4_t a, half4_t b, exl2_acc_t c) { return __builtin_amdgcn_mfma_f32_4x4x4f16(a, b, c, 4, ABID, 0); }
template <int MT, int EX2_NW, bool STAGED, bool NARROW, int DMODE = 0>
__device__ __forceinline__ void exl2_gemv2_body(const uint16_t* __restrict__ x, const uint32_t* __restrict__ qw,
                                                const uint16_t* __restrict__ scales, const uint16_t* __restrict__ zeros,
                                                const uint16_t* __restrict__ perm, const uint16_t* __restrict__ gmap,
                                                unsigned long long* __restrict__ gran, unsigned* __restrict__ gen,
                                                uint16_t* __restrict__ y, const Exl2Rows rows, const int M, const int K, const int N,
                                                const int chunks_per_slab, const int S, const int colblock, const int slab_idx,
                                                const int colblocks, unsigned epoch, unsigned* status, unsigned tag_skew, int spin_limit,
                                                const Exl2Groups grp = Exl2Groups{}, const uint16_t* __restrict__ xp = nullptr,
                                                const float2_t* __restrict__ cs = nullptr) {
    constexpr bool DIRECT = DMODE != 0;
    constexpr bool XP = DMODE == 2;
    // XW (two rows and more): the kernel in front writes xp and the chunk sums CHUNK-MAJOR ([chunk][row][32], rows beyond M repeating row M - 1): a
    // chunk's activations for all rows are 64 MT contiguous bytes that the WAVE requests with the chunk's packed words -- no workgroup staging and no
    // barrier in front of the first weight request (a workgroup of a group call lives for four chunks per wave: the staging round trip was a third
    // of it), and no LDS limit on the slab (x for 16 rows x 4096 k would be 128 KiB).  The matrix instruction's A operand is the same 4 x 4 block of x
    // for all sixteen column blocks of the wave: it is BROADCAST by the instruction (cbsz = 4, abid = the block that holds it), so lane l = (block b,
    // row r) keeps in ONE register pair the four k of step b % 8 (row group b / 8) of row r -- the whole [MT][32] block of the chunk is one 8-byte
    // load per lane (two at sixteen rows), no LDS ring, no 16-byte LDS reads per step (they were the LDS pipe's load at 8 / 16 rows); the chunk's
    // {offset sum, x sum} pairs are wave-uniform: scalar loads.
    constexpr bool XW = XP;  // (an earlier form of DMODE 2 copied the workgroup's K slab of xp into LDS: a staging round trip in front of every workgroup's first weight
                             //  request and 16-byte LDS reads per step -- sibling groups 12.5 / 15.9 / 20.4 us at 1 / 2 / 4 rows against 11.6 / 11.9 / 12.3, lists at 4 / 8 / 16 rows
                             //  1.86 / 4.53 / 8.37 us per layer against 1.41 / 1.81 / 2.83: profiles/r04_exl2_ablation.txt)
    static_assert(!DIRECT || !STAGED, "the direct forms stage nothing");
    extern __shared__ __attribute__((aligned(16))) uint32_t smem2[];
    const int tid = threadIdx.x, lane = tid & 63;
    // the tile's generation is read ONCE, at kernel entry, by every wave: the reducer advances the word as soon as it is done, and a
    // wave that read it only in the epilogue could see the NEXT generation, tag with it and never be matched
    unsigned gen_entry = 0;
    if (S > 1) gen_entry = __hip_atomic_load(gen + colblock, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = colblock * 64 + lane;
    const int nl = n < N ? n : N - 1;
    const int C = K >> 5;
    const int c_begin = slab_idx * chunks_per_slab;
    int c_end = c_begin + chunks_per_slab;
    if (c_end > C) c_end = C;
    // Staged in LDS once per workgroup: the slab's group-map entries and either its activations (STAGED: gathered through q_perm, M
    // rows) or its permutation indices (a wave then gathers its own chunk's 32 activations with the chunk's loads).  A chunk's
    // group constants can thus be requested TOGETHER with its packed words.  (Fetched per chunk they were two dependent global loads
    // behind the words -- PMC: 58 % of the wave time waiting, profiles/r02_pmc_exl2.txt.)
    const int slab_k = chunks_per_slab * 32;
    uint16_t* x_s = reinterpret_cast<uint16_t*>(smem2);                       // STAGED: [MT][slab_k] fp16, q_perm applied
    uint16_t* xs = reinterpret_cast<uint16_t*>(smem2) + wave * (4 * MT * 32);  // !STAGED: wave-private [set 0..3][MT][32]
    uint16_t* perm_s = reinterpret_cast<uint16_t*>(smem2) + EX2_NW * (4 * MT * 32);  // !STAGED: [slab_k]
    uint16_t* gmap_s = STAGED ? x_s + MT * slab_k : perm_s + slab_k;           // [chunks_per_slab * 2]
    int two_groups = 0;  // does any chunk of the slab straddle two groups (group sizes below 32)?
    if constexpr (!DIRECT) {
        const int nk = (c_end - c_begin) * 32;
        for (int i = tid; i < nk; i += EX2_NW * 64) {
            const int kx = perm ? (int)perm[c_begin * 32 + i] : c_begin * 32 + i;
            if constexpr (STAGED) {
#pragma unroll
                for (int m = 0; m < MT; m++) x_s[m * slab_k + i] = x[(long)(m < M ? m : 0) * K + kx];
            } else {
                perm_s[i] = (uint16_t)kx;
            }
        }
        for (int i = tid; i < (c_end - c_begin); i += EX2_NW * 64) {
            const uint16_t ga = gmap[2 * ((c_begin + i) * 32)], gb = gmap[2 * ((c_begin + i) * 32 + 16)];
            gmap_s[2 * i] = ga;
            gmap_s[2 * i + 1] = gb;
            two_groups |= (ga != gb);
        }
        two_groups = __builtin_amdgcn_readfirstlane(__syncthreads_or(two_groups));  // the slab's metadata is in LDS; workgroup-uniform flag
    }
    float yacc[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) yacc[m] = 0.f;
    typedef const __attribute__((address_space(4))) float2_t cfloat2_t;
    // XW: where in a chunk's [MT][32] block this lane's A-operand pair lies (halves): row 4 (lane / 32) + lane % 4 (the last row again beyond MT), k = 4 ((lane / 4) % 8)
    const int xa_row = 4 * (lane >> 5) + (lane & 3);
    const int xa_off = (xa_row < MT ? xa_row : MT - 1) * 32 + 4 * ((lane >> 2) & 7);
    const int xrow = MT > 1 ? ((lane & 3) < M ? (lane & 3) : M - 1) : 0;  // the row of x this lane feeds the matrix pipe with (MT <= 4; more rows: per row group)
    const Exl2Magic magic;
    uint2_t ones2 = uint2_t{0x3c003c00u, 0x3c003c00u};
    asm("" : "+v"(ones2));  // matrix operands are registers
    const half4_t ones4 = __builtin_bit_cast(half4_t, ones2);
    // Packed rows and group constants through buffer descriptors: the row / group part of every address is wave-uniform (scalar
    // offset), the column part one register per element size -- no 64-bit vector address arithmetic per load (it was 1.2 VALU per
    // weight pair).  The host admits only tensors below 4 GB (exl2_buffer_ok).
    const auto rsrc_of = [](const void* p) {
        const uint64_t b = (uint64_t)(uintptr_t)p;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)hi << 32) | lo), 0, (int)0xfffffffcu, 0x00020000);
    };
    const auto rq = rsrc_of(qw), rs = rsrc_of(scales), rz = rsrc_of(zeros);
    const uint32_t col4 = (uint32_t)nl * 4u, col2 = (uint32_t)nl * 2u;
    const uint32_t row_bytes = (uint32_t)N * 4u, grow_bytes = (uint32_t)N * 2u;
    // The wave's chunks are taken BAND BY BAND: inside a band the bit width -- hence the number of loads per chunk -- is a compile-time
    // constant, every issue is unconditional (the look-ahead index is clamped to the band's last chunk: a harmless re-load), so the
    // compiler can wait with an exact vmcnt for the OLDEST chunk only.  (With the width switched at run time and `if (c < c_end)`
    // around the issues every wait came out as vmcnt(0..3): the 4-deep prefetch was really 1-deep and each of a wave's 8 chunks paid
    // a full memory latency -- PMC: 58 % of the wave time waiting.)
    const uint16_t* pp = perm ? perm : gmap;  // DIRECT without q_perm: the index load stays (any readable address), its value is not used
    uint32_t pmask = perm ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(pmask));  // opaque: keeps the two cases one code path
    // chunks [cb0, cb1) of this slab lie in one band starting at row prow0; DIRECT: the band's first chunk cbs is in group gbase, 2^glg chunks per group
    auto band = [&](auto bits_tag, auto two_tag, int cb0, int cb1, int prow0, int cbs, int gbase, int glg) {
        cb0 = __builtin_amdgcn_readfirstlane(cb0);  // scalars, said so: the row offsets below go into soffset
        cb1 = __builtin_amdgcn_readfirstlane(cb1);
        prow0 = __builtin_amdgcn_readfirstlane(prow0);
        cbs = __builtin_amdgcn_readfirstlane(cbs);
        gbase = __builtin_amdgcn_readfirstlane(gbase);
        glg = __builtin_amdgcn_readfirstlane(glg);
        constexpr int BITS = decltype(bits_tag)::value;
        constexpr bool TWO = decltype(two_tag)::value;  // group constants per 16-k half (two loads more per chunk) or per chunk
        uint32_t ofs[16];
        exl2_offset_pairs16<BITS>(magic, ofs);
        struct Chunk {
            uint32_t w[BITS];
            uint32_t s[2], z[2];
            uint32_t xraw[STAGED ? 1 : MT];  // !STAGED: this lane's gathered activation(s) of the chunk
            uint32_t p;                      // DIRECT: this lane's permutation index for the NEXT chunk of this set
            uint2_t xa[XW ? (MT + 7) / 8 : 1];  // XW: this lane's A-operand pair(s): row (lane % 4) of row group (lane / 32) [+ 2], k = 4 ((lane / 4) % 8) .. + 3
            float2_t csv[XW && MT <= 4 ? MT : 1];  // XW, up to four rows: {offset sum, x sum} of every row (wave-uniform: scalar registers)
            uint2_t csl;                           // XW, eight / sixteen rows (fp32 bit patterns): the pair of row lane % MT (32 / 64 scalar registers per two chunks in flight do not exist)
        };
        auto issue_w = [&](int c, Chunk& ch) {
            const int prow = prow0 + (c - cb0) * BITS;
#pragma unroll
            for (int i = 0; i < BITS; i++) ch.w[i] = __builtin_amdgcn_raw_buffer_load_b32(rq, col4, (uint32_t)(prow + i) * row_bytes, 2);  // nt
        };
        auto group_of = [&](int c, int half) -> int {
            if constexpr (DIRECT) return gbase + ((c - cbs) >> glg);
            else return __builtin_amdgcn_readfirstlane((int)gmap_s[2 * (c - c_begin) + half]);
        };
        auto load_perm = [&](int c, Chunk& ch) {
            const uint32_t idx = (uint32_t)c * 32u + (uint32_t)(lane & 31);
            ch.p = pp[idx];  // raw: whatever touches the value here would wait for the load here
        };
        auto issue_p = [&](int c, Chunk& ch) {
            const uint32_t g0 = (uint32_t)group_of(c, 0) * grow_bytes;
            ch.s[0] = __builtin_amdgcn_raw_buffer_load_b16(rs, col2, g0, 0);
            ch.z[0] = __builtin_amdgcn_raw_buffer_load_b16(rz, col2, g0, 0);
            if constexpr (TWO) {  // static load count per chunk either way
                const uint32_t g1 = (uint32_t)group_of(c, 1) * grow_bytes;
                ch.s[1] = __builtin_amdgcn_raw_buffer_load_b16(rs, col2, g1, 0);
                ch.z[1] = __builtin_amdgcn_raw_buffer_load_b16(rz, col2, g1, 0);
            }
            if constexpr (XW) {
                const uint16_t* xc = xp + (long)c * (MT * 32);
                ch.xa[0] = *reinterpret_cast<const uint2_t*>(xc + xa_off);
                if constexpr (MT > 8) ch.xa[1] = *reinterpret_cast<const uint2_t*>(xc + xa_off + 8 * 32);  // row groups 2, 3
                if constexpr (MT <= 4) {
                    const cfloat2_t* cc = (const cfloat2_t*)cs + (long)c * MT;  // constant address space: scalar loads (written by the kernel in front)
#pragma unroll
                    for (int m = 0; m < MT; m++) ch.csv[m] = cc[m];
                } else {
                    ch.csl = reinterpret_cast<const uint2_t*>(cs)[(long)c * MT + (lane & (MT - 1))];
                }
            }
            if constexpr (!STAGED && !XP) {
                int pidx;
                if constexpr (DIRECT) {  // v_bfi_b32, no branch on `perm` around a load (a branch costs a wait per load)
                    const uint32_t idx = (uint32_t)c * 32u + (uint32_t)(lane & 31);
                    pidx = (int)((ch.p & pmask) | (idx & ~pmask));
                }
                else pidx = (int)perm_s[(c - c_begin) * 32 + (lane & 31)];
#pragma unroll
                for (int m = 0; m < MT; m++) ch.xraw[m] = x[(long)(m < M ? m : 0) * K + pidx];
            }
        };
        auto compute = [&](int set, int c, const Chunk& ch) {
            const uint16_t* xw = nullptr;
            int xstride = 0;
            const float2_t* csw = nullptr;
            // GB (one row, gathered per chunk): lane l < 32 holds x[q_perm[32 c + l]]; two DPP moves inside each quad put the four k of the quad into the
            // register pair of its first lane -- lane 4 b = row 0 of block b -- and the matrix instruction broadcasts block b's A operand for step b
            // (cbsz = 4, abid = b): no LDS round trip (2-byte store, wave barrier, four 16-byte reads per chunk) between the gather and the matrix pipe
            constexpr bool GB = !STAGED && !XP && MT == 1;
            uint2_t agb = uint2_t{0u, 0u};
            if constexpr (XW) {
                csw = ch.csv;
            } else if constexpr (STAGED) {
                xw = x_s + (c - c_begin) * 32;
                xstride = slab_k;
            } else if constexpr (GB) {
                const int xr = (int)ch.xraw[0];
                const int nb = __builtin_amdgcn_mov_dpp(xr, 0xF5, 0xf, 0xf, true);       // quad_perm [1, 1, 3, 3]: the neighbour's value
                const uint32_t pr = (uint32_t)xr | ((uint32_t)nb << 16);                  // lanes 0 / 2 of a quad: the pairs (k, k + 1) / (k + 2, k + 3)
                agb = uint2_t{pr, (uint32_t)__builtin_amdgcn_mov_dpp((int)pr, 0xAA, 0xf, 0xf, true)};  // quad_perm [2, 2, 2, 2]
            } else {
                uint16_t* xwr = xs + set * (MT * 32);
                if (lane < 32) {
#pragma unroll
                    for (int m = 0; m < MT; m++) xwr[m * 32 + lane] = (uint16_t)ch.xraw[m];
                }
                // same wave writes then reads: the LDS pipe keeps a wave's operations in order, the compiler must too (the 2-byte
                // stores and the 16-byte loads below have different types)
                asm volatile("" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                xw = xwr;
                xstride = 32;
            }
            uint32_t w8[8];
#pragma unroll
            for (int i = 0; i < 8; i++) w8[i] = i < BITS ? ch.w[i] : 0u;
            uint32_t T[16];
            // Everything after the field extraction runs on the MATRIX pipe.  v_mfma_f32_4x4x4_16B_f16 is sixteen independent 4x4x4
            // products: lane l = (block l / 4, column l % 4) supplies four k of ITS OWN column as the B operand -- the lane-per-column
            // layout of this kernel as it is -- and row l % 4 of x as the A operand, and receives D[0..3][its column]: up to four rows
            // of x for one instruction per four k (2 passes).  Three accumulations per chunk with the same A operands:
            //   dq = sum (offset_k + q_k) x_k   (B = the pairs as v_and_or_b32 left them: exact fp16 integers, exact products)
            //   dc = sum  offset_k        x_k   (B = the band's offset pattern, column-independent)
            //   dx = sum                  x_k   (B = 1.0)
            // and the group's constants are applied ONCE per chunk in fp32:  y += s (dq - dc) - z dx  ==  sum (q s - z) x.
            // Against the reference's per-weight fp16 rounding r = fp16(q s - z) (exl2/q_gemm_kernel.cuh:dot22_*) this differs by at
            // most half an fp16 ulp of each weight times |x| -- below what its own fp16 accumulation loses; the reconstruction
            // kernel keeps the exact rounding.  The vector ALU is left with the 16 v_and_or_b32 (+ shifts) per chunk: 1 / 3 of the
            // instructions of the per-weight form, which was issue-bound (profiles/r04_exl2_list_pmc_xp.txt).
            // Rows beyond M read row M - 1 again and are never stored.
            const exl2_acc_t zero4 = exl2_acc_t{0.f, 0.f, 0.f, 0.f};
            exl2_acc_t dq = zero4, dc = zero4, dx = zero4;
            if constexpr (XP && MT <= 4) {
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    const float2_t v = csw[m];
                    dc[m] = v.x;
                    dx[m] = v.y;
                }
            }
            auto apply = [&](int half) {  // the constants of the group the half (or the whole chunk) lies in
                const float sf = (float)__builtin_bit_cast(half_t, (uint16_t)ch.s[half]);
                const float zf = (float)__builtin_bit_cast(half_t, (uint16_t)ch.z[half]);
#pragma unroll
                for (int m = 0; m < (MT <= 4 ? MT : 4); m++) {
                    yacc[m] = __builtin_fmaf(sf, dq[m] - dc[m], yacc[m]);
                    yacc[m] = __builtin_fmaf(-zf, dx[m], yacc[m]);
                }
            };
            if constexpr (MT > 4) {
                // Eight / sixteen rows of x (the pre-permuted form only): MT / 4 matrix instructions per four k -- row group rg feeds rows 4 rg .. 4 rg + 3
                // as the A operand (lane l: row 4 rg + l % 4) against the SAME B operand, one extraction for all rows.
                static_assert(MT <= 4 || (XP && !TWO), "more than four rows: the pre-permuted form");
                constexpr int RG = MT / 4;
                exl2_acc_t dqr[RG];
#pragma unroll
                for (int rg = 0; rg < RG; rg++) dqr[rg] = zero4;
                exl2_static_for<0, 2>([&](auto hh) {
                    constexpr int half = decltype(hh)::value;
                    exl2_tpairs16<BITS, 8 * half, 8 * half + 8>(w8, magic, T);
                    exl2_static_for<0, RG>([&](auto rr) {
                        constexpr int rg = decltype(rr)::value;
                        exl2_static_for<0, 4>([&](auto ii) {
                            constexpr int i = decltype(ii)::value;
                            constexpr int j = 8 * half + 2 * i;
                            dqr[rg] = exl2_mfma4_bcast<(rg & 1) * 8 + 4 * half + i>(__builtin_bit_cast(half4_t, ch.xa[rg >> 1]), __builtin_bit_cast(half4_t, uint2_t{T[j], T[j + 1]}), dqr[rg]);
                        });
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
                const float sf = (float)__builtin_bit_cast(half_t, (uint16_t)ch.s[0]);
                const float zf = (float)__builtin_bit_cast(half_t, (uint16_t)ch.z[0]);
#pragma unroll
                for (int m = 0; m < MT; m++) {
                    const float so = __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)ch.csl.x, m));  // row m's pair sits in lane m
                    const float sx = __builtin_bit_cast(float, __builtin_amdgcn_readlane((int)ch.csl.y, m));
                    yacc[m] = __builtin_fmaf(sf, dqr[m >> 2][m & 3] - so, yacc[m]);
                    yacc[m] = __builtin_fmaf(-zf, sx, yacc[m]);
                }
                return;
            }
            // half a chunk at a time -- extraction of 8 pairs, their 4 (x 3) matrix instructions -- so that 8, not 16, operand pairs are live
            exl2_static_for<0, 2>([&](auto hh) {
                constexpr int half = decltype(hh)::value;
                exl2_tpairs16<BITS, 8 * half, 8 * half + 8>(w8, magic, T);
                if constexpr (XW) {
                    exl2_static_for<0, 4>([&](auto ii) {
                        constexpr int i = decltype(ii)::value;
                        constexpr int j = 8 * half + 2 * i;
                        dq = exl2_mfma4_bcast<4 * half + i>(__builtin_bit_cast(half4_t, ch.xa[0]), __builtin_bit_cast(half4_t, uint2_t{T[j], T[j + 1]}), dq);
                    });
                } else if constexpr (GB) {
                    exl2_static_for<0, 4>([&](auto ii) {
                        constexpr int i = decltype(ii)::value;
                        constexpr int j = 8 * half + 2 * i;
                        const half4_t av = __builtin_bit_cast(half4_t, agb);
                        dq = exl2_mfma4_bcast<4 * half + i>(av, __builtin_bit_cast(half4_t, uint2_t{T[j], T[j + 1]}), dq);
                        dc = exl2_mfma4_bcast<4 * half + i>(av, __builtin_bit_cast(half4_t, uint2_t{ofs[j], ofs[j + 1]}), dc);
                        dx = exl2_mfma4_bcast<4 * half + i>(av, ones4, dx);
                    });
                } else {
                const uint4_t xa = reinterpret_cast<const uint4_t*>(xw + xrow * xstride)[2 * half], xb = reinterpret_cast<const uint4_t*>(xw + xrow * xstride)[2 * half + 1];
                const uint32_t xd[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const half4_t av = __builtin_bit_cast(half4_t, uint2_t{xd[2 * i], xd[2 * i + 1]});
                    const int j = 8 * half + 2 * i;
                    dq = exl2_mfma4(av, __builtin_bit_cast(half4_t, uint2_t{T[j], T[j + 1]}), dq);
                    if constexpr (!XP) {  // DMODE 2 has the two column-independent sums from the kernel that permuted x
                        dc = exl2_mfma4(av, __builtin_bit_cast(half4_t, uint2_t{ofs[j], ofs[j + 1]}), dc);
                        dx = exl2_mfma4(av, ones4, dx);
                    }
                }
                }
                if constexpr (TWO) {
                    apply(half);
                    dq = zero4; dc = zero4; dx = zero4;
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if constexpr (!TWO) apply(0);
        };
        // this wave's chunks in the band: c = first, first + NW, ... < cb1
        int first = c_begin + wave;
        if (first < cb0) first += ((cb0 - first + EX2_NW - 1) / EX2_NW) * EX2_NW;
        if (first >= cb1) return;
        const int cnt = (cb1 - first + EX2_NW - 1) / EX2_NW;
        const int last = first + (cnt - 1) * EX2_NW;
        auto at = [&](int jj) { const int c = first + jj * EX2_NW; return __builtin_amdgcn_readfirstlane(c < last ? c : last); };  // clamped look-ahead; a scalar
        // words, group constants, the gather through the indices this set was given a round ago -- then the indices of its next chunk
        auto issue = [&](int c, int c_next, Chunk& ch) {
            issue_w(c, ch);
            issue_p(c, ch);
            if constexpr (DIRECT && !XP) load_perm(c_next, ch);
        };
        // Chunks in flight: as many as fit ~16 packed words (4 of <= 4-bit, 3 of 5-bit, 2 of 6- / 8-bit chunks): the same bytes in flight for
        // every band, and the register count of the kernel no longer set by its widest band (4 x 8 words: 32 registers for a band most
        // tensors do not have -- the mixed 6 / 5 / 4-bit instances ran at 96-128 registers, two workgroups per CU, half the rate of the
        // 3 / 2-bit ones).
        constexpr int DEPTH = (BITS >= 6 || (XW && MT > 1)) ? 2 : (BITS == 5 ? 3 : 4);  // the per-wave x form: two (its x rows and, from eight rows, the accumulators take the registers of the other two sets; measured with 3 / 4: slower)
        Chunk cs[DEPTH];
        if constexpr (DIRECT && !XP) {
            exl2_static_for<0, DEPTH>([&](auto i) { load_perm(at(decltype(i)::value), cs[decltype(i)::value]); });
            exl2_static_for<0, DEPTH>([&](auto i) { issue_w(at(decltype(i)::value), cs[decltype(i)::value]); });  // the first words do not wait for the indices
            exl2_static_for<0, DEPTH>([&](auto i) {
                constexpr int I = decltype(i)::value;
                issue_p(at(I), cs[I]);
                load_perm(at(I + DEPTH), cs[I]);
            });
        } else {
            exl2_static_for<0, DEPTH>([&](auto i) { issue(at(decltype(i)::value), 0, cs[decltype(i)::value]); });
        }
        int jj = 0;
        for (; jj + DEPTH < cnt; jj += DEPTH) {  // a further round follows: DEPTH full steps, each re-filling the set it has just consumed
            exl2_static_for<0, DEPTH>([&](auto i) {
                constexpr int I = decltype(i)::value;
                compute(I, at(jj + I), cs[I]);
                issue(at(jj + DEPTH + I), at(jj + 2 * DEPTH + I), cs[I]);
            });
        }
        exl2_static_for<0, DEPTH>([&](auto i) {  // last round: nothing left to request
            constexpr int I = decltype(i)::value;
            if (jj + I < cnt) compute(I, at(jj + I), cs[I]);
        });
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
            const int cbs = cb0;
            const int skip = cb0 < c_begin ? c_begin - cb0 : 0;  // clip to the slab
            cb0 += skip;
            if (cb1 > c_end) cb1 = c_end;
            if (cb0 < cb1) {
                const int p0 = prow_band + skip * bits;
#define BIE_BAND(B)                                                                        \
    do {                                                                                   \
        if constexpr (DIRECT) band(std::integral_constant<int, B>{}, std::false_type{}, cb0, cb1, p0, cbs, grp.gfirst[b], grp.glog[b]); \
        else if (two_groups) band(std::integral_constant<int, B>{}, std::true_type{}, cb0, cb1, p0, 0, 0, 0); \
        else band(std::integral_constant<int, B>{}, std::false_type{}, cb0, cb1, p0, 0, 0, 0);      \
    } while (0)
                if constexpr (NARROW) {
                    switch (b) {
                        case 3: BIE_BAND(4); break;
                        case 4: BIE_BAND(3); break;
                        case 5: BIE_BAND(2); break;
                        default: break;  // the host checked: no such rows
                    }
                } else {
                    switch (b) {
                        case 0: BIE_BAND(8); break;
                        case 1: BIE_BAND(6); break;
                        case 2: BIE_BAND(5); break;
                        case 3: BIE_BAND(4); break;
                        case 4: BIE_BAND(3); break;
                        default: BIE_BAND(2); break;
                    }
                }
#undef BIE_BAND
            }
        }
    }
    // block reduction over the waves in wave order (deterministic)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem2);
#pragma unroll
    for (int m = 0; m < MT; m++) red[(wave * MT + m) * 64 + lane] = yacc[m];
    __syncthreads();
    // wave w finishes rows w, w + EX2_NW, ... (eight and sixteen rows of x: more rows than waves)
    for (int om = tid >> 6; om < MT; om += EX2_NW) {
        const int ol = tid & 63, on = colblock * 64 + ol;
        float tot = 0.f;
#pragma unroll
        for (int wv = 0; wv < EX2_NW; wv++) tot += red[(wv * MT + om) * 64 + ol];
        if (S > 1) {
            // K slabs: the tagged-granule reduction of the lookup GEMV (mpq_gemv_lut.hip) -- slabs 0..S-2 publish {fp32, tag} with
            // one write-through store per column and retire; the last slab's workgroup (highest block ids: dispatched after every
            // publisher, which never waits) polls them and adds in slab order.  No finalize launch, no atomics, deterministic.
            const unsigned gen_next = gen_entry + 1u;
            const unsigned tag = epoch | (gen_next & 0xffu);
            const long ncat = (long)colblocks * 64, col = (long)colblock * 64 + ol;
            const int slab = slab_idx;
            if (slab != S - 1) {
                const unsigned long long gval = ((unsigned long long)tag << 32) | __float_as_uint(tot);
                __hip_atomic_store(gran + ((long)slab * MT + om) * ncat + col, gval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                continue;
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
            if (tid == 0) gen[colblock] = gen_next;  // the next launch (or a replay of this one) tags differently (a kernel boundary lies between)
        }
        if (om < M && on < N) y[(long)om * N + on] = f32_to_f16_bits(tot);
    }
}


template <int MT, int EX2_NW, bool NARROW, int DMODE>
__global__ __launch_bounds__(EX2_NW * 64, (EX2_NW == 8 ? 4 : 1)) void exl2_gemv2_kernel(const Exl2Call c, unsigned epoch, unsigned* status, unsigned tag_skew,
                                                                                          int spin_limit) {
    Exl2Groups grp;
#pragma unroll
    for (int i = 0; i < 6; i++) { grp.gfirst[i] = c.gfirst[i]; grp.glog[i] = c.glog[i]; }
    exl2_gemv2_body<MT, EX2_NW, (MT > 1 && DMODE == 0), NARROW, DMODE>(c.x, c.qw, c.scales, c.zeros, c.perm, c.gmap, c.gran, c.gen, c.y, c.rows, c.M, c.K, c.N,
                                          c.chunks_per_slab, c.S, (int)blockIdx.x, (int)blockIdx.y, c.colblocks, epoch, status, tag_skew, spin_limit, grp);
}

// ---- exl2 for 3 <= M <= 64 on the matrix pipe (the reference keeps these rows in its fused kernel, exl2/q_gemm_kernel.cuh:90-549;
// round 2 handed M >= 9 to reconstruct + a library GEMM: 35-52 us per layer) ------------------------------------------------------
// The load / extract / dequant half is the decode kernel's: lane = column, a wave takes 32-k chunks band by band with a 4-deep
// prefetch, v_pk_fma_f16 reproduces __hfma2(q, s, -z).  A lane then holds the 32 fp16 weights of ITS column; v_mfma_f32_16x16x32_f16
// wants lane (c, kb) to hold 8 consecutive k of column c, so the chunk goes through a wave-private LDS buffer (64 columns x 80 bytes:
// four 16-byte writes per lane, four 16-byte reads back, conflict-free -- the pitch is an odd number of 16-byte units; a wave's LDS
// operations complete in order, no barrier).  x is permuted ONCE per call by exl2_permute_x_kernel (x[:, q_perm] into the workspace;
// gathering the slab per workgroup as the decode kernel does cost 45 M two-byte loads at 64 x 4096 x 11008: 193 us); its fragments --
// lane (m, kb): 16 bytes -- are requested with the chunk's words, per chunk 4 * MB MFMAs (D = W_frag x x_frag: rows = columns n,
// columns = x rows m; rows >= M repeat row M - 1 and are never stored).  Waves are reduced through LDS 16 rows at a time, K slabs by
// the decode kernel's tagged granules.
__global__ __launch_bounds__(256) void exl2_permute_x_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ perm,
                                                             uint16_t* __restrict__ xp, int M, int K) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const int kx = perm[k];
    for (int m = blockIdx.y; m < M; m += gridDim.y) xp[(long)m * K + k] = x[(long)m * K + kx];
}

// The same for MANY rows (prefill): a workgroup stages whole rows of x in LDS with 16-byte loads, gathers from LDS (two-byte reads, any
// order) and writes 16 bytes per lane -- HBM sees x once and xp once, both streamed (the per-element form above touches a 64-byte line per
// two-byte read: 41 us at 4096 x 4096 against the ~13 us of 64 MB at the streaming rate).  K % 8 == 0; rows of up to 32768 elements.
__global__ __launch_bounds__(256) void exl2_permute_rows_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ perm,
                                                                uint16_t* __restrict__ xp, int M, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pr[];
    uint16_t* row = reinterpret_cast<uint16_t*>(smem_pr);
    const int K8 = K >> 3;
    for (int m = blockIdx.x; m < M; m += gridDim.x) {
        const uint4_t* src = reinterpret_cast<const uint4_t*>(x + (long)m * K);
        for (int i = threadIdx.x; i < K8; i += 256) reinterpret_cast<uint4_t*>(row)[i] = src[i];
        __syncthreads();
        uint4_t* dst = reinterpret_cast<uint4_t*>(xp + (long)m * K);
        for (int i = threadIdx.x; i < K8; i += 256) {
            const uint4_t p = reinterpret_cast<const uint4_t*>(perm)[i];  // eight source columns
            uint4_t o;
            o.x = (uint32_t)row[p.x & 0xffffu] | ((uint32_t)row[p.x >> 16] << 16);
            o.y = (uint32_t)row[p.y & 0xffffu] | ((uint32_t)row[p.y >> 16] << 16);
            o.z = (uint32_t)row[p.z & 0xffffu] | ((uint32_t)row[p.z >> 16] << 16);
            o.w = (uint32_t)row[p.w & 0xffffu] | ((uint32_t)row[p.w >> 16] << 16);
            dst[i] = o;
        }
        __syncthreads();
    }
}

constexpr int EXL2_T_PITCH = 40;  // halves per column of the transpose buffer (80 bytes)

// NARROW: no 8 / 6 / 5-bit rows in this tensor (host: rows7[2] == 0) -- the wide bands' prefetch sets (4 x 8 words) cost the 3/2-bit models
// 40-60 registers they never use; OCC = workgroups per CU the register budget is capped for.
template <int MB, int EX2_NW, int OCC, bool NARROW>
__global__ __launch_bounds__(EX2_NW * 64, OCC) void exl2_mfma_kernel(const Exl2Call c0, unsigned epoch, unsigned* status, unsigned tag_skew, int spin_limit) {
    const uint16_t* __restrict__ x = c0.x;
    const uint32_t* __restrict__ qw = c0.qw;
    const uint16_t* __restrict__ scales = c0.scales;
    const uint16_t* __restrict__ zeros = c0.zeros;
    const uint16_t* __restrict__ gmap = c0.gmap;
    const Exl2Rows rows = c0.rows;
    const int M = c0.M, K = c0.K, N = c0.N, chunks_per_slab = c0.chunks_per_slab, S = c0.S;
    const int colblock = blockIdx.x, slab_idx = blockIdx.y;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem2[];
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned gen_entry = 0;
    if (S > 1) gen_entry = __hip_atomic_load(c0.gen + colblock, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // once, at entry (see the decode kernel)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = colblock * 64 + lane;
    const int nl = n < N ? n : N - 1;
    const int C = K >> 5;
    const int c_begin = slab_idx * chunks_per_slab;
    int c_end = c_begin + chunks_per_slab;
    if (c_end > C) c_end = C;
    uint16_t* T = reinterpret_cast<uint16_t*>(smem2) + wave * (64 * EXL2_T_PITCH);          // wave-private [64 columns][EXL2_T_PITCH]
    uint16_t* gmap_s = reinterpret_cast<uint16_t*>(smem2) + EX2_NW * (64 * EXL2_T_PITCH);   // [chunks_per_slab * 2]
    const Exl2Magic magic;
    int two_groups = 0;
    {
        for (int i = tid; i < (c_end - c_begin); i += EX2_NW * 64) {
            const uint16_t ga = gmap[2 * ((c_begin + i) * 32)], gb = gmap[2 * ((c_begin + i) * 32 + 16)];
            gmap_s[2 * i] = ga;
            gmap_s[2 * i + 1] = gb;
            two_groups |= (ga != gb);
        }
        two_groups = __syncthreads_or(two_groups);
    }
    exl2_acc_t acc[4][MB];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int rb = 0; rb < MB; rb++) acc[j][rb] = exl2_acc_t{0.f, 0.f, 0.f, 0.f};
    const int c16 = lane & 15, kb = lane >> 4;
    const uint16_t* xrow[MB];  // x here is the PERMUTED activation matrix [M][K]
#pragma unroll
    for (int rb = 0; rb < MB; rb++) xrow[rb] = x + (long)min(16 * rb + c16, M - 1) * K + 8 * kb;
    auto band = [&](auto bits_tag, auto two_tag, int cb0, int cb1, int prow0) {
        constexpr int BITS = decltype(bits_tag)::value;
        constexpr bool TWO = decltype(two_tag)::value;
        struct Chunk {
            uint32_t w[BITS];
            uint32_t s[2], z[2];
        };
        auto group_of = [&](int c, int half) -> int { return __builtin_amdgcn_readfirstlane((int)gmap_s[2 * (c - c_begin) + half]); };
        auto issue = [&](int c, Chunk& ch) {
            const int prow = prow0 + (c - cb0) * BITS;
#pragma unroll
            for (int i = 0; i < BITS; i++) ch.w[i] = __builtin_nontemporal_load(qw + (long)(prow + i) * N + nl);
            const int g0 = group_of(c, 0);
            ch.s[0] = scales[(long)g0 * N + nl];
            ch.z[0] = zeros[(long)g0 * N + nl];
            if constexpr (TWO) {
                const int g1 = group_of(c, 1);
                ch.s[1] = scales[(long)g1 * N + nl];
                ch.z[1] = zeros[(long)g1 * N + nl];
            }
        };
        auto compute = [&](int c, const Chunk& ch) {
            uint4_t xf[MB];  // L2-resident: requested here, used after the ~100 VALU of the extraction below (one set, not one per prefetch slot)
#pragma unroll
            for (int rb = 0; rb < MB; rb++) xf[rb] = *reinterpret_cast<const uint4_t*>(xrow[rb] + c * 32);
            uint32_t w8[8];
#pragma unroll
            for (int i = 0; i < 8; i++) w8[i] = i < BITS ? ch.w[i] : 0u;
            half2_t Q[16];
            exl2_qpairs16<BITS>(w8, magic, Q);
            uint32_t r[16];
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const half_t sh = __builtin_bit_cast(half_t, (uint16_t)ch.s[TWO ? half : 0]);
                const half_t zh = __builtin_bit_cast(half_t, (uint16_t)ch.z[TWO ? half : 0]);
                const half2_t s2 = half2_t{sh, sh}, nz2 = half2_t{(half_t)-zh, (half_t)-zh};
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    r[8 * half + i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(Q[8 * half + i], s2, nz2));  // == __hfma2(q, s, -z)
                }
            }
            uint4_t* tw = reinterpret_cast<uint4_t*>(T + lane * EXL2_T_PITCH);
#pragma unroll
            for (int i = 0; i < 4; i++) tw[i] = uint4_t{r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]};
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            uint4_t wf[4];
#pragma unroll
            for (int j = 0; j < 4; j++) wf[j] = *reinterpret_cast<const uint4_t*>(T + (16 * j + c16) * EXL2_T_PITCH + 8 * kb);
#pragma unroll
            for (int rb = 0; rb < MB; rb++) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    acc[j][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, wf[j]), __builtin_bit_cast(half8_t, xf[rb]), acc[j][rb], 0, 0, 0);
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        };
        int first = c_begin + wave;
        if (first < cb0) first += ((cb0 - first + EX2_NW - 1) / EX2_NW) * EX2_NW;
        if (first >= cb1) return;
        const int cnt = (cb1 - first + EX2_NW - 1) / EX2_NW;
        const int last = first + (cnt - 1) * EX2_NW;
        auto at = [&](int jj) { const int c = first + jj * EX2_NW; return c < last ? c : last; };
        Chunk c0_, c1_, c2_, c3_;
        issue(at(0), c0_);
        issue(at(1), c1_);
        issue(at(2), c2_);
        issue(at(3), c3_);
        int jj = 0;
        for (; jj + 4 < cnt; jj += 4) {
            compute(at(jj), c0_);
            issue(at(jj + 4), c0_);
            compute(at(jj + 1), c1_);
            issue(at(jj + 5), c1_);
            compute(at(jj + 2), c2_);
            issue(at(jj + 6), c2_);
            compute(at(jj + 3), c3_);
            issue(at(jj + 7), c3_);
        }
        compute(at(jj), c0_);
        if (jj + 1 < cnt) compute(at(jj + 1), c1_);
        if (jj + 2 < cnt) compute(at(jj + 2), c2_);
        if (jj + 3 < cnt) compute(at(jj + 3), c3_);
    };
    {
        int kprev = 0, prow = 0;
#pragma unroll
        for (int b = 0; b < 6; b++) {
            const int khi = rows.r[b];
            const int bits = exl2_bits_of_band(b);
            int cb0 = kprev >> 5, cb1 = khi >> 5;
            const int prow_band = prow;
            prow += (cb1 - cb0) * bits;
            kprev = khi;
            const int skip = cb0 < c_begin ? c_begin - cb0 : 0;
            cb0 += skip;
            if (cb1 > c_end) cb1 = c_end;
            if (cb0 < cb1) {
                const int p0 = prow_band + skip * bits;
#define BIE_BAND(B)                                                                        \
    do {                                                                                   \
        if (two_groups) band(std::integral_constant<int, B>{}, std::true_type{}, cb0, cb1, p0); \
        else band(std::integral_constant<int, B>{}, std::false_type{}, cb0, cb1, p0);      \
    } while (0)
                if constexpr (NARROW) {
                    switch (b) {
                        case 3: BIE_BAND(4); break;
                        case 4: BIE_BAND(3); break;
                        case 5: BIE_BAND(2); break;
                        default: break;  // the host checked: no such rows
                    }
                } else {
                    switch (b) {
                        case 0: BIE_BAND(8); break;
                        case 1: BIE_BAND(6); break;
                        case 2: BIE_BAND(5); break;
                        case 3: BIE_BAND(4); break;
                        case 4: BIE_BAND(3); break;
                        default: BIE_BAND(2); break;
                    }
                }
#undef BIE_BAND
            }
        }
    }
    // ---- reduction over the waves, 16 x rows at a time.  D layout of the 16x16 MFMA: lane (kb', m) holds in acc[j][rb][r] the output of
    // column 16 j + 4 kb' + r for x row 16 rb + m.  red[wave][16][64]
    float* red = reinterpret_cast<float*>(smem2);
    const unsigned gen_next = gen_entry + 1u;
    const unsigned tag = epoch | (gen_next & 0xffu);
    const long ncat = (long)c0.colblocks * 64;
    __syncthreads();
#pragma unroll
    for (int rb = 0; rb < MB; rb++) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            *reinterpret_cast<float4_t*>(red + ((wave * 16 + c16) * 64 + 16 * j + 4 * kb)) = float4_t{acc[j][rb][0], acc[j][rb][1], acc[j][rb][2], acc[j][rb][3]};
        __syncthreads();
        for (int o = tid; o < 16 * 64; o += EX2_NW * 64) {
            const int om = o >> 6, ol = o & 63, m = 16 * rb + om, on = colblock * 64 + ol;
            float tot = 0.f;
#pragma unroll
            for (int wv = 0; wv < EX2_NW; wv++) tot += red[(wv * 16 + om) * 64 + ol];
            if (S > 1 && m < M) {  // m < M is wave-uniform (a wave covers one om)
                const long col = (long)colblock * 64 + ol;
                if (slab_idx != S - 1) {
                    const unsigned long long gval = ((unsigned long long)tag << 32) | __float_as_uint(tot);
                    __hip_atomic_store(c0.gran + ((long)slab_idx * M + m) * ncat + col, gval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    continue;
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
                            gv[jj] = __hip_atomic_load(c0.gran + ((long)sidx * M + m) * ncat + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
                    if (!ready) {  // never a silent number
                        v = __uint_as_float(0x7fc00000u);
                        if (ol == 0 && status) __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
                tot = v + tot;
            }
            if (m < M && on < N) c0.y[(long)m * N + on] = f32_to_f16_bits(tot);
        }
        __syncthreads();
    }
    if (S > 1 && slab_idx == S - 1 && tid == 0) c0.gen[colblock] = gen_next;
}

// K slabs of the matrix-pipe kernel: ~512 workgroups (column blocks x slabs), every wave at least 4 chunks (the prefetch depth)
static void exl2_mfma_plan(int M, int K, int N, int& cps, int& S, int& mb, int& nw) {
    mb = cdiv(M, 16);
    nw = mb >= 2 ? 4 : 8;  // 32-64 accumulator + 8-16 x-fragment registers per prefetch set: one wave per SIMD (512 registers)
    const int C = K / 32;
    const int colblocks = cdiv(N, 64);
    const int want_s = cdiv(nw == 8 ? 512 : 1024, colblocks);
    cps = cdiv(C, want_s);
    if (cps < 4 * nw) cps = 4 * nw;
    cps = cdiv(cps, nw) * nw;
    if (cps > 2048) cps = 2048;  // the group-map copy in LDS
    if (cps > C) cps = C;
    S = cdiv(C, cps);
}
static size_t exl2_mfma_lds_bytes(int mb, int nw, int cps) {
    const size_t a = (size_t)nw * 64 * EXL2_T_PITCH * 2 + (size_t)cps * 4;
    const size_t red = (size_t)nw * 16 * 64 * 4;
    return a > red ? a : red;
}
bool exl2_mfma_ok(int M, int K, int N) { return M >= 3 && M <= 64 && K % 32 == 0 && cdiv(N, 64) <= BIE_WS_COUNTERS; }
size_t exl2_mfma_granule_bytes(int M, int K, int N) {
    if (!exl2_mfma_ok(M, K, N)) return 0;
    int cps, S, mb, nw;
    exl2_mfma_plan(M, K, N, cps, S, mb, nw);
    return (S > 1 ? (size_t)(S - 1) * M * cdiv(N, 64) * 64 * 8 : 0) + 256 + (size_t)M * K * 2;  // granules, then x[:, q_perm]
}
static size_t exl2_mfma_xp_offset(int M, int K, int N) {
    int cps, S, mb, nw;
    exl2_mfma_plan(M, K, N, cps, S, mb, nw);
    const size_t g = S > 1 ? (size_t)(S - 1) * M * cdiv(N, 64) * 64 * 8 : 0;
    return (g + 255) / 256 * 256;
}

// ONE launch over a LIST of exl2 layers (bie_mbwq_exl2_list_*): block b -> {entry, column block | slab << 20} through a device table
// (the MPQ list's idea, mpq_list.hip): a 4096x4096 3/2-bit layer is 5 MB -- far too little for a launch of its own.
template <int MT, bool NARROW, int DMODE>
__device__ __forceinline__ void exl2_list_body(const Exl2Call* __restrict__ ent, const uint2_t* __restrict__ blk, unsigned epoch, unsigned* status,
                                               unsigned tag_skew, int spin_limit) {
    typedef const __attribute__((address_space(4))) uint2_t cu2_t;
    typedef const __attribute__((address_space(4))) Exl2Call ccall_t;
    const uint2_t rec = *((cu2_t*)(uintptr_t)(blk + blockIdx.x));
    ccall_t* c = (ccall_t*)(uintptr_t)(ent + rec.x);
    Exl2Rows rows;
#pragma unroll
    for (int i = 0; i < 6; i++) rows.r[i] = c->rows.r[i];
    Exl2Groups grp;
#pragma unroll
    for (int i = 0; i < 6; i++) { grp.gfirst[i] = c->gfirst[i]; grp.glog[i] = c->glog[i]; }
    exl2_gemv2_body<MT, 8, (MT > 1 && DMODE == 0), NARROW, DMODE>(c->x, c->qw, c->scales, c->zeros, c->perm, c->gmap, c->gran, c->gen, c->y, rows, c->M, c->K, c->N,
                                             c->chunks_per_slab, c->S, (int)(rec.y & 0xfffffu), (int)(rec.y >> 20), c->colblocks, epoch, status,
                                             tag_skew, spin_limit, grp, c->xp, c->cs);
}
template <int MT, bool NARROW, int DMODE>
__global__ __launch_bounds__(512, ((DMODE == 2 && MT == 1) ? 6 : 4)) void exl2_list_kernel(const Exl2Call* __restrict__ ent, const uint2_t* __restrict__ blk, unsigned epoch,
                                                           unsigned* status, unsigned tag_skew, int spin_limit) {
    exl2_list_body<MT, NARROW, DMODE>(ent, blk, epoch, status, tag_skew, spin_limit);
}
// The kernel in front of a DMODE 2 list launch, for every entry (and every row of x): xp = x[q_perm] (or a copy), and per 32-k chunk the two
// sums that do not depend on the column: cs[c] = {sum offset_k x_k, sum x_k} -- offset_k is the power of two the field of k carries
// in the pairs the decode kernel feeds the matrix pipe (exl2_tpairs16), a property of the chunk's bit width and of k's place in it.
// position k of every row of x: xp[m][k] = x[m][q_perm[k]]; the 32 lanes of a chunk reduce its two sums (cs[m][k / 32])
// M >= 2 (the per-wave form, instantiated for 2 / 4 / 8 / 16 rows): chunk-major, xp[chunk][row][32] and cs[chunk][row], rows beyond M repeating row M - 1
__device__ __forceinline__ void exl2_permute_rows(const uint16_t* __restrict__ x, const uint16_t* __restrict__ perm, uint16_t* __restrict__ xp,
                                                  float2_t* __restrict__ cs, const Exl2Rows& rows, int M, int K, int k, int m) {
    const int MT = exl2_xp_rows_d(M);
    const int src = perm ? (int)perm[k] : k;
    int bits = 2;
#pragma unroll
    for (int b = 5; b >= 0; b--)
        if (k < rows.r[b]) bits = exl2_bits_of_band(b);  // the first band whose end lies beyond k
    const float ofs = exl2_offset_of(bits, (k & 31) >> 1);
    {   // one (k, row) per thread: the grid's z dimension walks the rows the decode kernel is instantiated for (MT, or M below the threshold)
        const uint16_t xb = x[(long)(m < M ? m : M - 1) * K + src];
        xp[((long)(k >> 5) * MT + m) * 32 + (k & 31)] = xb;
        const float xv = f16_bits_to_f32(xb);
        float so = ofs * xv, sx = xv;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            so += __shfl_xor(so, d, 32);
            sx += __shfl_xor(sx, d, 32);
        }
        if ((k & 31) == 0) {
            cs[(long)(k >> 5) * MT + m] = float2_t{so, sx};
        }
    }
}
__global__ __launch_bounds__(256) void exl2_list_permute_kernel(const Exl2Call* __restrict__ ent) {
    const Exl2Call& e = ent[blockIdx.y];
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= e.K) return;  // K % 32 == 0: whole 32-lane groups leave together
    exl2_permute_rows(e.x, e.perm, const_cast<uint16_t*>(e.xp), const_cast<float2_t*>(e.cs), e.rows, e.M, e.K, k, (int)blockIdx.z);
}
// ---- a GROUP of up to 8 exl2 layers that consume the same one-row x (q / k / v, gate / up), everything in the kernel arguments ----------
// bie_mbwq_exl2_forward_grouped: no plan object and no device table -- the call descriptors travel in the kernel-argument segment and
// are read from there with scalar loads, so x and the outputs may be new tensors on every call.  Two launches: the permute kernel
// (xp_i = x[q_perm_i] and the two per-chunk sums, every member has its own q_perm) and the DMODE 2 decode body over all members.
constexpr int EXL2_GROUP_MAX = 8;
struct Exl2GroupArgs {
    int n, max_k;
    int first_block[EXL2_GROUP_MAX + 1];  // prefix sums of colblocks * S
    Exl2Call ent[EXL2_GROUP_MAX];
};
template <int MT, bool NARROW>
__global__ __launch_bounds__(512, (MT == 1 ? 6 : 4)) void exl2_group_kernel(const Exl2GroupArgs a, unsigned epoch, unsigned* status, unsigned tag_skew, int spin_limit) {
    typedef const __attribute__((address_space(4))) Exl2Call ccall_t;
    int ei = 0;
#pragma unroll
    for (int i = 1; i < EXL2_GROUP_MAX; i++)
        if (i < a.n && (int)blockIdx.x >= a.first_block[i]) ei = i;
    const __attribute__((address_space(4))) char* kp = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    ccall_t* c = (ccall_t*)(kp + __builtin_offsetof(Exl2GroupArgs, ent) + (size_t)ei * sizeof(Exl2Call));
    const int local = (int)blockIdx.x - a.first_block[ei];
    Exl2Rows rows;
#pragma unroll
    for (int i = 0; i < 6; i++) rows.r[i] = c->rows.r[i];
    Exl2Groups grp;
#pragma unroll
    for (int i = 0; i < 6; i++) { grp.gfirst[i] = c->gfirst[i]; grp.glog[i] = c->glog[i]; }
    const int cb = c->colblocks;
    exl2_gemv2_body<MT, 8, false, NARROW, 2>(c->x, c->qw, c->scales, c->zeros, c->perm, c->gmap, c->gran, c->gen, c->y, rows, c->M, c->K, c->N,
                                            c->chunks_per_slab, c->S, local % cb, local / cb, cb, epoch, status, tag_skew, spin_limit, grp, c->xp, c->cs);
}
__global__ __launch_bounds__(256) void exl2_group_permute_kernel(const Exl2GroupArgs a) {
    typedef const __attribute__((address_space(4))) Exl2Call ccall_t;
    const __attribute__((address_space(4))) char* kp = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    ccall_t* e = (ccall_t*)(kp + __builtin_offsetof(Exl2GroupArgs, ent) + (size_t)blockIdx.y * sizeof(Exl2Call));
    const int K = e->K;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    Exl2Rows rows;
#pragma unroll
    for (int i = 0; i < 6; i++) rows.r[i] = e->rows.r[i];
    exl2_permute_rows(e->x, e->perm, const_cast<uint16_t*>(e->xp), const_cast<float2_t*>(e->cs), rows, e->M, K, k, (int)blockIdx.z);
}

// decode (M <= 2): column blocks x K slabs ~ 512 workgroups of 8 waves (two per CU -> one round), slabs in whole multiples of
// 8 chunks so that the 8 waves of a workgroup get equal shares; at most BIE_WS_COUNTERS column blocks use the slab reduction
static void exl2_decode_plan(int M, int K, int N, int& cps, int& S, int& nw, bool direct = false) {
    const int C = K / 32, colblocks = cdiv(N, 64);
    const int CPS_MAX = M <= 2 ? 768 / M : 384;  // (callers pass M <= 2) the slab's x (q_perm applied, M rows) / group-map copy in LDS: 64 M + 4 bytes per chunk (<= 52 KiB)
    static const int nw16_min = [] { const char* ev = getenv("BIE_EXL2_NW16_MIN"); return ev ? atoi(ev) : 160; }();
    static const int want_wgs = [] { const char* ev = getenv("BIE_EXL2_PLAN_WGS"); return ev && atoi(ev) > 0 ? atoi(ev) : 512; }();
    static const int min_cpw = [] { const char* ev = getenv("BIE_EXL2_MIN_CPW"); return ev && atoi(ev) > 0 ? atoi(ev) : 4; }();
    if (colblocks >= nw16_min && (M == 1 || direct)) {  // wide layers: 16-wave workgroups, one K slab (more only when K is too long for the LDS copy).
        // M = 2 keeps the 8-wave form: 1024-thread workgroups cap a wave at 128 registers and the two-row variant spilled 1096 dwords
        nw = 16;
        S = cdiv(C, CPS_MAX);
        static const int s16 = [] { const char* ev = getenv("BIE_EXL2_NW16_SLABS"); return ev ? atoi(ev) : 0; }();  // tuning: K slabs wanted for 16-wave workgroups
        if (s16 > S) S = s16;
        cps = cdiv(cdiv(C, S), nw) * nw;
        if (cps > C) cps = C;
        S = cdiv(C, cps);
        return;
    }
    nw = 8;
    int want = (want_wgs + colblocks / 2) / colblocks;
    if (want < 1) want = 1;
    cps = cdiv(cdiv(C, want), nw) * nw;
    if (cps < min_cpw * nw) cps = min_cpw * nw;  // at least four chunks per wave: the depth of the kernel's prefetch
    if (cps > CPS_MAX) cps = CPS_MAX;
    if (cps > C) cps = C;
    S = cdiv(C, cps);
}

static int exl2_decode_mt(int M) { return M <= 1 ? 1 : 2; }  // rows of x the decode kernel is instantiated for

static int exl2_slabs(int K, int N) {
    const int C = K / 32;
    int S = cdiv(1024, cdiv(N, 256));
    if (S > C) S = C;
    if (S < 1) S = 1;
    int cps = cdiv(C, S);
    if (cps > 64) cps = 64;  // LDS: 8 rows x 64 chunks x 32 x 4 B = 64 KiB
    return cps;
}

// When a lone layer runs as a group of ONE (permute kernel + the pre-permuted decode body, two launches), measured (profiles/r04_exl2_ablation.txt):
//  one / two rows with a long K: 11008x4096 11.6 against 12.2 us, 14336x4096 12.2 against 13.6, 28672x8192 26.4 against 39.1; at K <= 8192 the direct launch is ahead at one row
//  two rows from 32 Mi weights: 4096x11008 10.6 against 13.4 us (4096x4096: 8.3 against 7.6, stays direct)
//  3 ... 16 rows always: 8.1-11.4 / 11.2-16.9 / 11.7-18.7 us against 11.3-12.9 / 20.1-21.6 / 17.5-19.4 of the fused matrix-pipe kernel (4096^2 / 4096x11008 / 11008x4096)
constexpr int EXL2_LONE_AS_GROUP_MIN_K = 10240;
constexpr long EXL2_LONE_AS_GROUP_M2_MIN_WEIGHTS = 32l << 20;
constexpr int EXL2_XP_MAX_M = 16;  // rows of x of the pre-permuted decode form (four per matrix instruction, up to four instructions per four k)
static size_t exl2_lone_group_bytes(int M, int K, int N);
int exl2_group_forward(int n, const bie_exl2_list_entry* e, const void* x, int M, float* head, char* body, hipStream_t st);

// prefill form of the mixed-bit layout: rows of x from which the dequantise-once image + dense GEMM (mpq_dense.hip) replaces the
// streaming kernels (which read the packed weight once per 16 rows).  BIE_EXL2_DENSE_MIN_M moves the switch (tools/).
bool mpq_dense_shape_ok(int K, int N);
size_t mpq_dense_workspace_bytes(int K, int N);
size_t mpq_dense_part_bytes(int M, int K, int N);
int mpq_dense_gemm_only_launch(const void* x, const void* img, const void* bias, void* y, int M, int K, int N, int dtype, hipStream_t st, int ldy, float* part);
static int exl2_dense_min_m() {
    static const int v = [] { const char* e = getenv("BIE_EXL2_DENSE_MIN_M"); return e ? atoi(e) : 49; }();
    const char* t = getenv("BIE_TUNING");
    if (t) { const char* e = getenv("BIE_EXL2_DENSE_MIN_M"); return e ? atoi(e) : 49; }
    return v;
}
static bool exl2_dense_ok(int M, int K, int N) { return M >= exl2_dense_min_m() && mpq_dense_shape_ok(K, N); }
static size_t exl2_dense_xp_bytes(int M, int K) { return ((size_t)M * K * 2 + 255) & ~(size_t)255; }
static size_t exl2_dense_bytes(int M, int K, int N) {  // [image][x[:, q_perm]][fp32 partial sums of the K splits]
    return exl2_dense_ok(M, K, N) ? mpq_dense_workspace_bytes(K, N) + exl2_dense_xp_bytes(M, K) + mpq_dense_part_bytes(M, K, N) : 0;
}

// exl2 = true: the mixed-bit forward (its prefill form keeps a dense fp16 image of the weights: 2*K*N bytes and more); false: the uniform q4 / q2
// forward, which never touches that image -- sized together (round 5) a q4 prefill on an 8192 x 28672 layer pinned 470 MB of per-stream workspace
// for the rest of the process
size_t mbwq_workspace_bytes(int M, int K, int N, bool exl2) {
    if (exl2 && exl2_dense_ok(M, K, N)) {
        const size_t f = exl2_dense_bytes(M, K, N), b0 = mpq_gemm_workspace_bytes(M, K, N);
        return f > b0 ? f : b0;
    }
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
    size_t d = 0;  // decode granules (unused when another kernel takes over).  The decode plan is defined for its own row counts only: asked about
    if (M <= 2) {  // 769 rows and more it divided by zero (768 / M chunks per slab) -- a host SIGFPE for big M on shapes the prefill form cannot take
        int cps2, S2, nw2;
        exl2_decode_plan(M, K, N, cps2, S2, nw2);
        if (S2 > 1) d = (size_t)(S2 - 1) * exl2_decode_mt(M) * cdiv(N, 64) * 64 * 8;
    }
    if (d > c) c = d;
    const size_t e = exl2_mfma_granule_bytes(M, K, N);  // 3 <= M <= 64: granules of the matrix-pipe kernel's K slabs
    if (e > c) c = e;
    const size_t g1 = exl2_lone_group_bytes(M, K, N);
    if (g1 > c) c = g1;
    const size_t gen = (size_t)cdiv(K, 512) * (M < MBWQ_GENERIC_M_CHUNK ? M : MBWQ_GENERIC_M_CHUNK) * N * sizeof(float);  // the any-shape kernel of the uniform q4 / q2 forward
    if (gen > c) c = gen;
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

// in place: the checkpoint's LSB-first streams -> the half-pair layout every exl2 kernel of this library reads
int mbwq_exl2_shuffle_launch(int32_t* qw, const int* rows6, int K, int N, hipStream_t st, bool inverse) {
    Exl2Rows rows;
    for (int i = 0; i < 6; i++) rows.r[i] = rows6[i];
    dim3 grid(cdiv(N, 256), K / 32);
    if (inverse)
        hipLaunchKernelGGL(exl2_shuffle_kernel<true>, grid, dim3(256), 0, st, (uint32_t*)qw, rows, K, N);
    else
        hipLaunchKernelGGL(exl2_shuffle_kernel<false>, grid, dim3(256), 0, st, (uint32_t*)qw, rows, K, N);
    return check_launch("exl2_shuffle_kernel");
}

static void exl2_fill_groups(Exl2Call& c, const int* rows_ext) {
    for (int b = 0; b < 6; b++) {
        c.gfirst[b] = rows_ext[BIE_EXL2_ROWS_GFIRST + b];
        c.glog[b] = rows_ext[BIE_EXL2_ROWS_GLOG + b];
    }
}
static bool exl2_xp_on() {
    static const bool on = [] { const char* ev = getenv("BIE_EXL2_XP"); return !ev || atoi(ev) != 0; }();
    return on;
}
static bool exl2_direct_on() {
    static const bool on = [] { const char* e = getenv("BIE_EXL2_DIRECT"); return !e || atoi(e) != 0; }();
    return on;
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
    // any other shape the layout itself allows (whole packed words: K a multiple of 32 / bits; whole groups): the one-column-per-lane kernel with the
    // same per-weight rounding, 32 rows per launch.  The reference takes such shapes too (its kernels bound-check K and N,
    // mbwq_linear_cuda_kernel.cu:740-830); this used to be a refusal (found by tests/sweeps/fuzz_other_ops.py).  A correctness path, not a fast one.
    if (K % (32 / bits) == 0 && (group_size >= K || K % group_size == 0)) {
        for (int m0 = 0; m0 < M; m0 += MBWQ_GENERIC_M_CHUNK) {
            const int mc = (M - m0) < MBWQ_GENERIC_M_CHUNK ? (M - m0) : MBWQ_GENERIC_M_CHUNK;
            const int rc = mpq_gemv_generic_launch((const uint16_t*)x + (size_t)m0 * K, qw, scales, zeros, nullptr, nullptr, (uint16_t*)y + (size_t)m0 * N,
                                                   part + BIE_WS_HEAD_BYTES / sizeof(float), mc, K, N, bits, group_size, 2, BIE_F16, st, p);
            if (rc) return rc;
        }
        return BIE_OK;
    }
    set_error("bie_mbwq_q4_forward: unsupported shape M=%d K=%d N=%d bits=%d group_size=%d (K must hold whole packed words and whole groups)", M,
              K, N, bits, group_size);
    return BIE_ERR_UNSUPPORTED;
}

int mbwq_exl2_forward_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const int16_t* perm,
                             const int16_t* gmap, const int* rows7, void* y, float* head, float* part, int M, int K, int N,
                             hipStream_t st) {
    Exl2Rows rows;
    for (int i = 0; i < 6; i++) rows.r[i] = rows7[i];
    if (exl2_dense_ok(M, K, N)) {  // prefill: dequantise once into the fragment image, x[:, q_perm], dense MFMA GEMM -- three launches, no vendor GEMM
        char* img = reinterpret_cast<char*>(part);
        hipLaunchKernelGGL(exl2_dequant_frag_kernel, dim3(cdiv(cdiv(N, 32) * 32, 256), K / 32), dim3(256), 0, st, (const uint32_t*)qw, (const uint16_t*)scales,
                           (const uint16_t*)zeros, (const uint16_t*)gmap, (uint4_t*)img, rows, K, N);
        int rc = check_launch("exl2_dequant_frag_kernel");
        if (rc) return rc;
        const void* xin = x;
        if (perm) {
            uint16_t* xp = reinterpret_cast<uint16_t*>(img + mpq_dense_workspace_bytes(K, N));
            if ((K & 7) == 0 && K <= 32768 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(perm) & 15) == 0)
                hipLaunchKernelGGL(exl2_permute_rows_kernel, dim3(M < 2048 ? M : 2048), dim3(256), (size_t)K * 2, st, (const uint16_t*)x, (const uint16_t*)perm, xp, M, K);
            else
                hipLaunchKernelGGL(exl2_permute_x_kernel, dim3(cdiv(K, 256), M < 256 ? M : 256), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)perm, xp, M, K);
            rc = check_launch("exl2_permute_rows_kernel");
            if (rc) return rc;
            xin = xp;
        }
        float* psum = mpq_dense_part_bytes(M, K, N) ? reinterpret_cast<float*>(img + mpq_dense_workspace_bytes(K, N) + exl2_dense_xp_bytes(M, K)) : nullptr;
        return mpq_dense_gemm_only_launch(xin, img, nullptr, y, M, K, N, BIE_F16, st, N, psum);
    }
    const bool slab_ok = !(cdiv(N, 64) > BIE_WS_COUNTERS && K / 32 > 768);  // K slabs need one generation word per column block
    const bool regular = (rows7[6] & BIE_EXL2_ROWS_REGULAR) && exl2_direct_on();
    static const int lone_min_k = [] { const char* ev = getenv("BIE_EXL2_LONE_AS_GROUP_MIN_K"); return ev ? atoi(ev) : EXL2_LONE_AS_GROUP_MIN_K; }();
    static const int lone_rows_lo = [] { const char* ev = getenv("BIE_EXL2_LONE_ROWS_LO"); return ev ? atoi(ev) : 3; }();
    static const int lone_rows_hi = [] { const char* ev = getenv("BIE_EXL2_LONE_ROWS_HI"); return ev ? atoi(ev) : EXL2_XP_MAX_M; }();
    const bool rows_as_group = M >= lone_rows_lo && M <= lone_rows_hi && M <= EXL2_XP_MAX_M;
    if (((M <= 2 && K >= lone_min_k) || (M == 2 && (long)K * N >= EXL2_LONE_AS_GROUP_M2_MIN_WEIGHTS) || rows_as_group) && regular && K % 32 == 0 && cdiv(N, 64) <= BIE_WS_COUNTERS && exl2_xp_on()) {
        bie_exl2_list_entry one{};
        one.x = x; one.qweight = qw; one.scales = scales; one.zeros = zeros; one.q_perm = perm; one.q_group_map = gmap; one.rows7 = rows7; one.y = y;
        one.K = K; one.N = N;
        return exl2_group_forward(1, &one, x, M, head, reinterpret_cast<char*>(part), st);
    }
    if (M <= 2 && slab_ok) {  // decode path.  (Three and four rows would ride on the same v_mfma_f32_4x4x4, but the direct form then gathers four rows
                              //  per chunk and spills: 14.2 / 35.2 / 39.3 us at M = 4 against 13.3 / 22.2 / 19.4 on the matrix-pipe kernel below.)  (The packed-fp16 kernel instantiated for 4 / 8 rows measured SLOWER than the fp32 kernel below:
                   //  33.3 / 46.2 us against 28.0 / 39.9 us at 4096x11008 M = 3 / 8 -- 178-256 registers, one wave per SIMD.)
        const int colblocks = cdiv(N, 64);
        int cps2, S, nw;
        exl2_decode_plan(M, K, N, cps2, S, nw, regular);
        const int MT = exl2_decode_mt(M);
        const bool direct = regular;
        size_t lds2 = (MT > 1 && !direct) ? (size_t)cps2 * (32 * MT + 2) * sizeof(uint16_t)   // the slab's gathered activations (M rows) + group map
                                          : (size_t)nw * 4 * MT * 32 * sizeof(uint16_t) + (direct ? 0 : (size_t)cps2 * 34 * sizeof(uint16_t));  // wave x buffers (+ q_perm + group map)
        const size_t red = (size_t)nw * MT * 64 * sizeof(float);
        if (lds2 < red) lds2 = red;
        dim3 grid2(colblocks, S);
        const unsigned epoch = next_launch_epoch();
        unsigned* gen = reinterpret_cast<unsigned*>(head) + BIE_WS_GEN_OFFSET;
        unsigned long long* gran = reinterpret_cast<unsigned long long*>(part);
        unsigned skew;
        int spin;
        test_forge_get(&skew, &spin);
        Exl2Call call{(const uint16_t*)x, (const uint32_t*)qw, (const uint16_t*)scales, (const uint16_t*)zeros, (const uint16_t*)perm,
                      (const uint16_t*)gmap, gran, gen, (uint16_t*)y, rows, M, K, N, cps2, S, colblocks, {}, {}, nullptr, nullptr};
        exl2_fill_groups(call, rows7);
        const bool narrow2 = rows7[2] == 0;  // no 8 / 6 / 5-bit rows
#define L2(MTV, NWV, DIR)                                                                                                                        \
    do {                                                                                                                                    \
        if (narrow2) hipLaunchKernelGGL((exl2_gemv2_kernel<MTV, NWV, true, DIR>), grid2, dim3(NWV * 64), lds2, st, call, epoch, device_status_word(), skew, spin); \
        else hipLaunchKernelGGL((exl2_gemv2_kernel<MTV, NWV, false, DIR>), grid2, dim3(NWV * 64), lds2, st, call, epoch, device_status_word(), skew, spin);      \
    } while (0)
        if (nw == 16) {  // the staged form has 16 waves for one row only
            if (MT == 1) { if (direct) L2(1, 16, 1); else L2(1, 16, 0); }
            else L2(2, 16, 1);
        } else if (MT == 1) {
            if (direct) L2(1, 8, 1); else L2(1, 8, 0);
        } else {
            if (direct) L2(2, 8, 1); else L2(2, 8, 0);
        }
#undef L2
        return check_launch("exl2_gemv2_kernel");
    }
    static const bool mfma_on = [] { const char* e = getenv("BIE_EXL2_MFMA"); return !e || atoi(e) != 0; }();
    if (mfma_on && exl2_mfma_ok(M, K, N)) {  // 3 <= M <= 64: the decode kernel's stream feeding v_mfma_f32_16x16x32_f16
        int cpsm, Sm, mb, nwm;
        exl2_mfma_plan(M, K, N, cpsm, Sm, mb, nwm);
        const size_t ldsm = exl2_mfma_lds_bytes(mb, nwm, cpsm);
        const int colblocks = cdiv(N, 64);
        dim3 gridm(colblocks, Sm);
        const unsigned epoch = next_launch_epoch();
        unsigned* gen = reinterpret_cast<unsigned*>(head) + BIE_WS_GEN_OFFSET;
        unsigned skew;
        int spin;
        test_forge_get(&skew, &spin);
        const uint16_t* xin = (const uint16_t*)x;
        if (perm) {  // x[:, q_perm] once per call, behind the granule area of the workspace
            uint16_t* xp = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(part) + exl2_mfma_xp_offset(M, K, N));
            hipLaunchKernelGGL(exl2_permute_x_kernel, dim3(cdiv(K, 256), M < 16 ? M : 16), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)perm, xp, M, K);
            int rc = check_launch("exl2_permute_x_kernel");
            if (rc) return rc;
            xin = xp;
        }
        Exl2Call call{xin, (const uint32_t*)qw, (const uint16_t*)scales, (const uint16_t*)zeros, nullptr,
                      (const uint16_t*)gmap, reinterpret_cast<unsigned long long*>(part), gen, (uint16_t*)y, rows, M, K, N, cpsm, Sm, colblocks, {}, {}, nullptr, nullptr};
        const bool narrow = rows7[2] == 0;  // cumulative end of the 5-bit band: no 8 / 6 / 5-bit rows
#define LM(MBV, NWV, OCCN, OCCW)                                                                                                                 \
    do {                                                                                                                                         \
        if (narrow) hipLaunchKernelGGL((exl2_mfma_kernel<MBV, NWV, OCCN, true>), gridm, dim3(NWV * 64), ldsm, st, call, epoch, device_status_word(), skew, spin); \
        else hipLaunchKernelGGL((exl2_mfma_kernel<MBV, NWV, OCCW, false>), gridm, dim3(NWV * 64), ldsm, st, call, epoch, device_status_word(), skew, spin);      \
    } while (0)
        switch (mb) {
            case 1: LM(1, 8, 1, 1); break;
            case 2: LM(2, 4, 2, 2); break;
            case 3: LM(3, 4, 2, 1); break;
            default: LM(4, 4, 2, 1); break;
        }
#undef LM
        return check_launch("exl2_mfma_kernel");
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


// ---- a LIST of exl2 layers in one launch --------------------------------------------------------------------------
int status_report(const char* fn);  // splitk.hip

struct Exl2List {
    int n = 0, M = 1, max_k = 0;
    bool narrow = true;  // no entry has 8 / 6 / 5-bit rows: the leaner kernel instance
    bool direct = false; // M == 1 and every entry has regular groups: nothing staged (exl2_gemv2_body, DMODE 1 / 2)
    bool xp = false;     // DMODE 2: x permuted once per forward by exl2_list_permute_kernel
    bool any_perm = false;
    unsigned grid = 0;
    size_t lds = 0;
    Exl2Call* d_ent = nullptr;
    uint2_t* d_blk = nullptr;
};

static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

// column blocks x K slabs, 8-wave workgroups; ~`want` workgroups in all (two per CU and four rounds), a slab at least 4 chunks
// per wave (the depth of the kernel's prefetch) and at most CPS_MAX chunks (the LDS copy of the slab's q_perm / group map)
static int exl2_xp_rows(int M) { return exl2_xp_rows_d(M); }  // rows the permute kernel writes (chunk-major and padded from five rows on)
static int exl2_rows_mt(int M) { return M <= 1 ? 1 : (M <= 2 ? 2 : (M <= 4 ? 4 : (M <= 8 ? 8 : 16))); }  // rows of x the list / group kernels are instantiated for
static bool exl2_all_regular(int n, const bie_exl2_list_entry* e) {
    for (int i = 0; i < n; i++)
        if (!e[i].rows7 || !(e[i].rows7[6] & BIE_EXL2_ROWS_REGULAR)) return false;
    return true;
}
static void exl2_list_plan(int n, const bie_exl2_list_entry* e, std::vector<int>& cps, std::vector<int>& S, long* blocks, size_t* gran_bytes,
                           size_t* lds, int Mrows, bool xp, int target_wgs = 0) {
    const int M = exl2_rows_mt(Mrows);  // everything below is sized for the instantiated row count
    const int CPS_MAX = xp ? (1 << 20) : 768 / M;  // the pre-permuted form: x travels per wave and chunk in registers, no slab of it in LDS
    long colblocks_all = 0;
    for (int i = 0; i < n; i++) colblocks_all += cdiv(e[i].N, 64);
    cps.resize(n); S.resize(n);
    *blocks = 0; *gran_bytes = 0; *lds = 0;
    for (int i = 0; i < n; i++) {
        const int C = e[i].K / 32, cb = cdiv(e[i].N, 64);
        static const int wgs_env = [] { const char* ev = getenv("BIE_EXL2_LIST_WGS"); return ev && atoi(ev) > 0 ? atoi(ev) : 0; }();
        const int wgs = wgs_env ? wgs_env : (target_wgs ? target_wgs : 2048);
        int want = (int)((wgs + colblocks_all / 2) / colblocks_all);
        if (want < 1) want = 1;
        int c = cdiv(cdiv(C, want), 8) * 8;
        if (c < 32) c = 32;
        if (c > CPS_MAX) c = CPS_MAX;
        if (c > C) c = C;
        cps[i] = c;
        S[i] = cdiv(C, c);
        *blocks += (long)cb * S[i];
        if (S[i] > 1) *gran_bytes += (size_t)(S[i] - 1) * M * cb * 64 * 8;  // M = the instantiated row count
        size_t l = M > 1 ? (size_t)c * (32 * M + 2) * sizeof(uint16_t) : (size_t)8 * 4 * M * 32 * sizeof(uint16_t) + (size_t)c * 34 * sizeof(uint16_t);
        if (xp) l = 0;  // nothing but the block reduction below
        const size_t red = (size_t)8 * M * 64 * sizeof(float);
        if (l < red) l = red;
        if (l > *lds) *lds = l;
    }
}

// does a list run the pre-permuted form (DMODE 2)?  Regular group structures in every entry.
static bool exl2_list_xp(int n, const bie_exl2_list_entry* e) { return exl2_all_regular(n, e) && exl2_direct_on() && exl2_xp_on(); }
// M <= 2 always (staged form); up to sixteen rows in the pre-permuted form (regular groups)
static bool exl2_list_ok(int n, const bie_exl2_list_entry* e, int M) {
    if (n <= 0 || !e || M < 1 || M > EXL2_XP_MAX_M) return false;
    for (int i = 0; i < n; i++) {
        if (e[i].K <= 0 || e[i].N <= 0 || e[i].K % 32 || !e[i].rows7) return false;
        if (cdiv(e[i].N, 64) >= (1 << 20)) return false;
    }
    return M <= 2 || exl2_list_xp(n, e);
}

size_t exl2_list_device_bytes(int n, const bie_exl2_list_entry* e, int M) {
    if (!exl2_list_ok(n, e, M)) return 0;
    std::vector<int> cps, S;
    long blocks; size_t gran, lds;
    exl2_list_plan(n, e, cps, S, &blocks, &gran, &lds, M, exl2_list_xp(n, e));
    long tiles = 0;
    for (int i = 0; i < n; i++) tiles += cdiv(e[i].N, 64);
    size_t xp = 0;
    for (int i = 0; i < n; i++) xp += align256((size_t)exl2_xp_rows(M) * e[i].K * 2) + align256((size_t)exl2_xp_rows(M) * (e[i].K / 32) * 8);
    return align256((size_t)n * sizeof(Exl2Call)) + align256((size_t)blocks * 8) + align256((size_t)tiles * 4) + align256(gran) + xp;
}

int exl2_list_create(Exl2List** out, int n, const bie_exl2_list_entry* e, int M, void* device_mem, size_t device_bytes) {
    BIE_REQUIRE(out && e && device_mem, BIE_ERR_INVALID_ARG, "bie_mbwq_exl2_list_create: NULL argument");
    BIE_REQUIRE(exl2_list_ok(n, e, M), BIE_ERR_UNSUPPORTED, "bie_mbwq_exl2_list_create: a list takes 1 <= M <= 2 (up to 16 when every entry's table carries the REGULAR mark) and K %% 32 == 0 (fp16 only)");
    for (int i = 0; i < n; i++) {
        BIE_REQUIRE(e[i].x && e[i].qweight && e[i].scales && e[i].zeros && e[i].q_group_map && e[i].y, BIE_ERR_INVALID_ARG, "bie_mbwq_exl2_list_create: NULL tensor pointer in entry %d", i);
        int prev = 0;
        for (int b = 0; b < 6; b++) {
            BIE_REQUIRE(e[i].rows7[b] >= prev && e[i].rows7[b] % 32 == 0, BIE_ERR_UNSUPPORTED, "bie_mbwq_exl2_list_create: entry %d: band boundary rows[%d]=%d must be a non-decreasing multiple of 32", i, b, e[i].rows7[b]);
            prev = e[i].rows7[b];
        }
        BIE_REQUIRE(e[i].rows7[5] == e[i].K, BIE_ERR_INVALID_ARG, "bie_mbwq_exl2_list_create: entry %d: rows[5]=%d must equal K=%d", i, e[i].rows7[5], e[i].K);
        BIE_REQUIRE((e[i].rows7[6] & BIE_EXL2_ROWS_SHUFFLED) && e[i].rows7[BIE_EXL2_ROWS_LEN - 1] == BIE_EXL2_ROWS_TAG, BIE_ERR_INVALID_ARG,
                    "bie_mbwq_exl2_list_create: entry %d: the band table is not the one bie_mbwq_exl2_shuffle wrote (qweight must pass through it once)", i);
    }
    BIE_REQUIRE((reinterpret_cast<uintptr_t>(device_mem) & 255) == 0, BIE_ERR_INVALID_ARG, "bie_mbwq_exl2_list_create: the device buffer must be 256-byte aligned");
    std::vector<int> cps, S;
    long blocks; size_t gran, lds;
    exl2_list_plan(n, e, cps, S, &blocks, &gran, &lds, M, exl2_list_xp(n, e));
    long tiles = 0;
    for (int i = 0; i < n; i++) tiles += cdiv(e[i].N, 64);
    const size_t o_blk = align256((size_t)n * sizeof(Exl2Call)), o_gen = o_blk + align256((size_t)blocks * 8), o_gran = o_gen + align256((size_t)tiles * 4);
    const size_t o_xp = o_gran + align256(gran);
    size_t xp_bytes = 0;
    for (int i = 0; i < n; i++) xp_bytes += align256((size_t)exl2_xp_rows(M) * e[i].K * 2) + align256((size_t)exl2_xp_rows(M) * (e[i].K / 32) * 8);
    BIE_REQUIRE(device_bytes >= o_xp + xp_bytes, BIE_ERR_WORKSPACE, "bie_mbwq_exl2_list_create: device buffer of %zu bytes required, got %zu", o_xp + xp_bytes, device_bytes);
    char* base = static_cast<char*>(device_mem);
    std::vector<Exl2Call> he(n);
    std::vector<uint2_t> hb((size_t)blocks);
    size_t b = 0, go = o_gran, xo = o_xp;
    long t0 = 0;
    int max_k = 0;
    bool any_perm = false;
    for (int i = 0; i < n; i++) {
        Exl2Call& c = he[i];
        const int cb = cdiv(e[i].N, 64);
        c.x = (const uint16_t*)e[i].x;
        if (e[i].K > max_k) max_k = e[i].K;
        c.qw = (const uint32_t*)e[i].qweight; c.scales = (const uint16_t*)e[i].scales; c.zeros = (const uint16_t*)e[i].zeros;
        c.perm = (const uint16_t*)e[i].q_perm; c.gmap = (const uint16_t*)e[i].q_group_map; c.y = (uint16_t*)e[i].y;
        c.gran = S[i] > 1 ? reinterpret_cast<unsigned long long*>(base + go) : nullptr;
        if (S[i] > 1) go += (size_t)(S[i] - 1) * exl2_rows_mt(M) * cb * 64 * 8;
        c.gen = reinterpret_cast<unsigned*>(base + o_gen) + t0;
        for (int k = 0; k < 6; k++) c.rows.r[k] = e[i].rows7[k];
        exl2_fill_groups(c, e[i].rows7);
        c.xp = reinterpret_cast<const uint16_t*>(base + xo);
        xo += align256((size_t)exl2_xp_rows(M) * e[i].K * 2);
        c.cs = reinterpret_cast<const float2_t*>(base + xo);
        xo += align256((size_t)exl2_xp_rows(M) * (e[i].K / 32) * 8);
        if (c.perm) any_perm = true;
        c.M = M; c.K = e[i].K; c.N = e[i].N; c.chunks_per_slab = cps[i]; c.S = S[i]; c.colblocks = cb;
        BIE_REQUIRE(S[i] < 4096, BIE_ERR_UNSUPPORTED, "bie_mbwq_exl2_list_create: entry %d needs %d K slabs (< 4096)", i, S[i]);
        for (int sl = 0; sl < S[i]; sl++)
            for (int t = 0; t < cb; t++) hb[b++] = uint2_t{(uint32_t)i, (uint32_t)t | ((uint32_t)sl << 20)};
        t0 += cb;
    }
    hipError_t err = hipMemset(base + o_gen, 0, o_xp - o_gen);
    if (err == hipSuccess) err = hipMemcpy(base, he.data(), (size_t)n * sizeof(Exl2Call), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(base + o_blk, hb.data(), (size_t)blocks * 8, hipMemcpyHostToDevice);
    BIE_REQUIRE(err == hipSuccess, BIE_ERR_HIP, "bie_mbwq_exl2_list_create: uploading the plan: %s", hipGetErrorString(err));
    Exl2List* pl = new Exl2List();
    pl->n = n; pl->M = M; pl->grid = (unsigned)blocks; pl->lds = lds; pl->max_k = max_k;
    pl->direct = M == 1 && exl2_direct_on();
    pl->any_perm = any_perm;
    for (int i = 0; i < n; i++) {
        if (e[i].rows7[2] != 0) pl->narrow = false;
        if (!(e[i].rows7[6] & BIE_EXL2_ROWS_REGULAR)) pl->direct = false;
    }
    pl->xp = exl2_list_xp(n, e);
    pl->d_ent = reinterpret_cast<Exl2Call*>(base);
    pl->d_blk = reinterpret_cast<uint2_t*>(base + o_blk);
    *out = pl;
    return BIE_OK;
}

int exl2_list_forward(Exl2List* p, hipStream_t st) {
    BIE_REQUIRE(p, BIE_ERR_INVALID_ARG, "bie_mbwq_exl2_list_forward: NULL plan");
    int rc = status_report("bie_mbwq_exl2_list_forward");
    if (rc) return rc;
    unsigned skew;
    int spin;
    test_forge_get(&skew, &spin);
    const unsigned epoch = next_launch_epoch();
#define LL(MTV, DIR)                                                                                                                                  \
    do {                                                                                                                                         \
        if (p->narrow) hipLaunchKernelGGL((exl2_list_kernel<MTV, true, DIR>), dim3(p->grid), dim3(512), p->lds, st, p->d_ent, p->d_blk, epoch, device_status_word(), skew, spin); \
        else hipLaunchKernelGGL((exl2_list_kernel<MTV, false, DIR>), dim3(p->grid), dim3(512), p->lds, st, p->d_ent, p->d_blk, epoch, device_status_word(), skew, spin);      \
    } while (0)
    if (p->xp) {
        hipLaunchKernelGGL(exl2_list_permute_kernel, dim3(cdiv(p->max_k, 256), p->n, exl2_xp_rows(p->M)), dim3(256), 0, st, p->d_ent);
        rc = check_launch("exl2_list_permute_kernel");
        if (rc) return rc;
    }
    if (p->xp) {
        switch (exl2_rows_mt(p->M)) {
            case 1: LL(1, 2); break;
            case 2: LL(2, 2); break;
            case 4: LL(4, 2); break;
            case 8: LL(8, 2); break;
            default: LL(16, 2); break;
        }
    } else if (p->M == 1) {
        if (p->direct) LL(1, 1); else LL(1, 0);
    } else {
        LL(2, 0);
    }
#undef LL
    return check_launch("exl2_list_kernel");
}

// ---- the group call (bie_mbwq_exl2_forward_grouped) -----------------------------------------------------------------------------
constexpr int EXL2_GROUP_WGS = 768;  // one round of three workgroups per CU (2 - 3 members: 18.0-18.7 us against 20.4-21.3 with the list's 2048)
static bool exl2_group_entry_ok(const bie_exl2_list_entry& e) {
    return e.K > 0 && e.N > 0 && e.K % 32 == 0 && e.rows7 && (e.rows7[6] & BIE_EXL2_ROWS_SHUFFLED) && e.rows7[BIE_EXL2_ROWS_LEN - 1] == BIE_EXL2_ROWS_TAG &&
           (e.rows7[6] & BIE_EXL2_ROWS_REGULAR) && e.rows7[5] == e.K;
}
bool exl2_group_ok(int n, const bie_exl2_list_entry* e, int M) {
    if (n < 1 || n > EXL2_GROUP_MAX || !e || !exl2_direct_on() || M < 1 || M > EXL2_XP_MAX_M) return false;
    long cbs = 0;
    for (int i = 0; i < n; i++) {
        if (!exl2_group_entry_ok(e[i])) return false;
        cbs += cdiv(e[i].N, 64);
    }
    return cbs <= BIE_WS_COUNTERS;  // one generation word per column block in the workspace head
}
// behind the 16 KiB head: granules of the K slabs, then per member xp (K fp16) and cs (K / 32 float2)
size_t exl2_group_workspace_bytes(int n, const bie_exl2_list_entry* e, int M) {
    if (!exl2_group_ok(n, e, M)) return 0;
    std::vector<int> cps, S;
    long blocks; size_t gran, lds;
    exl2_list_plan(n, e, cps, S, &blocks, &gran, &lds, M, true, EXL2_GROUP_WGS);
    size_t tot = align256(gran);
    for (int i = 0; i < n; i++) tot += align256((size_t)exl2_xp_rows(M) * e[i].K * 2) + align256((size_t)exl2_xp_rows(M) * (e[i].K / 32) * 8);
    return tot;
}
static size_t exl2_lone_group_bytes(int M, int K, int N) {  // upper bound without the band table: the plan only reads K and N
    if (M < 1 || M > EXL2_XP_MAX_M || K % 32 != 0 || K <= 0 || N <= 0) return 0;
    bie_exl2_list_entry one{};
    one.K = K; one.N = N;
    std::vector<int> cps, S;
    long blocks; size_t gran, lds;
    exl2_list_plan(1, &one, cps, S, &blocks, &gran, &lds, M, true, EXL2_GROUP_WGS);
    return align256(gran) + align256((size_t)exl2_xp_rows(M) * K * 2) + align256((size_t)exl2_xp_rows(M) * (K / 32) * 8);
}
int exl2_group_forward(int n, const bie_exl2_list_entry* e, const void* x, int M, float* head, char* body, hipStream_t st) {
    std::vector<int> cps, S;
    long blocks; size_t gran, lds;
    exl2_list_plan(n, e, cps, S, &blocks, &gran, &lds, M, true, EXL2_GROUP_WGS);
    const int MT = exl2_rows_mt(M);
    // the permute kernel in front (two launches); every workgroup permuting its own slab instead (one launch) measured slower: 12.7 / 18.7 / 19.6 against
    // 12.2 / 14.2 / 14.9 us for 3 x 4096x4096 / 2 x 4096x11008 / 2 x 11008x4096 -- the dependent gather in front of every workgroup's first word costs
    // more than the launch it saves (profiles/r04_exl2_ablation.txt)
    Exl2GroupArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n = n;
    size_t go = 0, xo = align256(gran);
    long t0 = 0;
    bool narrow = true;
    unsigned* gen = reinterpret_cast<unsigned*>(head) + BIE_WS_GEN_OFFSET;
    for (int i = 0; i < n; i++) {
        Exl2Call& c = a.ent[i];
        const int cb = cdiv(e[i].N, 64);
        c.x = (const uint16_t*)(x ? x : e[i].x);
        c.qw = (const uint32_t*)e[i].qweight; c.scales = (const uint16_t*)e[i].scales; c.zeros = (const uint16_t*)e[i].zeros;
        c.perm = (const uint16_t*)e[i].q_perm; c.gmap = (const uint16_t*)e[i].q_group_map; c.y = (uint16_t*)e[i].y;
        c.gran = S[i] > 1 ? reinterpret_cast<unsigned long long*>(body + go) : nullptr;
        if (S[i] > 1) go += (size_t)(S[i] - 1) * MT * cb * 64 * 8;
        c.gen = gen + t0;
        for (int k = 0; k < 6; k++) c.rows.r[k] = e[i].rows7[k];
        exl2_fill_groups(c, e[i].rows7);
        c.xp = reinterpret_cast<const uint16_t*>(body + xo);
        xo += align256((size_t)exl2_xp_rows(M) * e[i].K * 2);
        c.cs = reinterpret_cast<const float2_t*>(body + xo);
        xo += align256((size_t)exl2_xp_rows(M) * (e[i].K / 32) * 8);
        c.M = M; c.K = e[i].K; c.N = e[i].N; c.chunks_per_slab = cps[i]; c.S = S[i]; c.colblocks = cb;
        BIE_REQUIRE(S[i] < 4096, BIE_ERR_UNSUPPORTED, "bie_mbwq_exl2_forward_grouped: member %d needs %d K slabs (< 4096)", i, S[i]);
        a.first_block[i] = (int)(i == 0 ? 0 : a.first_block[i - 1] + (long)cdiv(e[i - 1].N, 64) * S[i - 1]);
        if (e[i].K > a.max_k) a.max_k = e[i].K;
        if (e[i].rows7[2] != 0) narrow = false;
        t0 += cb;
    }
    a.first_block[n] = (int)blocks;
    unsigned skew;
    int spin;
    test_forge_get(&skew, &spin);
    const unsigned epoch = next_launch_epoch();
    int rc = BIE_OK;
    hipLaunchKernelGGL(exl2_group_permute_kernel, dim3(cdiv(a.max_k, 256), n, exl2_xp_rows(M)), dim3(256), 0, st, a);
    rc = check_launch("exl2_group_permute_kernel");
    if (rc) return rc;
#define LG(MTV)                                                                                                                                             \
    do {                                                                                                                                                    \
        if (narrow) hipLaunchKernelGGL((exl2_group_kernel<MTV, true>), dim3((unsigned)blocks), dim3(512), lds, st, a, epoch, device_status_word(), skew, spin); \
        else hipLaunchKernelGGL((exl2_group_kernel<MTV, false>), dim3((unsigned)blocks), dim3(512), lds, st, a, epoch, device_status_word(), skew, spin);      \
    } while (0)
    switch (MT) {
        case 1: LG(1); break;
        case 2: LG(2); break;
        case 4: LG(4); break;
        case 8: LG(8); break;
        default: LG(16); break;
    }
#undef LG
    return check_launch("exl2_group_kernel");
}

void exl2_list_destroy(Exl2List* p) { delete p; }

}  // namespace bie
