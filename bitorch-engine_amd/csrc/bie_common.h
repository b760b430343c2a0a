// Internal helpers shared by the HIP translation units of libbie_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/bie_hip.h"

namespace bie {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));
typedef uint32_t uint2_t __attribute__((ext_vector_type(2)));

// ---- error plumbing --------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define BIE_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            ::bie::set_error(__VA_ARGS__); \
            return (code);                \
        }                                 \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long cdivl(long a, long b) { return (a + b - 1) / b; }

// ---- 16-bit float <-> float on device, bit-level (torch semantics: RNE) -----------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    // v_cvt_pk_bf16_f32 (RNE, NaN-preserving) -- one instruction for two values; used pairwise below
    bf16x2_t r = __builtin_convertvector(float2_t{f, 0.0f}, bf16x2_t);
    return __builtin_bit_cast(uint32_t, r) & 0xffffu;
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    bf16x2_t r = __builtin_convertvector(float2_t{lo, hi}, bf16x2_t);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float f16_bits_to_f32(uint32_t h) {
    return (float)__builtin_bit_cast(half_t, (uint16_t)h);
}
__device__ __forceinline__ uint32_t f32_to_f16_bits(float f) {
    return (uint32_t)__builtin_bit_cast(uint16_t, (half_t)f);
}

template <int DT> struct dt_traits;
template <> struct dt_traits<BIE_F16> {
    static __device__ __forceinline__ float load(const void* p, long i) { return f16_bits_to_f32(((const uint16_t*)p)[i]); }
    static __device__ __forceinline__ float round(float v) { return (float)(half_t)v; }
    static __device__ __forceinline__ void store(void* p, long i, float v) { ((uint16_t*)p)[i] = (uint16_t)f32_to_f16_bits(v); }
    static constexpr int bytes = 2;
};
template <> struct dt_traits<BIE_BF16> {
    static __device__ __forceinline__ float load(const void* p, long i) { return bf16_bits_to_f32(((const uint16_t*)p)[i]); }
    static __device__ __forceinline__ float round(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
    static __device__ __forceinline__ void store(void* p, long i, float v) { ((uint16_t*)p)[i] = (uint16_t)f32_to_bf16_bits(v); }
    static constexpr int bytes = 2;
};
template <> struct dt_traits<BIE_F32> {
    static __device__ __forceinline__ float load(const void* p, long i) { return ((const float*)p)[i]; }
    static __device__ __forceinline__ float round(float v) { return v; }
    static __device__ __forceinline__ void store(void* p, long i, float v) { ((float*)p)[i] = v; }
    static constexpr int bytes = 4;
};

// Every workspace starts with a 16 KiB head (zero on first use); every scratch user starts behind it.  Two protocols live in it
// and must not share words: the first half holds split-K ARRIVAL TICKETS (one per 64-column output tile, returned to zero
// by the last arriver: mpq_gemv.hip), the second half the tiles' GENERATION words of the tagged-granule reduction
// (monotonic, never reset: mpq_gemv_lut.hip).
constexpr size_t BIE_WS_HEAD_BYTES = 16384;
constexpr int BIE_WS_COUNTERS = (int)(BIE_WS_HEAD_BYTES / 8);  // 2048 tiles = 131072 output columns per launch, each protocol
constexpr int BIE_WS_GEN_OFFSET = BIE_WS_COUNTERS;             // in 32-bit words from the start of the workspace

unsigned next_launch_epoch();  // splitk.hip

// shared split-K epilogue (splitk.hip): y[m][n] = dt( sum_s part[s][m][n] ) (+ bias[n])
int launch_splitk_finalize(const float* part, const void* bias, void* y, int S, int M, int N, int dtype,
                           hipStream_t stream);

}  // namespace bie
