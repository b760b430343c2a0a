// Issue rate and dependent-accumulator behaviour of the matrix instructions this library uses, as bare streams (no memory, no LDS): bf16 32x32x16,
// i8 32x32x32, i8 16x16x64, FP4 32x32x64 (v_mfma_scale ... f8f6f4, cbsz = blgp = 4) and FP4 16x16x128, with 1 / 2 / 4 / 8 independent accumulators per
// wave, one wave per SIMD, every CU busy.  Reports nanoseconds and SHADER cycles (s_memtime) per MFMA.  Round 6: the W8A8 GEMM's counters showed the i8
// pipe 0.43 busy at 32 cycles per instruction while the loop could not go faster, and the FP4 conv ran 60-100 cycles per MFMA: which of these
// instructions really issue every 32 cycles?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
#define ITER 2048

__device__ unsigned long long g_cycles[256];

template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int seed) {
    f16v af[NACC]; i16v ai[NACC]; f4v a4[NACC]; i4v i4[NACC];
    for (int i = 0; i < NACC; i++) { for (int e = 0; e < 16; e++) { af[i][e] = 0.f; ai[i][e] = 0; } for (int e = 0; e < 4; e++) { a4[i][e] = 0.f; i4[i][e] = 0; } }
    bf8v a, b;
    for (int e = 0; e < 8; e++) { a[e] = (__bf16)(float)((threadIdx.x * 7 + e * 3 + seed) % 13 - 6); b[e] = (__bf16)(float)((threadIdx.x * 5 + e + seed) % 11 - 5); }
    i4v xa = {(int)(threadIdx.x * 2654435761u + seed), (int)(threadIdx.x * 40503u + 77), seed * 31 + 5, (int)threadIdx.x ^ 0x5a5a5a5a}, xb = {seed, (int)threadIdx.x * 3, 0x12345678, 0x0f1e2d3c};
    i8v fa = {0x2a2a2a2a, 0x2222aaaa, (int)0xa2a2a2a2, 0x22222222, 0, 0, 0, 0}, fb = {(int)0xaaaa2222, 0x2a2a2a2a, 0x22aa22aa, (int)0xa22aa22a, 0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < 8 / NACC; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                if constexpr (KIND == 0) af[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, af[i], 0, 0, 0);
                else if constexpr (KIND == 1) ai[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xa, xb, ai[i], 0, 0, 0);
                else if constexpr (KIND == 2) i4[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(xa, xb, i4[i], 0, 0, 0);
                else if constexpr (KIND == 3) af[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, af[i], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                else a4[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa, fb, a4[i], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < NACC; i++) { for (int e = 0; e < 16; e++) s += af[i][e] + (float)ai[i][e]; for (int e = 0; e < 4; e++) s += a4[i][e] + (float)i4[i][e]; }
    if (s == 12345.678f) out[0] = s;
    if (threadIdx.x == 0) g_cycles[blockIdx.x] = t1 - t0;
}

template <int KIND, int NACC>
void run(float* d, const char* name, double ops) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<KIND, NACC><<<256, 256>>>(d, 1);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 4; r++) k<KIND, NACC><<<256, 256>>>(d, 2 + r);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc[256];
    (void)hipMemcpyFromSymbol(cyc, HIP_SYMBOL(g_cycles), sizeof(cyc));
    double c = 0; for (int i = 0; i < 256; i++) c += (double)cyc[i]; c /= 256;
    const double n = (double)ITER * 8;  // MFMAs per wave = per SIMD
    printf("%-22s acc=%d: %7.2f ns per MFMA, %6.1f shader cycles per MFMA, clock %.2f GHz -> %7.1f T(FL)OP/s chip\n", name, NACC, ms / 4 * 1e6 / n, c / n, c / (ms / 4 * 1e6),
           ops * n * 1024 / (ms / 4 * 1e-3) / 1e12);
}

int main() {
    float* d; (void)hipMalloc(&d, 64);
#define ALL(KIND, NAME, OPS) run<KIND, 1>(d, NAME, OPS); run<KIND, 2>(d, NAME, OPS); run<KIND, 4>(d, NAME, OPS); run<KIND, 8>(d, NAME, OPS);
    ALL(0, "bf16 32x32x16", 32768.0)
    ALL(1, "i8 32x32x32", 65536.0)
    ALL(2, "i8 16x16x64", 32768.0)
    ALL(3, "fp4 32x32x64 (scale)", 131072.0)
    ALL(4, "fp4 16x16x128 (scale)", 65536.0)
    return 0;
}
