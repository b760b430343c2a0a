// Can one wave hide VALU work under its own MFMAs on gfx950?  Loop: 8 x (1 MFMA 32x32x16 bf16 + NV independent VALU ops),
// 1 or 2 waves per SIMD.  Reports cycles per MFMA slot (32 = MFMA pipe bound).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
#define ITER 4096

template <int NV, int KIND>
__global__ __launch_bounds__(512) void k(float* out, uint32_t seed) {
    f16v acc[8];
    for (int i = 0; i < 8; i++) for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    bf8v a, b;
    for (int e = 0; e < 8; e++) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(seed + e); }
    uint32_t r[8];
    for (int i = 0; i < 8; i++) r[i] = seed + threadIdx.x + i;
    uint32_t c0 = seed * 7, c1 = seed * 13;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; v++) {
                if constexpr (KIND == 0) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[(v + i) & 7]) : "v"(c0), "v"(c1));
                else asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[(v + i) & 7]) : "v"(c0));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) for (int e = 0; e < 16; e++) s += acc[i][e];
    uint32_t x = 0;
    for (int i = 0; i < 8; i++) x ^= r[i];
    if (s == 12345.f && x == 77) out[0] = s;
}

template <int NV, int KIND>
void run(float* d, int threads) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NV, KIND><<<256, threads>>>(d, 1);
    (void)hipEventRecord(e0);
    k<NV, KIND><<<256, threads>>>(d, 2);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double slots = (double)ITER * 8 * (threads / 256);  // MFMAs per SIMD
    printf("NV=%d kind=%d waves/SIMD=%d: %7.3f ms  %6.1f ns per MFMA slot per SIMD (%5.1f cycles @2.4GHz)  -> %6.1f TFLOP/s chip\n", NV, KIND, threads / 256, ms,
           ms * 1e6 / slots, ms * 1e6 / slots * 2.4, 32768.0 * slots * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
    float* d; (void)hipMalloc(&d, 64);
    run<0, 0>(d, 256); run<2, 0>(d, 256); run<4, 0>(d, 256); run<6, 0>(d, 256); run<8, 0>(d, 256); run<12, 0>(d, 256);
    run<4, 1>(d, 256); run<8, 1>(d, 256);
    run<0, 0>(d, 512); run<4, 0>(d, 512); run<6, 0>(d, 512); run<8, 0>(d, 512); run<12, 0>(d, 512);
    return 0;
}
