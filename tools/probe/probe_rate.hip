// Issue-rate probe for the VALU instructions the dequant paths are built from (gfx950).  Each test runs ITER x 16
// independent instructions per wave, 1024 blocks x 256 threads; reports cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t r[16];
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = 0x3f803f80u;
    uint64_t w0 = ((uint64_t)a << 32) | b, w1 = 0x3f8000003f800000ull;
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = a + i;
    uint64_t r2[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r2[i] = w0 + i;
    for (int it = 0; it < ITER; it++) {
#define X(i)                                                                                                   \
    if constexpr (OP == 0) asm volatile("v_cvt_f32_ubyte1_e32 %0, %1" : "=v"(r[i]) : "v"(r[i]));               \
    else if constexpr (OP == 1) asm volatile("v_cvt_pk_f32_fp8_e32 %0, %1" : "=v"(r2[i]) : "v"(r[i]));          \
    else if constexpr (OP == 2) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(c)); \
    else if constexpr (OP == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(a)); \
    else if constexpr (OP == 4) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r2[i]) : "v"(r2[i]), "v"(w1));    \
    else if constexpr (OP == 5) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[i]), "v"(a), "v"(b)); \
    else if constexpr (OP == 6) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(c));   \
    else if constexpr (OP == 7) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(c));  \
    else if constexpr (OP == 8) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[i]), "v"(a), "v"(b)); \
    else if constexpr (OP == 9) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(c));       \
    else if constexpr (OP == 10) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r2[i]) : "v"(r2[i]), "v"(w1));  \
    else if constexpr (OP == 11) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(r[i]) : "v"(r[i]));            \
    else if constexpr (OP == 12) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(c));     \
    else if constexpr (OP == 13) asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(c)); \
    else if constexpr (OP == 14) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r2[i]) : "v"(w0), "v"(w1)); \
    else if constexpr (OP == 15) asm volatile("v_bfe_u32 %0, %1, 4, 4" : "=v"(r[i]) : "v"(r[i]));
        REP16(X)
#undef X
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc += r[i] + (uint32_t)r2[i] + (uint32_t)(r2[i] >> 32);
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <int OP>
void run(const char* name, uint32_t* d) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 2048;  // 8 blocks of 4 waves per CU -> 8 waves per SIMD
    k<OP><<<blocks, 256>>>(d, 1);
    (void)hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, 2);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_simd = (double)blocks * 4 / (256.0 * 4) * ITER * 16;
    printf("%-28s %8.3f ms  -> %6.2f ns per wave-instr per SIMD (= %5.2f cycles @2.4GHz)\n", name, ms, ms * 1e6 / wave_instr_per_simd,
           ms * 1e6 / wave_instr_per_simd * 2.4);
}

int main() {
    uint32_t* d; (void)hipMalloc(&d, 4096);
    run<0>("v_cvt_f32_ubyte1", d);
    run<1>("v_cvt_pk_f32_fp8", d);
    run<2>("v_dot2_f32_bf16", d);
    run<3>("v_cvt_pk_bf16_f32", d);
    run<4>("v_pk_mul_f32", d);
    run<5>("v_perm_b32", d);
    run<6>("v_pk_fma_f16", d);
    run<7>("v_dot2_f32_f16", d);
    run<8>("v_and_or_b32", d);
    run<9>("v_fma_f32", d);
    run<10>("v_pk_add_f32", d);
    run<11>("v_lshlrev_b32", d);
    run<12>("v_pk_mul_f16", d);
    run<13>("v_cvt_scalef32_pk_bf16_fp8", d);
    run<14>("v_pk_fma_f32", d);
    run<15>("v_bfe_u32", d);
    return 0;
}
