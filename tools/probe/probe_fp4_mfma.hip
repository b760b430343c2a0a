// Issue rate of v_mfma_scale_f32_32x32x64_f8f6f4 (FP4 x FP4) on gfx950: cycles per MFMA (s_memtime) and wall-clock TFLOP/s for
// NACC independent accumulators per wave, W waves per SIMD, on `blocks` workgroups.  hipcc --offload-arch=gfx950 -O3 -o probe_fp4_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NACC, int FMT, int THREADS>
__global__ __launch_bounds__(THREADS) void k32(const int* in, float* out, unsigned long long* cyc, int iters) {
    v8i a, b;
    for (int i = 0; i < 8; i++) { a[i] = in[threadIdx.x + 64 * i]; b[i] = in[threadIdx.x + 64 * i + 7]; }
    v16f acc[NACC];
    for (int n = 0; n < NACC; n++) for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[n], FMT, FMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int n = 0; n < NACC; n++) for (int r = 0; r < 16; r++) s += acc[n][r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC, int FMT, int THREADS>
__global__ __launch_bounds__(THREADS) void k16(const int* in, float* out, unsigned long long* cyc, int iters) {
    v8i a, b;
    for (int i = 0; i < 8; i++) { a[i] = in[threadIdx.x + 64 * i]; b[i] = in[threadIdx.x + 64 * i + 7]; }
    v4f acc[NACC];
    for (int n = 0; n < NACC; n++) for (int r = 0; r < 4; r++) acc[n][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[n], FMT, FMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int n = 0; n < NACC; n++) for (int r = 0; r < 4; r++) s += acc[n][r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
typedef short v8s __attribute__((ext_vector_type(8)));
template <int NACC, int THREADS>
__global__ __launch_bounds__(THREADS) void kbf(const int* in, float* out, unsigned long long* cyc, int iters) {
    v8s a, b;
    for (int i = 0; i < 8; i++) { a[i] = (short)(0x3f80 ^ (in[threadIdx.x + 64 * i] & 0x807f)); b[i] = (short)(0x3f80 ^ (in[threadIdx.x + 64 * i + 7] & 0x807f)); }
    v16f acc[NACC];
    for (int n = 0; n < NACC; n++) for (int r = 0; r < 16; r++) acc[n][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int n = 0; n < NACC; n++) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int n = 0; n < NACC; n++) for (int r = 0; r < 16; r++) s += acc[n][r];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <class K>
static void run(const char* name, K kern, int threads, int blocks, int nacc, double flop_per_mfma, const int* in, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, in, out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * nacc;
    printf("%-44s blocks %4d waves/blk %d: %6.1f memtime-ticks per MFMA per wave, wall %8.1f us, %7.1f TFLOP/s, %5.1f ns per MFMA per wave\n", name, blocks, threads / 64,
           (double)c / nm, ms * 1e3, flop_per_mfma * nm * (threads / 64) * blocks / (ms * 1e-3) / 1e12, ms * 1e6 / nm);
}
int main() {
    int* in; float* out; unsigned long long* cyc;
    hipMalloc(&in, 1 << 16); hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 8 * 4096);
    int* h = (int*)malloc(1 << 16);
    srand(1);
    for (int i = 0; i < (1 << 14); i++) { unsigned v = 0; for (int j = 0; j < 8; j++) v |= ((rand() & 1) ? 0x2u : 0xau) << (4 * j); h[i] = (int)v; }
    hipMemcpy(in, h, 1 << 16, hipMemcpyHostToDevice);
    const double F32 = 2.0 * 32 * 32 * 64, F16 = 2.0 * 16 * 16 * 128;
    for (int blocks : {16, 256, 512}) {
        run("32x32x64 fp4, 16 acc, 1 wave/SIMD", k32<16, 4, 256>, 256, blocks, 16, F32, in, out, cyc);
        run("32x32x64 fp4,  4 acc, 1 wave/SIMD", k32<4, 4, 256>, 256, blocks, 4, F32, in, out, cyc);
        run("32x32x64 fp4,  4 acc, 2 waves/SIMD", k32<4, 4, 512>, 512, blocks, 4, F32, in, out, cyc);
        run("32x32x64 fp4,  4 acc, 4 waves/SIMD", k32<4, 4, 1024>, 1024, blocks, 4, F32, in, out, cyc);
        run("32x32x16 bf16 (random +-[1,2) values), 16 acc, 1 wave/SIMD", kbf<16, 256>, 256, blocks, 16, 2.0 * 32 * 32 * 16, in, out, cyc);
        run("32x32x64 fp8 (e4m3), 16 acc, 1 wave/SIMD", k32<16, 0, 256>, 256, blocks, 16, F32, in, out, cyc);
        run("16x16x128 fp4, 16 acc, 1 wave/SIMD", k16<16, 4, 256>, 256, blocks, 16, F16, in, out, cyc);
        run("16x16x128 fp4, 16 acc, 2 waves/SIMD", k16<16, 4, 512>, 512, blocks, 16, F16, in, out, cyc);
    }
    return 0;
}
