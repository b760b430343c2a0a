// Does v_cvt_scalef32_pk_bf16_fp8 multiply by the WHOLE fp32 scale and round once (RNE) to bf16?  If so, fl_bf16(q * s) of two W4 fields -- the
// first rounding of the reference's fl(fl(q * s) - z) -- is ONE instruction per pair: a field in a byte read as fp8 e4m3 is the subnormal q * 2^-9
// (exact), and the scale operand carries s * 2^9.  Also v_cvt_scalef32_pk_f32_fp8 (is the fp32 result the exact product?) and the half selection.
// Compares every (q0, q1) pair against software for a set of scales (bf16 values times 512: ordinary, tiny, huge).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void k(const uint32_t* src, const float* scale, uint32_t* o_lo, uint32_t* o_hi, f2* o_f32, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o_lo[i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(src[i], scale[i], false));
    o_hi[i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(src[i], scale[i], true));
    o_f32[i] = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(src[i], scale[i], false);
}
static float bf16_to_f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f_to_bf16_rne(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
int main() {
    std::vector<uint16_t> sb;  // scales as bf16 bit patterns
    for (uint32_t e = 0x30; e < 0x50; e++) for (uint32_t m = 0; m < 128; m += 7) sb.push_back((uint16_t)((e << 7) | m));  // 2^-31 .. 2^+1 region
    for (uint32_t m = 0; m < 128; m++) sb.push_back((uint16_t)((0x78u << 7) | m));                                       // ~2^-7: typical scales, every mantissa
    sb.push_back(0x0080); sb.push_back(0x0001); sb.push_back(0x7f00); sb.push_back(0x7e80); sb.push_back(0xbc12); sb.push_back(0x7f80); sb.push_back(0);
    std::vector<uint32_t> src; std::vector<float> sc;
    for (uint16_t s : sb)
        for (uint32_t q0 = 0; q0 < 16; q0++) for (uint32_t q1 = 0; q1 < 16; q1++) {
            src.push_back(q0 | (q1 << 8) | ((15 - q0) << 16) | ((q1 ^ 5u) << 24));
            sc.push_back(bf16_to_f(s) * 512.0f);
        }
    const int n = (int)src.size();
    uint32_t *d_src, *d_lo, *d_hi; float* d_sc; f2* d_f;
    hipMalloc(&d_src, n * 4); hipMalloc(&d_lo, n * 4); hipMalloc(&d_hi, n * 4); hipMalloc(&d_sc, n * 4); hipMalloc(&d_f, n * 8);
    hipMemcpy(d_src, src.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(d_sc, sc.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, d_src, d_sc, d_lo, d_hi, d_f, n);
    std::vector<uint32_t> lo(n), hi(n); std::vector<float> f(2 * n);
    hipMemcpy(lo.data(), d_lo, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hi.data(), d_hi, n * 4, hipMemcpyDeviceToHost); hipMemcpy(f.data(), d_f, n * 8, hipMemcpyDeviceToHost);
    long bad_lo = 0, bad_hi = 0, bad_f = 0, shown = 0, inexact_scale = 0;
    for (int i = 0; i < n; i++) {
        const float s512 = sc[i], s = s512 / 512.0f;
        if (s * 512.0f != s512 || std::isinf(s512)) inexact_scale++;
        const uint32_t q[4] = {src[i] & 15, (src[i] >> 8) & 15, (src[i] >> 16) & 15, (src[i] >> 24) & 15};
        uint16_t want[4]; float wf[4];
        for (int j = 0; j < 4; j++) { wf[j] = (float)q[j] * s; want[j] = f_to_bf16_rne(wf[j]); }  // q * s: exact in fp32 (12 significant bits) unless it overflows / underflows
        const uint32_t wl = want[0] | ((uint32_t)want[1] << 16), wh = want[2] | ((uint32_t)want[3] << 16);
        if (lo[i] != wl) { bad_lo++; if (shown++ < 12) printf("lo: scale %g q (%u,%u): got %08x want %08x\n", s, q[0], q[1], lo[i], wl); }
        if (hi[i] != wh) { bad_hi++; if (shown++ < 12) printf("hi: scale %g q (%u,%u): got %08x want %08x\n", s, q[2], q[3], hi[i], wh); }
        if (memcmp(&f[2 * i], &wf[0], 4) || memcmp(&f[2 * i + 1], &wf[1], 4)) { bad_f++; if (shown++ < 12) printf("f32: scale %g q (%u,%u): got %g %g want %g %g\n", s, q[0], q[1], f[2 * i], f[2 * i + 1], wf[0], wf[1]); }
    }
    printf("cases %d (scales %zu x 256 pairs): pk_bf16 low half mismatches %ld, high half (op_sel) %ld, pk_f32 %ld; scales whose x512 is inexact / infinite: %ld\n",
           n, sb.size(), bad_lo, bad_hi, bad_f, inexact_scale);
    return 0;
}
