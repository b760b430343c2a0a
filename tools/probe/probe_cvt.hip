// Probe of gfx950 conversion instructions (semantics are not documented in this image): run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

__global__ void probe(const uint32_t* in, float* out, float scale, float z) {
    const int i = threadIdx.x;
    const uint32_t v = in[i];
    // (a) packed fp8 -> 2 x f32
    float2_t lo = __builtin_amdgcn_cvt_pk_f32_fp8(v, false);
    float2_t hi = __builtin_amdgcn_cvt_pk_f32_fp8(v, true);
    out[i * 16 + 0] = lo.x; out[i * 16 + 1] = lo.y; out[i * 16 + 2] = hi.x; out[i * 16 + 3] = hi.y;
    // (b) scaled conversion fp8 -> bf16 pair
    bf16x2_t s0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, scale, false);
    bf16x2_t s1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, scale, true);
    out[i * 16 + 4] = (float)s0.x; out[i * 16 + 5] = (float)s0.y; out[i * 16 + 6] = (float)s1.x; out[i * 16 + 7] = (float)s1.y;
    // (c) dot2 as unpack-and-subtract
    const float nz = -z;
    bf16x2_t t = s0;
    uint32_t one_lo = 0x00003f80u, one_hi = 0x3f800000u;
    bf16x2_t sel_lo, sel_hi;
    memcpy(&sel_lo, &one_lo, 4); memcpy(&sel_hi, &one_hi, 4);
    out[i * 16 + 8] = __builtin_amdgcn_fdot2_f32_bf16(t, sel_lo, nz, false);
    out[i * 16 + 9] = __builtin_amdgcn_fdot2_f32_bf16(t, sel_hi, nz, false);
    out[i * 16 + 10] = (float)t.x - z; out[i * 16 + 11] = (float)t.y - z;
    // (d) scaled conversion to f32
    float2_t f0 = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(v, scale, false);
    out[i * 16 + 12] = f0.x; out[i * 16 + 13] = f0.y;
    out[i * 16 + 14] = 0; out[i * 16 + 15] = 0;
}

int main() {
    uint32_t h[64];
    for (int i = 0; i < 64; i++) h[i] = (uint32_t)((i * 4) & 0xF) | (((i * 4 + 1) & 0xF) << 8) | (((i * 4 + 2) & 0xF) << 16) | (((i * 4 + 3) & 0xF) << 24);
    uint32_t* d; float* o; float ho[64 * 16];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const float scales[3] = {512.0f, 512.0f * 0.0123f, 3.0f};
    for (int s = 0; s < 3; s++) {
        probe<<<1, 64>>>(d, o, scales[s], 0.07f);
        hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        printf("== scale %g\n", scales[s]);
        for (int i = 0; i < 4; i++) {
            printf("in %08x:", h[i]);
            for (int j = 0; j < 14; j++) printf(" %.9g", ho[i * 16 + j]);
            printf("\n");
        }
    }
    return 0;
}
