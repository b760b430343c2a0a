// v_mfma_f32_4x4x4_16b_f16 with the A-operand broadcast (cbsz / abid) on gfx950: which lanes' A registers feed which blocks?
// Expectation under test: with cbsz = 4 every one of the 16 blocks multiplies the A block held by lanes 4 abid .. 4 abid + 3 (lane 4 abid + i = row i,
// four k in its register pair) with ITS OWN B block; cbsz = 3: blocks 0-7 use block abid, blocks 8-15 block 8 + abid.  Prints the number of wrong lanes.
// hipcc --offload-arch=gfx950 -O2 -o probe_mfma_bcast probe_mfma_bcast.hip && ./probe_mfma_bcast
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int CBSZ, int ABID>
__global__ void k(const h4* A, const h4* B, f4* D) {
    const int l = threadIdx.x;
    f4 c = {0, 0, 0, 0};
    D[l] = __builtin_amdgcn_mfma_f32_4x4x4f16(A[l], B[l], c, CBSZ, ABID, 0);
}
template <int CBSZ, int ABID>
int run(const std::vector<_Float16>& a, const std::vector<_Float16>& b, h4* dA, h4* dB, f4* dD) {
    hipLaunchKernelGGL((k<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipDeviceSynchronize());
    std::vector<float> d(256);
    CK(hipMemcpy(d.data(), dD, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        const int blk = l / 4, j = l % 4;
        const int span = 1 << CBSZ;                       // blocks sharing one A block
        const int src = CBSZ ? (blk / span) * span + ABID : blk;
        for (int i = 0; i < 4; i++) {
            float want = 0;
            for (int kk = 0; kk < 4; kk++) want += (float)a[(src * 4 + i) * 4 + kk] * (float)b[l * 4 + kk];  // A: lane 4 src + i, B: lane l = (block, column j)
            if (std::fabs(want - d[l * 4 + i]) > 1e-3f * (1 + std::fabs(want))) { bad++; break; }
        }
        (void)j;
    }
    printf("cbsz=%d abid=%2d: %d lanes differ from the expectation\n", CBSZ, ABID, bad);
    return bad;
}
int main() {
    std::vector<_Float16> a(256), b(256);
    for (int i = 0; i < 256; i++) { a[i] = (_Float16)(float)((i * 37 + 11) % 29 - 14); b[i] = (_Float16)(float)((i * 53 + 5) % 23 - 11); }
    h4 *dA, *dB; f4* dD;
    CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, a.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, b.data(), 512, hipMemcpyHostToDevice));
    int bad = 0;
    bad += run<0, 0>(a, b, dA, dB, dD);
    bad += run<4, 0>(a, b, dA, dB, dD); bad += run<4, 1>(a, b, dA, dB, dD); bad += run<4, 7>(a, b, dA, dB, dD); bad += run<4, 15>(a, b, dA, dB, dD);
    bad += run<3, 0>(a, b, dA, dB, dD); bad += run<3, 5>(a, b, dA, dB, dD);
    bad += run<2, 3>(a, b, dA, dB, dD);
    printf(bad ? "MISMATCH\n" : "all as expected\n");
    return 0;
}
