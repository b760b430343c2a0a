// Deterministic split-K epilogue shared by the GEMV and the MFMA GEMM:
//   y[m][n] = fl16( sum_{s=0..S-1} part[s][m][n] ) (+ bias[n], rounded again like `out + bias` in torch)
// plus the thread-local error string of the C ABI.
#include "bie_common.h"
#include <atomic>
#include <string.h>

namespace bie {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* get_error() { return g_err; }

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return BIE_ERR_HIP;
    }
    return BIE_OK;
}

template <int DT>
__global__ __launch_bounds__(256) void splitk_finalize_kernel(const float* __restrict__ part, const void* __restrict__ bias,
                                                              void* __restrict__ y, int S, long MN, int N) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= MN) return;
    float v = 0.0f;
    for (int s0 = 0; s0 < S; s0 += 8) {  // 8 independent loads in flight, summed in slab order
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) t[j] = part[(long)((s0 + j < S) ? s0 + j : S - 1) * MN + i];
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (s0 + j < S) v += t[j];
    }
    float o = dt_traits<DT>::round(v);
    if (bias) o = o + dt_traits<DT>::load(bias, i % N);
    dt_traits<DT>::store(y, i, o);
}

// Upper 24 bits of the granule tags of one launch of a tagged-granule reduction (mpq_gemv_lut.hip, mbwq.hip): a process-wide call
// number, never 0 (tag 0 = granule never written), shared by every kernel that uses the generation words of a workspace head.
unsigned next_launch_epoch() {
    static std::atomic<unsigned> calls{0};
    unsigned e = (calls.fetch_add(1, std::memory_order_relaxed) + 1u) & 0xffffffu;
    if (e == 0) e = (calls.fetch_add(1, std::memory_order_relaxed) + 1u) & 0xffffffu;
    return e << 8;
}

int launch_splitk_finalize(const float* part, const void* bias, void* y, int S, int M, int N, int dtype, hipStream_t st) {
    const long MN = (long)M * N;
    dim3 grid((unsigned)cdivl(MN, 256));
    if (dtype == BIE_F16)
        hipLaunchKernelGGL(splitk_finalize_kernel<BIE_F16>, grid, dim3(256), 0, st, part, bias, y, S, MN, N);
    else if (dtype == BIE_BF16)
        hipLaunchKernelGGL(splitk_finalize_kernel<BIE_BF16>, grid, dim3(256), 0, st, part, bias, y, S, MN, N);
    else
        hipLaunchKernelGGL(splitk_finalize_kernel<BIE_F32>, grid, dim3(256), 0, st, part, bias, y, S, MN, N);
    return check_launch("splitk_finalize_kernel");
}

}  // namespace bie
