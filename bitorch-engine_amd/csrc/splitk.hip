// Deterministic split-K epilogue shared by the GEMV and the MFMA GEMM:
//   y[m][n] = fl16( sum_{s=0..S-1} part[s][m][n] ) (+ bias[n], rounded again like `out + bias` in torch)
// plus the thread-local error string of the C ABI.
#include "bie_common.h"
#include <atomic>
#include <string.h>
#include <mutex>

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

// ---- device status page: 4 KiB of host-mapped memory the kernels can raise bits in (a reducer whose granules never arrive,
// a dependent entry whose producer never finishes).  Read by the host without any synchronisation: the word is host memory.
static std::atomic<unsigned*> g_status_host{nullptr};
static unsigned* g_status_dev = nullptr;
static std::atomic<unsigned> g_forge_skew{0};
static std::atomic<int> g_forge_spin{0};
static std::atomic<int> g_forge_dep{0};

int status_init() {
    if (g_status_host.load(std::memory_order_acquire)) return BIE_OK;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (g_status_host.load(std::memory_order_acquire)) return BIE_OK;
    void* h = nullptr;
    hipError_t e = hipHostMalloc(&h, 4096, hipHostMallocMapped | hipHostMallocPortable);
    if (e != hipSuccess) { set_error("bie_status_init: hipHostMalloc: %s", hipGetErrorString(e)); return BIE_ERR_HIP; }
    memset(h, 0, 4096);
    void* d = nullptr;
    e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) { (void)hipHostFree(h); set_error("bie_status_init: hipHostGetDevicePointer: %s", hipGetErrorString(e)); return BIE_ERR_HIP; }
    g_status_dev = reinterpret_cast<unsigned*>(d);
    g_status_host.store(reinterpret_cast<unsigned*>(h), std::memory_order_release);
    return BIE_OK;
}
unsigned* device_status_word() { return g_status_host.load(std::memory_order_acquire) ? g_status_dev : nullptr; }
unsigned status_read(bool clear) {
    unsigned* h = g_status_host.load(std::memory_order_acquire);
    if (!h) return 0;
    return clear ? __atomic_exchange_n(h, 0u, __ATOMIC_ACQ_REL) : __atomic_load_n(h, __ATOMIC_ACQUIRE);
}
// called by every launching entry point of the reduction-carrying kernels: an earlier launch's fault is reported once
int status_report(const char* fn) {
    const unsigned s = status_read(true);
    if (s == 0) return BIE_OK;
    set_error("%s: an EARLIER launch raised device status 0x%x (%s%s): its outputs were poisoned with NaN; the status is now cleared", fn, s,
              (s & 1u) ? "a split-K reducer timed out waiting for partial sums; " : "", (s & 2u) ? "a dependent list entry timed out waiting for its producer" : "");
    return BIE_ERR_DEVICE;
}
void test_forge_set(unsigned tag_skew, int spin_limit) { g_forge_skew.store(tag_skew); g_forge_spin.store(spin_limit); }
void test_forge_dep_set(int extra) { g_forge_dep.store(extra); }
int test_forge_dep_get() { return g_forge_dep.load(std::memory_order_relaxed); }
void test_forge_get(unsigned* tag_skew, int* spin_limit) {
    *tag_skew = g_forge_skew.load(std::memory_order_relaxed);
    const int s = g_forge_spin.load(std::memory_order_relaxed);
    *spin_limit = s > 0 ? s : (1 << 22);
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
