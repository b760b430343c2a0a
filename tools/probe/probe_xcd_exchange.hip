// How long does a tagged 8-byte granule take from one workgroup to another -- the cross-workgroup reduction of the sliced decode launches (mpq_list.hip /
// mpq_gemv_lut.hip: publishers store {fp32, tag} write-through, the reducer polls) -- when both sit on the SAME XCD (same L2) and when they do not, and with
// which load?  Pairs of one-wave workgroups: the producer waits ~4 us, stamps the wall clock (s_memrealtime, 100 MHz) and stores the granule with an
// agent-scope relaxed atomic store (write-through); the consumer polls with (a) agent-scope atomic loads (sc1: what ships), (b) loads that may hit the local
// L2 (sc0 only).  Also prints the XCC id of the first 32 workgroups of a 1-D grid (is blockIdx % 8 the XCD?).
// usage: probe_xcd_exchange      (hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/probe_xcd_exchange tools/probe/probe_xcd_exchange.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long wall() { return __builtin_amdgcn_s_memrealtime(); }

// pairs: workgroup b < P is the consumer of producer b + dist; MODE 0: consumer polls with agent-scope loads, 1: with sc0 loads (L2 may answer)
template <int MODE>
__global__ __launch_bounds__(64) void k_pair(unsigned long long* gran, unsigned long long* out, unsigned* xcc, int P, int dist, unsigned tag) {
    const int b = blockIdx.x, lane = threadIdx.x;
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (lane == 0) xcc[b] = x & 0xf;
    const bool consumer = b < P;
    const int pair = consumer ? b : b - dist;
    if (!consumer && (pair < 0 || pair >= P)) return;
    unsigned long long* g = gran + (long)pair * 64 + lane;
    if (!consumer) {
        const unsigned long long t0 = wall();
        while (wall() - t0 < 400) __builtin_amdgcn_s_sleep(1);  // 4 us: the consumer is polling by now
        const unsigned long long ts = wall();
        __hip_atomic_store(g, ((unsigned long long)tag << 32) | (unsigned)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane == 0) out[pair * 4 + 0] = ts;
        return;
    }
    unsigned long long v;
    int spins = 0;
    bool ok;
    do {
        if constexpr (MODE == 0) v = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(g) : "memory");
        ok = __builtin_amdgcn_ballot_w64((unsigned)(v >> 32) != tag) == 0;
    } while (!ok && ++spins < 200000);
    const unsigned long long te = wall();
    if (lane == 0) { out[pair * 4 + 1] = te; out[pair * 4 + 2] = (unsigned long long)spins; out[pair * 4 + 3] = ok; }
}

template <int MODE>
static int run(const char* what, int dist, unsigned long long* gran, unsigned long long* out, unsigned* xcc, unsigned& tag) {
    const int P = 8, G = 32;
    unsigned long long h[P * 4];
    unsigned hx[G];
    double sum = 0, mn = 1e9, mx = 0;
    int same = 0, n = 0, fails = 0;
    for (int rep = 0; rep < 12; rep++) {
        tag++;
        hipLaunchKernelGGL(k_pair<MODE>, dim3(G), dim3(64), 0, 0, gran, out, xcc, P, dist, tag);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost));
        if (rep < 2) continue;
        for (int p = 0; p < P; p++) {
            if (!h[p * 4 + 3]) { fails++; continue; }
            const double us = (double)(long long)(h[p * 4 + 1] - h[p * 4 + 0]) * 0.01;
            sum += us; n++;
            if (us < mn) mn = us;
            if (us > mx) mx = us;
            same += hx[p] == hx[p + dist];
        }
    }
    printf("%-34s producer = consumer + %2d: %5.2f us store -> seen (min %5.2f, max %5.2f; %d pairs, %d on one XCD, %d never seen)\n", what, dist, n ? sum / n : 0.0, mn, mx, n, same, fails);
    return 0;
}

int main() {
    unsigned long long *gran, *out;
    unsigned* xcc;
    CK(hipMalloc(&gran, 64 * 64 * 8));
    CK(hipMemset(gran, 0, 64 * 64 * 8));
    CK(hipMalloc(&out, 64 * 4 * 8));
    CK(hipMalloc(&xcc, 64 * 4));
    unsigned tag = 100;
    hipLaunchKernelGGL(k_pair<0>, dim3(32), dim3(64), 0, 0, gran, out, xcc, 8, 8, ++tag);
    CK(hipDeviceSynchronize());
    unsigned hx[32];
    CK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost));
    printf("XCC id of workgroups 0..31:");
    for (int i = 0; i < 32; i++) printf(" %u", hx[i]);
    printf("\n");
    if (run<0>("agent-scope loads (sc1, shipped)", 8, gran, out, xcc, tag)) return 1;
    if (run<0>("agent-scope loads (sc1, shipped)", 9, gran, out, xcc, tag)) return 1;
    if (run<1>("sc0 loads (local L2 may answer)", 8, gran, out, xcc, tag)) return 1;
    if (run<1>("sc0 loads (local L2 may answer)", 9, gran, out, xcc, tag)) return 1;
    return 0;
}
