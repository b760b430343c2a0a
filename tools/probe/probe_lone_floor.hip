// What does a LONE decode launch cost on this chip before any dequantisation?  A chain of launches in one captured HIP graph, each over its own buffer
// (the set is > 0.8 GB: nothing stays in the 256 MiB memory-side cache), in three bodies:
//   empty   -- the same grid, every wave returns at once: the launch-to-launch floor of a graph's kernel nodes;
//   stream  -- the lone plan's memory picture and nothing else: a wave requests U units of 16 packed rows x 256 B (nontemporal dword loads through a buffer
//              descriptor, all in flight at once), adds them up, the workgroup's eight waves meet in LDS and store 64 floats: no table, no lookup, no
//              cross-workgroup reduction;
//   stream2 -- the same with the rows requested as 16-byte loads (four rows per instruction are NOT contiguous in the packed layout: this is the
//              picture a re-laid-out weight matrix would give, not one the reference's format allows -- an upper bound only).
// bie_mpq_forward at M = 1: 4096x4096 5.9 us, 4096->11008 9.6-10.3 us, gate/up (2 x 4096->11008 in one launch) 14.6-15.5 us.
// usage: probe_lone_floor            (hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/probe_lone_floor tools/probe/probe_lone_floor.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_empty(const uint32_t* w, float* y) {
    if (w == nullptr) y[0] = 0.f;
}

// rows: packed rows of the layer (K / 8), N columns; a workgroup = 8 waves = 8 consecutive units of one 64-column tile
template <int U>
__global__ __launch_bounds__(512) void k_stream(const uint32_t* __restrict__ w, float* __restrict__ y, int N, int rows, int tiles) {
    __shared__ float red[8 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x % tiles, slice = blockIdx.x / tiles;
    const int r0 = (slice * 8 + wave) * U * 16;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)((long)rows * N * 4), 0x00020000);
    uint32_t v[U * 16];
    unsigned soff = (unsigned)r0 * (unsigned)N * 4u;
    const unsigned col = (unsigned)(tile * 64 + lane) * 4u;
#pragma unroll
    for (int i = 0; i < U * 16; i++) {
        v[i] = r0 + i < rows ? __builtin_amdgcn_raw_buffer_load_b32(rsrc, col, soff, 2) : 0u;
        soff += (unsigned)N * 4u;
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < U * 16; i++) s ^= v[i];
    red[wave * 64 + lane] = (float)(s & 0xffff);
    __syncthreads();
    if (wave) return;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) t += red[i * 64 + lane];
    y[(long)slice * tiles * 64 + tile * 64 + lane] = t;
}

// the same bytes as contiguous 16-byte pieces per lane (1 KiB per wave instruction): not the packed layout, an upper bound
template <int U>
__global__ __launch_bounds__(512) void k_stream16(const u4* __restrict__ w, float* __restrict__ y, long n16) {
    __shared__ float red[8 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long base = ((long)blockIdx.x * 8 + wave) * (U * 4) * 64 + lane;
    u4 v[U * 4];
#pragma unroll
    for (int i = 0; i < U * 4; i++) {
        const long a = base + (long)i * 64;
        v[i] = a < n16 ? __builtin_nontemporal_load(w + a) : u4{0, 0, 0, 0};
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < U * 4; i++) s ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    red[wave * 64 + lane] = (float)(s & 0xffff);
    __syncthreads();
    if (wave) return;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) t += red[i * 64 + lane];
    y[(long)blockIdx.x * 64 + lane] = t;
}

struct Shape { const char* name; int K, N, layers; };

template <typename F>
static int time_chain(const char* what, const Shape& sh, int nl, F launch, double bytes) {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < nl; i++) launch(i, st);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = 0; r < 5; r++) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    float best = 1e30f, sum = 0.f;
    const int reps = 20;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        sum += ms;
        if (ms < best) best = ms;
    }
    const double us = sum / reps * 1e3 / nl, usb = best * 1e3 / nl;
    printf("%-22s %-9s %6.2f us per launch (best replay %6.2f)", sh.name, what, us, usb);
    if (bytes > 0) printf("   %5.2f TB/s over the whole launch", bytes / us / 1e6);
    printf("\n");
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    CK(hipStreamDestroy(st));
    return 0;
}

int main() {
    const Shape shapes[] = {{"4096x4096", 4096, 4096, 96}, {"4096x11008", 4096, 11008, 40}, {"11008x4096", 11008, 4096, 40}, {"gate/up 4096x22016", 4096, 22016, 20}};
    float* y;
    CK(hipMalloc(&y, 64 << 20));
    for (const Shape& sh : shapes) {
        const int rows = sh.K / 8, tiles = sh.N / 64, units = rows / 16;
        const size_t bytes = (size_t)rows * sh.N * 4;
        std::vector<uint32_t*> w(sh.layers);
        for (int i = 0; i < sh.layers; i++) {
            CK(hipMalloc(&w[i], bytes));
            CK(hipMemset(w[i], 0x5a + i, bytes));
        }
        CK(hipDeviceSynchronize());
        const int grid1 = tiles * ((units + 7) / 8), grid2 = tiles * ((units + 15) / 16);
        if (time_chain("empty", sh, sh.layers, [&](int i, hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(grid1), dim3(512), 0, st, (const uint32_t*)w[i], y); }, 0)) return 1;
        if (time_chain("stream U=1", sh, sh.layers, [&](int i, hipStream_t st) { hipLaunchKernelGGL(k_stream<1>, dim3(grid1), dim3(512), 0, st, (const uint32_t*)w[i], y, sh.N, rows, tiles); }, (double)bytes)) return 1;
        if (time_chain("stream U=2", sh, sh.layers, [&](int i, hipStream_t st) { hipLaunchKernelGGL(k_stream<2>, dim3(grid2), dim3(512), 0, st, (const uint32_t*)w[i], y, sh.N, rows, tiles); }, (double)bytes)) return 1;
        const long n16 = (long)(bytes / 16);
        const int g16 = (int)((n16 + 8 * 4 * 64 - 1) / (8 * 4 * 64));
        if (time_chain("16B U=1", sh, sh.layers, [&](int i, hipStream_t st) { hipLaunchKernelGGL(k_stream16<1>, dim3(g16), dim3(512), 0, st, (const u4*)w[i], y, n16); }, (double)bytes)) return 1;
        for (int i = 0; i < sh.layers; i++) CK(hipFree(w[i]));
    }
    return 0;
}
