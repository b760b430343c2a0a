// Cross-kernel weight prefetch probe (MI355X): does touching layer l+1's packed weights from kernel l (one dword per 128-byte
// line, nobody waits for the data) make kernel l+1's own reads faster -- through the XCD L2 (same block -> XCD mapping), or
// through the Infinity Cache only (mapping shifted by one XCD)?  And what does a dependent chain of such kernels cost per layer?
//
// consume kernel = the access pattern of the decode GEMV: grid = tiles x slices workgroups of 4 waves, a wave reads `rows`
// packed rows of its 64-column tile (256 contiguous bytes per wave-row, nt loads), sums them, block-reduces and writes one
// 256-byte result row that the NEXT launch reads first (a true y -> x dependency between consecutive launches).
// `work` = dependent FMAs per loaded dword (emulates the lookup arithmetic: ~10 cycles per weight-instruction).
//
// hipcc --offload-arch=gfx950 -O3 -o probe_prefetch probe_prefetch.hip && ./probe_prefetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Args {
    const uint32_t* w;   // [rows_total][N] dwords
    const float* xin;    // 64 floats written by the previous launch
    float* yout;         // [tiles*64]
    const char* pf;      // next layer's weights (or NULL)
    long pf_bytes;
    int N, rows_total, tiles, slices, pf_blocks, pf_mode, work, nt;
    int tail_blocks;     // > 0: the first tail_blocks COMPUTE blocks touch the first tail_bytes of pf after their own rows are summed (fire and forget)
    long tail_bytes;
    int tail_xcd;        // 1: a block touches only the 256-byte segments whose consumer shares its XCD; 2: page touch -- the first 8 blocks (one per XCD) touch one dword every tail_stride bytes of the WHOLE next layer (TLB warm-up)
    long tail_stride;
};

// pf_mode 0: linear split of the lines over the prefetch blocks; 1: XCD-aware (a block touches the 256-byte column segments whose
// consumer block has the same blockIdx % 8); 2: deliberately the NEXT XCD (Infinity Cache only); +4: touch every 64 bytes
__global__ __launch_bounds__(256) void consume(const Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x < a.pf_blocks) {
        if (a.pf == nullptr) return;
        const int mode = a.pf_mode & 3;
        const int step = (a.pf_mode & 4) ? 64 : 128;
        const long row_bytes = (long)a.N * 4;
        uint32_t v = 0;  // ONE destination register, live until the final wait: the compiler cannot hand it to anything else while loads are in flight
        if (mode == 0) {
            const long lines = a.pf_bytes / step;
            const long per = (lines + a.pf_blocks - 1) / a.pf_blocks;
            const long l0 = (long)blockIdx.x * per;
            for (long i = threadIdx.x; i < per; i += 256) {
                const long l = l0 + i;
                if (l < lines) asm volatile("global_load_dword %0, %1, off" : "+v"(v) : "v"(a.pf + l * step));
            }
        } else {
            // segment = 256 bytes of one row = one (row, tile); consumer of tile t (slice-major grid, tiles % 8 == 0) runs on XCD t % 8
            const int xcd = (blockIdx.x + (mode == 2 ? 1 : 0)) & 7;
            const int same = a.pf_blocks / 8;          // prefetch blocks of this XCD class
            const int idx = blockIdx.x / 8;            // my index among them
            const long rows = a.pf_bytes / row_bytes;
            const int tiles_x = a.tiles / 8;           // tiles of this class per row
            const long segs = rows * tiles_x;
            const long per = (segs + same - 1) / same;
            const int lps = 256 / step;                // lines per segment
            for (long i = threadIdx.x; i < per * lps; i += 256) {
                const long s = (long)idx * per + i / lps;
                if (s < segs) {
                    const long r = s / tiles_x;
                    const int t = (int)(s % tiles_x) * 8 + xcd;
                    asm volatile("global_load_dword %0, %1, off" : "+v"(v) : "v"(a.pf + r * row_bytes + (long)t * 256 + (i % lps) * step));
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v)::"memory");
        return;
    }
    const int b = blockIdx.x - a.pf_blocks;
    const int tile = b % a.tiles, slice = b / a.tiles;
    const int rows_per_wave = a.rows_total / (a.slices * 4);
    const int r0 = (slice * 4 + wave) * rows_per_wave;
    const uint32_t* p = a.w + (long)r0 * a.N + tile * 64 + lane;
    const float x = a.xin[lane] * 1e-30f;  // the dependency on the previous launch
    float acc = x;
    for (int r = 0; r < rows_per_wave; r += 16) {
        uint32_t v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = a.nt ? __builtin_nontemporal_load(p + (long)(r + u) * a.N) : p[(long)(r + u) * a.N];
#pragma unroll
        for (int u = 0; u < 16; u++) {
            float f = __uint_as_float(v[u] & 0x3f7fffffu);
            for (int k = 0; k < a.work; k++) f = __builtin_fmaf(f, 0.999f, x);
            acc += f;
        }
    }
    if (a.tail_xcd == 2 && a.pf != nullptr) {
        if (b < 8 && wave == 0) {
            uint32_t v = 0;
            for (long off = (long)lane * a.tail_stride; off < a.pf_bytes; off += 64 * a.tail_stride)
                asm volatile("global_load_dword %0, %1, off" : "+v"(v) : "v"(a.pf + off));
            asm volatile("" : "+v"(v));
        }
    } else if (b < a.tail_blocks && a.pf != nullptr) {
        asm volatile("" : "+v"(acc));  // after the wave's own rows
        uint32_t v = 0;
        if (a.tail_xcd && (a.tiles & 7) == 0) {
            const long row_bytes = (long)a.N * 4;
            const int xcd = b & 7, same = a.tail_blocks / 8, idx = b / 8, tiles_x = a.tiles / 8;
            const long segs = a.tail_bytes / row_bytes * tiles_x;   // (row, tile of my class)
            const long per = (segs + same - 1) / same;
            for (long i = threadIdx.x; i < per * 2; i += 256) {
                const long sg = (long)idx * per + i / 2;
                if (sg < segs) asm volatile("global_load_dword %0, %1, off" : "+v"(v) : "v"(a.pf + (sg / tiles_x) * row_bytes + ((sg % tiles_x) * 8 + xcd) * 256L + (i & 1) * 128));
            }
        } else {
            const long lines = a.tail_bytes / 128, per = (lines + a.tail_blocks - 1) / a.tail_blocks;
            for (long i = threadIdx.x; i < per; i += 256) {
                const long l = (long)b * per + i;
                if (l < lines) asm volatile("global_load_dword %0, %1, off" : "+v"(v) : "v"(a.pf + l * 128));
            }
        }
        asm volatile("" : "+v"(v));  // not waited for: s_endpgm drains
    }
    __shared__ float red[4][64];
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && slice == 0) a.yout[tile * 64 + lane] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

static float run_chain(const std::vector<uint32_t*>& bufs, float* y0, float* y1, int N, int rows, int slices, int pf_blocks, int pf_mode, bool pf, int work,
                       int nt, bool same_buffer, int reps, int tail_blocks = 0, long tail_bytes = 0, int tail_xcd = 0, long tail_stride = 0) {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int tiles = N / 64;
    auto enqueue = [&]() {
        for (size_t i = 0; i < bufs.size(); i++) {
            Args a;
            a.w = same_buffer ? bufs[0] : bufs[i];
            a.xin = (i & 1) ? y1 : y0;
            a.yout = (i & 1) ? y0 : y1;
            a.pf = pf ? (const char*)bufs[(i + 1) % bufs.size()] : nullptr;
            a.pf_bytes = (long)rows * N * 4;
            a.N = N; a.rows_total = rows; a.tiles = tiles; a.slices = slices; a.pf_blocks = pf_blocks; a.pf_mode = pf_mode; a.work = work; a.nt = nt;
            a.tail_blocks = tail_blocks; a.tail_bytes = tail_bytes; a.tail_xcd = tail_xcd; a.tail_stride = tail_stride;
            hipLaunchKernelGGL(consume, dim3(tiles * slices + pf_blocks), dim3(256), 0, st, a);
        }
    };
    enqueue();
    CK(hipStreamSynchronize(st));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    enqueue();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; i++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(st));
    return ms * 1e3f / (reps * bufs.size());
}

int main(int argc, char** argv) {
    const int NB = 48;
    float *y0, *y1;
    CK(hipMalloc(&y0, 1 << 20)); CK(hipMalloc(&y1, 1 << 20));
    CK(hipMemset(y0, 0, 1 << 20)); CK(hipMemset(y1, 0, 1 << 20));
    struct Shape { int N, rows, slices; const char* name; };
    const Shape shapes[] = {{4096, 512, 8, "4096x4096 (8.4 MB)"}, {11008, 512, 4, "4096x11008 (22.5 MB, tiles%8=4)"}, {12288, 512, 4, "4096x12288 q/k/v (25.2 MB)"},
                            {4096, 1536, 8, "12288x4096 (25.2 MB)"}, {22016, 512, 4, "4096x22016 gate/up (45.1 MB)"}};
    for (const Shape& s : shapes) {
        std::vector<uint32_t*> bufs(NB);
        const size_t bytes = (size_t)s.rows * s.N * 4;
        for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0x3c, bytes)); }
        const int nb = (int)(bufs.size());
        (void)nb;
        printf("== %s, %d rotating buffers (%.0f MB), grid %d x 256, us per launch (dependent chain, hipGraph)\n", s.name, NB, bytes * NB / 1e6, s.N / 64 * s.slices);
        for (int work : {0, 4, 10}) {
            const float cold = run_chain(bufs, y0, y1, s.N, s.rows, s.slices, 0, 0, false, work, 1, false, 10);
            const float cold2 = run_chain(bufs, y0, y1, s.N, s.rows, s.slices, 0, 0, false, work, 1, false, 10);
            printf("  work %2d: no prefetch %6.2f %6.2f |", work, cold, cold2);
            for (long stride : {2L << 20, 64L << 10, 4L << 10}) {
                const float t = run_chain(bufs, y0, y1, s.N, s.rows, s.slices, 0, 0, true, work, 1, false, 10, 8, 0, 2, stride);
                printf(" page touch every %4ld KiB by one wave per XCD %6.2f |", stride >> 10, t);
            }
            printf("\n");
        }
        for (auto b : bufs) CK(hipFree(b));
    }
    return 0;
}
