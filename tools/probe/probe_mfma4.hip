// Hazards of v_mfma_f32_4x4x4_16b_f16 on gfx950, measured (the compiler's wait-state tables returned wrong sums in the exl2 decode kernel).
// One wave; every sequence is ONE inline-asm block, so no wait state is inserted behind our back.  For each test the minimal number of
// wait states n (s_nop n-1) that gives the right answer in 64 lanes x 200 repetitions is printed ("-" = wrong at every n tried).
//   raw_valu   mfma -> n -> v_mov reads the result
//   war_b/a    mfma -> n -> v_mov OVERWRITES its B / A operand (result read much later)
//   raw_src_b/a  v_mov_b32 x 2 write the B / A operand -> n -> mfma
//   chain      mfma -> n -> mfma with SrcC = the first result (same registers)
//   chain_x    mfma d1 -> n -> mfma d2 with SrcC = d1 (different destination)
//   dst_ovl_b  destination registers = {B operand, ...};  dst_ovl_c  destination v[12:15] with SrcC v[14:17]
//   raw_vmov   mfma -> n -> v_mov_b32 reads the result (raw_valu reads it with a global store)
// hipcc --offload-arch=gfx950 -O2 -o probe_mfma4 probe_mfma4.hip && ./probe_mfma4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define NOPS(n) "s_nop " #n "\n\t"
template <int TEST, int N>
__global__ void probe(const u2* A, const u2* B, const f4* C, f4* out) {
    const int l = threadIdx.x & 63;
    out += (size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64;
    u2 a = A[l], b = B[l];
    if (gridDim.x > 1) {  // contention: every wave of the chip keeps its SIMD's matrix pipe busy right up to the sequence under test
        f4 t0 = {0, 0, 0, 0}, t1 = t0, t2 = t0, t3 = t0;
        for (int r = 0; r < 8 + (int)(blockIdx.x & 7); r++)
            asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %4, %5, %0\n\tv_mfma_f32_4x4x4_16b_f16 %1, %4, %5, %1\n\tv_mfma_f32_4x4x4_16b_f16 %2, %4, %5, %2\n\t"
                         "v_mfma_f32_4x4x4_16b_f16 %3, %4, %5, %3\n\t" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(a), "v"(b));
        asm volatile("" ::"v"(t0), "v"(t1), "v"(t2), "v"(t3));
    }
    f4 c = C[l], d = {0, 0, 0, 0}, d2 = {0, 0, 0, 0};
    u2 junk = u2{0x7e007e00u, 0x7e007e00u};  // NaNs
#define W(n_) if constexpr (N == n_)
#define SEQ(pre, post)                                                                                                   \
    W(0) asm volatile(pre post : "+v"(d), "+v"(d2), "+v"(a), "+v"(b), "+v"(c) : "v"(junk));                               \
    W(1) asm volatile(pre NOPS(0) post : "+v"(d), "+v"(d2), "+v"(a), "+v"(b), "+v"(c) : "v"(junk));                       \
    W(2) asm volatile(pre NOPS(1) post : "+v"(d), "+v"(d2), "+v"(a), "+v"(b), "+v"(c) : "v"(junk));                       \
    W(3) asm volatile(pre NOPS(2) post : "+v"(d), "+v"(d2), "+v"(a), "+v"(b), "+v"(c) : "v"(junk));                       \
    W(4) asm volatile(pre NOPS(3) post : "+v"(d), "+v"(d2), "+v"(a), "+v"(b), "+v"(c) : "v"(junk));                       \
    W(5) asm volatile(pre NOPS(4) post : "+v"(d), "+v"(d2), "+v"(a), "+v"(b), "+v"(c) : "v"(junk));                       \
    W(6) asm volatile(pre NOPS(5) post : "+v"(d), "+v"(d2), "+v"(a), "+v"(b), "+v"(c) : "v"(junk));                       \
    W(7) asm volatile(pre NOPS(6) post : "+v"(d), "+v"(d2), "+v"(a), "+v"(b), "+v"(c) : "v"(junk));                       \
    W(8) asm volatile(pre NOPS(7) post : "+v"(d), "+v"(d2), "+v"(a), "+v"(b), "+v"(c) : "v"(junk));                       \
    W(10) asm volatile(pre NOPS(7) NOPS(1) post : "+v"(d), "+v"(d2), "+v"(a), "+v"(b), "+v"(c) : "v"(junk));              \
    W(12) asm volatile(pre NOPS(7) NOPS(3) post : "+v"(d), "+v"(d2), "+v"(a), "+v"(b), "+v"(c) : "v"(junk));              \
    W(16) asm volatile(pre NOPS(7) NOPS(7) post : "+v"(d), "+v"(d2), "+v"(a), "+v"(b), "+v"(c) : "v"(junk));              \
    W(20) asm volatile(pre NOPS(7) NOPS(7) NOPS(3) post : "+v"(d), "+v"(d2), "+v"(a), "+v"(b), "+v"(c) : "v"(junk));
#define LONG NOPS(7) NOPS(7) NOPS(7) NOPS(7)
    if constexpr (TEST == 0) {  // raw_valu: d = mfma(a,b,c); n; d2.x = d.x ...
        f4* po = out + l;
        W(0) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %3\n\t" "global_store_dwordx4 %4, %0, off\n\t" LONG : "+v"(d) : "v"(a), "v"(b), "v"(c), "v"(po) : "memory");
        W(1) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %3\n\t" NOPS(0) "global_store_dwordx4 %4, %0, off\n\t" LONG : "+v"(d) : "v"(a), "v"(b), "v"(c), "v"(po) : "memory");
        W(2) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %3\n\t" NOPS(1) "global_store_dwordx4 %4, %0, off\n\t" LONG : "+v"(d) : "v"(a), "v"(b), "v"(c), "v"(po) : "memory");
        W(3) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %3\n\t" NOPS(2) "global_store_dwordx4 %4, %0, off\n\t" LONG : "+v"(d) : "v"(a), "v"(b), "v"(c), "v"(po) : "memory");
        W(4) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %3\n\t" NOPS(3) "global_store_dwordx4 %4, %0, off\n\t" LONG : "+v"(d) : "v"(a), "v"(b), "v"(c), "v"(po) : "memory");
        W(5) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %3\n\t" NOPS(4) "global_store_dwordx4 %4, %0, off\n\t" LONG : "+v"(d) : "v"(a), "v"(b), "v"(c), "v"(po) : "memory");
        W(6) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %3\n\t" NOPS(5) "global_store_dwordx4 %4, %0, off\n\t" LONG : "+v"(d) : "v"(a), "v"(b), "v"(c), "v"(po) : "memory");
        W(7) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %3\n\t" NOPS(6) "global_store_dwordx4 %4, %0, off\n\t" LONG : "+v"(d) : "v"(a), "v"(b), "v"(c), "v"(po) : "memory");
        W(8) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %3\n\t" NOPS(7) "global_store_dwordx4 %4, %0, off\n\t" LONG : "+v"(d) : "v"(a), "v"(b), "v"(c), "v"(po) : "memory");
        W(10) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %3\n\t" NOPS(7) NOPS(1) "global_store_dwordx4 %4, %0, off\n\t" LONG : "+v"(d) : "v"(a), "v"(b), "v"(c), "v"(po) : "memory");
        W(12) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %3\n\t" NOPS(7) NOPS(3) "global_store_dwordx4 %4, %0, off\n\t" LONG : "+v"(d) : "v"(a), "v"(b), "v"(c), "v"(po) : "memory");
        W(16) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %3\n\t" NOPS(7) NOPS(7) "global_store_dwordx4 %4, %0, off\n\t" LONG : "+v"(d) : "v"(a), "v"(b), "v"(c), "v"(po) : "memory");
        W(20) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %3\n\t" NOPS(7) NOPS(7) NOPS(3) "global_store_dwordx4 %4, %0, off\n\t" LONG : "+v"(d) : "v"(a), "v"(b), "v"(c), "v"(po) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    } else if constexpr (TEST == 1) {  // war_b
        SEQ("v_mfma_f32_4x4x4_16b_f16 %0, %2, %3, %4\n\t", "v_pk_mov_b32 %3, %5, %5\n\t" LONG)
    } else if constexpr (TEST == 2) {  // war_a
        SEQ("v_mfma_f32_4x4x4_16b_f16 %0, %2, %3, %4\n\t", "v_pk_mov_b32 %2, %5, %5\n\t" LONG)
    } else if constexpr (TEST == 3 || TEST == 9) {  // raw_src_b / raw_src_a: a vector instruction writes the operand -> n -> mfma reads it
        const u2 src = TEST == 3 ? b : a, other = TEST == 3 ? a : b;
#define RS(n_, nops) W(n_) { if constexpr (TEST == 3) asm volatile("v_mov_b32 v20, %5\n\tv_mov_b32 v21, %5\n\t" NOPS(7) "v_mov_b32 v20, %3\n\tv_mov_b32 v21, %4\n\t" nops "v_mfma_f32_4x4x4_16b_f16 %0, %1, v[20:21], %2\n\t" LONG : "=&v"(d) : "v"(other), "v"(c), "v"(src.x), "v"(src.y), "v"(junk.x) : "v20", "v21"); \
                       else asm volatile("v_mov_b32 v20, %5\n\tv_mov_b32 v21, %5\n\t" NOPS(7) "v_mov_b32 v20, %3\n\tv_mov_b32 v21, %4\n\t" nops "v_mfma_f32_4x4x4_16b_f16 %0, v[20:21], %1, %2\n\t" LONG : "=&v"(d) : "v"(other), "v"(c), "v"(src.x), "v"(src.y), "v"(junk.x) : "v20", "v21"); }
        RS(0, "") RS(1, NOPS(0)) RS(2, NOPS(1)) RS(3, NOPS(2)) RS(4, NOPS(3)) RS(5, NOPS(4)) RS(6, NOPS(5)) RS(7, NOPS(6)) RS(8, NOPS(7)) RS(10, NOPS(7) NOPS(1)) RS(12, NOPS(7) NOPS(3)) RS(16, NOPS(7) NOPS(7)) RS(20, NOPS(7) NOPS(7) NOPS(3))
    } else if constexpr (TEST == 4) {  // chain: d = mfma(a,b,c); n; d = mfma(a,b,d)
        SEQ("v_mfma_f32_4x4x4_16b_f16 %0, %2, %3, %4\n\t", "v_mfma_f32_4x4x4_16b_f16 %0, %2, %3, %0\n\t" LONG)
    } else if constexpr (TEST == 5) {  // chain_x: d2 = mfma(a,b,c); n; d = mfma(a,b,d2)
        SEQ("v_mfma_f32_4x4x4_16b_f16 %1, %2, %3, %4\n\t", "v_mfma_f32_4x4x4_16b_f16 %0, %2, %3, %1\n\t" LONG)
    } else if constexpr (TEST == 6) {  // dst_ovl_b: destination = c's registers... use b inside d: emulate with d tied to (b, x, x, x)
        // d[0:1] hold b on entry; the instruction names d as destination AND its low half as B
        float o0, o1, o2, o3;
        asm volatile("v_mov_b32 v10, %6\n\tv_mov_b32 v11, %7\n\t" NOPS(7) "v_mfma_f32_4x4x4_16b_f16 v[10:13], %4, v[10:11], %5\n\t" LONG
                     "v_mov_b32 %0, v10\n\tv_mov_b32 %1, v11\n\tv_mov_b32 %2, v12\n\tv_mov_b32 %3, v13\n\t"
                     : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(a), "v"(c), "v"(b.x), "v"(b.y) : "v10", "v11", "v12", "v13");
        d = f4{o0, o1, o2, o3};
    } else if constexpr (TEST == 7) {  // dst_ovl_c: destination v[12:15], SrcC v[14:17] (overlap by half, what the allocator did with one live row)
        float o0, o1, o2, o3;
        asm volatile("v_mov_b32 v14, %6\n\tv_mov_b32 v15, %7\n\tv_mov_b32 v16, %8\n\tv_mov_b32 v17, %9\n\t" NOPS(7) "v_mfma_f32_4x4x4_16b_f16 v[12:15], %4, %5, v[14:17]\n\t" LONG
                     "v_mov_b32 %0, v12\n\tv_mov_b32 %1, v13\n\tv_mov_b32 %2, v14\n\tv_mov_b32 %3, v15\n\t"
                     : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(a), "v"(b), "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w) : "v12", "v13", "v14", "v15", "v16", "v17");
        d = f4{o0, o1, o2, o3};
    } else if constexpr (TEST == 8) {  // raw_valu proper: mfma -> n -> v_mov reads element 0 of the result
        float o0 = 0.f;
#define RV(n_, nops) W(n_) asm volatile("v_mov_b32 v14, %3\n\tv_mov_b32 v15, %4\n\tv_mov_b32 v16, %5\n\tv_mov_b32 v17, %6\n\t" NOPS(7) "v_mfma_f32_4x4x4_16b_f16 v[14:17], %1, %2, v[14:17]\n\t" nops "v_mov_b32 %0, v14\n\t" LONG : "=&v"(o0) : "v"(a), "v"(b), "v"(c.x), "v"(c.y), "v"(c.z), "v"(c.w) : "v14", "v15", "v16", "v17");
        RV(0, "") RV(1, NOPS(0)) RV(2, NOPS(1)) RV(3, NOPS(2)) RV(4, NOPS(3)) RV(5, NOPS(4)) RV(6, NOPS(5)) RV(7, NOPS(6)) RV(8, NOPS(7)) RV(10, NOPS(7) NOPS(1)) RV(12, NOPS(7) NOPS(3)) RV(16, NOPS(7) NOPS(7)) RV(20, NOPS(7) NOPS(7) NOPS(3))
        d = f4{o0, 0, 0, 0};
    }
    out[l] = d;
}

static float h2f(uint16_t h) {
    const int s = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    float v = e == 0 ? std::ldexp((float)m, -24) : std::ldexp((float)(m | 1024), e - 25);
    return s ? -v : v;
}

int main() {
    std::vector<u2> A(64), B(64);
    std::vector<f4> C(64), want(64), want2(64);
    auto h = [](int i) { return (uint16_t)(0x3c00 + 37 * (i % 23) + 256 * (i % 5)); };  // fp16 values 1 .. 4, all normal
    for (int l = 0; l < 64; l++) {
        A[l] = u2{(uint32_t)h(3 * l) | ((uint32_t)h(3 * l + 1) << 16), (uint32_t)h(5 * l + 2) | ((uint32_t)h(7 * l + 3) << 16)};
        B[l] = u2{(uint32_t)h(11 * l + 4) | ((uint32_t)h(13 * l + 5) << 16), (uint32_t)h(17 * l + 6) | ((uint32_t)h(19 * l + 7) << 16)};
        C[l] = f4{(float)l, (float)(2 * l), (float)(3 * l), 1.f};
    }
    auto quad = [&](const u2& v, int k) { return h2f((uint16_t)(k < 2 ? (v.x >> (16 * k)) : (v.y >> (16 * (k - 2))))); };
    for (int l = 0; l < 64; l++) {
        const int b = l / 4;
        for (int i = 0; i < 4; i++) {
            float s = 0;
            for (int k = 0; k < 4; k++) s += quad(A[4 * b + i], k) * quad(B[l], k);
            want[l][i] = C[l][i] + s;
            want2[l][i] = C[l][i] + 2 * s;
        }
    }
    u2 *dA, *dB; f4 *dC, *dO;
    CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dC, 1024)); CK(hipMalloc(&dO, (size_t)16 * 64 * 2048 * 8));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice));
    int BLOCKS = 1, THREADS = 64, REPS = 200;
    auto run = [&](auto kern, const std::vector<f4>& w, int elems) {
        int bad = 0;
        const int waves = BLOCKS * (THREADS / 64);
        std::vector<f4> o((size_t)64 * waves);
        for (int rep = 0; rep < REPS; rep++) {
            hipLaunchKernelGGL(kern, dim3(BLOCKS), dim3(THREADS), 0, 0, dA, dB, dC, dO);
            CK(hipMemcpy(o.data(), dO, o.size() * 16, hipMemcpyDeviceToHost));
            for (int wv = 0; wv < waves; wv++)
                for (int l = 0; l < 64; l++)
                    for (int i = 0; i < elems; i++)
                        if (!(std::fabs(o[(size_t)wv * 64 + l][i] - w[l][i]) <= 1e-3f * std::fabs(w[l][i]))) bad++;
        }
        return bad;
    };
#define ROW(name, T, w, el) printf("%-10s", name); \
    printf(" n=0:%d", run(probe<T, 0>, w, el)); printf(" 1:%d", run(probe<T, 1>, w, el)); printf(" 2:%d", run(probe<T, 2>, w, el)); printf(" 3:%d", run(probe<T, 3>, w, el)); \
    printf(" 4:%d", run(probe<T, 4>, w, el)); printf(" 5:%d", run(probe<T, 5>, w, el)); printf(" 6:%d", run(probe<T, 6>, w, el)); printf(" 7:%d", run(probe<T, 7>, w, el)); \
    printf(" 8:%d", run(probe<T, 8>, w, el)); printf(" 10:%d", run(probe<T, 10>, w, el)); printf(" 12:%d", run(probe<T, 12>, w, el)); printf(" 16:%d", run(probe<T, 16>, w, el)); printf(" 20:%d\n", run(probe<T, 20>, w, el));
    for (int pass = 0; pass < 2; pass++) {
    if (pass == 1) { BLOCKS = 2048; THREADS = 512; REPS = 3; }
    printf("%s: bad values of %d waves x 64 lanes x elems x %d runs, by number of wait states n between the two instructions\n", pass ? "WHOLE CHIP, 8 waves per SIMD, matrix pipe busy" : "ONE WAVE", BLOCKS * THREADS / 64, REPS);
    ROW("raw_valu", 0, want, 1)
    ROW("war_b", 1, want, 4)
    ROW("war_a", 2, want, 4)
    ROW("raw_src_b", 3, want, 4)
    ROW("raw_src_a", 9, want, 4)
    ROW("chain", 4, want2, 4)
    ROW("chain_x", 5, want2, 4)
    printf("%-10s %d\n", "dst_ovl_b", run(probe<6, 0>, want, 4));
    printf("%-10s %d\n", "dst_ovl_c", run(probe<7, 0>, want, 4));
    ROW("raw_vmov", 8, want, 1)
    }
    return 0;
}
