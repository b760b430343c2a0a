// Probe for the table-lookup dequant redesign (gfx950):
//  (1) semantics: ds_read_u16_d16_hi into a register that held something (low half preserved or zeroed?), ds_bpermute_b32 with garbage
//      above address bit 7, v_perm_b32 sign-replicate selectors;
//  (2) VALU issue rates of the candidate address-formation / pairing instructions (ns per wave-instruction per SIMD, 8 waves/SIMD);
//  (3) LDS rates per CU: ds_read_b32 / ds_read_u16_d16_hi / ds_read_u16 / ds_read_b64 / ds_bpermute_b32, conflict-free table layout.
// hipcc --offload-arch=gfx950 -O3 -o bin/probe_lut probe_lut.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

__global__ void semantics(uint32_t* out) {
    __shared__ uint32_t tab[512];
    const int lane = threadIdx.x;
    tab[lane] = 0x11110000u * (lane & 15) + 0xbe00u + lane;  // low half = 0xbe00 + lane, high half = 0x1111 * (lane % 16)
    tab[lane + 64] = 0xdead0000u + lane;
    __syncthreads();
    uint32_t r = 0xaaaa5555u, addr = lane * 4;
    asm volatile("ds_read_u16_d16_hi %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(r) : "v"(addr) : "memory");
    out[lane] = r;  // expect hi = 0xbe00 + lane; low = 0x5555 (preserved) or 0 (zeroed)
    uint32_t r2 = 0xaaaa5555u;
    asm volatile("ds_read_u16_d16 %0, %1 offset:2\n\ts_waitcnt lgkmcnt(0)" : "+v"(r2) : "v"(addr) : "memory");
    // ds_write_b16_d16_hi: stores the HIGH half of the data register
    uint32_t wv = 0xc0de0000u + lane, waddr = 512 * 4 / 2 + lane * 4;
    asm volatile("ds_write_b16_d16_hi %0, %1 offset:2\n\tds_write_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(waddr), "v"(wv) : "memory");
    __syncthreads();
    out[200 + lane] = tab[256 + lane];  // expect (0xc0de << 16) | lane
    // SDWA byte insert: addr.byte1 = src.byte2, other bytes preserved
    uint32_t sd = 0x000000a4u + 0 * lane, src = 0x11223344u + (lane << 16);
    asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(sd) : "v"(src));
    out[264 + lane] = sd;  // expect 0x0000(22+lane)a4
    out[64 + lane] = r2;  // low = 0x1111 * (lane % 16); high = 0xaaaa or 0
    // bpermute: lane reads the value of lane ((addr / 4) % 64)?  address with garbage above bit 7
    uint32_t val = 0x1000 + lane, baddr = ((uint32_t)(63 - lane) << 2) | 0xabcd00u, got;
    asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(got) : "v"(baddr), "v"(val));
    out[128 + lane] = got;  // expect 0x1000 + 63 - lane if the upper address bits are ignored
    // v_perm_b32 selectors 8..11: sign of byte 1/3/5/7? (documented: 8 = sign of byte 1 of S1 ... ) -- record what each gives
    uint32_t s0 = 0x80017f02u, s1 = 0x03ff8004u, pr;
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pr) : "v"(s0), "v"(s1), "v"(0x0b0a0908u));
    out[192] = pr;
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pr) : "v"(s0), "v"(s1), "v"(0x0f0e0d0cu));
    out[193] = pr;
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pr) : "v"(s0), "v"(s1), "v"(0x07060504u));
    out[194] = pr;
    asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(pr) : "v"(s0), "v"(s1), "v"(0x03020100u));
    out[195] = pr;
}

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t r[16];
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1;
    float fa = 1.0001f, fb = 0.5f;
    uint64_t rr[16];
    const uint64_t sp = __builtin_amdgcn_readfirstlane(seed) * 0x100000001ull;
#pragma unroll
    for (int i = 0; i < 16; i++) { r[i] = a + i; rr[i] = a * 0x100000003ull + i; }
    for (int it = 0; it < ITER; it++) {
#define X(i)                                                                                                   \
    if constexpr (OP == 0) asm volatile("v_add_u32 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(a));             \
    else if constexpr (OP == 1) asm volatile("v_or_b32 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(a));          \
    else if constexpr (OP == 2) asm volatile("v_and_b32 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(a));         \
    else if constexpr (OP == 3) asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(fa));        \
    else if constexpr (OP == 4) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(fa));        \
    else if constexpr (OP == 5) asm volatile("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(r[i]) : "v"(r[i]), "v"(a)); \
    else if constexpr (OP == 6) asm volatile("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(r[i]) : "v"(r[i]), "v"(a)); \
    else if constexpr (OP == 7) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[i]), "v"(a), "v"(b)); \
    else if constexpr (OP == 8) asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[i]), "v"(a), "v"(b)); \
    else if constexpr (OP == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(r[(i + 1) & 15]));          \
    else if constexpr (OP == 10) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(fa), "v"(fb));        \
    else if constexpr (OP == 11) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(fa), "s"(seed));   \
    else if constexpr (OP == 12) asm volatile("v_lshrrev_b32 %0, 8, %1" : "=v"(r[i]) : "v"(r[i]));             \
    else if constexpr (OP == 13) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));    \
    else if constexpr (OP == 14) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(r[i]), "v"(a), "s"(seed)); \
    else if constexpr (OP == 15) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2" : "+v"(r[i]) : "v"(a)); \
    else if constexpr (OP == 16) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(r[i]) : "v"(a)); \
    else if constexpr (OP == 17) asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r[i]) : "v"(r[i]), "v"(a)); \
    else if constexpr (OP == 18) asm volatile("v_lshlrev_b32 %0, 8, %1" : "=v"(r[i]) : "v"(r[i])); \
    else if constexpr (OP == 19) asm volatile("v_and_b32 %0, 0xff00, %1" : "=v"(r[i]) : "v"(r[i])); \
    else if constexpr (OP == 20) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r[i]) : "v"(r[i]), "v"(a)); \
    else if constexpr (OP == 21) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(rr[i]) : "v"(rr[i]), "v"(rr[(i + 1) & 15])); \
    else if constexpr (OP == 22) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(rr[i]) : "v"(rr[(i + 1) & 15]), "s"(sp)); \
    else if constexpr (OP == 23) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r[i]) : "v"(a), "v"(fa));
        REP16(X)
#undef X
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc += r[i] + (uint32_t)rr[i] + (uint32_t)(rr[i] >> 32);
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

// LDS forms: every wave issues ITER x 16 reads of the conflict-free table layout tab[q][lane] (q varies per lane)
template <int OP>
__global__ __launch_bounds__(256) void lds(uint32_t* out, uint32_t seed) {
    __shared__ uint32_t tab[4 * 16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 16 * 64; i += 256) tab[i] = i * seed;
    __syncthreads();
    uint32_t addr[16], r[16];
    uint64_t r2[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t q = (lane * 7 + i * 5 + seed) & 15;
        addr[i] = ((wave * 16 + q) << 8) | (lane << 2);
        if (OP == 3) addr[i] &= ~7u;
        r[i] = 0;
        r2[i] = 0;
    }
    uint32_t val = lane;
    for (int it = 0; it < ITER; it++) {
#define X(i)                                                                                                   \
    if constexpr (OP == 0) asm volatile("ds_read_b32 %0, %1" : "=v"(r[i]) : "v"(addr[i]));                     \
    else if constexpr (OP == 1) asm volatile("ds_read_u16_d16_hi %0, %1" : "+v"(r[i]) : "v"(addr[i]));         \
    else if constexpr (OP == 2) asm volatile("ds_read_u16 %0, %1" : "=v"(r[i]) : "v"(addr[i]));                \
    else if constexpr (OP == 3) asm volatile("ds_read_b64 %0, %1" : "=v"(r2[i]) : "v"(addr[i]));               \
    else if constexpr (OP == 4) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(r[i]) : "v"(addr[i]), "v"(val)); \
    else if constexpr (OP == 5) asm volatile("ds_read_u16_d16_hi %0, %1 offset:2" : "+v"(r[i]) : "v"(addr[i]));
        REP16(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) acc += r[i] + (uint32_t)r2[i];
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <typename F>
static void timeit(const char* name, F launch, double wave_instr_per_unit, const char* unit) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(1);
    (void)hipEventRecord(e0);
    launch(2);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s %8.3f ms  -> %6.2f ns per wave-instr per %s (= %5.2f cycles @2.4GHz)\n", name, ms, ms * 1e6 / wave_instr_per_unit, unit,
           ms * 1e6 / wave_instr_per_unit * 2.4);
}

int main() {
    uint32_t* d; (void)hipMalloc(&d, 4096);
    semantics<<<1, 64>>>(d);
    uint32_t h[512];
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("ds_write_b16_d16_hi (offset 2) + ds_write_b16: lane0 %08x lane3 %08x (want c0de0000 c0de0003)\n", h[200], h[203]);
    printf("v_mov_b32_sdwa BYTE_1 <- BYTE_2, preserve: lane0 %08x lane3 %08x (want 000022a4 000025a4)\n", h[264], h[267]);
    printf("ds_read_u16_d16_hi into 0xaaaa5555: lane0 %08x lane1 %08x lane5 %08x (hi = 0xbe00+lane; low 5555 = preserved, 0000 = zeroed)\n", h[0], h[1], h[5]);
    printf("ds_read_u16_d16 (offset 2) into 0xaaaa5555: lane0 %08x lane1 %08x lane5 %08x (low = 0x1111*lane; high aaaa = preserved)\n", h[64], h[65], h[69]);
    printf("ds_bpermute with garbage above bit 7: lane0 %08x lane1 %08x lane62 %08x (want 103f 103e 1001)\n", h[128], h[129], h[190]);
    printf("v_perm selectors 0b0a0908 -> %08x, 0f0e0d0c -> %08x, 07060504 -> %08x, 03020100 -> %08x  (S0 = 80017f02, S1 = 03ff8004)\n", h[192], h[193], h[194], h[195]);
    const int blocks = 2048;
    const double per_simd = (double)blocks * 4 / (256.0 * 4) * ITER * 16;
#define RUN(OP, NAME) timeit(NAME, [&](uint32_t s) { k<OP><<<blocks, 256>>>(d, s); }, per_simd, "SIMD")
    RUN(0, "v_add_u32"); RUN(1, "v_or_b32"); RUN(2, "v_and_b32"); RUN(3, "v_add_f32"); RUN(4, "v_mul_f32"); RUN(5, "v_lshl_or_b32");
    RUN(6, "v_lshl_add_u32"); RUN(7, "v_mad_u32_u24"); RUN(8, "v_bfi_b32"); RUN(9, "v_mov_b32"); RUN(10, "v_fmac_f32");
    RUN(11, "v_fma_f32 (sgpr operand)"); RUN(12, "v_lshrrev_b32"); RUN(13, "v_dot2c_f32_bf16"); RUN(14, "v_perm_b32 (sgpr selector)");
    RUN(15, "v_mov_b32_sdwa byte insert, PRESERVE"); RUN(16, "v_mov_b32_sdwa byte insert, PAD"); RUN(17, "v_or_b32_sdwa src0 BYTE_2"); RUN(18, "v_lshlrev_b32");
    RUN(19, "v_and_b32 literal"); RUN(20, "v_cvt_pk_bf16_f32"); RUN(21, "v_pk_mul_f32"); RUN(22, "v_pk_fma_f32 (sgpr pair)"); RUN(23, "v_fma_mix_f32 hi-half f16 weight");
    const double per_cu = (double)blocks * 4 / 256.0 * ITER * 16;
#define RUNL(OP, NAME) timeit(NAME, [&](uint32_t s) { lds<OP><<<blocks, 256>>>(d, s); }, per_cu, "CU")
    RUNL(0, "ds_read_b32 tab[q][lane]"); RUNL(1, "ds_read_u16_d16_hi"); RUNL(5, "ds_read_u16_d16_hi offset:2"); RUNL(2, "ds_read_u16");
    RUNL(3, "ds_read_b64 (8-byte aligned)"); RUNL(4, "ds_bpermute_b32");
    return 0;
}
