// Pieces of a hand-ordered MFMA pipeline on gfx950 (one in-order wave per SIMD): LDS fragment reads, counted waits tied to the registers
// they make valid, compile-time loops for "one item per MFMA shadow" schedules.  Used by binary_fp4.hip (FP4 GEMM) and mpq_dense.hip
// (bf16 / fp16 GEMM on dequantised weight fragments).
#pragma once
#include "bie_common.h"

namespace bie {

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef int v4i_t __attribute__((ext_vector_type(4)));

// ---- LDS fragment reads (hand-issued: the compiler must not order them against the LDS-DMA by its own alias rules) ----
template <int OFF>
__device__ __forceinline__ v4i_t lds_read16(uint32_t addr) {
    v4i_t r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}
template <int N, int BASE>
__device__ __forceinline__ void read_frags(v4i_t (&f)[N], uint32_t addr) {
    static_assert(N == 1 || N == 2 || N == 4, "1, 2 or 4 fragments");
    f[0] = lds_read16<BASE>(addr);
    if constexpr (N >= 2) f[1] = lds_read16<BASE + 2048>(addr);
    if constexpr (N >= 4) {
        f[2] = lds_read16<BASE + 4096>(addr);
        f[3] = lds_read16<BASE + 6144>(addr);
    }
}
// s_waitcnt lgkmcnt(CNT) tied to the fragment registers it makes valid
template <int CNT, int NA, int NB>
__device__ __forceinline__ void wait_frags(v4i_t (&a)[NA], v4i_t (&b)[NB]) {
    if constexpr (NA == 4 && NB == 4)
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(CNT) : "memory");
    else if constexpr (NA == 2 && NB == 2)
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : "n"(CNT) : "memory");
    else if constexpr (NA == 2 && NB == 1)
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]) : "n"(CNT) : "memory");
    else {
        static_assert(NA == 1 && NB == 1, "wait_frags: add the register list of this tile");
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(b[0]) : "n"(CNT) : "memory");
    }
}

__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }  // >= 18 wait states: XDL write -> VALU read

constexpr int imin(int a, int b) { return a < b ? a : b; }
// Workgroup -> output tile.  Workgroup b runs on XCD b & 7: every XCD gets one contiguous run of the tile sequence (runs differ by one tile
// when the grid is not a multiple of 8), and the sequence goes down gm tile rows before moving one tile column on, so the workgroups an XCD
// runs at once share gm row panels and (its CUs / gm) column panels in that XCD's L2 instead of one row panel and a column panel each
// (dense W4A16 GEMM, M = 4096, 4096 -> 11008: gm 1 -> 4 is 2-10 % of the launch, profiles/r03_dense_gm_ab.txt).
#ifndef BIE_PIPE_GM
#define BIE_PIPE_GM 4
#endif
__device__ __forceinline__ void pipe_tile(int bid, int nblk, int tiles_n, int gm, int& tile_m, int& tile_n) {
    const int xcd = bid & 7, per = nblk >> 3, rem = nblk & 7;
    bid = xcd * per + (xcd < rem ? xcd : rem) + (bid >> 3);
    const int grp = bid / (gm * tiles_n), first_m = grp * gm, left = nblk / tiles_n - first_m, rows = left < gm ? left : gm, r = bid - grp * gm * tiles_n;
    tile_n = r / rows;
    tile_m = first_m + (r - tile_n * rows);
}

template <int I> struct ic_t { static constexpr int value = I; };
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(ic_t<I>{});
        static_for<I + 1, N>(f);
    }
}

}  // namespace bie
