// W4A16 decode GEMV (M <= 2) by TABLE LOOKUP -- the bandwidth-bound half of bie_mpq_forward for bf16 layers.
// Replaces quant_mm_kernel[_asym] (reference layers/qlinear/nbit/cuda/mpq_linear_cuda_kernel.cu:67-451).
//
// Why a table.  The reference's CPU path rounds every dequantised weight twice to the layer dtype
// (fl(fl(q*s) - z), layers/qlinear/nbit/cuda/utils.py:36-51); gfx950 has no packed bf16 ALU, so reproducing the two
// roundings per weight costs ~5 VALU instructions per weight and made the dot2 kernel (mpq_gemv.hip) VALU-issue bound at
// 0.24 of the HBM roofline.  But inside one (group, column) a 4-bit weight takes only 16 values: a lane (= one output
// column) computes the 16 doubly-rounded values ONCE per group (128 k), parks them in LDS as fp32 (bf16 << 16), and every
// weight then costs
//     1 VALU  v_perm_b32       {lane*4, byte_k(w'), 0, 0} = the LDS address (wave*16 + q)*256 + lane*4: no shift, no add.
//                              w' = the word with the other nibble of every byte replaced by the wave index: 3 VALU per
//                              word, i.e. per 8 weights
//     1 LDS   ds_read_b32      tab[wave][q][lane]: bank = lane mod 32 whatever q is -> conflict-free by construction
//     1 VALU  v_fma_f32        acc += x_k * T[q], x_k wave-uniform (SGPR operand), full-rate fp32 FMA
// The table holds the exact reference values, so the result is the reference's up to fp32 summation order.
//
// Shape of the launch.  A workgroup = NW waves on one 64-column tile; each wave owns a run of whole groups (normally ONE:
// 16 packed rows, all loaded up front: dword non-temporal loads, a wave-row = 256 contiguous bytes), builds its own 4 KiB
// table, and the waves are summed through LDS in wave order.  K is additionally split over the grid (slice-major block
// ids).  Cross-workgroup reduction WITHOUT atomics or drains: slices 0..S-2 publish their column sums as 8-byte
// {value, tag} granules with one write-through (sc1) store and retire; the workgroup of the LAST slice (highest block
// ids: dispatched after every publisher, so the wait cannot starve them) polls the granules with bypassing loads until
// every tag equals this launch's tag, adds them in slice order (deterministic) and writes y.  A tag = {24 bits: number of
// the launch CALL (a host counter baked into the arguments: granule addresses of differently shaped launches alias, their
// tags cannot), 8 bits: the tile's generation word in the workspace head + 1 (advanced by the reducer when it is done: a
// REPLAY of a captured launch carries the same call number but the next generation)}; nothing ever has to be reset.  `bie_mpq_forward_grouped` passes several weight sets that share x
// (q/k/v, gate/up): their column tiles are concatenated into one grid.
#include "mpq_dequant.cuh"
#include "mpq_list.h"
#include <stdlib.h>
#include <atomic>

#pragma clang fp contract(off)

namespace bie {

unsigned* device_status_word();                            // splitk.hip
void test_forge_get(unsigned* tag_skew, int* spin_limit);  // splitk.hip
// mpq_list.hip: the layer-list kernel with its entries in the kernel arguments (W4, M = 1, bf16: what bie_mpq_forward[_grouped] run on)
bool mpq_list_inline_ok(int M, int K, long n_total, int w_bit, int group_size, int zm, int dtype);
size_t mpq_list_inline_part_floats(int M, int K, int group_size, int tiles_total, int w_bit);
int mpq_list_inline_launch(int nsets, const int32_t* const* qw, const void* const* scales, const void* const* zeros, const void* const* bias,
                           void* const* y, const int* N, const void* x, unsigned* gen, float* gran, int K, int group_size, int zm, hipStream_t st);

constexpr int LUT_MAX_SETS = 8;

struct LutSet {
    const uint32_t* qw;
    const uint16_t* scales;
    const void* zeros;
    const uint16_t* bias;
    uint16_t* y;
    int N;
    int tile_begin;  // first column tile of this set in the concatenated grid
};

struct LutArgs {
    LutSet set[LUT_MAX_SETS];
    const uint16_t* x;
    unsigned long long* gran;  // [S-1][M][tiles_total * 64] {fp32 partial, tag}
    unsigned* gen;             // generation word per column tile (workspace head)
    int nsets, M, K, G, tiles_total, S, groups_per_wave;  // G / groups_per_wave count UNITS of RPG rows: H units per group
    int hshift;  // log2(H)
    unsigned epoch;  // (launch-call number mod 2^24) << 8: the upper 24 bits of every granule tag of this launch
    unsigned* status;   // host-mapped status word (NULL before bie_status_init): bit 0 = a reducer gave up
    unsigned tag_skew;  // testing aid (bie_test_forge_reducer): the reducer expects tag ^ tag_skew
    int spin_limit;
};

// tuning aid (BIE_GEMV_LAB=5, LAB builds only: make lab): per-wave timestamps {start, weights landed, compute done, end, xcc/cu id}
#ifdef BIE_LAB_BUILD
__device__ unsigned long long g_lut_stamps[65536 * 5];
#define BIE_LAB5(LABV) ((LABV) == 5)
#else
__device__ unsigned long long g_lut_stamps[1];  // never touched: LAB == 5 is not instantiated in the product build
#define BIE_LAB5(LABV) false
#endif

typedef __attribute__((address_space(3))) uint32_t lds_u32;
// constant address space: x is never written by this kernel, and uniform loads from it always take the scalar path
typedef const __attribute__((address_space(4))) uint32_t const_u32;

// LDS byte address {byte 0 = lane * 4, byte 1 = byte BYTE of w (= wave * 16 + q), bytes 2-3 = 0} in ONE v_perm_b32
// (selector 0x0c = the constant 0x00; an SDWA move with dst_unused:UNUSED_PRESERVE does the same and measured no faster)
template <int BYTE>
__device__ __forceinline__ uint32_t lut_addr(uint32_t lane_addr, uint32_t w) {
    return __builtin_amdgcn_perm(w, lane_addr, 0x0c0c0400u + ((uint32_t)BYTE << 8));
}

__device__ __forceinline__ float lds_f32(uint32_t byte_addr) {
    return __uint_as_float(*reinterpret_cast<const lds_u32*>(byte_addr));
}

// the 16 values a 4-bit weight of this (group, column) can take, rounded exactly like the reference
template <int DT, int ZM>
__device__ __forceinline__ float lut_entry(uint32_t q, float s, float z, int zq1) {
    if constexpr (ZM == ZM_ASYM) return dequant_scalar_asym<DT>(q, s, zq1);
    else if constexpr (ZM == ZM_FUSED) return dt_traits<DT>::round(__builtin_fmaf((float)q, s, -z));
    else return dequant_scalar_sym<DT>(q, s, z);
}

// Cross-workgroup reduction of a column tile (called by wave 0 of every workgroup) + the store of y; see the file header.
template <int DT, int MT, bool REDUCE>
__device__ __forceinline__ void lut_cross_wg_reduce(const LutArgs& a, const LutSet& ls, float (&tot)[MT], int tile, int slice, int lane,
                                                    int n, int N, unsigned tag, unsigned gen_next) {
    // ---- cross-workgroup reduction (wave 0 only) -----------------------------------------------------------------
    const bool owner = n < N;
    const long ncat = (long)a.tiles_total * 64;
    const long col = (long)tile * 64 + lane;
    if (a.S > 1 && REDUCE) {
        if (slice != a.S - 1) {  // publisher: one 8-byte write-through store per column, no drain, no atomic
#pragma unroll
            for (int m = 0; m < MT; m++) {
                const unsigned long long gval = ((unsigned long long)tag << 32) | __float_as_uint(tot[m]);
                __hip_atomic_store(a.gran + ((long)slice * MT + m) * ncat + col, gval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        // reducer: poll until every granule of this column carries this launch's tag, then add in slice order
        const unsigned want = tag ^ a.tag_skew;
        bool poisoned = false;
#pragma unroll
        for (int m = 0; m < MT; m++) {
            float v = 0.0f;
            for (int s0 = 0; s0 < a.S - 1; s0 += 8) {
                unsigned long long gv[8];
                bool ready;
                int spins = 0;
                do {
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) {
                        const int sidx = (s0 + jj < a.S - 1) ? s0 + jj : a.S - 2;
                        gv[jj] = __hip_atomic_load(a.gran + ((long)sidx * MT + m) * ncat + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    ready = true;
#pragma unroll
                    for (int jj = 0; jj < 8; jj++) ready = ready && ((unsigned)(gv[jj] >> 32) == want);
                    ready = __builtin_amdgcn_ballot_w64(!ready) == 0;  // wave-uniform: every lane's granules are in
                    if (!ready) __builtin_amdgcn_s_sleep(2);
                } while (!ready && ++spins < a.spin_limit);  // bounded: publishers never wait, but HIP promises no dispatch order
                if (!ready) poisoned = true;  // wave-uniform
#pragma unroll
                for (int jj = 0; jj < 8; jj++)
                    if (s0 + jj < a.S - 1) v += __uint_as_float((unsigned)gv[jj]);
            }
            tot[m] = v + tot[m];
        }
        if (lane == 0) a.gen[tile] = gen_next;  // a replay of this launch gets a different tag; visible at the kernel boundary
        if (poisoned) {  // never a silent number: NaN in y and a bit in the status page the next C-ABI call reports
#pragma unroll
            for (int m = 0; m < MT; m++) tot[m] = __uint_as_float(0x7fc00000u);
            if (lane == 0 && a.status) __hip_atomic_fetch_or(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (owner) {
#pragma unroll
        for (int m = 0; m < MT; m++) {
            float o = dt_traits<DT>::round(tot[m]);
            if (ls.bias) o = o + dt_traits<DT>::load(ls.bias, n);
            dt_traits<DT>::store(ls.y, (long)m * N + n, o);
        }
    }
}

// MT == M (1 or 2); RPG = packed rows per quantisation group (group_size / 8); NW = waves per workgroup;
// LAB (tuning aids): 2 = stream only, 3 = no cross-workgroup reduction, 4 = stream + reduction (no tables / lookups)
// RD = rows of every group dequantised directly on the VALU (mpq_dequant.cuh) instead of through the table: the table path is
// LDS-throughput bound (one ds_read_b32 per weight, 32 lookups per clock per CU), the direct path VALU bound; splitting the
// rows between them balances the two pipes
// WB = 2 (W2A16): a packed word holds 16 two-bit weights = 8 nibbles, and a nibble is a PAIR index (q[2i] | q[2i+1] << 2): the
// table holds the 16 possible pairs of dequantised weights as packed 16-bit halves and one v_dot2(c)_f32_{bf16,f16} against the
// packed x pair -- which is how x arrives from the scalar loads anyway -- consumes two weights: per word the same 2 + 8 + 8
// instructions as W4 for twice the weights, fp16 and bf16 alike (RPG = group_size / 16).
template <int DT, int ZM, int MT, int RPG, int NW, int LAB, int RD = 0, int WB = 4>
__global__ __launch_bounds__(NW * 64) void mpq_gemv_lut_kernel(const LutArgs a) {
    constexpr int NB = 32 / WB;      // weights per packed word
    constexpr int XD = NB / 2;       // x dwords (16-bit pairs) per packed word
    static_assert(WB == 4 || (WB == 2 && LAB == 0 && RD == 0), "the tuning variants exist for W4 only");
    __shared__ __attribute__((aligned(4096))) uint32_t tab[NW * 16 * 64];  // the only LDS object: starts at LDS address 0

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x % a.tiles_total;
    const int slice = blockIdx.x / a.tiles_total;
    int si = 0;
#pragma unroll
    for (int i = 1; i < LUT_MAX_SETS; i++)
        if (i < a.nsets && tile >= a.set[i].tile_begin) si = i;
    const LutSet& ls = a.set[si];
    const int N = ls.N;
    const int n = (tile - ls.tile_begin) * 64 + lane;
    const int nl = n < N ? n : N - 1;  // clamp: out-of-range lanes load valid memory and are never stored
    const int g0 = (slice * NW + wave) * a.groups_per_wave;
    int g1 = g0 + a.groups_per_wave;
    if (g1 > a.G) g1 = a.G;
    unsigned tag = 0;
    unsigned gen_next = 0;
    if (a.S > 1) {  // uniform; the generation only changes when this launch's reducer is done
        gen_next = a.gen[tile] + 1u;
        tag = a.epoch | (gen_next & 0xffu);
    }

    const uint32_t* wcol = ls.qw + nl;
    auto load_group = [&](uint32_t (&dst)[RPG], int g) {
#pragma unroll
        for (int u = 0; u < RPG; u++) dst[u] = __builtin_nontemporal_load(wcol + (long)(g * RPG + u) * N);
    };
    const int zero_width = N / NB;
    auto load_params = [&](int unit, uint32_t& sb, uint32_t& zb) {
        const int g = unit >> a.hshift;  // a group may be split into H units of RPG rows, each with its own wave (and table)
        sb = ls.scales[(long)g * N + nl];
        if constexpr (ZM == ZM_ASYM) {
            const uint32_t zw = reinterpret_cast<const uint32_t*>(ls.zeros)[(long)g * zero_width + nl / NB];
            zb = ((zw >> ((nl % NB) * WB)) & ((1u << WB) - 1u)) + 1u;
        } else {
            zb = reinterpret_cast<const uint16_t*>(ls.zeros)[(long)g * N + nl];
        }
    };

    float acc[MT][2];  // even / odd nibbles: two independent FMA chains
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m][0] = acc[m][1] = 0.0f;

    // LDS byte address of tab[wave][q][lane] = ((wave * 16 + q) << 8) | (lane << 2): byte 1 carries (wave, q)
    const uint32_t lane_addr = lane * 4;
    const uint32_t wavepat = (uint32_t)wave * 0x10101010u;
    uint32_t* mytab = tab + wave * (16 * 64) + lane;
    uint32_t m0f;  // VOP3 takes no 32-bit literal: the nibble mask lives in a register
    asm("v_mov_b32 %0, 0x0f0f0f0f" : "=v"(m0f));

    auto process_group = [&](const uint32_t (&w)[RPG], int g, uint32_t sb, uint32_t zb) {
        // the activations of the group are wave-uniform: scalar loads, issued before the table is built so that nothing
        // but LDS traffic is pending in the lookup phase
        uint32_t xs[MT][RPG * XD];
#pragma unroll
        for (int m = 0; m < MT; m++) {
            const_u32* xd = (const_u32*)(uintptr_t)(a.x + (long)m * a.K + (long)g * (RPG * NB));
#pragma unroll
            for (int i = 0; i < RPG * XD; i++) xs[m][i] = xd[i];
        }
        // ---- the 16-entry table of this (group, column)
        if constexpr (WB == 2) {
            float s, z = 0.0f;
            int zq1 = 0;
            if constexpr (DT == BIE_BF16) s = bf16_bits_to_f32(sb); else s = f16_bits_to_f32(sb);
            if constexpr (ZM == ZM_ASYM) zq1 = (int)zb;
            else if constexpr (DT == BIE_BF16) z = bf16_bits_to_f32(zb); else z = f16_bits_to_f32(zb);
            uint32_t v[4];  // the four dequantised values as 16-bit patterns
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float t = lut_entry<DT, ZM>((uint32_t)q, s, z, zq1);
                if constexpr (DT == BIE_BF16) v[q] = __float_as_uint(t) >> 16; else v[q] = f32_to_f16_bits(t);
            }
#pragma unroll
            for (int p2 = 0; p2 < 16; p2++) mytab[p2 * 64] = v[p2 & 3] | (v[p2 >> 2] << 16);
        } else if constexpr (LAB == 0 || LAB == 3 || LAB == 5) {
            if constexpr (DT == BIE_BF16 && ZM == ZM_SYM) {
                // a_q = fl(q*s): v_mul_f32 (exact) + v_cvt_pk_bf16_f32; T_q = fl(a_q - z): unpack-and-subtract on the dot unit
                // (bf16_pairs_sub) + v_cvt_pk_bf16_f32; the entry is the fp32 value of the bf16 weight (bf16 << 16)
                const float s = bf16_bits_to_f32(sb), nz = -bf16_bits_to_f32(zb);
                const uint32_t sel0 = sel_lo_hi<0>(), sel1 = sel_lo_hi<1>();
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t A0 = pack_bf16x2((float)(4 * j) * s, (float)(4 * j + 1) * s);
                    const uint32_t A1 = pack_bf16x2((float)(4 * j + 2) * s, (float)(4 * j + 3) * s);
                    float d[4];
                    bf16_pairs_sub(A0, A1, sel0, sel1, nz, d);
                    const uint32_t T0 = pack_bf16x2(d[0], d[1]), T1 = pack_bf16x2(d[2], d[3]);
                    // two entries per LDS instruction (ds_write2st64_b32: the rows of the table are 64 dwords apart)
                    mytab[(4 * j + 0) * 64] = T0 << 16;
                    mytab[(4 * j + 1) * 64] = T0 & 0xffff0000u;
                    mytab[(4 * j + 2) * 64] = T1 << 16;
                    mytab[(4 * j + 3) * 64] = T1 & 0xffff0000u;
                }
            } else {
                float s, z = 0.0f;
                int zq1 = 0;
                if constexpr (DT == BIE_BF16) s = bf16_bits_to_f32(sb); else s = f16_bits_to_f32(sb);
                if constexpr (ZM == ZM_ASYM) zq1 = (int)zb;
                else if constexpr (DT == BIE_BF16) z = bf16_bits_to_f32(zb); else z = f16_bits_to_f32(zb);
#pragma unroll
                for (int q = 0; q < 16; q++) mytab[q * 64] = __float_as_uint(lut_entry<DT, ZM>((uint32_t)q, s, z, zq1));
            }
        }
        // the activations have landed (in SGPRs) before the first lookup is issued: no scalar load is pending in the lookup
        // phase, so the LDS reads can be waited for with counted lgkmcnt
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int i = 0; i < RPG * XD; i += 8)
                asm volatile("" ::"s"(xs[m][i]), "s"(xs[m][i + 1]), "s"(xs[m][i + 2]), "s"(xs[m][i + 3]), "s"(xs[m][i + 4]), "s"(xs[m][i + 5]),
                             "s"(xs[m][i + 6]), "s"(xs[m][i + 7]));
        if constexpr (LAB == 2 || LAB == 4) {  // tuning aid: stream only
#pragma unroll
            for (int u = 0; u < RPG; u++) acc[0][0] += __uint_as_float(w[u] & 0x3f7fffffu) + __uint_as_float(sb << 16) + __uint_as_float(xs[0][u]);
            return;
        }
        // ---- 8 lookups + FMAs per packed word, one row ahead (lgkmcnt is a 4-bit counter: at most 15 LDS reads can be
        // waited for individually)
        auto lookup = [&](float (&t)[8], int u) {
            uint32_t we, wo;  // bytes (wave, q) of the even / odd nibbles: (w & 0x0f0f0f0f) | wavepat in ONE v_and_or_b32
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(we) : "v"(w[u]), "v"(m0f), "s"(wavepat));
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(wo) : "v"(w[u] >> 4), "v"(m0f), "s"(wavepat));
            t[0] = lds_f32(lut_addr<0>(lane_addr, we));
            t[1] = lds_f32(lut_addr<0>(lane_addr, wo));
            t[2] = lds_f32(lut_addr<1>(lane_addr, we));
            t[3] = lds_f32(lut_addr<1>(lane_addr, wo));
            t[4] = lds_f32(lut_addr<2>(lane_addr, we));
            t[5] = lds_f32(lut_addr<2>(lane_addr, wo));
            t[6] = lds_f32(lut_addr<3>(lane_addr, we));
            t[7] = lds_f32(lut_addr<3>(lane_addr, wo));
        };
        auto fmas = [&](const float (&t)[8], int u) {
            if constexpr (WB == 2) {  // t[i] = the packed pair (w[2i], w[2i+1]) of nibble i; x dword i of the word = (x[2i], x[2i+1])
#pragma unroll
                for (int m = 0; m < MT; m++)
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        acc[m][i & 1] = dot2_acc<DT>(__float_as_uint(t[i]), xs[m][u * 8 + i], acc[m][i & 1]);
                return;
            }
            if constexpr (DT == BIE_F16) {  // fp16 x straight from the SGPR pair: v_fma_mix_f32 converts the selected half on the fly
#pragma unroll
                for (int m = 0; m < MT; m++)
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        const uint32_t p = xs[m][u * 4 + i];
                        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(acc[m][0]) : "v"(t[2 * i]), "s"(p));
                        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(acc[m][1]) : "v"(t[2 * i + 1]), "s"(p));
                    }
                return;
            }
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t p = xs[m][u * 4 + i];
                    float xlo, xhi;
                    if constexpr (DT == BIE_BF16) { xlo = __uint_as_float(p << 16); xhi = __uint_as_float(p & 0xffff0000u); }
                    else { xlo = f16_bits_to_f32(p & 0xffffu); xhi = f16_bits_to_f32(p >> 16); }
                    acc[m][0] = __builtin_fmaf(xlo, t[2 * i], acc[m][0]);
                    acc[m][1] = __builtin_fmaf(xhi, t[2 * i + 1], acc[m][1]);
                }
        };
        // pin(): the DAG linearisation is free to sink the (unchained) FMAs below every later LDS read; a volatile asm that
        // consumes the accumulators keeps row u's FMAs between the reads of row u+1 and those of row u+2
        auto pin = [&]() {
#pragma unroll
            for (int m = 0; m < MT; m++) asm volatile("" : "+v"(acc[m][0]), "+v"(acc[m][1]));
        };
        // direct rows: dequantise the word in registers (both reference roundings, mpq_dequant.cuh) and feed v_dot2
        ColParams<DT, ZM> cp;
        if constexpr (RD > 0) cp = make_col_params<DT, 4, ZM>(sb, zb);
        auto direct = [&](int u) {
            uint32_t wp[4];
            dequant_word<DT, 4, ZM>(w[u], cp, wp);
#pragma unroll
            for (int m = 0; m < MT; m++) {
                uint32_t xp[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {  // pair order of dequant_word: slot p holds k = pair_src_k(p)
                    const int ka = pair_src_k<DT, 4>(2 * i), kb = pair_src_k<DT, 4>(2 * i + 1);
                    const uint32_t da = xs[m][u * 4 + (ka >> 1)], db = xs[m][u * 4 + (kb >> 1)];
                    const uint32_t lo = (ka & 1) ? (da >> 16) : (da & 0xffffu);
                    const uint32_t hi = (kb & 1) ? (db & 0xffff0000u) : (db << 16);
                    xp[i] = lo | hi;
                }
#pragma unroll
                for (int i = 0; i < 4; i++) acc[m][i & 1] = dot2_acc<DT>(wp[i], xp[i], acc[m][i & 1]);
            }
        };
        // table rows 0 .. RL-1, direct rows RL .. RPG-1; a direct row is issued behind the LDS reads of a table row, so its
        // VALU work covers their latency
        constexpr int RL = RPG - RD;
        float ta[8], tb[8];
        if constexpr (RL > 0) lookup(ta, 0);
#pragma unroll
        for (int u = 0; u < RL; u += 2) {
            if (u + 1 < RL) lookup(tb, u + 1);
            if (RL + u < RPG) direct(RL + u);
            fmas(ta, u);
            pin();
            if (u + 1 < RL) {
                if (u + 2 < RL) lookup(ta, u + 2);
                if (RL + u + 1 < RPG) direct(RL + u + 1);
                fmas(tb, u + 1);
                pin();
            }
        }
#pragma unroll
        for (int u = 2 * RL; u < RPG; u++) direct(u);  // more direct rows than table rows
    };

    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if constexpr (BIE_LAB5(LAB)) st0 = wall_clock64();
    uint32_t wa[RPG], wb[RPG];
    uint32_t sa = 0, za = 0, sb2 = 0, zb2 = 0;
    if (g0 < g1) {
        // the group constants are requested BEFORE the weight rows (loads return in order): the table is built while the
        // rows are still in flight instead of after the last of them has landed
        load_params(g0, sa, za);
        if (g0 + 1 < g1) load_params(g0 + 1, sb2, zb2);
        asm volatile("" ::: "memory");
        load_group(wa, g0);
        if (g0 + 1 < g1) load_group(wb, g0 + 1);
    }
    if constexpr (BIE_LAB5(LAB)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st1 = wall_clock64();
    }
    for (int g = g0; g < g1; g += 2) {
        process_group(wa, g, sa, za);
        if (g + 2 < g1) { load_params(g + 2, sa, za); load_group(wa, g + 2); }
        if (g + 1 < g1) {
            process_group(wb, g + 1, sb2, zb2);
            if (g + 3 < g1) { load_params(g + 3, sb2, zb2); load_group(wb, g + 3); }
        }
    }

    if constexpr (BIE_LAB5(LAB)) {
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]));
        st2 = wall_clock64();
        const long wid = (long)blockIdx.x * NW + wave;
        if (lane == 0 && wid < 65536) {
            g_lut_stamps[wid * 5 + 0] = st0;
            g_lut_stamps[wid * 5 + 1] = st1;
            g_lut_stamps[wid * 5 + 2] = st2;
            unsigned xcc, hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            g_lut_stamps[wid * 5 + 4] = ((unsigned long long)xcc << 32) | hwid;
        }
    }
    // ---- workgroup reduction through LDS (the tables are dead), wave order --------------------------------------
    float tot[MT];
    if constexpr (NW > 1) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(tab);
#pragma unroll
        for (int m = 0; m < MT; m++) red[(wave * MT + m) * 64 + lane] = acc[m][0] + acc[m][1];
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int m = 0; m < MT; m++) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; w++) v += red[(w * MT + m) * 64 + lane];
            tot[m] = v;
        }
    } else {
#pragma unroll
        for (int m = 0; m < MT; m++) tot[m] = acc[m][0] + acc[m][1];
    }

    lut_cross_wg_reduce<DT, MT, (LAB == 0 || LAB == 4 || LAB == 5)>(a, ls, tot, tile, slice, lane, n, N, tag, gen_next);
    if constexpr (BIE_LAB5(LAB)) {
        const long wid = (long)blockIdx.x * NW;
        if (lane == 0 && wid < 65536) g_lut_stamps[wid * 5 + 3] = wall_clock64();
    }
}

// =====================================================================================================================
// Cooperative form: the FOUR waves of a workgroup work on the SAME quantisation group at a time.  Each wave builds 4 of the 16
// table entries and looks up RPG/4 of the group's rows, so the time from "the group's rows have landed" to "the group is
// summed" is a quarter of the one-wave-per-group form's (whose ~2-3 us per group sat exposed behind the weight stream), and
// a workgroup walks its 128/RPG ... groups in arrival order.  Two tables (double buffer, one s_barrier per group; the buffer
// bit travels in the (q) byte of the prepared word, so the lookup is still one v_perm_b32 + one ds_read_b32).  All weight
// rows of the workgroup's groups are requested up front (32 dwords per lane in flight).
// =====================================================================================================================
template <int DT, int ZM, int MT, int RPG, int LAB>
__global__ __launch_bounds__(256) void mpq_gemv_lutc_kernel(const LutArgs a) {
    constexpr int NB = 8, NW = 4;
    constexpr int RPW = RPG / NW;   // rows of a group per wave
    constexpr int GW = 32 / RPW;    // groups per workgroup: 32 row loads per lane in flight
    __shared__ __attribute__((aligned(4096))) uint32_t tab[2 * 16 * 64];  // the only LDS object: starts at LDS address 0

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x % a.tiles_total;
    const int slice = blockIdx.x / a.tiles_total;
    int si = 0;
#pragma unroll
    for (int i = 1; i < LUT_MAX_SETS; i++)
        if (i < a.nsets && tile >= a.set[i].tile_begin) si = i;
    const LutSet& ls = a.set[si];
    const int N = ls.N;
    const int n = (tile - ls.tile_begin) * 64 + lane;
    const int nl = n < N ? n : N - 1;  // clamp: out-of-range lanes load valid memory and are never stored
    const int gb = slice * GW;
    unsigned tag = 0;
    unsigned gen_next = 0;
    if (a.S > 1) {
        gen_next = a.gen[tile] + 1u;
        tag = a.epoch | (gen_next & 0xffu);
    }

    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if constexpr (BIE_LAB5(LAB)) st0 = wall_clock64();
    // ---- every row of this wave, every group constant: requested now
    const uint32_t* wcol = ls.qw + nl;
    uint32_t w[GW][RPW];
    uint32_t sb[GW], zb[GW];
    const int zero_width = N / NB;
#pragma unroll
    for (int j = 0; j < GW; j++) {
        const int g = (gb + j < a.G) ? gb + j : a.G - 1;  // clamped: valid memory, never used
#pragma unroll
        for (int u = 0; u < RPW; u++) w[j][u] = __builtin_nontemporal_load(wcol + (long)(g * RPG + wave * RPW + u) * N);
        sb[j] = ls.scales[(long)g * N + nl];
        if constexpr (ZM == ZM_ASYM) {
            const uint32_t zw = reinterpret_cast<const uint32_t*>(ls.zeros)[(long)g * zero_width + nl / NB];
            zb[j] = ((zw >> ((nl % NB) * 4)) & 15u) + 1u;
        } else {
            zb[j] = reinterpret_cast<const uint16_t*>(ls.zeros)[(long)g * N + nl];
        }
        // issue order = group order: the loads return in order, so group j must not wait behind a later group's request
        asm volatile("" ::: "memory");
    }
    if constexpr (BIE_LAB5(LAB)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st1 = wall_clock64();
    }

    float acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m][0] = acc[m][1] = 0.0f;
    const uint32_t lane_addr = lane * 4;
    uint32_t m0f;
    asm("v_mov_b32 %0, 0x0f0f0f0f" : "=v"(m0f));
    const float qbase = (float)(4 * wave);

#pragma unroll
    for (int j = 0; j < GW; j++) {
        const int g = gb + j;
        if (g >= a.G) break;  // workgroup-uniform
        const int buf = j & 1;
        uint32_t* mytab = tab + buf * (16 * 64) + (4 * wave) * 64 + lane;  // this wave's 4 entries
        // activations of this wave's rows: wave-uniform scalar loads
        uint32_t xs[MT][RPW * 4];
#pragma unroll
        for (int m = 0; m < MT; m++) {
            const_u32* xd = (const_u32*)(uintptr_t)(a.x + (long)m * a.K + (long)(g * RPG + wave * RPW) * NB);
#pragma unroll
            for (int i = 0; i < RPW * 4; i++) xs[m][i] = xd[i];
        }
        // ---- entries q = 4*wave .. 4*wave+3 of the group's table
        if constexpr (LAB == 0 || LAB == 3 || LAB == 5) {
            if constexpr (DT == BIE_BF16 && ZM == ZM_SYM) {
                const float s = bf16_bits_to_f32(sb[j]), nz = -bf16_bits_to_f32(zb[j]);
                const uint32_t sel0 = sel_lo_hi<0>(), sel1 = sel_lo_hi<1>();
                const uint32_t A0 = pack_bf16x2(qbase * s, (qbase + 1.0f) * s);
                const uint32_t A1 = pack_bf16x2((qbase + 2.0f) * s, (qbase + 3.0f) * s);
                float d[4];
                bf16_pairs_sub(A0, A1, sel0, sel1, nz, d);
                const uint32_t T0 = pack_bf16x2(d[0], d[1]), T1 = pack_bf16x2(d[2], d[3]);
                mytab[0 * 64] = T0 << 16;
                mytab[1 * 64] = T0 & 0xffff0000u;
                mytab[2 * 64] = T1 << 16;
                mytab[3 * 64] = T1 & 0xffff0000u;
            } else {
                float s, z = 0.0f;
                int zq1 = 0;
                if constexpr (DT == BIE_BF16) s = bf16_bits_to_f32(sb[j]); else s = f16_bits_to_f32(sb[j]);
                if constexpr (ZM == ZM_ASYM) zq1 = (int)zb[j];
                else if constexpr (DT == BIE_BF16) z = bf16_bits_to_f32(zb[j]); else z = f16_bits_to_f32(zb[j]);
#pragma unroll
                for (int e = 0; e < 4; e++) mytab[e * 64] = __float_as_uint(lut_entry<DT, ZM>((uint32_t)(4 * wave + e), s, z, zq1));
            }
        }
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int i = 0; i < RPW * 4; i += 4)
                asm volatile("" ::"s"(xs[m][i]), "s"(xs[m][i + 1]), "s"(xs[m][i + 2]), "s"(xs[m][i + 3]));
        // table complete (and every wave is done with the buffer two groups back): LDS-only wait, the weight loads stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if constexpr (LAB == 2 || LAB == 4) {
#pragma unroll
            for (int u = 0; u < RPW; u++) acc[0][0] += __uint_as_float(w[j][u] & 0x3f7fffffu) + __uint_as_float(sb[j] << 16) + __uint_as_float(xs[0][u]);
            continue;
        }
        const uint32_t bufpat = buf ? 0x10101010u : 0u;  // the buffer bit rides in the high nibble of every (q) byte
#pragma unroll
        for (int u = 0; u < RPW; u++) {
            uint32_t we, wo;
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(we) : "v"(w[j][u]), "v"(m0f), "s"(bufpat));
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(wo) : "v"(w[j][u] >> 4), "v"(m0f), "s"(bufpat));
            float t[8];
            t[0] = lds_f32(lut_addr<0>(lane_addr, we));
            t[1] = lds_f32(lut_addr<0>(lane_addr, wo));
            t[2] = lds_f32(lut_addr<1>(lane_addr, we));
            t[3] = lds_f32(lut_addr<1>(lane_addr, wo));
            t[4] = lds_f32(lut_addr<2>(lane_addr, we));
            t[5] = lds_f32(lut_addr<2>(lane_addr, wo));
            t[6] = lds_f32(lut_addr<3>(lane_addr, we));
            t[7] = lds_f32(lut_addr<3>(lane_addr, wo));
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t p = xs[m][u * 4 + i];
                    float xlo, xhi;
                    if constexpr (DT == BIE_BF16) { xlo = __uint_as_float(p << 16); xhi = __uint_as_float(p & 0xffff0000u); }
                    else { xlo = f16_bits_to_f32(p & 0xffffu); xhi = f16_bits_to_f32(p >> 16); }
                    acc[m][0] = __builtin_fmaf(xlo, t[2 * i], acc[m][0]);
                    acc[m][1] = __builtin_fmaf(xhi, t[2 * i + 1], acc[m][1]);
                }
        }
    }
    if constexpr (BIE_LAB5(LAB)) {
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]));
        st2 = wall_clock64();
        const long wid = (long)blockIdx.x * NW + wave;
        if (lane == 0 && wid < 65536) {
            g_lut_stamps[wid * 5 + 0] = st0;
            g_lut_stamps[wid * 5 + 1] = st1;
            g_lut_stamps[wid * 5 + 2] = st2;
        }
    }

    // ---- workgroup reduction through LDS (the tables are dead), wave order ------------------------------------------
    float tot[MT];
    __syncthreads();
    float* red = reinterpret_cast<float*>(tab);
#pragma unroll
    for (int m = 0; m < MT; m++) red[(wave * MT + m) * 64 + lane] = acc[m][0] + acc[m][1];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int m = 0; m < MT; m++) {
        float v = 0.0f;
#pragma unroll
        for (int ww = 0; ww < NW; ww++) v += red[(ww * MT + m) * 64 + lane];
        tot[m] = v;
    }
    lut_cross_wg_reduce<DT, MT, (LAB == 0 || LAB == 4 || LAB == 5)>(a, ls, tot, tile, slice, lane, n, N, tag, gen_next);
    if constexpr (BIE_LAB5(LAB)) {
        const long wid = (long)blockIdx.x * NW;
        if (lane == 0 && wid < 65536) g_lut_stamps[wid * 5 + 3] = wall_clock64();
    }
}

// =====================================================================================================================
// Matrix-pipe form (M <= 16; 17 <= M <= 32 with two row blocks, RB = 2): the products go to the otherwise idle MFMA unit.  v_mfma_f32_16x16x32 wants, per lane, the 8
// consecutive-k weights of ONE column -- exactly one packed W4 word.  Lane (kb, c) = (lane >> 4, lane & 15) loads FOUR adjacent
// columns 4c .. 4c+3 of packed row 4*rq + kb with one 16-byte load (a wave instruction = 4 full 256-byte row segments), the
// four words are the A fragments of four MFMAs: fragment f's row c is column 4c + f of the tile.  A word's eight table values
// are read as 16-bit halves (ds_read_u16) and paired with one v_lshl_or_b32 each into the four operand registers.  The
// activations enter as the B operand: lane (kb, m) supplies x[m][32*rq + 8*kb .. +8] (one 16-byte buffer load; lanes m >= M
// read out of bounds = 0), so D accumulates all M rows at the price of one: lane (kb', m) ends with acc[f][r] = column
// 16*kb' + 4*r + f.  Per word: 2 + 8 + 4 VALU, 8 LDS reads, one MFMA; no FMAs, no scalar unpacking of x.
// Table layout (per wave 8 KiB): byte ((f >> 1) << 12) | (q << 8) | (rep << 6) | (c << 2) | ((f & 1) << 1) -- the f part is an
// instruction offset (the columns of a pair share a dword: even column in the low half; bf16 with two row blocks: (f & 1) << 7, a dword per
// entry), q is the byte v_perm_b32 drops into the address,
// and rep = kb & 1 holds a second copy so that the 32 lanes the LDS serves per clock (two kb values x 16 c) fall on 32 different banks.
// Lane (kb, c) builds the entries q = 4*kb .. 4*kb+3 of its four columns.
// =====================================================================================================================
typedef float lutm_acc_t __attribute__((ext_vector_type(4)));

template <int DT>
__device__ __forceinline__ lutm_acc_t lutm_mfma(const uint32_t (&w)[4], uint4_t xf, lutm_acc_t c) {
    const uint4_t wv = {w[0], w[1], w[2], w[3]};
    if constexpr (DT == BIE_F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, wv), __builtin_bit_cast(half8_t, xf), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wv), __builtin_bit_cast(bf16x8_t, xf), c, 0, 0, 0);
}

template <int OFF>
__device__ __forceinline__ void lutm_issue8(uint32_t (&l)[8], uint32_t ca, uint32_t we, uint32_t wo) {
    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(l[0]) : "v"(lut_addr<0>(ca, we)), "n"(OFF) : "memory");
    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(l[1]) : "v"(lut_addr<0>(ca, wo)), "n"(OFF) : "memory");
    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(l[2]) : "v"(lut_addr<1>(ca, we)), "n"(OFF) : "memory");
    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(l[3]) : "v"(lut_addr<1>(ca, wo)), "n"(OFF) : "memory");
    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(l[4]) : "v"(lut_addr<2>(ca, we)), "n"(OFF) : "memory");
    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(l[5]) : "v"(lut_addr<2>(ca, wo)), "n"(OFF) : "memory");
    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(l[6]) : "v"(lut_addr<3>(ca, we)), "n"(OFF) : "memory");
    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(l[7]) : "v"(lut_addr<3>(ca, wo)), "n"(OFF) : "memory");
}

// One unit (RPG packed rows of this wave's 64 columns) against RB 16-row blocks of x: the weights' reference values into the A operand of
// v_mfma_f32_16x16x32, by table (bf16) or by packed-fp16 arithmetic (fp16).  XPERM: the fp16 form still has to put the activations' eight k
// of a fragment into the pair order (the x-sharing form does it once per workgroup while staging)
// eight consecutive k (16 bytes of x) in the order the fp16 arithmetic form leaves a word's fields in: (k0, k4), (k1, k5), (k2, k6), (k3, k7)
__device__ __forceinline__ uint4_t lutm_pair_order(const uint4_t v) {
    return uint4_t{__builtin_amdgcn_perm(v.z, v.x, 0x05040100u), __builtin_amdgcn_perm(v.z, v.x, 0x07060302u),
                   __builtin_amdgcn_perm(v.w, v.y, 0x05040100u), __builtin_amdgcn_perm(v.w, v.y, 0x07060302u)};
}

struct LutmCtx { int kb, c; uint32_t* mytab; uint32_t lane_addr, wavepat, m0f; };

// xfrag(rq, dst): the RB fragments of row quad rq (registers already loaded, or an LDS read issued here -- the table form calls it where the
// read is older than the lookups its next counted wait leaves in flight); after_rq(rq): every word of w[rq] has been consumed (the x-sharing
// form loads the next unit's row quad into the same registers there)
template <int DT, int ZM, int RPG, int RB, bool DIRECT, bool XPERM, bool PAIR2 = false, class XF, class AR>
__device__ __forceinline__ void lutm_process_unit(const LutmCtx& cx, uint4_t (&w)[RPG / 4], XF&& xfrag, const uint32_t (&sb)[4],
                                                  const uint32_t (&zb)[4], lutm_acc_t (&acc)[RB][4], AR&& after_rq) {
    constexpr int RQ = RPG / 4;
    constexpr bool PAIRCOL = RB == 1 || DT == BIE_F16 || PAIR2;  // table layout: below (PAIR2: the pair form for bf16 with two row blocks too)
    const int kb = cx.kb, c = cx.c;
    uint32_t* const mytab = cx.mytab;
    const uint32_t lane_addr = cx.lane_addr, wavepat = cx.wavepat, m0f = cx.m0f;
    if constexpr (DIRECT) {
        // fp16: no table.  The reference's value of a weight IS packed-fp16 arithmetic on the exact integer -- (1024 + q) by one v_and_or_b32
        // per PAIR of fields, v_pk_add_f16 (exact), v_pk_mul_f16 = fl(q * s), v_pk_add_f16 = fl(. - z) (fused: one v_pk_fma_f16; asym: the
        // exact difference, one v_pk_mul_f16) -- 2 vector instructions per weight against 2.75 + the table build, and no LDS.  A pair is
        // (k_i, k_{i+4}) of the word, so the activations' eight k are put in the same order by four v_perm_b32 per fragment.
        // fields 0 / 2 of a byte pair sit in the mantissa of 1024 (0x6400 | q), fields 1 / 3 four bits higher in the mantissa of 64
        // (0x5400 | q << 4 = 64 + q): one shift per word instead of three
        half2_t s2[4], zlo[4], zhi[4];  // sym / fused: zlo = z; asym: the exact offsets 1024 + (zq + 1) and 64 + (zq + 1)
#pragma unroll
        for (int f = 0; f < 4; f++) {
            const half_t sh = __builtin_bit_cast(half_t, (uint16_t)sb[f]);
            s2[f] = half2_t{sh, sh};
            if constexpr (ZM == ZM_ASYM) {
                const half_t o1 = (half_t)(1024.0f + (float)zb[f]), o2 = (half_t)(64.0f + (float)zb[f]);
                zlo[f] = half2_t{o1, o1};
                zhi[f] = half2_t{o2, o2};
            } else {
                const half_t zh = __builtin_bit_cast(half_t, (uint16_t)zb[f]);
                zlo[f] = half2_t{zh, zh};
            }
        }
        uint32_t mlo, mhi;
        asm("v_mov_b32 %0, 0x000f000f" : "=v"(mlo));
        asm("v_mov_b32 %0, 0x00f000f0" : "=v"(mhi));
        const uint32_t k1024u = 0x64006400u, k64u = 0x54005400u;
        const half2_t k1024 = __builtin_bit_cast(half2_t, k1024u), k64 = __builtin_bit_cast(half2_t, k64u);
        uint4_t xq[2][RB];  // the fragments of the current and the next row quad (an LDS read of the x-sharing form gets a row quad of cover)
        xfrag(0, xq[0]);
#pragma unroll
        for (int rq = 0; rq < RQ; rq++) {
            if (rq + 1 < RQ) xfrag(rq + 1, xq[(rq + 1) & 1]);
            uint4_t xp[RB];
#pragma unroll
            for (int b = 0; b < RB; b++) xp[b] = XPERM ? lutm_pair_order(xq[rq & 1][b]) : xq[rq & 1][b];
#pragma unroll
            for (int f = 0; f < 4; f++) {
                const uint32_t word = f == 0 ? w[rq].x : (f == 1 ? w[rq].y : (f == 2 ? w[rq].z : w[rq].w));
                const uint32_t word8 = word >> 8;
                uint32_t b4[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {  // b4[i] = the reference's values of fields (i, i + 4)
                    uint32_t P;
                    if (i & 1) asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(P) : "v"(i < 2 ? word : word8), "v"(mhi), "s"(k64u));
                    else asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(P) : "v"(i < 2 ? word : word8), "v"(mlo), "s"(k1024u));
                    const half2_t a = __builtin_bit_cast(half2_t, P);
                    half2_t r;
                    if constexpr (ZM == ZM_ASYM) {
                        r = (a - ((i & 1) ? zhi[f] : zlo[f])) * s2[f];  // exact integer difference, one rounding
                    } else {
                        const half2_t q = a - ((i & 1) ? k64 : k1024);  // exact
                        if constexpr (ZM == ZM_FUSED) r = __builtin_elementwise_fma(q, s2[f], -zlo[f]);
                        else r = q * s2[f] - zlo[f];  // fl(q * s), then fl(. - z)
                    }
                    b4[i] = __builtin_bit_cast(uint32_t, r);
                }
#pragma unroll
                for (int rb = 0; rb < RB; rb++) acc[rb][f] = lutm_mfma<DT>(b4, xp[rb], acc[rb][f]);
            }
            after_rq(rq);
            // one row quad at a time: left to itself hipcc interleaves all four (every v_and_or_b32 first), 130 registers in the lone-launch
            // kernel -- one 8-wave workgroup per CU instead of two, 13.1 against the table form's 11.1 us per 4096x11008 launch
            __builtin_amdgcn_sched_barrier(0);
        }
        return;
    }
    // PAIRCOL (one 16-row block of x, and fp16 at either size): the entries of a column pair share a dword.  bf16 with two row blocks keeps one
    // dword per entry: the pair form measured 2-5 % slower there (profiles/r05_lutm_colpair_ab.txt)
    if constexpr (PAIRCOL) {
        // ---- this lane's 16 entries (q = 4*kb .. 4*kb+3 of its four columns), each stored for both bank replicas.  The entries of a column
        // PAIR (f, f + 1) share a dword -- low half the even column, high half the odd one -- so the sixteen entries leave as eight stores
        // and come out of ONE conversion per pair; a lookup of the odd column is the same read two bytes further
#pragma unroll
        for (int fp = 0; fp < 2; fp++) {
            float s[2], z[2] = {0.0f, 0.0f};
            int zq1[2] = {0, 0};
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int f = 2 * fp + h;
                if constexpr (DT == BIE_BF16) s[h] = bf16_bits_to_f32(sb[f]); else s[h] = f16_bits_to_f32(sb[f]);
                if constexpr (ZM == ZM_ASYM) zq1[h] = (int)zb[f];
                else if constexpr (DT == BIE_BF16) z[h] = bf16_bits_to_f32(zb[f]); else z[h] = f16_bits_to_f32(zb[f]);
            }
            uint32_t bits[4];
            if constexpr (DT == BIE_BF16 && ZM == ZM_SYM) {
                // fl(q * s) of both columns in one v_cvt_pk_bf16_f32, unpacked by shift / mask, the second rounding's subtraction in fp32, one more
                // v_cvt_pk_bf16_f32: 8 operations per pair of entries instead of 11.  (NOT the dot-unit subtraction of the one-row list kernel,
                // 1 * lo + 0 * hi - z: the halves here are two different COLUMNS, and 0 * inf = NaN would carry one column's infinite scale into
                // its neighbour's entries -- caught by the special-value tests)
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const float qf = (float)(4 * kb + e);
                    const uint32_t A = pack_bf16x2(qf * s[0], qf * s[1]);
                    bits[e] = pack_bf16x2(__uint_as_float(A << 16) - z[0], __uint_as_float(A & 0xffff0000u) - z[1]);
                }
            } else if constexpr (DT == BIE_F16) {
                // fp16: both columns' entries in ONE packed-fp16 pass (fl(q * s): v_pk_mul_f16, fl(. - z): v_pk_add_f16 -- sym, the reference's two roundings;
                // fused: v_pk_fma_f16; asym: the exact integer difference times s, one rounding)
                const half2_t s2 = half2_t{__builtin_bit_cast(half_t, (uint16_t)sb[2 * fp]), __builtin_bit_cast(half_t, (uint16_t)sb[2 * fp + 1])};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int q = 4 * kb + e;
                    half2_t r;
                    if constexpr (ZM == ZM_ASYM) {
                        r = half2_t{(half_t)(float)(q - zq1[0]), (half_t)(float)(q - zq1[1])} * s2;
                    } else {
                        const half2_t z2 = half2_t{__builtin_bit_cast(half_t, (uint16_t)zb[2 * fp]), __builtin_bit_cast(half_t, (uint16_t)zb[2 * fp + 1])};
                        const half2_t q2 = half2_t{(half_t)(float)q, (half_t)(float)q};
                        if constexpr (ZM == ZM_FUSED) r = __builtin_elementwise_fma(q2, s2, -z2);
                        else r = q2 * s2 - z2;
                    }
                    bits[e] = __builtin_bit_cast(uint32_t, r);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {  // both values are representable in the 16-bit type: the conversion is exact
                    const uint32_t q = (uint32_t)(4 * kb + e);
                    bits[e] = pack_bf16x2(lut_entry<DT, ZM>(q, s[0], z[0], zq1[0]), lut_entry<DT, ZM>(q, s[1], z[1], zq1[1]));
                }
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                uint32_t* p = mytab + fp * 1024 + (4 * kb + e) * 64 + c;
                p[0] = bits[e];
                p[16] = bits[e];
            }
        }
    } else {
        // ---- this lane's 16 entries (q = 4*kb .. 4*kb+3 of its four columns), each stored for both bank replicas;
        // 16-bit value in the low half of a dword
#pragma unroll
        for (int f = 0; f < 4; f++) {
            float s, z = 0.0f;
            int zq1 = 0;
            if constexpr (DT == BIE_BF16) s = bf16_bits_to_f32(sb[f]); else s = f16_bits_to_f32(sb[f]);
            if constexpr (ZM == ZM_ASYM) zq1 = (int)zb[f];
            else if constexpr (DT == BIE_BF16) z = bf16_bits_to_f32(zb[f]); else z = f16_bits_to_f32(zb[f]);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t q = (uint32_t)(4 * kb + e);
                const float t = lut_entry<DT, ZM>(q, s, z, zq1);
                uint32_t bits;
                if constexpr (DT == BIE_BF16) bits = __float_as_uint(t) >> 16; else bits = f32_to_f16_bits(t);
                uint32_t* p = mytab + (f >> 1) * 1024 + q * 64 + (f & 1) * 32 + c;
                p[0] = bits;
                p[16] = bits;
            }
        }
    }
    // ---- fragment steps st = 4*rq + f: the eight 16-bit lookups of step st+1 are in flight while step st feeds the MFMA
    // (plain ds_read_u16 + one v_lshl_or_b32 per pair: with SRAM-ECC register files a d16 load does not preserve the other
    // half of its destination, so a d16 / d16_hi pair cannot share a register)
    uint32_t la[8], lb[8];
    auto issue = [&](uint32_t (&l)[8], int rq, int f) {
        const uint32_t word = f == 0 ? w[rq].x : (f == 1 ? w[rq].y : (f == 2 ? w[rq].z : w[rq].w));
        uint32_t we, wo;
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(we) : "v"(word), "v"(m0f), "s"(wavepat));
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(wo) : "v"(word >> 4), "v"(m0f), "s"(wavepat));
        constexpr int ODD = PAIRCOL ? 2 : 128;  // the odd column of a pair: the high half of the same dword / a block of its own
        if (f == 0) lutm_issue8<0>(l, lane_addr, we, wo);
        else if (f == 1) lutm_issue8<ODD>(l, lane_addr, we, wo);
        else if (f == 2) lutm_issue8<4096>(l, lane_addr, we, wo);
        else lutm_issue8<4096 + ODD>(l, lane_addr, we, wo);
    };
    auto wait_pack = [&](uint32_t (&l)[8], bool more, uint32_t (&b)[4]) {
        if (more) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]), "+v"(l[4]), "+v"(l[5]), "+v"(l[6]), "+v"(l[7])::"memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]), "+v"(l[4]), "+v"(l[5]), "+v"(l[6]), "+v"(l[7])::"memory");
#pragma unroll
        for (int i = 0; i < 4; i++) b[i] = (l[2 * i + 1] << 16) | l[2 * i];
    };
    uint4_t xq[2][RB];  // the fragments of the current and the next row quad
    xfrag(0, xq[0]);
    issue(la, 0, 0);
#pragma unroll
    for (int st = 0; st < 4 * RQ; st++) {
        const int rq = st >> 2, f = st & 3;
        const bool more = st + 1 < 4 * RQ;
        uint32_t b[4];
        if (more && f == 3) xfrag(rq + 1, xq[(rq + 1) & 1]);  // ahead of the next issue: complete at this step's counted wait
        if (st & 1) {
            if (more) issue(la, (st + 1) >> 2, (st + 1) & 3);
            if (f == 2) after_rq(rq);  // word 3 of this row quad has just gone out
            wait_pack(lb, more, b);
        } else {
            if (more) issue(lb, (st + 1) >> 2, (st + 1) & 3);
            if (f == 2) after_rq(rq);
            wait_pack(la, more, b);
        }
#pragma unroll
        for (int rb = 0; rb < RB; rb++) acc[rb][f] = lutm_mfma<DT>(b, xq[rq & 1][rb], acc[rb][f]);
    }
}

// what one workgroup of the matrix-pipe form needs about its layer: filled from the kernel arguments (one launch per layer / set of
// layers sharing x) or from a device-resident ListEntry (bie_mpq_list_*, 3 <= M <= 32: many layers in one launch)
struct LutmView {
    const uint32_t* qw; const uint16_t* scales; const void* zeros; const uint16_t* bias; uint16_t* y; const uint16_t* x;
    unsigned long long* gran;  // [S-1][M][ncat] granules of the launch / of the list entry
    unsigned* gen;             // generation words, indexed by the tile number the granule columns are counted in
    int N, M, K, G, S, gpw, hshift;
    long ncat;                 // granule columns: tiles * 64
    unsigned qw_bytes = 0;     // bytes of qw (the x-sharing form reads the rows through a buffer descriptor)
};

// the scale / zero patterns of a lane's four adjacent columns n4 .. n4 + 3 for `unit` (asym: zq + 1 of the packed zero fields)
template <int ZM>
__device__ __forceinline__ void lutm_load_params(const LutmView& lv, const int n4, const int unit, uint32_t (&sb)[4], uint32_t (&zb)[4]) {
    constexpr int NB = 8;
    const int N = lv.N;
    const int g = unit >> lv.hshift;
    const uint2_t s2 = *reinterpret_cast<const uint2_t*>(lv.scales + (long)g * N + n4);
    sb[0] = s2.x & 0xffffu; sb[1] = s2.x >> 16; sb[2] = s2.y & 0xffffu; sb[3] = s2.y >> 16;
    if constexpr (ZM == ZM_ASYM) {
        const uint32_t zw = reinterpret_cast<const uint32_t*>(lv.zeros)[(long)g * (N / NB) + n4 / NB];
#pragma unroll
        for (int f = 0; f < 4; f++) zb[f] = ((zw >> (((n4 % NB) + f) * 4)) & 15u) + 1u;
    } else {
        const uint2_t z2 = *reinterpret_cast<const uint2_t*>(reinterpret_cast<const uint16_t*>(lv.zeros) + (long)g * N + n4);
        zb[0] = z2.x & 0xffffu; zb[1] = z2.x >> 16; zb[2] = z2.y & 0xffffu; zb[3] = z2.y >> 16;
    }
}

// One row of a column tile, lane = column: a publisher slice writes its partial sum as a tagged granule; the tile's last slice collects the
// others' (spinning on the tags, bounded: a reducer that never sees them returns NaN and raises the status bit), rounds, adds the bias, stores
template <int DT>
__device__ __forceinline__ void lutm_finish_row(const LutmView& lv, float tot, const int m, const int slice, const unsigned tag, const unsigned tag_skew,
                                                const int spin_limit, unsigned* status, const long col, const long ncat, const bool owner, const int n,
                                                const int lane) {
    const int M = lv.M, N = lv.N;
    if (lv.S > 1) {
        if (slice != lv.S - 1) {
            const unsigned long long gval = ((unsigned long long)tag << 32) | __float_as_uint(tot);
            __hip_atomic_store(lv.gran + ((long)slice * M + m) * ncat + col, gval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        float v = 0.0f;
        for (int s0 = 0; s0 < lv.S - 1; s0 += 8) {
            unsigned long long gv[8];
            bool ready;
            int spins = 0;
            do {
#pragma unroll
                for (int jj = 0; jj < 8; jj++) {
                    const int sidx = (s0 + jj < lv.S - 1) ? s0 + jj : lv.S - 2;
                    gv[jj] = __hip_atomic_load(lv.gran + ((long)sidx * M + m) * ncat + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                ready = true;
#pragma unroll
                for (int jj = 0; jj < 8; jj++) ready = ready && ((unsigned)(gv[jj] >> 32) == (tag ^ tag_skew));
                ready = __builtin_amdgcn_ballot_w64(!ready) == 0;
                if (!ready) __builtin_amdgcn_s_sleep(2);
            } while (!ready && ++spins < spin_limit);
#pragma unroll
            for (int jj = 0; jj < 8; jj++)
                if (s0 + jj < lv.S - 1) v += __uint_as_float((unsigned)gv[jj]);
            if (!ready) {  // wave-uniform: never a silent number
                v = __uint_as_float(0x7fc00000u);
                if (lane == 0 && status) __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        tot = v + tot;
    }
    if (owner) {
        float o = dt_traits<DT>::round(tot);
        if (lv.bias) o = o + dt_traits<DT>::load(lv.bias, n);
        dt_traits<DT>::store(lv.y, (long)m * N + n, o);
    }
}

// RB: 16-row blocks of x served by one pass over the weights (RB = 2: 17 <= M <= 32 -- the lookups and the pairing are shared, a word
// costs one more MFMA, the activations one more 16-byte load per row quad; list launches only)
template <int DT, int ZM, int RPG, int NW, bool PF, int RB = 1>  // PF: the next unit's loads in flight under the current one (costs ~40 registers)
__device__ __forceinline__ void lutm_body(const LutmView& lv, const int tile_local, const int gtile, const int slice, const unsigned epoch,
                                          unsigned* status, const unsigned tag_skew, const int spin_limit) {
    constexpr int RQ = RPG / 4;  // row quads (32 k) per unit
#ifdef BIE_LUTM_F16_TABLE
    constexpr bool DIRECT = false;
#else
    constexpr bool DIRECT = DT == BIE_F16;  // fp16: packed-fp16 arithmetic instead of the table (process_unit)
#endif
    static_assert(NW <= 8 && RPG % 4 == 0, "wave bits of the table address / whole row quads");
    __shared__ __attribute__((aligned(8192))) uint32_t tab[NW * 2048];  // the only LDS object: starts at LDS address 0

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, kb = lane >> 4;
    const int N = lv.N;  // N % 4 == 0 on this path: a lane's four columns are all in range or all out
    const int M = lv.M;
    const int nt0 = tile_local * 64;
    const int n4 = nt0 + 4 * c < N ? nt0 + 4 * c : N - 4;  // clamp: out-of-range columns load valid memory and are never stored
    const int g0 = (slice * NW + wave) * lv.gpw;
    int g1 = g0 + lv.gpw;
    if (g1 > lv.G) g1 = lv.G;
    unsigned tag = 0, gen_next = 0;
    if (lv.S > 1) {
        gen_next = lv.gen[gtile] + 1u;
        tag = epoch | (gen_next & 0xffu);
    }

    // x as a raw buffer: rows >= M are out of bounds and read as 0 (the unused columns of the MFMA's second operand)
    const uint64_t xb = (uint64_t)(uintptr_t)lv.x;
    const uint32_t xlo = __builtin_amdgcn_readfirstlane((uint32_t)xb), xhi = __builtin_amdgcn_readfirstlane((uint32_t)(xb >> 32));  // unsigned: no sign extension
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)xhi << 32) | xlo), 0,
                                                         __builtin_amdgcn_readfirstlane((uint32_t)((long)M * lv.K * 2)), 0x00020000);
    uint32_t xvoff[RB];
#pragma unroll
    for (int b = 0; b < RB; b++) xvoff[b] = 16 * b + c < M ? (uint32_t)((16 * b + c) * lv.K * 2 + kb * 16) : 0x80000000u;

    const uint32_t* wcol = lv.qw + n4;
    auto load_params = [&](int unit, uint32_t (&sb)[4], uint32_t (&zb)[4]) { lutm_load_params<ZM>(lv, n4, unit, sb, zb); };
    auto load_unit = [&](uint4_t (&w)[RQ], uint4_t (&xf)[RB][RQ], int unit) {
#pragma unroll
        for (int rq = 0; rq < RQ; rq++) {
            const long row = (long)unit * RPG + 4 * rq + kb;
            w[rq] = __builtin_nontemporal_load(reinterpret_cast<const uint4_t*>(wcol + row * N));
#pragma unroll
            for (int b = 0; b < RB; b++)
                xf[b][rq] = __builtin_bit_cast(uint4_t, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xvoff[b], (uint32_t)((unit * RPG + 4 * rq) * 16), 0));
        }
    };

    lutm_acc_t acc[RB][4];
#pragma unroll
    for (int b = 0; b < RB; b++)
#pragma unroll
        for (int f = 0; f < 4; f++) acc[b][f] = lutm_acc_t{0.0f, 0.0f, 0.0f, 0.0f};

    const uint32_t lane_addr = (uint32_t)((kb & 1) * 64 + c * 4);
    const uint32_t wavepat = (uint32_t)wave * 0x20202020u;  // byte = (wave << 5) | q
    uint32_t* mytab = tab + wave * 2048;
    uint32_t m0f;
    asm("v_mov_b32 %0, 0x0f0f0f0f" : "=v"(m0f));

    const LutmCtx cx{kb, c, mytab, lane_addr, wavepat, m0f};
    auto process_unit = [&](uint4_t (&w)[RQ], const uint4_t (&xf)[RB][RQ], const uint32_t (&sb)[4], const uint32_t (&zb)[4]) {
        lutm_process_unit<DT, ZM, RPG, RB, DIRECT, true>(
            cx, w,
            [&](int rq, uint4_t (&dst)[RB]) {
#pragma unroll
                for (int b = 0; b < RB; b++) dst[b] = xf[b][rq];
            },
            sb, zb, acc, [](int) {});
    };

    // one unit per wave is the normal plan (its rows, activations and constants are requested up front, constants first so that
    // the table is built under the row latency); further units of a wave (very wide layers) are taken one after the other
    uint4_t wa[RQ], wn[RQ];
    uint4_t xa[RB][RQ], xn[RB][RQ];
    uint32_t sa[4] = {0, 0, 0, 0}, za[4] = {0, 0, 0, 0}, sn[4] = {0, 0, 0, 0}, zn[4] = {0, 0, 0, 0};
    if (!PF || g1 - g0 <= 1) {  // the per-layer plan: one unit per wave (more only on very wide layers)
        for (int g = g0; g < g1; g++) {
            load_params(g, sa, za);
            asm volatile("" ::: "memory");
            load_unit(wa, xa, g);
            process_unit(wa, xa, sa, za);
        }
    } else {  // several units per wave (list launches, very wide layers): the next unit's rows and constants are in flight under the current one
        load_params(g0, sa, za);
        load_unit(wa, xa, g0);
        for (int g = g0; g < g1; g += 2) {
            const int gn = g + 1 < g1 ? g + 1 : g1 - 1;  // clamped: an unconditional issue keeps the waits exact
            load_params(gn, sn, zn);
            load_unit(wn, xn, gn);
            process_unit(wa, xa, sa, za);
            if (g + 1 >= g1) break;
            const int gnn = g + 2 < g1 ? g + 2 : g1 - 1;
            load_params(gnn, sa, za);
            load_unit(wa, xa, gnn);
            process_unit(wn, xn, sn, zn);
        }
    }

    // ---- workgroup reduction through LDS (the tables are dead).  D layout of the 16x16 MFMA: lane (kb', m) holds in acc[f][r] the
    // output of tile column 16 * kb' + 4 * r + f for x row m.  red[wave][m][64]
    __syncthreads();
    float* red = reinterpret_cast<float*>(tab);
#pragma unroll
    for (int b = 0; b < RB; b++)
        if (16 * b + c < M) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float4_t v = {acc[b][0][r], acc[b][1][r], acc[b][2][r], acc[b][3][r]};
                *reinterpret_cast<float4_t*>(red + ((wave * M + 16 * b + c) * 64 + 16 * kb + 4 * r)) = v;
            }
        }
    __syncthreads();
    // wave w finishes x rows m = w, w + NW, ...: lane = column of the tile
    const int n = nt0 + lane;
    const bool owner = n < N;
    const long ncat = lv.ncat;
    const long col = (long)gtile * 64 + lane;
    for (int m = wave; m < M; m += NW) {
        float tot = 0.0f;
#pragma unroll
        for (int ww = 0; ww < NW; ww++) tot += red[(ww * M + m) * 64 + lane];
        lutm_finish_row<DT>(lv, tot, m, slice, tag, tag_skew, spin_limit, status, col, ncat, owner, n, lane);
    }
    if (lv.S > 1 && slice == lv.S - 1 && threadIdx.x == 0) lv.gen[gtile] = gen_next;  // read only by the next launch
}


template <int DT, int ZM, int RPG, int NW, int RB>
__global__ __launch_bounds__(NW * 64, (RB == 1 ? 2 : 1)) void mpq_gemv_lutm_kernel(const LutArgs a) {
    const int tile = blockIdx.x % a.tiles_total;
    const int slice = blockIdx.x / a.tiles_total;
    int si = 0;
#pragma unroll
    for (int i = 1; i < LUT_MAX_SETS; i++)
        if (i < a.nsets && tile >= a.set[i].tile_begin) si = i;
    const LutSet& ls = a.set[si];
    const LutmView v{ls.qw, ls.scales, ls.zeros, ls.bias, ls.y, a.x, a.gran, a.gen, ls.N, a.M, a.K, a.G, a.S, a.groups_per_wave, a.hshift,
                     (long)a.tiles_total * 64};
    lutm_body<DT, ZM, RPG, NW, false, RB>(v, tile - ls.tile_begin, tile, slice, a.epoch, a.status, a.tag_skew, a.spin_limit);
}

// ONE launch over a LIST of layers, 3 <= M <= 32 (bie_mpq_list_*): block b -> {entry, tile | slice << 20}; the entry's granules and
// generation words are its own (tile numbers local to the entry).
template <int DT, int ZM, int RPG, int NW, bool PF, int RB>
__global__ __launch_bounds__(NW * 64, (RB == 1 ? 2 : (NW == 4 && RPG <= 16 ? 3 : 1))) void mpq_lutm_list_kernel(const ListEntry* __restrict__ ent, const uint2_t* __restrict__ blk, const int M,
                                                                  const unsigned epoch, unsigned* status, const unsigned tag_skew, const int spin_limit) {
    typedef const __attribute__((address_space(4))) uint2_t cu2_t;
    typedef const __attribute__((address_space(4))) ListEntry cent_t;
    const uint2_t rec = *((cu2_t*)(uintptr_t)(blk + blockIdx.x));
    cent_t* e = (cent_t*)(uintptr_t)(ent + rec.x);
    const int tile = (int)(rec.y & 0xfffffu), slice = (int)(rec.y >> 20);
    const LutmView v{e->qw, e->scales, e->zeros, e->bias, e->y, e->x, e->gran, e->gen, e->N, M, e->K, e->G, e->S, e->gpw, e->hshift, (long)e->tiles * 64};
    lutm_body<DT, ZM, RPG, NW, PF, RB>(v, tile, tile, slice, epoch, status, tag_skew, spin_limit);
}

// =====================================================================================================================
// x-sharing list form (3 <= M <= 32 over a list with enough layers to fill the chip without slicing K finely): the four waves of a
// workgroup take four ADJACENT column tiles over the SAME k units, and the unit's M x (RPG * 8) block of x enters the workgroup ONCE --
// whole 128-byte lines (eight rows x 128 bytes per wave instruction), written to LDS in fragment order (a wave's B operand of row quad rq,
// row block b is the 1 KiB at ((rq * RB + b) * 64 + lane) * 16: lane-linear ds_read_b128) -- instead of once per wave in fragment-shaped
// 64-byte row pieces from L2.  In the k-split form above every wave pulls its own M x 256 bytes per unit: at 32 rows that is twice the weight
// bytes through the texture path, and the launch ran at 9 TB/s of L2 -> CU traffic whatever the decode cost (the fp16 arithmetic form took a
// third of the instructions out of the 16-row launch and all of its LDS work, and moved the time by 8 %; at 32 rows by nothing).
// A wave owns its 64 columns over the workgroup's whole k range: no cross-wave sum; K slices (S > 1) meet through the tagged granules as above.
// One barrier per unit: stage (u + 1) is loaded under the work on u and written to the other buffer before the barrier.
// =====================================================================================================================
template <int DT, int ZM, int RPG, int RB, int NW>
__device__ __forceinline__ void lutm_xs_body(const LutmView& lv, const int tile0, const int slice, const unsigned epoch, unsigned* status,
                                             const unsigned tag_skew, const int spin_limit) {
    constexpr int RQ = RPG / 4;
#ifdef BIE_LUTM_F16_TABLE
    constexpr bool DIRECT = false;
#else
    constexpr bool DIRECT = DT == BIE_F16;
#endif
    // LDS: [tables][two stage buffers]; the closing [M][64] transposes of the waves (RB * 1024 words each) reuse everything from word 0.
    // bf16 tables: the pair form (always, here) fills only 128 of the 256 bytes of a table row (q), so the tables of a PAIR of waves interleave
    // in one 8 KiB block -- wave w: block w >> 1, byte offset (w & 1) * 128 -- and four waves take 16 KiB instead of 32
    constexpr int REDW = RB * 1024;
    constexpr int TABW = DIRECT ? 0 : (NW / 2) * 2048;
    constexpr int XW = RB * RQ * 256;                // words of one stage buffer: RB * RQ fragments of 1 KiB
    constexpr int XOFF = TABW;
    constexpr int LDSW = TABW + 2 * XW > NW * REDW ? TABW + 2 * XW : NW * REDW;
    static_assert(NW % 2 == 0, "the bf16 tables interleave by wave pairs");
    __shared__ __attribute__((aligned(8192))) uint32_t tab[LDSW];  // the only LDS object: starts at LDS address 0

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, kb = lane >> 4;
    const int N = lv.N, M = lv.M;
    const int tiles = (int)(lv.ncat >> 6);
    const bool active = tile0 + wave < tiles;  // a wave past the last tile of a ragged quad works on the last tile and stores nothing
    const int tile = active ? tile0 + wave : tiles - 1;
    const int nt0 = tile * 64;
    const int n4 = nt0 + 4 * c < N ? nt0 + 4 * c : N - 4;
    const int g0 = slice * lv.gpw;
    int g1 = g0 + lv.gpw;
    if (g1 > lv.G) g1 = lv.G;
    unsigned tag = 0, gen_next = 0;
    if (lv.S > 1) {
        gen_next = lv.gen[tile] + 1u;
        tag = epoch | (gen_next & 0xffu);
    }

    const uint64_t xb = (uint64_t)(uintptr_t)lv.x;
    const uint32_t xlo = __builtin_amdgcn_readfirstlane((uint32_t)xb), xhi = __builtin_amdgcn_readfirstlane((uint32_t)(xb >> 32));
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)xhi << 32) | xlo), 0,
                                                         __builtin_amdgcn_readfirstlane((uint32_t)((long)M * lv.K * 2)), 0x00020000);
    // staging: lane (j8, c3) = (lane >> 3, lane & 7) moves the 16 bytes of row 8 * o + c3, k chunk 8 * h + j8 of the unit; (o, h) from
    // the wave and the pass.  Eight contiguous lanes = eight rows of one chunk = 128 contiguous bytes of a fragment: conflict-free ds_write_b128
    constexpr int HALVES = RPG >= 8 ? RPG / 8 : 1;
    constexpr int COMBOS = RB * 2 * HALVES;
    constexpr int SLD = (COMBOS + NW - 1) / NW;
    const int c3 = lane & 7, j8 = lane >> 3;
    uint32_t svoff[SLD], sdst[SLD];
    bool svalid[SLD];
#pragma unroll
    for (int i = 0; i < SLD; i++) {
        const int combo = i * NW + wave;
        const int o = combo % (RB * 2), h = combo / (RB * 2);
        const int r = o * 8 + c3, j = h * 8 + j8;
        svalid[i] = combo < COMBOS && j < RPG;
        svoff[i] = (svalid[i] && r < M) ? (uint32_t)(r * lv.K * 2 + j * 16) : 0x80000000u;  // rows >= M: zeros
        sdst[i] = (uint32_t)(((((j >> 2) * RB + (r >> 4)) * 64 + (j & 3) * 16 + (r & 15)) * 4));
    }
    uint32_t* const xst = tab + XOFF;
    auto stage_load = [&](uint4_t (&xs)[SLD], int unit) {
#pragma unroll
        for (int i = 0; i < SLD; i++) xs[i] = __builtin_bit_cast(uint4_t, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, svoff[i], (uint32_t)(unit * RPG * 16), 0));
    };
    auto stage_store = [&](const uint4_t (&xs)[SLD], int buf) {
#pragma unroll
        for (int i = 0; i < SLD; i++)
            if (svalid[i]) {
                uint4_t v = xs[i];
                if constexpr (DIRECT) v = lutm_pair_order(v);  // once per workgroup, not once per wave and fragment
                *reinterpret_cast<uint4_t*>(xst + buf * XW + sdst[i]) = v;
            }
    };
    // a wave's B operand of (row quad rq, row block b): the 1 KiB at ((rq * RB + b) * 64 + lane) * 16 of the stage buffer
    const uint32_t xfrag_addr = (uint32_t)(XOFF * 4 + lane * 16);
    auto read_frags = [&](int rq, uint4_t (&dst)[RB], int buf) {
        if constexpr (DIRECT) {
#pragma unroll
            for (int b = 0; b < RB; b++) dst[b] = *reinterpret_cast<const uint4_t*>(xst + buf * XW + ((rq * RB + b) * 64 + lane) * 4);
        } else {
            // the table form counts its own lgkmcnt: the fragment reads go out as asm too, where lutm_process_unit asks for them (in-order
            // LDS returns: the next counted wait of the lookups covers them)
            const uint32_t a = xfrag_addr + (uint32_t)(buf * XW * 4);
#pragma unroll
            for (int b = 0; b < RB; b++) asm volatile("ds_read_b128 %0, %1" : "=v"(dst[b]) : "v"(a + (uint32_t)((rq * RB + b) * 1024)) : "memory");
        }
    };

    auto load_params = [&](int unit, uint32_t (&sb)[4], uint32_t (&zb)[4]) { lutm_load_params<ZM>(lv, n4, unit, sb, zb); };
    // the rows through a buffer descriptor: lane (kb, c) reads 16 bytes of row 4 * rq + kb at a scalar offset per (unit, row quad).  The
    // request for the unit after the workgroup's last goes out all the same -- against a descriptor of no records, which fetches nothing --
    // so that every wait of the loop is counted exactly (a branch around a load makes the compiler drain the queue at the join)
    const uint64_t qb = (uint64_t)(uintptr_t)lv.qw;
    const uint32_t qlo = __builtin_amdgcn_readfirstlane((uint32_t)qb), qhi = __builtin_amdgcn_readfirstlane((uint32_t)(qb >> 32));
    const auto qrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)qhi << 32) | qlo), 0, __builtin_amdgcn_readfirstlane((int)lv.qw_bytes), 0x00020000);
    const auto qnone = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)(((uint64_t)qhi << 32) | qlo), 0, 0, 0x00020000);
    const uint32_t wvoff = (uint32_t)((kb * N + n4) * 4);
    auto load_row_quad = [&](int unit, int rq, bool real) {
        const uint32_t soff = (uint32_t)(((long)unit * RPG + 4 * rq) * N * 4);
        return __builtin_bit_cast(uint4_t, __builtin_amdgcn_raw_buffer_load_b128(real ? qrsrc : qnone, wvoff, soff, 2));  // aux 2 = nt
    };

    lutm_acc_t acc[RB][4];
#pragma unroll
    for (int b = 0; b < RB; b++)
#pragma unroll
        for (int f = 0; f < 4; f++) acc[b][f] = lutm_acc_t{0.0f, 0.0f, 0.0f, 0.0f};

    uint32_t m0f;
    asm("v_mov_b32 %0, 0x0f0f0f0f" : "=v"(m0f));
    const LutmCtx cx{kb, c, tab + (wave >> 1) * 2048 + (wave & 1) * 32, (uint32_t)((kb & 1) * 64 + c * 4 + (wave & 1) * 128), (uint32_t)(wave >> 1) * 0x20202020u, m0f};

    // The rows in a ring of DEPTH register sets: row quad rq of unit g + DEPTH is requested into the set of unit g as soon as unit g has consumed
    // it (DEPTH units ahead of its use); the next unit's constants and its stage of x at the top of the step.  The trip count is the
    // workgroup's: every wave meets every barrier
#ifdef BIE_LUTM_XS_DEPTH
    constexpr int DEPTH = BIE_LUTM_XS_DEPTH;
#else
    constexpr int DEPTH = 1;
#endif
    uint4_t w[DEPTH][RQ], xs[SLD];
    uint32_t sa[4] = {0, 0, 0, 0}, za[4] = {0, 0, 0, 0}, sn[4] = {0, 0, 0, 0}, zn[4] = {0, 0, 0, 0};
    load_params(g0, sn, zn);
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
#pragma unroll
        for (int rq = 0; rq < RQ; rq++) w[d][rq] = load_row_quad(g0 + d < g1 ? g0 + d : g0, rq, g0 + d < g1);
    stage_load(xs, g0);
    stage_store(xs, 0);
    __syncthreads();
    auto step = [&](uint4_t (&wc)[RQ], int g) {
        const int p = (g - g0) & 1;
        const bool more = g + 1 < g1;           // workgroup-uniform
        const int gn = more ? g + 1 : g;        // the last step asks for its own constants and stage again (cache hits; stored to the idle buffer)
        const bool ring = g + DEPTH < g1;
        const int gr = ring ? g + DEPTH : g;
#pragma unroll
        for (int f = 0; f < 4; f++) { sa[f] = sn[f]; za[f] = zn[f]; }
        load_params(gn, sn, zn);
        stage_load(xs, gn);
        asm volatile("" ::: "memory");  // the requests stay where they are written: hipcc otherwise sinks them to their first use (lower register
                                        // pressure), and the loop runs without a unit of rows in flight
        // bf16 with two row blocks: column-pair table entries here (6.9 against 7.3 us per 4096x11008 layer at 32 rows, profiles/r06_lutm_xs.txt;
        // in the k-split form, at 157 registers, the pair form was the slower one)
        lutm_process_unit<DT, ZM, RPG, RB, DIRECT, false, true>(
            cx, wc, [&](int rq, uint4_t (&dst)[RB]) { read_frags(rq, dst, p); }, sa, za, acc,
            [&](int rq) {
                asm volatile("" ::: "memory");
                wc[rq] = load_row_quad(gr, rq, ring);
                asm volatile("" ::: "memory");
            });
        stage_store(xs, p ^ 1);
        __syncthreads();
    };
    for (int g = g0; g < g1; g += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
            if (g + d < g1) step(w[d], g + d);
    }

    // ---- a wave's own [M][64] through LDS (D layout of the 16x16 MFMA: lane (kb', m) holds in acc[f][r] column 16 * kb' + 4 * r + f of row m)
    float* red = reinterpret_cast<float*>(tab + wave * REDW);
#pragma unroll
    for (int b = 0; b < RB; b++)
        if (16 * b + c < M) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const float4_t v = {acc[b][0][r], acc[b][1][r], acc[b][2][r], acc[b][3][r]};
                *reinterpret_cast<float4_t*>(red + ((16 * b + c) * 64 + 16 * kb + 4 * r)) = v;
            }
        }
    __syncthreads();
    if (!active) return;
    const int n = nt0 + lane;
    const bool owner = n < N;
    const long ncat = lv.ncat;
    const long col = (long)tile * 64 + lane;
    for (int m = 0; m < M; m++) {
        float tot = red[m * 64 + lane];
        lutm_finish_row<DT>(lv, tot, m, slice, tag, tag_skew, spin_limit, status, col, ncat, owner, n, lane);
    }
    if (lv.S > 1 && slice == lv.S - 1 && lane == 0) lv.gen[tile] = gen_next;  // read only by the next launch
}

// block b -> {entry, first tile of the quad | slice << 20}
// four four-wave workgroups per CU (<= 128 registers; LDS 16 .. 32 KiB each): bf16 at 32 rows 6.8 -> 6.6 us per 4096x11008 layer against the 136-140
// registers hipcc takes when left alone (profiles/r06_lutm_xs.txt)
template <int DT, int ZM, int RPG, int RB, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 4 && RPG <= 16 ? 4 : 1)) void mpq_lutm_xs_list_kernel(const ListEntry* __restrict__ ent, const uint2_t* __restrict__ blk, const int M, const unsigned epoch,
                                                               unsigned* status, const unsigned tag_skew, const int spin_limit) {
    typedef const __attribute__((address_space(4))) uint2_t cu2_t;
    typedef const __attribute__((address_space(4))) ListEntry cent_t;
    const uint2_t rec = *((cu2_t*)(uintptr_t)(blk + blockIdx.x));
    cent_t* e = (cent_t*)(uintptr_t)(ent + rec.x);
    const int tile0 = (int)(rec.y & 0xfffffu), slice = (int)(rec.y >> 20);
    const LutmView v{e->qw, e->scales, e->zeros, e->bias, e->y, e->x, e->gran, e->gen, e->N, M, e->K, e->G, e->S, e->gpw, e->hshift, (long)e->tiles * 64, e->qw_bytes};
    lutm_xs_body<DT, ZM, RPG, RB, NW>(v, tile0, slice, epoch, status, tag_skew, spin_limit);
}

#ifdef BIE_LAB_BUILD
// tuning aid: copy the stamps of the last BIE_GEMV_LAB=5 launch to the host (synchronises the device)
extern "C" int bie_debug_lut_stamps(unsigned long long* out, int n_waves) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lut_stamps), (size_t)n_waves * 5 * sizeof(unsigned long long));
}
#endif

// ---- host side --------------------------------------------------------------------------------------------
static int lut_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// W4, implicit groups of 32 / 64 / 128 / 256 that tile K exactly.  FMA form: bf16, M <= 2.  Matrix-pipe form: fp16 / bf16, M <= 16 per layer (17 .. 32 in list launches only).
// Measured (4096x11008 / 4096x4096, us per launch): bf16 M = 1: FMA form 10.05 / 5.92, matrix-pipe 11.66 / 6.84 -> FMA form;
// M = 2: 15.1 / 7.3 against 11.8 / 6.8; M = 4, 8, 16: 12.2 / 7.3, 13.0 / 7.9, 15.7 / 10.0 against the MFMA GEMM's 18.7-20.2 /
// 10.5-12.2 -> matrix-pipe form for 2 <= M <= 16.  fp16 has no FMA form: matrix-pipe from M = 1 (11.5 us against the dot2
// kernel's 11.9).
static bool lut_use_mfma(int M, int dtype) {
    static const int v = lut_env("BIE_LUT_MFMA", 1);
    static const int lo = lut_env("BIE_LUT_MFMA_MIN_M", 2);
    static const int hi = lut_env("BIE_LUT_MFMA_MAX_M", 16);
    static const int lo16 = lut_env("BIE_LUT_MFMA_MIN_M_F16", 2);
    return v != 0 && M >= (dtype == BIE_F16 ? lo16 : lo) && M <= hi && M <= 32;  // hi defaults to 16; 17 .. 32 (two row blocks) is an A/B arm: BIE_LUT_MFMA_MAX_M=32 with BIE_LUT_MAX_M=32
}
#include "mpq_lut_rb2_table.inc"

// 17 .. 32 rows of a LONE W4 call on the matrix-pipe decode kernel with two row blocks (RB = 2) instead of the fused MFMA GEMM + its finalize launch: round 6's
// fp16 arithmetic form made it the faster one on most 4096- and 8192-wide layers (4096x4096 fp16: 10.2 / 11.6 / 14.2 against 12.3 / 13.5 / 15.0 us at 17 / 24 / 32 rows;
// 4096x8192: 0.75-0.82 of the GEMM's time) and it is 1.2-1.7 x slower on narrow or very wide ones -- so the answer is the measurement, per dtype and exact shape
// (profiles/r06_lone_rb2_sweep.txt -> mpq_lut_rb2_table.inc); shapes outside the table keep the GEMM.  BIE_LUT_RB2=0: never.
bool mpq_lut_rb2_ok(int M, int K, int N, int dtype) {
    static const int on = lut_env("BIE_LUT_RB2", 1);
    if (!on || M <= 16 || M > 32 || (dtype != BIE_F16 && dtype != BIE_BF16)) return false;
    for (const auto& e : kLutRb2)
        if (e[0] == K && e[1] == N) return M <= e[dtype == BIE_F16 ? 2 : 3];
    return false;
}

// ... and SIBLING SETS of 17 .. 32 rows in one grouped launch of the same instance: q/k/v (3 x 4096x4096) 16.1 / 17.9 / 20.8 us against 31.3 / 35.7 / 42.8 for the members'
// lone calls in fp16 (bf16: 21.5 / 24.0 / 27.3 against 34.6 / 38.7 / 45.3), an 8192 -> 8192 + 1024 + 1024 set 0.55-0.73 of the lone calls' time; gate/up (2 x 4096x11008): fp16
// 0.79-0.80, bf16 1.01-1.15 -- so fp16 always, bf16 (table form) up to 16384 output columns in the set (profiles/r06_grouped_rb2_probe.txt).
bool mpq_lut_rb2_grouped_ok(int M, int K, long n_total, int dtype) {
    static const int on = lut_env("BIE_LUT_RB2", 1);
    (void)K;
    if (!on || M <= 16 || M > 32) return false;
    return dtype == BIE_F16 || (dtype == BIE_BF16 && n_total <= 16384);
}

bool mpq_gemv_lut_ok(int M, int K, int w_bit, int group_size, int dtype, bool has_gidx, int N) {
    static const int enabled = lut_env("BIE_GEMV_LUT", 1);
    static const int w2 = lut_env("BIE_GEMV_LUT_W2", 1);
    if (!enabled || has_gidx || M < 1) return false;
    if (dtype != BIE_BF16 && dtype != BIE_F16) return false;
    const int gs = group_size > K ? K : group_size;
    if (w_bit == 2) {  // pair lookup + dot2: fp16 and bf16, M <= 2, groups of 64 / 128 / 256 (RPG = gs / 16 >= 4)
        if (!w2 || M > 2) return false;
        if (gs != 64 && gs != 128 && gs != 256) return false;
        return K % gs == 0;
    }
    if (w_bit != 4) return false;
    if (M > 16 && !(N > 0 && mpq_lut_rb2_ok(M, K, N, dtype)) && !lut_use_mfma(M, dtype)) return false;  // 17 .. 32 rows: measured shapes only (or the A/B arm BIE_LUT_MFMA_MAX_M=32)
    if (M <= 16 && !lut_use_mfma(M, dtype) && M > 2) return false;  // the FMA form: M <= 2, bf16 and fp16 (v_fma_mix_f32)
    if (gs != 32 && gs != 64 && gs != 128 && gs != 256) return false;
    return K % gs == 0;
}

struct LutPlan {
    int rpg, G, gpw, nw, S, coop, H;  // rpg / G in UNITS (a group = H units) once H > 1
};

// A wave takes one unit = one group of 128 (16 packed rows); NW (8) waves per workgroup.  Small layers that would leave the
// chip short of waves split every group into H = 2 or 4 units (each wave then builds the group's table for 8 or 4 rows: the
// per-wave critical path, which is what a 4096x4096 launch spends its time on, shrinks accordingly); big grids give a wave
// several units (bounds the granule traffic).
static LutPlan lut_plan(int M, int dtype, int K, int group_size, int tiles_total, int w_bit = 4) {
    static const int min_rows = lut_env("BIE_LUT_ROWS", 16);
    static const int nw_env = lut_env("BIE_LUT_NW", 8);
    static const int max_wg = lut_env("BIE_LUT_MAX_WG", 2048);
    static const int coop = lut_env("BIE_LUT_COOP", 0);  // measured slower (12.1 vs 10.1 us at 4096x11008): kept as a tuning variant
    static const int force_h = lut_env("BIE_LUT_H", 0);
    static const int want_waves = lut_env("BIE_LUT_WANT_WAVES", 4096);
    LutPlan p;
    const int gs = group_size > K ? K : group_size;
    p.rpg = gs / (32 / w_bit);
    p.G = K / gs;
    const bool mfma = w_bit == 4 && (M > 16 || lut_use_mfma(M, dtype));
    p.coop = coop && !mfma && w_bit == 4 && dtype == BIE_BF16;
    p.H = 1;
    if (p.coop) {  // four waves per group, 128 / rpg groups per workgroup (mpq_gemv_lutc_kernel)
        p.nw = 4;
        p.gpw = 128 / p.rpg;
        p.S = cdiv(p.G, p.gpw);
        return p;
    }
    p.nw = (mfma || w_bit != 4 || dtype == BIE_F16) ? 8 : (nw_env == 4 ? 4 : (nw_env == 2 ? 2 : 8));
    int H = 1;
    if (force_h > 0) H = force_h;
    else
        while (H < 4 && p.rpg / (2 * H) >= 4 && (long)tiles_total * p.G * H < want_waves) H *= 2;
    while (H > 1 && (p.rpg % H || p.rpg / H < 4)) H /= 2;
    p.H = H;
    p.rpg /= H;
    p.G *= H;
    // W2: one 8-row group per wave (measured 6.9 us against 8.3 us at 4096x11008 with two groups per wave)
    int gpw = H > 1 ? 1 : cdiv(w_bit == 2 && min_rows > 8 ? 8 : min_rows, p.rpg);
    const int by_grid = (int)cdivl((long)tiles_total * p.G, (long)max_wg * p.nw);
    if (by_grid > gpw) gpw = by_grid;
    if (gpw > p.G) gpw = p.G;
    p.gpw = gpw;
    p.S = cdiv(p.G, gpw * p.nw);
    return p;
}

// granule area behind the workspace head, counted in floats (a granule = 8 bytes)
size_t mpq_gemv_lut_part_floats(int M, int K, int group_size, int tiles_total, int w_bit) {
    size_t need = 0;  // the dtype is not known where workspaces are sized: the larger of the two plans
    for (int dtype : {BIE_F16, BIE_BF16}) {
        const LutPlan p = lut_plan(M, dtype, K, group_size, tiles_total, w_bit);
        const size_t f = p.S > 1 ? (size_t)(p.S - 1) * M * tiles_total * 64 * 2 : 0;
        if (f > need) need = f;
    }
    const size_t inl = mpq_list_inline_part_floats(M, K, group_size, tiles_total, w_bit);
    return inl > need ? inl : need;
}

// fp16 W4 FMA form
template <int ZM, int MT>
static void lut4h_launch_rpg(const LutArgs& a, int rpg, int grid, hipStream_t st) {
    switch (rpg) {
        case 4: hipLaunchKernelGGL((mpq_gemv_lut_kernel<BIE_F16, ZM, MT, 4, 8, 0>), dim3(grid), dim3(512), 0, st, a); break;
        case 8: hipLaunchKernelGGL((mpq_gemv_lut_kernel<BIE_F16, ZM, MT, 8, 8, 0>), dim3(grid), dim3(512), 0, st, a); break;
        case 16: hipLaunchKernelGGL((mpq_gemv_lut_kernel<BIE_F16, ZM, MT, 16, 8, 0>), dim3(grid), dim3(512), 0, st, a); break;
        default: hipLaunchKernelGGL((mpq_gemv_lut_kernel<BIE_F16, ZM, MT, 32, 8, 0>), dim3(grid), dim3(512), 0, st, a); break;
    }
}
static void lut4h_launch(const LutArgs& a, int rpg, int grid, int M, int zm, hipStream_t st) {
    if (M == 1) {
        if (zm == ZM_ASYM) lut4h_launch_rpg<ZM_ASYM, 1>(a, rpg, grid, st);
        else if (zm == ZM_FUSED) lut4h_launch_rpg<ZM_FUSED, 1>(a, rpg, grid, st);
        else lut4h_launch_rpg<ZM_SYM, 1>(a, rpg, grid, st);
    } else {
        if (zm == ZM_ASYM) lut4h_launch_rpg<ZM_ASYM, 2>(a, rpg, grid, st);
        else if (zm == ZM_FUSED) lut4h_launch_rpg<ZM_FUSED, 2>(a, rpg, grid, st);
        else lut4h_launch_rpg<ZM_SYM, 2>(a, rpg, grid, st);
    }
}

// W2A16: pair-lookup instances (8-wave workgroups, no tuning variants)
template <int DT, int ZM, int MT>
static void lut2_launch_rpg(const LutArgs& a, int rpg, int grid, hipStream_t st) {
    switch (rpg) {
        case 4: hipLaunchKernelGGL((mpq_gemv_lut_kernel<DT, ZM, MT, 4, 8, 0, 0, 2>), dim3(grid), dim3(512), 0, st, a); break;
        case 8: hipLaunchKernelGGL((mpq_gemv_lut_kernel<DT, ZM, MT, 8, 8, 0, 0, 2>), dim3(grid), dim3(512), 0, st, a); break;
        default: hipLaunchKernelGGL((mpq_gemv_lut_kernel<DT, ZM, MT, 16, 8, 0, 0, 2>), dim3(grid), dim3(512), 0, st, a); break;
    }
}
template <int DT>
static void lut2_launch(const LutArgs& a, int rpg, int grid, int M, int zm, hipStream_t st) {
    if (M == 1) {
        if (zm == ZM_ASYM) lut2_launch_rpg<DT, ZM_ASYM, 1>(a, rpg, grid, st);
        else if (zm == ZM_FUSED) lut2_launch_rpg<DT, ZM_FUSED, 1>(a, rpg, grid, st);
        else lut2_launch_rpg<DT, ZM_SYM, 1>(a, rpg, grid, st);
    } else {
        if (zm == ZM_ASYM) lut2_launch_rpg<DT, ZM_ASYM, 2>(a, rpg, grid, st);
        else if (zm == ZM_FUSED) lut2_launch_rpg<DT, ZM_FUSED, 2>(a, rpg, grid, st);
        else lut2_launch_rpg<DT, ZM_SYM, 2>(a, rpg, grid, st);
    }
}

template <int DT, int ZM, int MT, int NW>
static void lut_launch_rpg(const LutArgs& a, int rpg, int grid, hipStream_t st) {
    static const int lab = lut_env("BIE_GEMV_LAB", 0);
    static const int rd = lut_env("BIE_LUT_RD", 0);
#define BIE_LUT(RPGV)                                                                                                       \
    do {                                                                                                                    \
        if (lab == 0 && rd > 0 && RPGV == 16 && MT == 1 && ZM == ZM_SYM) {                                                  \
            if (rd == 4) hipLaunchKernelGGL((mpq_gemv_lut_kernel<DT, ZM_SYM, 1, 16, NW, 0, 4>), dim3(grid), dim3(NW * 64), 0, st, a); \
            else if (rd == 6) hipLaunchKernelGGL((mpq_gemv_lut_kernel<DT, ZM_SYM, 1, 16, NW, 0, 6>), dim3(grid), dim3(NW * 64), 0, st, a); \
            else hipLaunchKernelGGL((mpq_gemv_lut_kernel<DT, ZM_SYM, 1, 16, NW, 0, 8>), dim3(grid), dim3(NW * 64), 0, st, a); \
        } else if (lab == 0 || RPGV != 16 || MT != 1 || ZM != ZM_SYM)                                                       \
            hipLaunchKernelGGL((mpq_gemv_lut_kernel<DT, ZM, MT, RPGV, NW, 0>), dim3(grid), dim3(NW * 64), 0, st, a);        \
        else if (lab == 2)                                                                                                  \
            hipLaunchKernelGGL((mpq_gemv_lut_kernel<DT, ZM_SYM, 1, 16, NW, 2>), dim3(grid), dim3(NW * 64), 0, st, a);       \
        else if (lab == 3)                                                                                                  \
            hipLaunchKernelGGL((mpq_gemv_lut_kernel<DT, ZM_SYM, 1, 16, NW, 3>), dim3(grid), dim3(NW * 64), 0, st, a);       \
        else if (lab == 5)                                                                                                  \
            hipLaunchKernelGGL((mpq_gemv_lut_kernel<DT, ZM_SYM, 1, 16, NW, 5>), dim3(grid), dim3(NW * 64), 0, st, a);       \
        else                                                                                                                \
            hipLaunchKernelGGL((mpq_gemv_lut_kernel<DT, ZM_SYM, 1, 16, NW, 4>), dim3(grid), dim3(NW * 64), 0, st, a);       \
    } while (0)
    switch (rpg) {
        case 4: BIE_LUT(4); break;
        case 8: BIE_LUT(8); break;
        case 16: BIE_LUT(16); break;
        default: BIE_LUT(32); break;
    }
#undef BIE_LUT
}

template <int NW>
static void lut_launch_nw(const LutArgs& a, int rpg, int grid, int M, int zm, hipStream_t st) {
    if (M == 1) {
        if (zm == ZM_ASYM) lut_launch_rpg<BIE_BF16, ZM_ASYM, 1, NW>(a, rpg, grid, st);
        else if (zm == ZM_FUSED) lut_launch_rpg<BIE_BF16, ZM_FUSED, 1, NW>(a, rpg, grid, st);
        else lut_launch_rpg<BIE_BF16, ZM_SYM, 1, NW>(a, rpg, grid, st);
    } else {
        if (zm == ZM_ASYM) lut_launch_rpg<BIE_BF16, ZM_ASYM, 2, NW>(a, rpg, grid, st);
        else if (zm == ZM_FUSED) lut_launch_rpg<BIE_BF16, ZM_FUSED, 2, NW>(a, rpg, grid, st);
        else lut_launch_rpg<BIE_BF16, ZM_SYM, 2, NW>(a, rpg, grid, st);
    }
}

template <int ZM, int MT>
static void lutc_launch_rpg(const LutArgs& a, int rpg, int grid, hipStream_t st) {
    static const int lab = lut_env("BIE_GEMV_LAB", 0);
#define BIE_LUTC(RPGV, LABV) hipLaunchKernelGGL((mpq_gemv_lutc_kernel<BIE_BF16, ZM, MT, RPGV, LABV>), dim3(grid), dim3(256), 0, st, a)
    if (lab != 0 && rpg == 16 && MT == 1 && ZM == ZM_SYM) {
        if (lab == 2) hipLaunchKernelGGL((mpq_gemv_lutc_kernel<BIE_BF16, ZM_SYM, 1, 16, 2>), dim3(grid), dim3(256), 0, st, a);
        else if (lab == 3) hipLaunchKernelGGL((mpq_gemv_lutc_kernel<BIE_BF16, ZM_SYM, 1, 16, 3>), dim3(grid), dim3(256), 0, st, a);
        else if (lab == 4) hipLaunchKernelGGL((mpq_gemv_lutc_kernel<BIE_BF16, ZM_SYM, 1, 16, 4>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((mpq_gemv_lutc_kernel<BIE_BF16, ZM_SYM, 1, 16, 5>), dim3(grid), dim3(256), 0, st, a);
        return;
    }
    switch (rpg) {
        case 4: BIE_LUTC(4, 0); break;
        case 8: BIE_LUTC(8, 0); break;
        case 16: BIE_LUTC(16, 0); break;
        default: BIE_LUTC(32, 0); break;
    }
#undef BIE_LUTC
}

static void lutc_launch(const LutArgs& a, int rpg, int grid, int M, int zm, hipStream_t st) {
    if (M == 1) {
        if (zm == ZM_ASYM) lutc_launch_rpg<ZM_ASYM, 1>(a, rpg, grid, st);
        else if (zm == ZM_FUSED) lutc_launch_rpg<ZM_FUSED, 1>(a, rpg, grid, st);
        else lutc_launch_rpg<ZM_SYM, 1>(a, rpg, grid, st);
    } else {
        if (zm == ZM_ASYM) lutc_launch_rpg<ZM_ASYM, 2>(a, rpg, grid, st);
        else if (zm == ZM_FUSED) lutc_launch_rpg<ZM_FUSED, 2>(a, rpg, grid, st);
        else lutc_launch_rpg<ZM_SYM, 2>(a, rpg, grid, st);
    }
}

template <int DT, int RB>
static void lutm_launch_rb(const LutArgs& a, int rpg, int grid, int zm, hipStream_t st) {
#define BIE_LUTM(ZMV)                                                                                                          \
    switch (rpg) {                                                                                                             \
        case 4: hipLaunchKernelGGL((mpq_gemv_lutm_kernel<DT, ZMV, 4, 8, RB>), dim3(grid), dim3(512), 0, st, a); break;         \
        case 8: hipLaunchKernelGGL((mpq_gemv_lutm_kernel<DT, ZMV, 8, 8, RB>), dim3(grid), dim3(512), 0, st, a); break;         \
        case 16: hipLaunchKernelGGL((mpq_gemv_lutm_kernel<DT, ZMV, 16, 8, RB>), dim3(grid), dim3(512), 0, st, a); break;       \
        default: hipLaunchKernelGGL((mpq_gemv_lutm_kernel<DT, ZMV, 32, 8, RB>), dim3(grid), dim3(512), 0, st, a); break;       \
    }
    if (zm == ZM_ASYM) { BIE_LUTM(ZM_ASYM) }
    else if (zm == ZM_FUSED) { BIE_LUTM(ZM_FUSED) }
    else { BIE_LUTM(ZM_SYM) }
#undef BIE_LUTM
}
// Per-layer launches stay at M <= 16: with two row blocks (RB = 2, one workgroup per CU) a lone layer measured 22.7-28.9 us at
// 4096x11008 for M = 17 .. 32 against 20.2-21.3 us for the MFMA GEMM -- the form pays only in list launches (9.4-10.5 us per layer).
template <int DT>
static void lutm_launch(const LutArgs& a, int rpg, int grid, int zm, hipStream_t st) {
    if (a.M > 16) lutm_launch_rb<DT, 2>(a, rpg, grid, zm, st);
    else lutm_launch_rb<DT, 1>(a, rpg, grid, zm, st);
}

// the list form of the matrix-pipe kernel (mpq_list.hip builds the entries and the block table)
template <int DT, bool PF, int RB, int NW = 8>
static void lutm_list_launch_dt(const ListEntry* ent, const uint2_t* blk, unsigned grid, int M, int rpg, int zm, unsigned epoch, unsigned* status,
                                unsigned skew, int spin, hipStream_t st) {
#define BIE_LUTML(ZMV)                                                                                                                               \
    switch (rpg) {                                                                                                                                   \
        case 4: hipLaunchKernelGGL((mpq_lutm_list_kernel<DT, ZMV, 4, NW, PF, RB>), dim3(grid), dim3(NW * 64), 0, st, ent, blk, M, epoch, status, skew, spin); break;   \
        case 8: hipLaunchKernelGGL((mpq_lutm_list_kernel<DT, ZMV, 8, NW, PF, RB>), dim3(grid), dim3(NW * 64), 0, st, ent, blk, M, epoch, status, skew, spin); break;   \
        case 16: hipLaunchKernelGGL((mpq_lutm_list_kernel<DT, ZMV, 16, NW, PF, RB>), dim3(grid), dim3(NW * 64), 0, st, ent, blk, M, epoch, status, skew, spin); break; \
        default: hipLaunchKernelGGL((mpq_lutm_list_kernel<DT, ZMV, 32, NW, PF, RB>), dim3(grid), dim3(NW * 64), 0, st, ent, blk, M, epoch, status, skew, spin); break; \
    }
    if (zm == ZM_ASYM) { BIE_LUTML(ZM_ASYM) }
    else { BIE_LUTML(ZM_SYM) }
#undef BIE_LUTML
}
// waves per workgroup of the list form (the plan in mpq_list.hip is made for this number): FOUR.  17 <= M <= 32 (two row blocks, ~132
// registers): three four-wave workgroups fit a CU (3 waves per SIMD, 96 KiB of tables) where ONE eight-wave workgroup does (2 per SIMD) --
// 10.3 -> 7.7 us per 4096x11008 layer at 32 rows.  M <= 16: same residency either way, but a wave walks twice the units (prologue, barriers and
// the workgroup reduction amortised): 6.48 -> 5.96 us at 16 rows (profiles/r05_lutm_list_nw_ab.txt).  BIE_LUTM_NW32 / BIE_LUTM_NW16 = 8: the old plan.
int mpq_lutm_list_nw(int M) {
    static const int nw32 = lut_env("BIE_LUTM_NW32", 4);
    static const int nw16 = lut_env("BIE_LUTM_NW16", 4);
    return M > 16 ? (nw32 == 8 ? 8 : 4) : (nw16 == 8 ? 8 : 4);
}
int mpq_lutm_list_launch(const ListEntry* ent, const uint2_t* blk, unsigned grid, int M, int rpg, int zm, int dtype, int nw, hipStream_t st) {
    unsigned skew;
    int spin;
    test_forge_get(&skew, &spin);
    const unsigned epoch = next_launch_epoch();
    // (the next-unit prefetch variant, PF = true, needs 139 registers: one 8-wave workgroup per CU instead of two -- measured 8.4 against
    //  6.7 us per 4096x11008 layer at M = 8, profiles/r03_z_lutm_list_ab.txt; not instantiated)
    // 17 <= M <= 32: two 16-row blocks of x per pass over the weights
    if (M > 16 && nw == 4) {
        if (dtype == BIE_F16) lutm_list_launch_dt<BIE_F16, false, 2, 4>(ent, blk, grid, M, rpg, zm, epoch, device_status_word(), skew, spin, st);
        else lutm_list_launch_dt<BIE_BF16, false, 2, 4>(ent, blk, grid, M, rpg, zm, epoch, device_status_word(), skew, spin, st);
    } else if (M > 16) {
        if (dtype == BIE_F16) lutm_list_launch_dt<BIE_F16, false, 2>(ent, blk, grid, M, rpg, zm, epoch, device_status_word(), skew, spin, st);
        else lutm_list_launch_dt<BIE_BF16, false, 2>(ent, blk, grid, M, rpg, zm, epoch, device_status_word(), skew, spin, st);
    } else if (nw == 4) {
        if (dtype == BIE_F16) lutm_list_launch_dt<BIE_F16, false, 1, 4>(ent, blk, grid, M, rpg, zm, epoch, device_status_word(), skew, spin, st);
        else lutm_list_launch_dt<BIE_BF16, false, 1, 4>(ent, blk, grid, M, rpg, zm, epoch, device_status_word(), skew, spin, st);
    } else if (dtype == BIE_F16) lutm_list_launch_dt<BIE_F16, false, 1>(ent, blk, grid, M, rpg, zm, epoch, device_status_word(), skew, spin, st);
    else lutm_list_launch_dt<BIE_BF16, false, 1>(ent, blk, grid, M, rpg, zm, epoch, device_status_word(), skew, spin, st);
    return check_launch("mpq_lutm_list_kernel");
}

// the x-sharing form (lutm_xs_body): workgroups of NW waves over NW adjacent tiles; the block table holds the first tile of each
template <int DT, int RB, int NW>
static void lutm_xs_list_launch_dt(const ListEntry* ent, const uint2_t* blk, unsigned grid, int M, int rpg, int zm, unsigned epoch, unsigned* status,
                                   unsigned skew, int spin, hipStream_t st) {
#define BIE_LUTMX(ZMV)                                                                                                                                \
    switch (rpg) {                                                                                                                                    \
        case 4: hipLaunchKernelGGL((mpq_lutm_xs_list_kernel<DT, ZMV, 4, RB, NW>), dim3(grid), dim3(NW * 64), 0, st, ent, blk, M, epoch, status, skew, spin); break;   \
        case 8: hipLaunchKernelGGL((mpq_lutm_xs_list_kernel<DT, ZMV, 8, RB, NW>), dim3(grid), dim3(NW * 64), 0, st, ent, blk, M, epoch, status, skew, spin); break;   \
        case 16: hipLaunchKernelGGL((mpq_lutm_xs_list_kernel<DT, ZMV, 16, RB, NW>), dim3(grid), dim3(NW * 64), 0, st, ent, blk, M, epoch, status, skew, spin); break; \
        default: hipLaunchKernelGGL((mpq_lutm_xs_list_kernel<DT, ZMV, 32, RB, NW>), dim3(grid), dim3(NW * 64), 0, st, ent, blk, M, epoch, status, skew, spin); break; \
    }
    if (zm == ZM_ASYM) { BIE_LUTMX(ZM_ASYM) }
    else { BIE_LUTMX(ZM_SYM) }
#undef BIE_LUTMX
}
int mpq_lutm_xs_list_launch(const ListEntry* ent, const uint2_t* blk, unsigned grid, int M, int rpg, int zm, int dtype, int nw, hipStream_t st) {
    unsigned skew;
    int spin;
    test_forge_get(&skew, &spin);
    const unsigned epoch = next_launch_epoch();
#define BIE_LUTMX_NW(NWV)                                                                                                                             \
    if (M > 16) {                                                                                                                                     \
        if (dtype == BIE_F16) lutm_xs_list_launch_dt<BIE_F16, 2, NWV>(ent, blk, grid, M, rpg, zm, epoch, device_status_word(), skew, spin, st);         \
        else lutm_xs_list_launch_dt<BIE_BF16, 2, NWV>(ent, blk, grid, M, rpg, zm, epoch, device_status_word(), skew, spin, st);                        \
    } else {                                                                                                                                          \
        if (dtype == BIE_F16) lutm_xs_list_launch_dt<BIE_F16, 1, NWV>(ent, blk, grid, M, rpg, zm, epoch, device_status_word(), skew, spin, st);         \
        else lutm_xs_list_launch_dt<BIE_BF16, 1, NWV>(ent, blk, grid, M, rpg, zm, epoch, device_status_word(), skew, spin, st);                        \
    }
    if (nw == 8) { BIE_LUTMX_NW(8) }
    else { BIE_LUTMX_NW(4) }
#undef BIE_LUTMX_NW
    return check_launch("mpq_lutm_xs_list_kernel");
}

// sets: n weight sets sharing x (one for a plain forward).  `gen` = the workspace's head, `gran` = granule area.
int mpq_gemv_lut_launch(int nsets, const int32_t* const* qw, const void* const* scales, const void* const* zeros,
                        const void* const* bias, void* const* y, const int* N, const void* x, unsigned* gen, float* gran,
                        int M, int K, int group_size, int zm, int dtype, hipStream_t st, int w_bit) {
    long n_total = 0;
    for (int i = 0; i < nsets; i++) n_total += N[i];
    if (mpq_list_inline_ok(M, K, n_total, w_bit, group_size, zm, dtype))  // big W4 / M = 1 / bf16 launches: the list kernel's D16 form, entries in the kernel arguments
        return mpq_list_inline_launch(nsets, qw, scales, zeros, bias, y, N, x, gen, gran, K, group_size, zm, st);
    LutArgs a;
    int tiles = 0;
    for (int i = 0; i < LUT_MAX_SETS; i++) {
        const int j = i < nsets ? i : nsets - 1;
        a.set[i].qw = reinterpret_cast<const uint32_t*>(qw[j]);
        a.set[i].scales = reinterpret_cast<const uint16_t*>(scales[j]);
        a.set[i].zeros = zeros[j];
        a.set[i].bias = bias ? reinterpret_cast<const uint16_t*>(bias[j]) : nullptr;  // entries may be NULL too
        a.set[i].y = reinterpret_cast<uint16_t*>(y[j]);
        a.set[i].N = N[j];
        a.set[i].tile_begin = tiles;
        if (i < nsets) tiles += cdiv(N[j], 64);
    }
    const LutPlan p = lut_plan(M, dtype, K, group_size, tiles, w_bit);
    a.x = reinterpret_cast<const uint16_t*>(x);
    a.gran = reinterpret_cast<unsigned long long*>(gran);
    a.gen = gen;
    a.nsets = nsets;
    a.M = M;
    a.K = K;
    a.G = p.G;
    a.tiles_total = tiles;
    a.S = p.S;
    a.groups_per_wave = p.gpw;
    a.hshift = p.H == 4 ? 2 : (p.H == 2 ? 1 : 0);
    a.epoch = next_launch_epoch();
    a.status = device_status_word();
    test_forge_get(&a.tag_skew, &a.spin_limit);
    const int grid = tiles * p.S;
    if (w_bit == 2) {
        if (dtype == BIE_F16) lut2_launch<BIE_F16>(a, p.rpg, grid, M, zm, st);
        else lut2_launch<BIE_BF16>(a, p.rpg, grid, M, zm, st);
        return check_launch("mpq_gemv_lut_kernel<W2>");
    }
    if ((M > 16 || lut_use_mfma(M, dtype)) && !p.coop) {
        if (dtype == BIE_F16) lutm_launch<BIE_F16>(a, p.rpg, grid, zm, st);
        else lutm_launch<BIE_BF16>(a, p.rpg, grid, zm, st);
        return check_launch("mpq_gemv_lutm_kernel");
    }
    if (dtype == BIE_F16) {  // fp16 FMA form: 8-wave workgroups, no tuning variants
        lut4h_launch(a, p.rpg, grid, M, zm, st);
        return check_launch("mpq_gemv_lut_kernel<f16>");
    }
    if (p.coop) lutc_launch(a, p.rpg, grid, M, zm, st);
    else if (p.nw == 4) lut_launch_nw<4>(a, p.rpg, grid, M, zm, st);
    else if (p.nw == 2) lut_launch_nw<2>(a, p.rpg, grid, M, zm, st);
    else lut_launch_nw<8>(a, p.rpg, grid, M, zm, st);
    return check_launch("mpq_gemv_lut_kernel");
}

}  // namespace bie
