// W{1,2,4,8}A16 decode GEMV / skinny GEMM (M <= 8) for gfx950 -- the bandwidth-bound half of
// bie_mpq_forward.  Replaces quant_mm_kernel[_asym] (reference
// layers/qlinear/nbit/cuda/mpq_linear_cuda_kernel.cu:67-451).
//
// Three kernels, every packed word read exactly once and coalesced:
//   * mpq_gemv3_kernel -- the decode path (no q_perm, whole 8-row batches inside one group).  One column per lane, a wave
//     owns 64 columns x a K range, 16 waves per block; dword non-temporal loads, up to three 8-row batches in flight; x is
//     wave-uniform and comes through scalar loads (no LDS staging, no barrier before the weight stream); dequantisation in
//     registers in 16-bit pairs (mpq_dequant.cuh) feeding v_dot2_f32_{f16,bf16} with fp32 accumulators; block reduction
//     through LDS.  K is also split over blockIdx.y (plan_gemv3: the split minimising the busiest CU's share of a column
//     block -- the dispatcher balances the small equal blocks dynamically); the last block to arrive at a column tile
//     (relaxed agent-scope ticket, write-through partials, per-wave vmcnt(0) drain, batched bypassing reads) sums the
//     slabs in slab order and writes y: deterministic, no finalize launch.
//   * mpq_gemv_kernel -- the earlier 4-columns-per-lane variant (dwordx4 loads, x slab staged in LDS, optionally gathered
//     through q_perm): MBWQ act-order and the shapes the v3 kernel does not take.
//   * mpq_gemv_generic_kernel -- any N / group size, explicit g_idx, fp32: correctness path (partials + splitk_finalize).
#include "mpq_dequant.cuh"
#include <stdlib.h>

#pragma clang fp contract(off)

namespace bie {

// mpq_gemv_lut.hip
bool mpq_gemv_lut_ok(int M, int K, int w_bit, int group_size, int dtype, bool has_gidx, int N = 0);  // N > 0: a lone call (17 .. 32 rows on measured shapes)
size_t mpq_gemv_lut_part_floats(int M, int K, int group_size, int tiles_total, int w_bit);
int mpq_gemv_lut_launch(int nsets, const int32_t* const* qw, const void* const* scales, const void* const* zeros,
                        const void* const* bias, void* const* y, const int* N, const void* x, unsigned* counters, float* part,
                        int M, int K, int group_size, int zm, int dtype, hipStream_t st, int w_bit);

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

constexpr int GEMV_THREADS = 256;
constexpr int GEMV_COLS = 256;  // columns per block: 64 lanes x 4
constexpr int GEMV_COUNTER_FLOATS = BIE_WS_COUNTERS;  // tickets: one arrival counter per column tile (first half of the head)
constexpr int GEMV_HEAD_FLOATS = (int)(BIE_WS_HEAD_BYTES / sizeof(float));  // the partial slabs start behind the WHOLE head

template <int DT, int WBIT, int MT, int ZM, int U>
__global__ __launch_bounds__(GEMV_THREADS) void mpq_gemv_kernel(
    const uint16_t* __restrict__ x, const uint32_t* __restrict__ qw, const uint16_t* __restrict__ scales,
    const void* __restrict__ zeros, const uint16_t* __restrict__ bias, const uint16_t* __restrict__ perm,
    float* __restrict__ part, unsigned* __restrict__ counters, uint16_t* __restrict__ y, int M, int K, int N,
    int group_size, int rows_per_slab, int R, int S, int lab) {
    constexpr int NB = 32 / WBIT;
    constexpr int NP = NB / 2;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * GEMV_COLS + lane * 4;
    const bool n_ok = n0 < N;  // N % 4 == 0 is guaranteed by the launcher
    const int slab = blockIdx.y;
    const int r_begin = slab * rows_per_slab;
    const int slab_k = rows_per_slab * NB;

    const int rpw = rows_per_slab / 4;
    const int rw_begin = r_begin + wave * rpw;
    int rw_end = rw_begin + rpw;
    if (rw_end > R) rw_end = R;

    // ---- weights first: the first U row loads are in flight while x is staged (nothing below depends on LDS yet)
    uint4_t wq[U];
    auto load_rows = [&](uint4_t (&dst)[U], int r) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (n_ok && r + u < rw_end) {
                const uint4_t* p = reinterpret_cast<const uint4_t*>(qw + (long)(r + u) * N + n0);
                dst[u] = __builtin_nontemporal_load(p);
            }
        }
    };
    load_rows(wq, rw_begin);

    // ---- stage the x slab into LDS in pair order (16-bit elements) -----------------------------
    {
        uint16_t* xs16 = reinterpret_cast<uint16_t*>(smem);
        const int total = MT * slab_k;
        for (int idx = tid; idx < total; idx += GEMV_THREADS) {
            const int m = idx / slab_k;
            const int kk = idx - m * slab_k;
            const int row = kk / NB, p = kk % NB;
            const int k = (r_begin + row) * NB + pair_src_k<DT, WBIT>(p);
            uint16_t v = 0;
            if (m < M && k < K) v = x[(long)m * K + (perm ? (int)perm[k] : k)];
            xs16[idx] = v;
        }
    }
    __syncthreads();

    float acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[m][c] = 0.0f;

    if (n_ok) {
        ColParams<DT, ZM> cp[4];
        int g_prev = -1;
        const int zero_width = N / NB;
        for (int r = rw_begin; r < rw_end; r += U) {
            uint4_t wn[U];
            if (r + U < rw_end) load_rows(wn, r + U);  // prefetch the next step
            const int g = (r * NB) / group_size;
            if (g != g_prev) {
                g_prev = g;
                const uint2_t sv = *reinterpret_cast<const uint2_t*>(scales + (long)g * N + n0);
                const uint32_t sb[4] = {sv.x & 0xffffu, sv.x >> 16, sv.y & 0xffffu, sv.y >> 16};
                if constexpr (ZM == ZM_ASYM) {
                    const uint32_t zw = reinterpret_cast<const uint32_t*>(zeros)[(long)g * zero_width + n0 / NB];
                    constexpr uint32_t M1 = (WBIT == 32) ? 0xffffffffu : ((1u << WBIT) - 1u);
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const uint32_t zq1 = ((zw >> (((n0 % NB) + c) * WBIT)) & M1) + 1u;
                        cp[c] = make_col_params<DT, WBIT, ZM>(sb[c], zq1);
                    }
                } else {
                    const uint2_t zv = *reinterpret_cast<const uint2_t*>(reinterpret_cast<const uint16_t*>(zeros) + (long)g * N + n0);
                    const uint32_t zb[4] = {zv.x & 0xffffu, zv.x >> 16, zv.y & 0xffffu, zv.y >> 16};
#pragma unroll
                    for (int c = 0; c < 4; c++) cp[c] = make_col_params<DT, WBIT, ZM>(sb[c], zb[c]);
                }
            }
            if (lab) {
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (r + u < rw_end) acc[0][0] += __uint_as_float((wq[u].x ^ wq[u].y ^ wq[u].z ^ wq[u].w) & 0x3f7fffffu);
            } else
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (r + u < rw_end) {
                    const int row_local = r + u - r_begin;
                    uint32_t xp[MT][NP];
#pragma unroll
                    for (int m = 0; m < MT; m++) {
                        const uint32_t* src = smem + ((m * slab_k) >> 1) + row_local * NP;
#pragma unroll
                        for (int i = 0; i < NP; i++) xp[m][i] = src[i];
                    }
                    const uint32_t words[4] = {wq[u].x, wq[u].y, wq[u].z, wq[u].w};
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        uint32_t wp[NP];
                        dequant_word<DT, WBIT, ZM>(words[c], cp[c], wp);
#pragma unroll
                        for (int i = 0; i < NP; i++)
#pragma unroll
                            for (int m = 0; m < MT; m++) acc[m][c] = dot2_acc<DT>(wp[i], xp[m][i], acc[m][c]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) wq[u] = wn[u];
        }
    }

    // ---- reduce the 4 waves through LDS, fixed order ----------------------------------------------
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int m = 0; m < MT; m++) {
        float4_t v = {acc[m][0], acc[m][1], acc[m][2], acc[m][3]};
        *reinterpret_cast<float4_t*>(red + (wave * MT + m) * GEMV_COLS + lane * 4) = v;
    }
    __syncthreads();
    const int n = blockIdx.x * GEMV_COLS + tid;
    float tot[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) {
        float v = red[(0 * MT + m) * GEMV_COLS + tid];
        v += red[(1 * MT + m) * GEMV_COLS + tid];
        v += red[(2 * MT + m) * GEMV_COLS + tid];
        v += red[(3 * MT + m) * GEMV_COLS + tid];
        tot[m] = v;
    }
    if (S == 1) {
        if (n < N) {
#pragma unroll
            for (int m = 0; m < MT; m++)
                if (m < M) {
                    float o = dt_traits<DT>::round(tot[m]);
                    if (bias) o = o + dt_traits<DT>::load(bias, n);
                    dt_traits<DT>::store(y, (long)m * N + n, o);
                }
        }
        return;
    }
    // ---- split-K: publish this slab's partial write-through (sc1), take a ticket; the LAST arriver of the column tile
    // sums the S slabs in slab order (deterministic) and writes y.  Placement-independent hand-off (agent scope):
    // sc1 stores + per-wave vmcnt(0) drain + barrier + one relaxed agent ticket; the reducer reads with sc1 loads.
    if (n < N) {
#pragma unroll
        for (int m = 0; m < MT; m++)
            if (m < M) __hip_atomic_store(part + ((long)slab * M + m) * N + n, tot[m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // also: every wave is done reading `red`
    if (tid == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(counters + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        smem[0] = ticket;
    }
    __syncthreads();
    if (smem[0] != (unsigned)(S - 1)) return;
    if (tid == 0) __hip_atomic_store(counters + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next call
    if (n < N) {
#pragma unroll
        for (int m = 0; m < MT; m++) {
            if (m < M) {
                // 16 independent sc1 loads in flight per batch (a dependent load-add chain costs one memory round
                // trip PER slab), summed in slab order
                float v = 0.0f;
                for (int s0 = 0; s0 < S; s0 += 16) {
                    float t[16];
#pragma unroll
                    for (int jj = 0; jj < 16; jj++) {
                        const int sidx = (s0 + jj < S) ? s0 + jj : S - 1;
                        t[jj] = __hip_atomic_load(part + ((long)sidx * M + m) * N + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int jj = 0; jj < 16; jj++)
                        if (s0 + jj < S) v += t[jj];
                }
                float o = dt_traits<DT>::round(v);
                if (bias) o = o + dt_traits<DT>::load(bias, n);
                dt_traits<DT>::store(y, (long)m * N + n, o);
            }
        }
    }
}

// =====================================================================================================
// v3 decode kernel: one COLUMN per lane, a wave owns 64 columns x a contiguous K range.
//   * the activations of a packed row are wave-uniform -> the compiler fetches them with scalar loads
//     (s_load_dwordx4, scalar cache) and feeds v_dot2 from SGPRs: no LDS staging, no barrier in front of
//     the weight stream;
//   * each lane streams dwords (a wave-row = 256 contiguous bytes), 16 loads in flight per lane, double
//     buffered; the block's NW waves split K, are reduced through LDS, and when the grid would be too
//     small the K range is additionally split over blockIdx.y with the ticketed (sc1) last-arriver sum.
// =====================================================================================================
constexpr int G3_U = 8;  // rows per batch; two batches (16 row loads) in flight per lane

template <int DT, int WBIT>
__device__ __forceinline__ void load_x_pairs(const uint16_t* __restrict__ xrow, uint32_t (&xp)[16 / WBIT]) {
    // xrow points at the NB activations of one packed row (wave-uniform address -> scalar loads); pair order
    constexpr int NB = 32 / WBIT, NP = NB / 2;
    const uint32_t* xd = reinterpret_cast<const uint32_t*>(xrow);
    uint32_t d[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) d[i] = xd[i];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const int ka = pair_src_k<DT, WBIT>(2 * i), kb = pair_src_k<DT, WBIT>(2 * i + 1);
        const uint32_t a = (ka & 1) ? (d[ka >> 1] >> 16) : (d[ka >> 1] & 0xffffu);
        const uint32_t b = (kb & 1) ? (d[kb >> 1] & 0xffff0000u) : (d[kb >> 1] << 16);
        xp[i] = a | b;
    }
}

// Requirements (checked by the launcher): R % 8 == 0, rows_per_wave % 8 == 0, rows-per-group % 8 == 0, so that a
// batch of 8 rows is all-or-nothing and lies inside one quantisation group.
template <int DT, int WBIT, int MT, int ZM, int NW>
__global__ __launch_bounds__(NW * 64) void mpq_gemv3_kernel(
    const uint16_t* __restrict__ x, const uint32_t* __restrict__ qw, const uint16_t* __restrict__ scales,
    const void* __restrict__ zeros, const uint16_t* __restrict__ bias, float* __restrict__ part,
    unsigned* __restrict__ counters, uint16_t* __restrict__ y, int M, int K, int N, int rows_per_group, int rows_per_wave,
    int R, int S, int lab) {
    constexpr int NB = 32 / WBIT;
    constexpr int NP = NB / 2;
    constexpr int U = G3_U;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x * 64 + lane;
    const int nl = n < N ? n : N - 1;  // clamp: out-of-range lanes load valid memory and are never stored
    const int slab = blockIdx.y;
    const int rb = (slab * NW + wave) * rows_per_wave;
    int re = rb + rows_per_wave;
    if (re > R) re = R;

    float acc[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m] = 0.0f;

    const uint32_t* wcol = qw + nl;
    auto load_batch = [&](uint32_t (&dst)[U], int r) {
#pragma unroll
        for (int u = 0; u < U; u++) dst[u] = __builtin_nontemporal_load(wcol + (long)(r + u) * N);
    };
    const int zero_width = N / NB;
    auto load_params = [&](int g, uint32_t& sb, uint32_t& zb) {
        sb = scales[(long)g * N + nl];
        if constexpr (ZM == ZM_ASYM) {
            constexpr uint32_t M1 = (WBIT == 32) ? 0xffffffffu : ((1u << WBIT) - 1u);
            const uint32_t zw = reinterpret_cast<const uint32_t*>(zeros)[(long)g * zero_width + nl / NB];
            zb = ((zw >> ((nl % NB) * WBIT)) & M1) + 1u;
        } else {
            zb = reinterpret_cast<const uint16_t*>(zeros)[(long)g * N + nl];
        }
    };

    ColParams<DT, ZM> cp;
    uint32_t sb_next = 0, zb_next = 0;
    int g_cur = -1;
    int switch_row = rb;  // first row of the next group
    const int g_last = (R - 1) / rows_per_group;
    auto compute_batch = [&](const uint32_t (&w)[U], int r) {
        if (r >= switch_row) {  // wave-uniform; a batch never straddles a group
            const int g = r / rows_per_group;
            uint32_t sb, zb;
            if (g == g_cur + 1 && g_cur >= 0) { sb = sb_next; zb = zb_next; }
            else load_params(g, sb, zb);
            cp = make_col_params<DT, WBIT, ZM>(sb, zb);
            g_cur = g;
            switch_row = (g + 1) * rows_per_group;
            if (g < g_last) load_params(g + 1, sb_next, zb_next);  // prefetch the next group's constants
        }
        if (lab) {  // tuning aid (BIE_GEMV_LAB=1): stream only, no dequant / dot -- measures the memory side alone
#pragma unroll
            for (int u = 0; u < U; u++) acc[0] += __uint_as_float(w[u] & 0x3f7fffffu);
            return;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint32_t wp[NP];
            dequant_word<DT, WBIT, ZM>(w[u], cp, wp);
#pragma unroll
            for (int m = 0; m < MT; m++) {
                if (m < M) {
                    uint32_t xp[NP];
                    load_x_pairs<DT, WBIT>(x + (long)m * K + (long)(r + u) * NB, xp);
#pragma unroll
                    for (int i = 0; i < NP; i++) acc[m] = dot2_acc<DT>(wp[i], xp[i], acc[m]);
                }
            }
        }
    };

    if (rb < re) {
        // three batches (24 row loads, 6 KiB per wave) in flight: the HBM stream keeps running while a batch is
        // dequantised
        uint32_t wa[U], wb[U], wc[U];
        load_batch(wa, rb);
        if (rb + U < re) load_batch(wb, rb + U);
        for (int r = rb; r < re; r += 3 * U) {
            if (r + 2 * U < re) load_batch(wc, r + 2 * U);
            compute_batch(wa, r);
            if (r + 3 * U < re) load_batch(wa, r + 3 * U);
            if (r + U < re) compute_batch(wb, r + U);
            if (r + 4 * U < re) load_batch(wb, r + 4 * U);
            if (r + 2 * U < re) compute_batch(wc, r + 2 * U);
        }
    }

    // ---- block reduction over the NW waves (fixed order) ----------------------------------------------------
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int m = 0; m < MT; m++) red[(wave * MT + m) * 64 + lane] = acc[m];
    __syncthreads();
    // thread t < 64*MT handles (m = t / 64, column = t % 64)
    const int om = tid >> 6, ol = tid & 63;
    const int on = blockIdx.x * 64 + ol;
    const bool owner = (tid < 64 * MT) && (om < M) && (on < N);
    float tot = 0.0f;
    if (tid < 64 * MT) {
#pragma unroll
        for (int w = 0; w < NW; w++) tot += red[(w * MT + om) * 64 + ol];
    }
    if (S == 1) {
        if (owner) {
            float o = dt_traits<DT>::round(tot);
            if (bias) o = o + dt_traits<DT>::load(bias, on);
            dt_traits<DT>::store(y, (long)om * N + on, o);
        }
        return;
    }
    if (owner) __hip_atomic_store(part + ((long)slab * M + om) * N + on, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(counters + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        smem[0] = ticket;
    }
    __syncthreads();
    if (smem[0] != (unsigned)(S - 1)) return;
    if (tid == 0) __hip_atomic_store(counters + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (owner) {
        float v = 0.0f;
        for (int s0 = 0; s0 < S; s0 += 8) {
            float t[8];
#pragma unroll
            for (int jj = 0; jj < 8; jj++) {
                const int sidx = (s0 + jj < S) ? s0 + jj : S - 1;
                t[jj] = __hip_atomic_load(part + ((long)sidx * M + om) * N + on, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int jj = 0; jj < 8; jj++)
                if (s0 + jj < S) v += t[jj];
        }
        float o = dt_traits<DT>::round(v);
        if (bias) o = o + dt_traits<DT>::load(bias, on);
        dt_traits<DT>::store(y, (long)om * N + on, o);
    }
}

struct Gemv3Plan {
    int NW, S, rows_per_wave;
};

// S (and with it the workspace) depends on (K, N, w_bit) only.
static Gemv3Plan plan_gemv3(int K, int N, int w_bit) {
    static const int nw_env = env_int("BIE_GEMV3_NW", 16);
    static const int force_s = env_int("BIE_GEMV3_S", 0);  // tuning knob
    const int NB = 32 / w_bit;
    const int R = K / NB;
    Gemv3Plan p;
    p.NW = nw_env == 8 ? 8 : 16;
    const int tiles = cdiv(N, 64);
    // Split K over blockIdx.y so that the busiest CU gets the smallest share of a column block: with `tiles * S` blocks on
    // 256 CUs that share is ceil(tiles * S / 256) / S (N = 11008: 172 tiles -> 1.0 unsplit, 0.75 at S = 4; N = 4096:
    // 64 tiles -> 0.25 at S = 4).  The ticketed in-kernel reduction costs about a microsecond, modelled as a flat penalty
    // (measured, tools/gemv_split_n11008.sh / gemv_small_n.sh: 4096x11008 14.8 -> 13.0 us, 4096x4096 9.6 -> 7.8 us,
    // 11008x4096 16.3 -> 14.2 us).  A wave needs at least one batch of 8 packed rows.
    int S = 1;
    if (force_s > 0) {
        S = force_s;
    } else {
        double best = 1e30;
        for (int s = 1; s <= 16; s++) {
            if (s > 1 && R / (8 * p.NW * s) < 1) break;
            const double share = (double)cdiv(tiles * s, 256) / s;
            const double cost = share + (s > 1 ? 0.08 + 0.005 * s : 0.0);
            if (cost < best - 1e-9) { best = cost; S = s; }
        }
    }
    int rpw = cdiv(cdiv(R, p.NW * S), 8) * 8;  // whole batches of 8 rows
    if (rpw < 8) rpw = 8;
    p.rows_per_wave = rpw;
    p.S = cdiv(R, rpw * p.NW);
    return p;
}

template <int DT, int WBIT, int MT, int ZM>
static int launch_gemv3(const Gemv3Plan& pl, const void* x, const int32_t* qw, const void* scales, const void* zeros,
                        const void* bias, float* ws, void* y, int M, int K, int N, int group_size, hipStream_t st) {
    constexpr int NB = 32 / WBIT;
    const int R = K / NB;
    dim3 grid(cdiv(N, 64), pl.S);
    float* part = ws + GEMV_HEAD_FLOATS;
    unsigned* counters = reinterpret_cast<unsigned*>(ws);
    static const int lab = env_int("BIE_GEMV_LAB", 0);
#define BIE_G3(NWV)                                                                                                    \
    hipLaunchKernelGGL((mpq_gemv3_kernel<DT, WBIT, MT, ZM, NWV>), grid, dim3(NWV * 64), (size_t)NWV * MT * 64 * sizeof(float), \
                       st, (const uint16_t*)x, (const uint32_t*)qw, (const uint16_t*)scales, zeros, (const uint16_t*)bias,  \
                       part, counters, (uint16_t*)y, M, K, N, (group_size > K ? K : group_size) / NB, pl.rows_per_wave, R, pl.S, lab)
    if (pl.NW == 8) BIE_G3(8); else BIE_G3(16);
#undef BIE_G3
    return check_launch("mpq_gemv3_kernel");
}

template <int DT, int WBIT, int ZM>
static int launch_gemv3_m(const Gemv3Plan& pl, int MT, const void* x, const int32_t* qw, const void* scales, const void* zeros,
                          const void* bias, float* ws, void* y, int M, int K, int N, int group_size, hipStream_t st) {
    switch (MT) {
        case 1: return launch_gemv3<DT, WBIT, 1, ZM>(pl, x, qw, scales, zeros, bias, ws, y, M, K, N, group_size, st);
        case 2: return launch_gemv3<DT, WBIT, 2, ZM>(pl, x, qw, scales, zeros, bias, ws, y, M, K, N, group_size, st);
        case 4: return launch_gemv3<DT, WBIT, 4, ZM>(pl, x, qw, scales, zeros, bias, ws, y, M, K, N, group_size, st);
        default: return launch_gemv3<DT, WBIT, 8, ZM>(pl, x, qw, scales, zeros, bias, ws, y, M, K, N, group_size, st);
    }
}

template <int DT, int WBIT>
static int launch_gemv3_z(const Gemv3Plan& pl, int MT, int zm, const void* x, const int32_t* qw, const void* scales,
                          const void* zeros, const void* bias, float* ws, void* y, int M, int K, int N, int group_size,
                          hipStream_t st) {
    if (zm == ZM_ASYM) return launch_gemv3_m<DT, WBIT, ZM_ASYM>(pl, MT, x, qw, scales, zeros, bias, ws, y, M, K, N, group_size, st);
    if (zm == ZM_FUSED) return launch_gemv3_m<DT, WBIT, ZM_FUSED>(pl, MT, x, qw, scales, zeros, bias, ws, y, M, K, N, group_size, st);
    return launch_gemv3_m<DT, WBIT, ZM_SYM>(pl, MT, x, qw, scales, zeros, bias, ws, y, M, K, N, group_size, st);
}

template <int DT>
static int launch_gemv3_w(const Gemv3Plan& pl, int MT, int w_bit, int zm, const void* x, const int32_t* qw, const void* scales,
                          const void* zeros, const void* bias, float* ws, void* y, int M, int K, int N, int group_size,
                          hipStream_t st) {
    switch (w_bit) {
        case 1: return launch_gemv3_z<DT, 1>(pl, MT, zm, x, qw, scales, zeros, bias, ws, y, M, K, N, group_size, st);
        case 2: return launch_gemv3_z<DT, 2>(pl, MT, zm, x, qw, scales, zeros, bias, ws, y, M, K, N, group_size, st);
        case 4: return launch_gemv3_z<DT, 4>(pl, MT, zm, x, qw, scales, zeros, bias, ws, y, M, K, N, group_size, st);
        default: return launch_gemv3_z<DT, 8>(pl, MT, zm, x, qw, scales, zeros, bias, ws, y, M, K, N, group_size, st);
    }
}

// ---- generic fallback: any N, any group size, explicit g_idx (act-order), fp32 too ---------------
// One column per lane, one slab of k per block.y, scalar dequant.  Correctness path, not a fast path.
template <int DT>
__global__ __launch_bounds__(256) void mpq_gemv_generic_kernel(
    const void* __restrict__ x, const uint32_t* __restrict__ qw, const void* __restrict__ scales,
    const void* __restrict__ zeros, const int32_t* __restrict__ g_idx, float* __restrict__ part, int M, int K,
    int N, int w_bit, int group_size, int asym, int k_per_slab, const uint16_t* __restrict__ perm) {  // asym: 0 / 1, or 2 = the uniform MBWQ rounding fl(q * s - z) in one step; perm: x is read at perm[k] (MBWQ act-order)
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int m = blockIdx.z;
    const int k_begin = blockIdx.y * k_per_slab;
    int k_end = k_begin + k_per_slab;
    if (k_end > K) k_end = K;
    if (n >= N) return;
    const int nb = 32 / w_bit;
    const uint32_t mask = (w_bit == 32) ? 0xffffffffu : ((1u << w_bit) - 1u);
    const int zero_width = N / nb;
    float acc = 0.0f;
    for (int k = k_begin; k < k_end; k++) {
        const int g = g_idx ? g_idx[k] : (k / group_size);
        const uint32_t word = qw[(long)(k / nb) * N + n];
        const uint32_t q = (word >> ((k % nb) * w_bit)) & mask;
        const float s = dt_traits<DT>::load(scales, (long)g * N + n);
        float w;
        if (asym == 1) {
            const uint32_t zw = reinterpret_cast<const uint32_t*>(zeros)[(long)g * zero_width + n / nb];
            const int zq1 = (int)((zw >> ((n % nb) * w_bit)) & mask) + 1;
            w = dequant_scalar_asym<DT>(q, s, zq1);
        } else if (asym == 2) {
            w = dt_traits<DT>::round(__builtin_fmaf((float)q, s, -dt_traits<DT>::load(zeros, (long)g * N + n)));
        } else {
            w = dequant_scalar_sym<DT>(q, s, dt_traits<DT>::load(zeros, (long)g * N + n));
        }
        acc = __builtin_fmaf(w, dt_traits<DT>::load(x, (long)m * K + (perm ? (int)perm[k] : k)), acc);
    }
    part[((long)blockIdx.y * M + m) * N + n] = acc;
}

// ---- launchers -------------------------------------------------------------------------------------
struct GemvPlan {
    int U, rows_per_slab, S;
};

// rows_per_slab is a multiple of 32 (4 waves x 8 rows) so that it does not depend on the group size;
// S (and with it the workspace size) is a function of (K, N, w_bit, MT) only.
static GemvPlan plan_gemv(int K, int N, int w_bit, int group_size, int MT) {
    const int NB = 32 / w_bit;
    const int R = K / NB;  // packed rows
    const int rows_per_group = (group_size > K ? K : group_size) / NB;
    int U = 8;
    while (U > 1 && (rows_per_group % U) != 0) U >>= 1;
    const int unit = 32;
    const int tiles_n = cdiv(N, GEMV_COLS);
    static const int target = []() { const char* e = getenv("BIE_GEMV_TARGET_BLOCKS"); return e ? atoi(e) : 1024; }();
    int S = cdiv(target, tiles_n);  // aim at ~4 blocks per CU
    int max_rows = (32768 / (2 * MT)) / NB;  // x slab <= 32 KiB of LDS
    max_rows = (max_rows / unit) * unit;
    int rows = cdiv(cdiv(R, S), unit) * unit;
    if (rows < unit) rows = unit;
    if (rows > max_rows) rows = max_rows;
    S = cdiv(R, rows);
    return GemvPlan{U, rows, S};
}

template <int DT, int WBIT, int MT, int ZM>
static int launch_gemv_u(const GemvPlan& pl, const void* x, const int32_t* qw, const void* scales, const void* zeros,
                         const void* bias, const uint16_t* perm, float* part, void* y, int M, int K, int N, int group_size, hipStream_t st) {
    constexpr int NB = 32 / WBIT;
    const int R = K / NB;
    dim3 grid(cdiv(N, GEMV_COLS), pl.S);
    size_t lds = (size_t)MT * pl.rows_per_slab * NB * 2;
    const size_t red = (size_t)4 * MT * GEMV_COLS * sizeof(float);
    if (lds < red) lds = red;
    static const int lab = env_int("BIE_GEMV_LAB", 0);
#define BIE_GEMV_LAUNCH(UU)                                                                                      \
    hipLaunchKernelGGL((mpq_gemv_kernel<DT, WBIT, MT, ZM, UU>), grid, dim3(GEMV_THREADS), lds, st,              \
                       (const uint16_t*)x, (const uint32_t*)qw, (const uint16_t*)scales, zeros, (const uint16_t*)bias, perm, \
                       part + GEMV_HEAD_FLOATS, reinterpret_cast<unsigned*>(part), (uint16_t*)y, M, K, N, group_size, pl.rows_per_slab, R, pl.S, lab)
    switch (pl.U) {
        case 8: BIE_GEMV_LAUNCH(8); break;
        case 4: BIE_GEMV_LAUNCH(4); break;
        case 2: BIE_GEMV_LAUNCH(2); break;
        default: BIE_GEMV_LAUNCH(1); break;
    }
#undef BIE_GEMV_LAUNCH
    return check_launch("mpq_gemv_kernel");
}

template <int DT, int WBIT, int ZM>
static int launch_gemv_m(const GemvPlan& pl, int MT, const void* x, const int32_t* qw, const void* scales,
                         const void* zeros, const void* bias, const uint16_t* perm, float* part, void* y, int M, int K, int N, int group_size,
                         hipStream_t st) {
    switch (MT) {
        case 1: return launch_gemv_u<DT, WBIT, 1, ZM>(pl, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
        case 2: return launch_gemv_u<DT, WBIT, 2, ZM>(pl, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
        case 4: return launch_gemv_u<DT, WBIT, 4, ZM>(pl, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
        default: return launch_gemv_u<DT, WBIT, 8, ZM>(pl, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
    }
}

template <int DT, int WBIT>
static int launch_gemv_a(const GemvPlan& pl, int MT, int zm, const void* x, const int32_t* qw, const void* scales,
                         const void* zeros, const void* bias, const uint16_t* perm, float* part, void* y, int M, int K, int N, int group_size,
                         hipStream_t st) {
    if (zm == ZM_ASYM) return launch_gemv_m<DT, WBIT, ZM_ASYM>(pl, MT, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
    if (zm == ZM_FUSED) return launch_gemv_m<DT, WBIT, ZM_FUSED>(pl, MT, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
    return launch_gemv_m<DT, WBIT, ZM_SYM>(pl, MT, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
}

template <int DT>
static int launch_gemv_w(const GemvPlan& pl, int MT, int w_bit, int zm, const void* x, const int32_t* qw,
                         const void* scales, const void* zeros, const void* bias, const uint16_t* perm, float* part, void* y, int M, int K,
                         int N, int group_size, hipStream_t st) {
    switch (w_bit) {
        case 1: return launch_gemv_a<DT, 1>(pl, MT, zm, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
        case 2: return launch_gemv_a<DT, 2>(pl, MT, zm, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
        case 4: return launch_gemv_a<DT, 4>(pl, MT, zm, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
        default: return launch_gemv_a<DT, 8>(pl, MT, zm, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
    }
}

// Fast path eligibility: fp16/bf16, implicit groups, group rows align with packed rows, N % 4 == 0.
bool mpq_gemv_fast_ok(int M, int K, int N, int w_bit, int group_size, int dtype, bool has_gidx) {
    if (dtype != BIE_F16 && dtype != BIE_BF16) return false;
    if (has_gidx) return false;
    if (M > 8) return false;
    const int NB = 32 / w_bit;
    const int gs = group_size > K ? K : group_size;
    if (K % NB || gs % NB || (N & 3)) return false;
    if (cdiv(N, 64) > GEMV_COUNTER_FLOATS) return false;  // one ticket per 64-column tile (v3 / LUT kernels)
    return true;
}

size_t mpq_gemv_workspace_bytes(int M, int K, int N, int w_bit) {
    size_t lut = 0;  // the group size is not known here: take the largest slab count any supported group size gives
    if ((w_bit == 4 && M <= 32) || (w_bit == 2 && M <= 2))  // W4: 17 .. 32 rows may take the matrix-pipe decode kernel with two row blocks (mpq_lut_rb2_ok)
        for (int gs = 32; gs <= 256; gs *= 2)
            if (K % gs == 0) {
                const size_t f = mpq_gemv_lut_part_floats(M, K, gs, cdiv(N, 64), w_bit);
                if (f > lut) lut = f;
            }
    lut = lut ? lut * sizeof(float) + BIE_WS_HEAD_BYTES : 0;
    const int MT = M <= 1 ? 1 : (M <= 2 ? 2 : (M <= 4 ? 4 : 8));
    const GemvPlan pl = plan_gemv(K, N, w_bit, K, MT);
    size_t fast = pl.S > 1 ? (size_t)pl.S * M * N * sizeof(float) + BIE_WS_HEAD_BYTES : 0;
    const Gemv3Plan p3 = plan_gemv3(K, N, w_bit);
    const size_t fast3 = p3.S > 1 ? (size_t)p3.S * M * N * sizeof(float) + BIE_WS_HEAD_BYTES : 0;
    if (fast3 > fast) fast = fast3;
    if (lut > fast) fast = lut;
    const size_t generic = (size_t)cdiv(K, 512) * M * N * sizeof(float);
    return fast > generic ? fast : generic;
}

int mpq_gemv_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const void* bias, void* y,
                    float* part, int M, int K, int N, int w_bit, int group_size, int zm, int dtype, const uint16_t* perm,
                    hipStream_t st) {
    if (perm == nullptr && cdiv(N, 64) <= BIE_WS_COUNTERS && mpq_gemv_lut_ok(M, K, w_bit, group_size, dtype, false, N)) {  // W4: table-lookup / matrix-pipe decode kernels
        const void* sc1[1] = {scales};
        const void* ze1[1] = {zeros};
        const void* bi1[1] = {bias};
        void* y1[1] = {y};
        return mpq_gemv_lut_launch(1, &qw, sc1, ze1, bias ? bi1 : nullptr, y1, &N, x, reinterpret_cast<unsigned*>(part) + BIE_WS_GEN_OFFSET,
                                   part + BIE_WS_HEAD_BYTES / sizeof(float), M, K, group_size, zm, dtype, st, w_bit);
    }
    const int MT = M <= 1 ? 1 : (M <= 2 ? 2 : (M <= 4 ? 4 : 8));
    static const int use_v3 = env_int("BIE_GEMV_V3", 1);
    const int NBv = 32 / w_bit;
    const int rpg = (group_size > K ? K : group_size) / NBv;
    if (perm == nullptr && use_v3 && (K / NBv) % 8 == 0 && rpg % 8 == 0 && (K & 1) == 0) {
        const Gemv3Plan p3 = plan_gemv3(K, N, w_bit);
        if (dtype == BIE_F16) return launch_gemv3_w<BIE_F16>(p3, MT, w_bit, zm, x, qw, scales, zeros, bias, part, y, M, K, N, group_size, st);
        return launch_gemv3_w<BIE_BF16>(p3, MT, w_bit, zm, x, qw, scales, zeros, bias, part, y, M, K, N, group_size, st);
    }
    const GemvPlan pl = plan_gemv(K, N, w_bit, group_size, MT);
    int rc;
    if (dtype == BIE_F16)
        rc = launch_gemv_w<BIE_F16>(pl, MT, w_bit, zm, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
    else
        rc = launch_gemv_w<BIE_BF16>(pl, MT, w_bit, zm, x, qw, scales, zeros, bias, perm, part, y, M, K, N, group_size, st);
    return rc;
}

int mpq_gemv_generic_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const int32_t* g_idx,
                            const void* bias, void* y, float* part, int M, int K, int N, int w_bit, int group_size,
                            int asym, int dtype, hipStream_t st, const uint16_t* perm) {
    const int k_per_slab = 512;
    const int S = cdiv(K, k_per_slab);
    dim3 grid(cdiv(N, 256), S, M);
    if (dtype == BIE_F16)
        hipLaunchKernelGGL(mpq_gemv_generic_kernel<BIE_F16>, grid, dim3(256), 0, st, x, (const uint32_t*)qw, scales, zeros, g_idx, part, M, K, N, w_bit, group_size, asym, k_per_slab, perm);
    else if (dtype == BIE_BF16)
        hipLaunchKernelGGL(mpq_gemv_generic_kernel<BIE_BF16>, grid, dim3(256), 0, st, x, (const uint32_t*)qw, scales, zeros, g_idx, part, M, K, N, w_bit, group_size, asym, k_per_slab, perm);
    else
        hipLaunchKernelGGL(mpq_gemv_generic_kernel<BIE_F32>, grid, dim3(256), 0, st, x, (const uint32_t*)qw, scales, zeros, g_idx, part, M, K, N, w_bit, group_size, asym, k_per_slab, perm);
    int rc = check_launch("mpq_gemv_generic_kernel");
    if (rc) return rc;
    return launch_splitk_finalize(part, bias, y, S, M, N, dtype, st);
}

}  // namespace bie
