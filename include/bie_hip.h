/*
 * bie_hip.h -- C ABI of libbie_hip.so: the MI355X (gfx950) low-bit Q-Linear / Q-Conv engine.
 *
 * This is the drop-in boundary for the hot path of GreenBitAI/bitorch-engine: every entry point
 * below replaces one function of the reference's pybind11 extension modules (cited per function,
 * paths relative to the reference tree).  Rules of the boundary:
 *   - extern "C", plain device pointers + sizes + enums; no torch / HIP types in signatures
 *     (`stream` is a hipStream_t passed as void*; NULL = the default stream);
 *   - the CALLER allocates every output and scratch buffer (torch.empty on the host side);
 *   - stream-ordered, no hidden synchronisation, no allocation, graph-capturable
 *     (the reference launches on the legacy default stream and cudaMalloc's per call);
 *   - returns 0 on success, a negative bie_status otherwise and records a message retrievable with
 *     bie_last_error() (the reference calls exit(EXIT_FAILURE) on unsupported arguments:
 *     layers/qlinear/nbit/cuda/mpq_linear_cuda_kernel.cu:506-508,573-576);
 *   - thread-safe and re-entrant (no global mutable state besides the thread-local error string).
 *
 * Tensor conventions (identical to the reference's state_dict layout):
 *   x        [M, K]            dtype (row-major, K contiguous)
 *   qweight  [K*w_bit/32, N]   int32; value k of column n = (qweight[k/(32/w)][n] >> ((k%(32/w))*w)) & (2^w-1)
 *   scales   [G, N]            dtype, G = ceil(K / group_size)
 *   zeros    [G, N]            dtype (symmetric / GBA)  or  int32 [G, N*w_bit/32] packed along N (asymmetric / GPTQ)
 *   g_idx    [K]               int32 group of row k, or NULL for the implicit k / group_size
 *   y        [M, N]            dtype
 */
#ifndef BIE_HIP_H
#define BIE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BIE_VERSION 300 /* 0.3.0 */

typedef enum { BIE_F16 = 0, BIE_BF16 = 1, BIE_F32 = 2 } bie_dtype;

typedef enum {
    BIE_OK = 0,
    BIE_ERR_INVALID_ARG = -1, /* NULL pointer, non-positive size, misaligned shape */
    BIE_ERR_UNSUPPORTED = -2, /* bit width / dtype / shape this build has no kernel for */
    BIE_ERR_WORKSPACE = -3,   /* scratch buffer smaller than bie_*_workspace_bytes() */
    BIE_ERR_HIP = -4,         /* a HIP runtime call failed (launch error) */
    BIE_ERR_DEVICE = -5       /* an EARLIER launch raised the device status page (see bie_device_status) */
} bie_status;

int bie_version(void);
/* message of the last failing call made by THIS thread ("" if none) */
const char* bie_last_error(void);

/* Device status page.  The decode kernels reduce split-K partial sums INSIDE the launch (tagged granules, no atomics); a
 * reducer whose partial sums do not arrive within its spin bound -- or a dependent list entry whose producer never finishes --
 * stores NaN in y and raises a bit in a 4 KiB host-mapped page instead of returning a silently wrong number (the reference's
 * half-precision atomicAdd split-K, mpq_linear_cuda_kernel.cu:440-450, cannot fail this way; an in-kernel hand-off can).
 * bie_status_init allocates the page (host-mapped pinned memory; call it ONCE, outside stream capture; without it the kernels
 * still poison y but cannot report).  bie_device_status returns the bits (1 = reducer timeout, 2 = dependency timeout) without
 * any device synchronisation and clears them if `clear`; every launching bie_mpq_* / bie_mbwq_exl2_forward call checks the page
 * first and fails with BIE_ERR_DEVICE (clearing it) if an earlier launch raised it. */
int bie_status_init(void);
unsigned bie_device_status(int clear);
/* (the fault-injection hooks the fail-loud tests use are NOT part of this ABI: include/bie_hip_testing.h) */

/* ------------------------------------------------------------------------------------------ */
/* MPQ (GPTQ-style) W{1,2,4,8}A16 linear                                                       */
/* ------------------------------------------------------------------------------------------ */

/* Scratch needed by bie_mpq_forward (split-K partial sums behind a 16 KiB head of arrival counters).
 * CONTRACT: the first BIE_WORKSPACE_HEAD_BYTES of a workspace must be ZERO the first time it is used;
 * the kernels return them to zero, so one memset at allocation time (or bie_workspace_init) is enough.
 * One workspace per stream (two launches that may run concurrently must not share one). */
#define BIE_WORKSPACE_HEAD_BYTES 16384
/* Stream-ordered memset of the counter head: call once after allocating a workspace that is not already
 * zero-filled, or to recover a workspace after a launch that did not complete (device reset, aborted graph). */
int bie_workspace_init(void* workspace, size_t workspace_bytes, void* stream);
size_t bie_mpq_workspace_bytes(int M, int K, int N, int w_bit);
/* Which prefill form a call of these rows takes (host-only, depends on (M, K, N) and the process environment like the sizing function above): 1 = the
 * dense two-launch form (dequantise once into the fragment image + dense MFMA GEMM: the reference's own split, mpq_layer.py:59-63), 0 = the fused
 * kernels.  Round 6: inside the measured grid the answer is the measurement (csrc/mpq_dense_table.inc). */
int bie_mpq_prefill_form(int M, int K, int N);
/* Which kernel family bie_mpq_forward takes for these rows (host-only; implicit groups, no g_idx): 2 = the decode kernels (lookup / matrix pipe: M <= 16, and
 * 17 .. 32 rows on the layer shapes where round 6 measured them ahead of the GEMM, csrc/mpq_lut_rb2_table.inc), 1 = the dense prefill form, 0 = the fused MFMA GEMM
 * (or the generic kernels for shapes it cannot take). */
int bie_mpq_rows_form(int M, int K, int N, int w_bit, int group_size, int dtype);
/* Rows of x up to which bie_mpq_forward_grouped runs a sibling set of n_total output columns as ONE decode launch (host-only): W4 16, or 32 where round 6 measured the
 * two-row-block instance ahead of the members' own calls (fp16; bf16 up to 16384 columns); W2 2.  More rows are served member by member. */
int bie_mpq_grouped_max_rows(int K, long n_total, int w_bit, int dtype);
/* The same for a call that passes an EXPLICIT g_idx which is not a permutation of k // group_size (unequal groups) with M > 32: room for
 * the dequantised MFMA fragment image, so that bie_mpq_forward runs "per-k dequantise + dense MFMA GEMM" instead of the row-chunked generic
 * kernel.  The reference's branch for those calls is unpack_qweight(g_idx) + torch.matmul (layers/qlinear/nbit/cuda/mpq_layer.py:59-63,
 * utils.py:36-51); with a workspace of only bie_mpq_workspace_bytes the call still works (generic kernel). */
size_t bie_mpq_workspace_bytes_gidx(int M, int K, int N, int w_bit);

/* y = x . dequant(qweight) (+ bias).
 * Replaces q_linear_cuda.mpq_forward (layers/qlinear/nbit/cuda/q_linear_cuda.cpp:258-270 ->
 * mpq_linear_cuda_kernel.cu:603-626) AND the M > 32 branch unpack_qweight + torch.matmul
 * (layers/qlinear/nbit/cuda/mpq_layer.py:59-63): one fused kernel family for every M.
 * The dequantised weight is rounded exactly like the reference's CPU path
 * (sym: fl(fl(q*s) - z); asym: fl(s * (q - (zq+1)))), products are accumulated in fp32 and the
 * result is rounded once to dtype; split-K partials are reduced in a fixed order (deterministic;
 * the reference uses half-precision atomicAdd, mpq_linear_cuda_kernel.cu:440-450).
 * bias may be NULL.  workspace may be NULL iff bie_mpq_workspace_bytes(...) == 0. */
int bie_mpq_forward(const void* x, const int32_t* qweight, const void* scales, const void* zeros,
                    const int32_t* g_idx, const void* bias, void* y, void* workspace,
                    size_t workspace_bytes, int M, int K, int N, int w_bit, int group_size, int asym,
                    int dtype, void* stream);

/* bie_mpq_forward whose result is a column range of a wider row-major destination: row m goes to y + m * ldy (elements; ldy >= N,
 * ldy % 4 == 0, y 8-byte aligned).  A column-sharded layer lets the GEMM epilogue store its shard straight into
 * out[:, lo:hi] instead of copying it there.  Implicit groups only (no g_idx).  Served by the MFMA GEMM range (M above the decode
 * kernels', shapes without a split-K plan); everything else returns BIE_ERR_UNSUPPORTED and launches nothing -- the caller keeps
 * its tight buffer + copy.  ldy == N is bie_mpq_forward.  No reference counterpart (its GEMM writes a fresh tensor,
 * mpq_layer.py:59-65). */
int bie_mpq_forward_pitched(const void* x, const int32_t* qweight, const void* scales, const void* zeros, const void* bias, void* y,
                            int ldy, void* workspace, size_t workspace_bytes, int M, int K, int N, int w_bit, int group_size, int asym,
                            int dtype, void* stream);

/* Several weight sets that share ONE activation x[M, K] (q/k/v projections, gate/up of an MLP): set i has
 * qweight[i] [K*w/32, N[i]], scales[i] / zeros[i] [G, N[i]], bias[i] (array or entry may be NULL), y[i] [M, N[i]].
 * Semantically n_sets calls of bie_mpq_forward with g_idx = NULL; for decode (M <= 2, bf16, W4) the column tiles of
 * all sets are concatenated into ONE launch, which amortises the fixed per-launch cost of a memory-bound kernel that
 * only lasts a few microseconds.  The pointer arrays live in HOST memory and are consumed during the call.
 * No reference counterpart: the reference launches one quant_mm_kernel per layer
 * (layers/qlinear/nbit/cuda/mpq_layer.py:65); this is the MI355X-side answer to its per-launch overhead. */
#define BIE_MAX_GROUPED_SETS 8
size_t bie_mpq_grouped_workspace_bytes(int n_sets, const int* N, int M, int K, int w_bit);
int bie_mpq_forward_grouped(const void* x, int n_sets, const int32_t* const* qweight, const void* const* scales,
                            const void* const* zeros, const void* const* bias, void* const* y, const int* N,
                            void* workspace, size_t workspace_bytes, int M, int K, int w_bit, int group_size,
                            int asym, int dtype, void* stream);

/* A LIST of decode layers (M <= 2; w_bit 4: M <= 32 -- 17 .. 32 rows as two row blocks per pass over the weights -- batched / speculative decode) in ONE launch.  Entry i computes y_i[M, N_i] = x_i[M, K_i] . dequant(qweight_i) (+ bias_i)
 * exactly as bie_mpq_forward with g_idx = NULL would; entries may differ in K, N and in their x / y buffers, while w_bit (4 or
 * 2), group_size, dtype (fp16 / bf16) and asym are common to the list.  `depends_on` >= 0 names an EARLIER entry whose y buffer
 * is this entry's x (a chain y_l -> x_{l+1}): the dependent entry's workgroups request their weight rows first and then wait for
 * the producer's completion count, so the weight stream of layer l+1 runs under the compute and reduction of layer l
 * (M <= 2 only).  Rows 3...16 (and 2 when the entries are independent, N % 4 == 0, x 16-byte aligned) take the lookup /
 * matrix-pipe kernel (v_mfma_f32_16x16x32); they need N % 4 == 0, K % 8 == 0 and a 16-byte aligned x.
 * No reference counterpart: the reference issues one default-stream quant_mm_kernel launch per layer
 * (layers/qlinear/nbit/cuda/mpq_linear_cuda_kernel.cu:482-577, mpq_layer.py:65); an 8.9 MB layer is over before the chip is
 * full, so the per-layer launch can never be bandwidth-bound.  This entry point is what makes decode HBM-bound on MI355X.
 *   bie_mpq_list_device_bytes: size of the caller-allocated DEVICE buffer the plan lives in (descriptor table, block table,
 *       generation words, completion counters, granules).
 *   bie_mpq_list_create: validates, plans and uploads the tables (blocking copy; not capturable) and returns a host handle.
 *       The tensor POINTERS are frozen in the plan; their contents may change between launches.
 *   bie_mpq_list_forward: one kernel launch (+ one memset node when the list has dependencies); stream-ordered, capturable.
 *   bie_mpq_list_destroy: frees the host handle (the device buffer is the caller's). */
typedef struct bie_mpq_list bie_mpq_list_t;
typedef struct {
    const void* x;           /* [M, K] dtype */
    const int32_t* qweight;  /* [K*w_bit/32, N] */
    const void* scales;      /* [K/group_size, N] dtype */
    const void* zeros;       /* [K/group_size, N] dtype, or int32 [K/group_size, N*w_bit/32] if asym */
    const void* bias;        /* [N] dtype or NULL */
    void* y;                 /* [M, N] dtype */
    int K, N;
    int depends_on;          /* index of an earlier entry whose y is this entry's x, or -1 */
    int reserved;
} bie_mpq_list_entry;
size_t bie_mpq_list_device_bytes(int n_entries, const bie_mpq_list_entry* entries, int M, int w_bit, int group_size);
int bie_mpq_list_create(bie_mpq_list_t** plan, int n_entries, const bie_mpq_list_entry* entries, int M, int w_bit,
                        int group_size, int asym, int dtype, void* device_mem, size_t device_bytes);
int bie_mpq_list_forward(bie_mpq_list_t* plan, void* stream);
int bie_mpq_list_launches(const bie_mpq_list_t* plan); /* kernel launches one forward issues (1, or 1 + a memset node) */
int bie_mpq_list_form(const bie_mpq_list_t* plan);     /* which kernel the plan chose: 0 lookup + FMA (M <= 2), 1 matrix pipe with K split over
                                                         * the waves of a workgroup, 2 matrix pipe with x shared by the four column tiles of a workgroup */
void bie_mpq_list_destroy(bie_mpq_list_t* plan);

/* out[K, N] (dtype) = dequantised weight.  Bit-exact twin of unpack_qweight layer_type 1
 * (layers/qlinear/nbit/cuda/utils.py:30-51). */
int bie_mpq_dequant(const int32_t* qweight, const void* scales, const void* zeros,
                    const int32_t* g_idx, void* out, int K, int N, int w_bit, int group_size,
                    int asym, int dtype, void* stream);

/* out[K*w/32, N] (int32) = packed quantisation of weight[K, N].  Bit-exact twin of pack_fp_weight
 * (layers/qlinear/nbit/cuda/utils.py:72-147). */
int bie_mpq_pack(const void* weight, const void* scales, const void* zeros, const int32_t* g_idx,
                 int32_t* out, int K, int N, int w_bit, int group_size, int asym, int dtype,
                 void* stream);

/* grad_x[M, K] = grad_y[M, N] . dequant(qweight)^T.  Replaces q_linear_cuda.mpq_grad_input
 * (mpq_linear_cuda_kernel.cu:1198-1223). */
int bie_mpq_grad_input(const void* grad_y, const int32_t* qweight, const void* scales,
                       const void* zeros, const int32_t* g_idx, void* grad_x, int M, int K, int N,
                       int w_bit, int group_size, int asym, int dtype, void* stream);

/* Act-order (explicit g_idx) preparation.  The reference resolves g_idx[k] per weight inside quant_mm_kernel
 * (mpq_linear_cuda_kernel.cu:300-317).  Here a layer whose g_idx is a permutation of k / group_size is re-ordered once:
 * out[K*w/32, N] holds field k' = field perm[k'] of qweight (perm = stable argsort of g_idx, int32[K]); bit-exact.
 * bie_mpq_forward on (out, g_idx = NULL) and bie_gather_cols(x, perm) then equals the g_idx forward of the original. */
int bie_mpq_sort_rows(const int32_t* qweight, const int32_t* perm, int32_t* out, int K, int N,
                      int w_bit, void* stream);

/* out[M, K] = x[:, perm]  (fp16 / bf16 / fp32 by dtype). */
int bie_gather_cols(const void* x, const int32_t* perm, void* out, int M, int K, int dtype,
                    void* stream);

/* ------------------------------------------------------------------------------------------ */
/* MBWQ: uniform 4/2-bit (GPTQ-like) and mixed 8/6/5/4/3/2-bit (exl2 layout) linear, fp16 only   */
/* ------------------------------------------------------------------------------------------ */

/* Band table rows[7] = {rows_8, rows_6, rows_5, rows_4, rows_3, rows_2 (cumulative k), kernel_p}
 * from the (bits, first packed row) pairs in q_groups.  HOST pointers, pure host code.
 * Replaces the table computation of q_linear_cuda.mbwq_trans_qweight
 * (mbwq_linear_cuda_kernel.cu:559-600; its shuffle kernel is a no-op, exl2/config.h:16-21). */
int bie_mbwq_rows(const int16_t* q_groups_host, int groups, int K, int* rows7_host);

/* The load-time step of the mixed-bit layout: re-arranges qweight IN PLACE (device tensor, [rows_packed, N]) from the
 * checkpoint's LSB-first chunk streams into the layout every bie_mbwq_exl2_* kernel reads, and writes the EXTENDED band table
 * rows_host[BIE_EXL2_ROWS_LEN] (HOST) that those calls take.  This is the reference's shuffle hook: shuffle_kernel
 * (mbwq_linear_cuda_kernel.cu:63-86), launched by q_linear_cuda.mbwq_trans_qweight (:602-625) on the tensor it returns; the
 * reference's build leaves its shuffle functions empty (exl2/config.h:16-21), this library uses the hook for a layout made for
 * v_and_or_b32 + packed fp16 ("half-pair" layout): 32 values of a chunk = 16 pairs (q[2j], q[2j+1]), a pair
 * sits at the same bit range of the low and of the high half of one word.  Call it ONCE per tensor; a second call scrambles it
 * (as a second shuffle_kernel pass would).  The kernels refuse a table without the SHUFFLED mark (a plain bie_mbwq_rows table).
 *   rows_host[0..5]   cumulative k ends of the 8/6/5/4/3/2-bit bands, [6] kernel_p | BIE_EXL2_ROWS_SHUFFLED | BIE_EXL2_ROWS_REGULAR
 *   rows_host[7..12]  first group of each band, [13..18] log2(chunks per group) of each band -- valid with REGULAR: every band's
 *                     groups hold the same power-of-two number of whole 32-k chunks (the last one may be shorter); decode then
 *                     needs no staged group map (exl2_gemv2_body<DIRECT>)
 *   rows_host[19]     BIE_EXL2_ROWS_TAG */
#define BIE_EXL2_ROWS_LEN 20
#define BIE_EXL2_ROWS_SHUFFLED 0x100
#define BIE_EXL2_ROWS_REGULAR 0x200
#define BIE_EXL2_ROWS_GFIRST 7
#define BIE_EXL2_ROWS_GLOG 13
#define BIE_EXL2_ROWS_TAG 0x45584c32
int bie_mbwq_exl2_shuffle(int32_t* qweight, const int16_t* q_groups_host, int groups, int K, int N, int* rows_host, void* stream);
/* The table alone (pure host code, nothing launched): what bie_mbwq_exl2_shuffle writes to rows_host.  For a tensor that a previous run
 * re-arranged and saved: load it as it is and take its table from here. */
int bie_mbwq_exl2_table(const int16_t* q_groups_host, int groups, int K, int* rows_host);
/* The way back, IN PLACE: a tensor bie_mbwq_exl2_shuffle re-arranged -> the checkpoint's LSB-first chunk streams (bit-exact inverse).
 * rows_host = the table that shuffle returned.  Used to WRITE checkpoints: a state_dict of a prepared layer holds the reference's
 * format (its shuffle is a no-op, exl2/config.h:16-21, so what the reference saves after prepare_params is the stream), never
 * this library's private layout. */
int bie_mbwq_exl2_unshuffle(int32_t* qweight, const int* rows_host, int K, int N, void* stream);

/* out[K, N] fp16: W[q_perm ? q_perm[k] : k][n] = fma(s, q, -z).  Replaces
 * q_linear_cuda.mbwq_q42fp_weight (mbwq_linear_cuda_kernel.cu:656-710, kernels :314-501). */
int bie_mbwq_q4_dequant(const int32_t* qweight, const void* scales, const void* zeros,
                        const int16_t* q_perm, void* out, int K, int N, int bits, int group_size,
                        void* stream);

/* out[K, N] fp16 for the mixed-bit layout, addressed exactly like the reference kernels do:
 * q_group_map is the DEVICE int16[2K] array of (group, rows-left-in-group) pairs built by
 * make_group_map (layers/qlinear/nbit/cuda/utils.py:150-187); rows7_host the HOST band table of
 * bie_mbwq_exl2_shuffle (BIE_EXL2_ROWS_LEN ints; qweight as that call left it).  Replaces q_linear_cuda.mbwq_exl2fp_weight
 * (mbwq_linear_cuda_kernel.cu:849-897, kernel :92-308). */
int bie_mbwq_exl2_dequant(const int32_t* qweight, const void* scales, const void* zeros,
                          const int16_t* q_perm, const int16_t* q_group_map, const int* rows7_host,
                          void* out, int K, int N, int groups, void* stream);

size_t bie_mbwq_workspace_bytes(int M, int K, int N);     /* bie_mbwq_exl2_forward (covers bie_mbwq_q4_forward too) */
size_t bie_mbwq_q4_workspace_bytes(int M, int K, int N);  /* bie_mbwq_q4_forward alone: without the mixed-bit prefill form's dense weight image */

/* y[M, N] fp16 = x[:, q_perm] . dequant.  Replace q_linear_cuda.mbwq_q4_forward
 * (mbwq_linear_cuda_kernel.cu:742-825) and q_linear_cuda.mbwq_exl2_forward (:926-1007).
 * bie_mbwq_exl2_forward: one pass over the packed weight for M <= 64 (M <= 2 the decode kernel; 3 <= M <= 16 with a table carrying
 * BIE_EXL2_ROWS_REGULAR the permute kernel + the pre-permuted decode body, a group of one; otherwise up to 64 rows the same stream on
 * the 16x16x32 matrix instruction, after one x[:, q_perm] launch into the workspace; the reference's fused range is M <= 32,
 * exl2/q_gemm_kernel.cuh:90-549);
 * larger M is served in passes of 8 rows -- callers reconstruct (bie_mbwq_exl2_dequant) and use a dense GEMM there, as the
 * reference does (:947-957).  q_perm may be NULL (no act-order). */
int bie_mbwq_q4_forward(const void* x, const int32_t* qweight, const void* scales,
                        const void* zeros, const int16_t* q_perm, void* y, void* workspace,
                        size_t workspace_bytes, int M, int K, int N, int bits, int group_size,
                        void* stream);
int bie_mbwq_exl2_forward(const void* x, const int32_t* qweight, const void* scales,
                          const void* zeros, const int16_t* q_perm, const int16_t* q_group_map,
                          const int* rows7_host, void* y, void* workspace, size_t workspace_bytes,
                          int M, int K, int N, int groups, void* stream);

/* A LIST of exl2 decode layers (M <= 2; M <= 16 when every entry's table carries BIE_EXL2_ROWS_REGULAR; fp16) in ONE launch: entry i is exactly one bie_mbwq_exl2_forward call (its own x, packed
 * matrix, band table, q_perm / q_group_map, y).  A 4096x4096 3/2-bit layer is 5 MB -- a lone launch of that size never leaves its
 * start-up transient (0.13-0.17 of the HBM roofline per layer launch); the list form walks the column blocks of every layer in one
 * grid.  Same contract as bie_mpq_list_*: caller-allocated device buffer of bie_mbwq_exl2_list_device_bytes, create uploads the
 * tables (blocking, not capturable), forward is one stream-ordered capturable launch, destroy frees the host handle.
 * No reference counterpart (one gemm_half_q_half_kernel launch per layer, mbwq_linear_cuda_kernel.cu:926-1007). */
typedef struct bie_exl2_list bie_exl2_list_t;
typedef struct {
    const void* x;              /* [M, K] fp16 */
    const int32_t* qweight;     /* [rows_packed, N], after bie_mbwq_exl2_shuffle */
    const void* scales;         /* [groups, N] fp16 */
    const void* zeros;          /* [groups, N] fp16 */
    const int16_t* q_perm;      /* [K] or NULL */
    const int16_t* q_group_map; /* [2K] device */
    const int* rows7;           /* HOST band table of bie_mbwq_exl2_shuffle, BIE_EXL2_ROWS_LEN ints (read during create) */
    void* y;                    /* [M, N] fp16 */
    int K, N;
    int reserved0, reserved1;   /* 0 */
} bie_exl2_list_entry;
size_t bie_mbwq_exl2_list_device_bytes(int n_entries, const bie_exl2_list_entry* entries, int M);
int bie_mbwq_exl2_list_create(bie_exl2_list_t** plan, int n_entries, const bie_exl2_list_entry* entries, int M, void* device_mem,
                              size_t device_bytes);
int bie_mbwq_exl2_list_forward(bie_exl2_list_t* plan, void* stream);
void bie_mbwq_exl2_list_destroy(bie_exl2_list_t* plan);

/* Up to 8 exl2 layers that consume the SAME activation x[M, K], M <= 16 (q / k / v, gate / up -- every layer has its own q_perm) in two
 * stream-ordered launches and without a plan object: the member descriptors travel in the kernel arguments, so x and the outputs
 * ([M, N_i] each) may be new buffers on every call (members[i].x is ignored).  Members need tables carrying BIE_EXL2_ROWS_REGULAR; fp16.
 * The rows beyond the first cost no weight traffic: four rows ride on one v_mfma_f32_4x4x4 (measured against lone calls: ahead at every row count up to 16).
 * Workspace: zero-filled once (head words as for bie_mbwq_exl2_forward; a buffer may serve both), of
 * bie_mbwq_exl2_grouped_workspace_bytes (0 = not groupable: call bie_mbwq_exl2_forward per member).  The reference launches
 * gemm_half_q_half_kernel once per layer (mbwq_linear_cuda_kernel.cu:926-1007): at 4096x4096 three launches take 3 x 6.7 us here,
 * the group 11.6 us. */
size_t bie_mbwq_exl2_grouped_workspace_bytes(int n_members, const bie_exl2_list_entry* members, int M);
int bie_mbwq_exl2_forward_grouped(const void* x, int M, int n_members, const bie_exl2_list_entry* members, void* workspace,
                                  size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Binary (1-bit W / 1-bit A) linear + conv2d, XNOR-popcount                                     */
/* ------------------------------------------------------------------------------------------ */

/* out[rows, K/8] uint8, bit j of byte b = (a[r][8b+j] >= 0).  Row packing of
 * binary_linear.cpp:43-54 (_get_binary_row) == binary_linear_cutlass_kernel.cu:44-90. */
int bie_binary_pack_rows_u8(const void* a, uint8_t* out, long rows, long K, int dtype, void* stream);

/* out[K/8, N] uint8 column bit-planes of w[N, K]: byte[kb*N + n] bit j = (w[n][8kb+j] >= 0).
 * Replaces binary_linear_cpp.w_pack (binary_linear.cpp:454-465). */
int bie_binary_pack_cols_u8(const void* w, uint8_t* out, long N, long K, int dtype, void* stream);

/* The reference CUDA layer's packed-weight images, bit for bit, so that BinaryLinearCuda checkpoints packed on CUDA load here:
 * w [N, K] values (dtype 0=f16 1=bf16 2=f32 3=int8 sign carriers) -> image of N*K/8 bytes; 32 consecutive k of one column per
 * 32-bit word, MSB first, words serialised big-endian.  BTC32 (bmm_type 2, or 3 with K % 128 == 0 and N % 8 == 0):
 * BMMA_toBit32Col_new tiles, binary_linear_cuda_kernel.cu:59-152; BSTC32 (otherwise): ToBit32RowUd, :186-300; dispatch and
 * byte order: _get_binary_weight_cuda :830-882, uint32_to_uint8 :33-41.  The unpack entry points turn an image into the
 * row-packed LSB-first [N, K/8] operand bie_binary_linear_forward(w_layout = 0) consumes. */
int bie_binary_pack_btc32(const void* w, uint8_t* image, long N, long K, int dtype, void* stream);
int bie_binary_pack_bstc32(const void* w, uint8_t* image, long N, long K, int dtype, void* stream);
int bie_binary_unpack_btc32(const uint8_t* image, uint8_t* rowpacked, long N, long K, void* stream);
int bie_binary_unpack_bstc32(const uint8_t* image, uint8_t* rowpacked, long N, long K, void* stream);

/* y[M, N] fp32 = (K - 2*popcount(xbits ^ wbits)) * scale.
 * w_layout 0: wpacked is row-packed [N, K/8] (binary_linear_cutlass / our native layout)
 * w_layout 1: wpacked is column bit-planes [K/8, N] (binary_linear_cpp.w_pack layout).
 * xpacked [M, K/8] row-packed.  Replaces binary_linear_cpp.forward (binary_linear.cpp:494-512),
 * binary_linear_cuda.forward (binary_linear_cuda_kernel.cu:629-660) and
 * binary_linear_cutlass.forward (binary_linear_cutlass_kernel.cu:604-625). */
int bie_binary_linear_forward(const uint8_t* xpacked, const uint8_t* wpacked, float* y, long M,
                              long N, long K, int w_layout, float scale, void* stream);

/* The same product on the MATRIX pipe (large M): +1 / -1 are exact FP4 (E2M1) values, so v_mfma_scale_f32_32x32x64_f8f6f4 on
 * FP4 images of the sign matrices accumulates K - 2*popcount(xbits ^ wbits) exactly in fp32 (K < 2^24) -- the same integers as
 * bie_binary_linear_forward at ~4x the rate of the v_xor + v_bcnt contraction (CDNA4 has no 1-bit MFMA).  Replaces the same
 * reference functions: binary_linear_cuda_kernel.cu:155-181,308-393,629-660, binary_linear_cutlass_kernel.cu:293-332,604-625.
 *   bie_binary_fp4_image_bytes(rows, K): size of an image (rows padded to 32, K to 128, 4 bits per element, 1 KiB MFMA fragments).
 *   bie_binary_fp4_image: row-packed sign bits [rows, K/8] (LSB first, the operand format above) -> image.
 *   bie_binary_fp4_image_from_values: values [rows, K] (dtype 0=f16 1=bf16 2=f32 3=int8), sign = ((v + bias[k]) >= 0) with the
 *     sum rounded in the tensor dtype (bias may be NULL) -> image; the bit packing of BinaryLinearCuda's set_activation
 *     (layers/qlinear/binary/cuda/layer.py:283) and bie_binary_pack_rows_u8 folded into the image pass.
 *   bie_binary_linear_forward_fp4: y[M, N] fp32 = (K - 2*popcount) * scale from two images (x: M rows, w: N rows).
 *   bie_binary_linear_layer_fp4: the BinaryLinearCuda layer epilogue in the GEMM: y[M, N] (dtype 0=f16 1=bf16 2=f32) =
 *     dt(dt(dt(K - 2*popcount) * scale_a) * scale_w), scale_a / scale_w device scalars of that dtype or NULL (= 1) -- the roundings
 *     of `forward(...).to(input.dtype) * scale_a * scale_w` (layers/qlinear/binary/cuda/layer.py:58-63), as bie_binary_linear_fused.
 *     With bie_binary_fp4_image_from_values(x, bias_a) in front: the whole layer forward at large M in two launches. */
size_t bie_binary_fp4_image_bytes(long rows, long K);
int bie_binary_fp4_image(const uint8_t* rowpacked, uint8_t* image, long rows, long K, void* stream);
int bie_binary_fp4_image_from_values(const void* values, const void* bias, uint8_t* image, long rows, long K, int dtype, void* stream);
int bie_binary_linear_forward_fp4(const uint8_t* ximage, const uint8_t* wimage, float* y, long M, long N, long K, float scale, void* stream);
int bie_binary_linear_layer_fp4(const uint8_t* ximage, const uint8_t* wimage, const void* scale_a, const void* scale_w, void* y, long M, long N,
                                long K, int dtype, void* stream);

/* The same convolution on the matrix pipe for large batches (C % 32 == 0): channel-minor sign bits of x, the FP4 image of the
 * (output pixel) x (tap, channel) matrix -- padding = -1.0 -- and the FP4 GEMM above with an NCHW epilogue; bit-identical to
 * bie_binary_conv2d_forward.  wimage = bie_binary_fp4_image of the tap-major weight words of bie_binary_conv_weight_taps read as
 * row-packed bytes (rows = OC, K = ksize*ksize*C): the k order (tap, channel) only has to be the same on both sides.
 * Replaces binary_conv_cpp.forward (binary_conv.cpp:319-365 im2binary_col + :464-530). */
size_t bie_binary_conv2d_fp4_workspace_bytes(int B, int C, int H, int W, int ksize, int stride, int pad, int dilation);
int bie_binary_conv2d_forward_fp4(const void* x, const uint8_t* wimage, float* y, void* workspace, size_t workspace_bytes, int B, int C,
                                  int H, int W, int OC, int ksize, int stride, int pad, int dilation, float scale, int dtype,
                                  void* stream);

/* The same convolution as ONE launch and without a workspace (round 6; C in {128, 256, 512}, k in {1, 3}, dilation 1, output rows of at
 * most 64 pixels -- bie_binary_conv2d_fused_ok says whether a geometry is in range): a workgroup = (image, 64 or 128 output channels, a
 * range of whole output rows) sign-packs the input rows it needs straight from x into a zero-bordered LDS bit image, keeps its K quarter of
 * the weights in registers (lane = output channel) and runs XNOR-popcount against uniform-address LDS reads; the quarters meet in LDS and y
 * leaves pixel-contiguous.  wlanes = bie_binary_conv_weight_lanes of the tap-major words of bie_binary_conv_weight_taps
 * ([ceil(OC/64)][4 quarters][k*k*C/128][64 lanes] uint32, bie_binary_conv_weight_lanes_bytes, once per weight tensor).  Bit-identical to
 * bie_binary_conv2d_forward.  Replaces binary_conv_cpp.forward (binary_conv.cpp:319-365 im2binary_col + :464-530) and the implicit-GEMM
 * convolution of binary_conv2d_cutlass_kernel.cu:122-183. */
int bie_binary_conv2d_fused_ok(int B, int C, int H, int W, int OC, int ksize, int stride, int pad, int dilation);
size_t bie_binary_conv_weight_lanes_bytes(int OC, int C, int ksize);
int bie_binary_conv_weight_lanes(const uint32_t* wtaps, uint32_t* wlanes, int OC, int C, int ksize, void* stream);
int bie_binary_conv2d_forward_fused(const void* x, const uint32_t* wlanes, float* y, int B, int C, int H, int W, int OC, int ksize, int stride,
                                    int pad, int dilation, float scale, int dtype, void* stream);

/* The matrix-pipe convolution as ONE launch (round 6; C % 64 == 0, k in {1, 3}, dilation 1, output rows of at most 128 pixels --
 * bie_binary_conv2d_mfma_ok): a workgroup sign-packs its input rows from x into a -1.0-bordered FP4 NHWC image in LDS and gathers the
 * pixel operand fragments of v_mfma_scale_f32_32x32x64_f8f6f4 from it, tap by tap -- the FP4 im2col matrix of
 * bie_binary_conv2d_forward_fp4 (and its two extra launches) never exists.  wimage as for bie_binary_conv2d_forward_fp4 (bie_binary_fp4_image
 * of the tap-major weight words, rows = OC, K = ksize*ksize*C).  Bit-identical to bie_binary_conv2d_forward.
 * Replaces binary_conv_cpp.forward (binary_conv.cpp:319-365, :464-530) / binary_conv2d_cutlass_kernel.cu:122-183. */
int bie_binary_conv2d_mfma_ok(int B, int C, int H, int W, int OC, int ksize, int stride, int pad, int dilation);
int bie_binary_conv2d_forward_mfma(const void* x, const uint8_t* wimage, float* y, int B, int C, int H, int W, int OC, int ksize, int stride,
                                   int pad, int dilation, float scale, int dtype, void* stream);

/* `batch` independent XNOR GEMMs in ONE launch: y[b][M, N] = (K - 2*popc(x[b] ^ w[b])) * scale, both operands row-packed
 * uint8 [rows, K/8]; strides in BYTES (packed operands) / ELEMENTS (y) between consecutive matrices.
 * Replaces binary_linear_cutlass.matmul -> binary_batched_forward_cutlass
 * (layers/qlinear/binary/cutlass/binary_linear_cutlass_kernel.cu:336-393, 700-738). */
int bie_binary_matmul_batched(const uint8_t* xpacked, const uint8_t* wpacked, float* y, long batch, long M, long N, long K,
                              long stride_x, long stride_w, long stride_y, float scale, void* stream);

/* One launch per BinaryLinearCuda layer forward (M <= 64; M <= 512 when K % 512 == 0):
 *   y[M, N] (dtype) = dt( dt( dt(K - 2*popcount(xbits ^ wbits)) * scale_a ) * scale_w ),  xbits = ((x + bias_a) >= 0)
 * x [M, K] raw activations, bias_a [K] or NULL, wpacked row-packed [N, K/8] (all three 16-byte aligned), scale_a / scale_w device
 * scalars of the same dtype or NULL (= 1).  The roundings are those of BinaryLinearForward.forward
 * (layers/qlinear/binary/cuda/layer.py:58-63: forward(...).to(input.dtype), then out*scale_a*scale_w) after set_activation
 * (:283).  y_f32 != 0: y is float[M, N] = K - 2*popcount, no rounding, scales must be NULL (what binary_linear_cuda.forward
 * itself returns, binary_linear_cuda_kernel.cu:629-660).  bie_binary_linear_fused_ok says whether a shape is in range (1 <= M <= 64 with K % 32 == 0, or 5 <= M <= 512 with K % 512 == 0). */
int bie_binary_linear_fused_ok(long M, long N, long K);
int bie_binary_linear_fused(const void* x, const void* bias_a, const uint8_t* wpacked,
                            const void* scale_a, const void* scale_w, void* y, long M, long N,
                            long K, int dtype, int y_f32, void* stream);

/* y[B, OC, OH, OW] fp32 = scale * sum over (c,i,j) of sign(x)*sign(w), padding counted as -1.
 * x [B, C, H, W] dtype; wpacked [OC, C*k*k/8] row-packed over the flattened (c,i,j) index
 * (C*k*k % 8 == 0).  workspace holds the bit-im2col image (bie_binary_conv2d_workspace_bytes).
 * Replaces binary_conv_cpp.forward (layers/qconv/binary/cpp/binary_conv.cpp:464-530,
 * im2binary_col :319-365). */
size_t bie_binary_conv2d_workspace_bytes(int B, int C, int H, int W, int OC, int ksize, int stride,
                                         int pad, int dilation);
int bie_binary_conv2d_forward(const void* x, const uint8_t* wpacked, float* y, void* workspace,
                              size_t workspace_bytes, int B, int C, int H, int W, int OC, int ksize,
                              int stride, int pad, int dilation, float scale, int dtype, void* stream);

/* The same convolution without the bit-im2col image: the packed weights re-laid once per tensor as
 * wtaps[OC][k*k][ceil(C/32)] uint32 (channel bits of one tap contiguous; bie_binary_conv_weight_taps, 16-byte aligned output),
 * activations sign-packed channel-minor into the workspace (bie_binary_conv2d_workspace_bytes covers both forms), one
 * XNOR-popcount pass.  bie_binary_conv2d_taps_ok: 1 when k input rows of W pixels x C channels fit a wave's 16 KiB LDS slab
 * (otherwise use bie_binary_conv2d_forward).  Same result bit for bit (binary_conv.cpp:464-530). */
int bie_binary_conv2d_taps_ok(int C, int W, int ksize);
int bie_binary_conv_weight_taps(const uint8_t* wpacked, uint32_t* wtaps, int OC, int C, int ksize,
                                void* stream);
int bie_binary_conv2d_forward_taps(const void* x, const uint32_t* wtaps, float* y, void* workspace,
                                   size_t workspace_bytes, int B, int C, int H, int W, int OC,
                                   int ksize, int stride, int pad, int dilation, float scale,
                                   int dtype, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* functions/cuda pack / unpack helpers (functions/cuda/functions_cuda_kernel.cu)                */
/* ------------------------------------------------------------------------------------------ */

/* sign -> uint8, LSB first (tensor_pack_to_uint8, :107-117, hosts :290-297). dtype may also be
 * 3 = int8 sign carriers. */
int bie_pack_sign_u8(const void* a, uint8_t* out, long n_bytes, int dtype, void* stream);
/* uint8 -> +-1 * scale[i / packed_dim] (uint8_to_unpacked_tensor, :121-134) */
int bie_unpack_u8_scaled(const uint8_t* in, const float* scale, float* out, long n_bytes,
                         long packed_dim, void* stream);
/* two int32 -> one int8, FIRST element in the HIGH nibble (q4_pack, :137-159) */
int bie_q4_pack(const int32_t* in, int8_t* out, long n_out, void* stream);
/* int8 -> two unsigned nibbles as int32 (q4_unpack, :162-181) */
int bie_q4_unpack(const int8_t* in, int32_t* out, long n_in, void* stream);
/* int8 -> two sign-extended nibbles * scale, fp32 (q4_unpack_and_scaling, :184-207) */
int bie_q4_unpack_scale(const int8_t* in, float* out, long n_in, float scale, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* W4A4 / W8A8 integer GEMM (the reference's CUTLASS int4 / int8 path) */
/* ------------------------------------------------------------------------------------------ */

/* q = clamp(round_half_away(x / max(scale, 1e-5)), -8, 7), two values per int8, FIRST value in the
 * high nibble.  Replaces q4_quantization_and_bit_packing_kernel / q_linear_cutlass.q4_w_pack
 * (layers/qlinear/nbit/cutlass/q4_linear_cutlass_kernel.cu:74-170, 418-470). */
int bie_q4_quantize_pack(const void* x, int8_t* out, long n_out, float scale, int dtype, void* stream);

/* y[b][M, N] (dtype) = fl( fl( sum_k a[m][k] * w[n][k] ) * fl(scale_a * scale_w) ) on nibble-packed operands
 * a [batch][M, K/2], w [batch][N, K/2] (K % 64 == 0, N % 4 == 0); batch strides in elements of each tensor.
 * Replaces q_linear_cutlass.q4_forward / q4_matmul (q4_linear_cutlass_kernel.cu:520-680). */
int bie_q4_gemm(const int8_t* a_packed, const int8_t* w_packed, void* y, int M, int N, int K, float scale_a,
                float scale_w, int dtype, int batch, long stride_a, long stride_w, long stride_y,
                void* stream);

/* y[M, N] fp32 = ((float) sum_k a[m][k] * w[n][k]) * scale_a * scale_w on int8 operands a [M, K], w [N, K].
 * Replaces q_linear_cutlass.q8_forward (q8_linear_cutlass_kernel.cu:186-230). */
int bie_q8_gemm(const int8_t* a, const int8_t* w, float* y, int M, int N, int K, float scale_a, float scale_w,
                void* stream);

/* y[batch][M][N] (int32) = A[batch][M][K] . W[batch][N][K]^T on packed 4-bit (bits = 4: K/2 bytes per row, first value in the high
 * nibble) or int8 (bits = 8) operands: the raw accumulators, which is what the reference's q4_gemm / q8_gemm return
 * (layers/qlinear/nbit/cutlass/q4_linear_cutlass_kernel.cu:526-555) and its backward entry points q4_backward :719-743,
 * q4_matmul_backward :901-941, q8_backward q8_linear_cutlass_kernel.cu:283-308 are built from.  Strides in bytes of A / W per batch
 * element and in int32 elements for y.  K % 64 == 0, N % 4 == 0. */
int bie_int_gemm_i32(const int8_t* a, const int8_t* w, int32_t* y, int M, int N, int K, int bits, int batch, long stride_a,
                     long stride_w, long stride_y, void* stream);

/* W4A4 convolution on nibble-packed NHWC operands: a [B, H, W, C/2], w [OC, KS, KS, C/2] (C % 8 == 0, OC % 4 == 0), zero padding,
 * y [B, OH, OW, OC] (dtype) with the bie_q4_gemm epilogue; OH = (H + 2*pad - dil*(KS-1) - 1)/stride + 1.  The workspace holds the
 * packed im2col matrix (no counters, need not be zeroed).  Replaces q4_conv_cutlass.forward
 * (layers/qconv/nbit/cutlass/q4_conv_cutlass_kernel.cu:441-510; the CUTLASS implicit-GEMM :178-345). */
size_t bie_q4_conv2d_workspace_bytes(int B, int H, int W, int C, int OC, int ksize, int stride, int pad, int dilation);
int bie_q4_conv2d_forward(const int8_t* a_packed, const int8_t* w_packed, void* y, void* workspace, size_t workspace_bytes, int B,
                          int H, int W, int C, int OC, int ksize, int stride, int pad, int dilation, float scale_a, float scale_w,
                          int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BIE_HIP_H */
