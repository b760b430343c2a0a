// extern "C" surface of libbie_hip.so (see include/bie_hip.h for the contract and the reference
// functions each entry point replaces).  Argument validation + dispatch only; kernels live in the
// sibling translation units.
#include "bie_common.h"
#include <stdlib.h>

namespace bie {
const char* get_error();
// mpq_gemv.hip
bool mpq_gemv_fast_ok(int M, int K, int N, int w_bit, int group_size, int dtype, bool has_gidx);
size_t mpq_gemv_workspace_bytes(int M, int K, int N, int w_bit);
int mpq_gemv_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const void* bias, void* y,
                    float* part, int M, int K, int N, int w_bit, int group_size, int zm, int dtype, const uint16_t* perm,
                    hipStream_t st);
int mpq_gemv_generic_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const int32_t* g_idx,
                            const void* bias, void* y, float* part, int M, int K, int N, int w_bit, int group_size,
                            int asym, int dtype, hipStream_t st, const uint16_t* perm = nullptr);
// mpq_gemv_lut.hip
bool mpq_gemv_lut_ok(int M, int K, int w_bit, int group_size, int dtype, bool has_gidx, int N = 0);  // N > 0: a lone call (17 .. 32 rows on measured shapes)
bool mpq_lut_rb2_grouped_ok(int M, int K, long n_total, int dtype);
size_t mpq_gemv_lut_part_floats(int M, int K, int group_size, int tiles_total, int w_bit);
int mpq_gemv_lut_launch(int nsets, const int32_t* const* qw, const void* const* scales, const void* const* zeros,
                        const void* const* bias, void* const* y, const int* N, const void* x, unsigned* counters, float* part,
                        int M, int K, int group_size, int zm, int dtype, hipStream_t st, int w_bit);
// mpq_list.hip
struct MpqList;
size_t mpq_list_device_bytes(int n, const bie_mpq_list_entry* ent, int M, int w_bit, int group_size);
int mpq_list_create(MpqList** out, int n, const bie_mpq_list_entry* ent, int M, int w_bit, int group_size, int asym, int dtype,
                    void* device_mem, size_t device_bytes);
int mpq_list_forward(MpqList* p, hipStream_t st);
int mpq_list_launches(const MpqList* p);
int mpq_list_form(const MpqList* p);
void mpq_list_destroy(MpqList* p);
// splitk.hip
int status_init();
unsigned status_read(bool clear);
int status_report(const char* fn);
void test_forge_set(unsigned tag_skew, int spin_limit);
void test_forge_dep_set(int extra);
// mpq_gemm.hip
bool mpq_gemm_ok(int M, int K, int N, int w_bit, int group_size, int dtype, bool has_gidx);
size_t mpq_gemm_workspace_bytes(int M, int K, int N);
int mpq_gemm_launch_ld(const void* x, const int32_t* qw, const void* scales, const void* zeros, const void* bias, void* y,
                       float* part, int M, int K, int N, int w_bit, int group_size, int zm, int dtype, const uint16_t* perm,
                       hipStream_t st, int ldy);
bool mpq_gemm_pitch_ok(int M, int K, int N, int ldy);
int mpq_gemm_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const void* bias, void* y,
                    float* part, int M, int K, int N, int w_bit, int group_size, int zm, int dtype, const uint16_t* perm,
                    hipStream_t st);
// mpq_util.hip
int mpq_dequant_launch(const int32_t* qw, const void* scales, const void* zeros, const int32_t* g_idx, void* out, int K,
                       int N, int w_bit, int group_size, int asym, int dtype, hipStream_t st);
int mpq_pack_launch(const void* weight, const void* scales, const void* zeros, const int32_t* g_idx, int32_t* out, int K,
                    int N, int w_bit, int group_size, int asym, int dtype, hipStream_t st);
int mpq_sort_rows_launch(const int32_t* qw, const int32_t* perm, int32_t* out, int K, int N, int w_bit, hipStream_t st);
int gather_cols_launch(const void* x, const int32_t* perm, void* out, int M, int K, int elem_bytes, hipStream_t st);
int mpq_grad_input_launch(const void* gy, const int32_t* qw, const void* scales, const void* zeros, const int32_t* g_idx,
                          void* gx, int M, int K, int N, int w_bit, int group_size, int asym, int dtype, hipStream_t st);
// mpq_dense.hip
bool mpq_dense_shape_ok(int K, int N);
size_t mpq_dense_workspace_bytes(int K, int N);
bool mpq_dense_ok(int M, int K, int N);
int mpq_dense_gidx_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const int32_t* g_idx, const void* bias, void* y, void* scratch,
                          int M, int K, int N, int w_bit, int asym, int dtype, hipStream_t st);
// mbwq.hip
size_t mbwq_workspace_bytes(int M, int K, int N, bool exl2);
int mbwq_q4_dequant_launch(const int32_t* qw, const void* scales, const void* zeros, const int16_t* perm, void* out, int K,
                           int N, int bits, int group_size, hipStream_t st);
int mbwq_exl2_shuffle_launch(int32_t* qw, const int* rows6, int K, int N, hipStream_t st, bool inverse);
int mbwq_exl2_dequant_launch(const int32_t* qw, const void* scales, const void* zeros, const int16_t* perm,
                             const int16_t* gmap, const int* rows7, void* out, int K, int N, hipStream_t st);
int mbwq_q4_forward_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const int16_t* perm,
                           void* y, float* part, int M, int K, int N, int bits, int group_size, hipStream_t st);
int mbwq_exl2_forward_launch(const void* x, const int32_t* qw, const void* scales, const void* zeros, const int16_t* perm,
                             const int16_t* gmap, const int* rows7, void* y, float* head, float* part, int M, int K, int N,
                             hipStream_t st);
struct Exl2List;
size_t exl2_list_device_bytes(int n, const bie_exl2_list_entry* e, int M);
int exl2_list_create(Exl2List** out, int n, const bie_exl2_list_entry* e, int M, void* device_mem, size_t device_bytes);
int exl2_list_forward(Exl2List* p, hipStream_t st);
void exl2_list_destroy(Exl2List* p);
bool exl2_group_ok(int n, const bie_exl2_list_entry* e, int M);
size_t exl2_group_workspace_bytes(int n, const bie_exl2_list_entry* e, int M);
int exl2_group_forward(int n, const bie_exl2_list_entry* e, const void* x, int M, float* head, char* body, hipStream_t st);
// binary.hip
int pack_rows_launch(const void* a, uint8_t* out, long n_bytes, int dtype, hipStream_t st);
int pack_cols_launch(const void* w, uint8_t* out, long N, long K, int dtype, hipStream_t st);
int binary_image_pack_launch(const void* w, uint8_t* image, long N, long K, int layout, int dtype, hipStream_t st);
int binary_image_unpack_launch(const uint8_t* image, uint8_t* rowpacked, long N, long K, int layout, hipStream_t st);
bool binary_linear_fused_ok(long M, long N, long K);
int binary_linear_fused_launch(const void* x, const void* bias_a, const uint8_t* wp, const void* sa, const void* sw, void* y, long M,
                               long N, long K, int dtype, int y_f32, hipStream_t st);
size_t binary_conv_taps_lds_bytes(int C, int W, int ks);
size_t binary_conv_taps_workspace_bytes(int B, int C, int H, int W);
int binary_conv_weight_taps_launch(const uint8_t* wpacked, uint32_t* wtaps, int OC, int C, int ks, hipStream_t st);
int binary_conv_taps_launch(const void* x, const uint32_t* wtaps, float* y, void* ws, int B, int C, int H, int W, int OC, int ks,
                            int stride, int pad, int dil, float scale, int dtype, hipStream_t st);
int binary_linear_launch(const uint8_t* xp, const uint8_t* wp, float* y, long M, long N, long K, int w_layout, float scale,
                         hipStream_t st);
// binary_conv_fused.hip
bool binary_conv_fused_ok(int B, int C, int H, int W, int OC, int ks, int stride, int pad, int dil);
size_t binary_conv_weight_lanes_words(int OC, int C, int ks);
int binary_conv_weight_lanes_launch(const uint32_t* wtaps, uint32_t* wl, int OC, int C, int ks, hipStream_t st);
int binary_conv_fused_launch(const void* x, const uint32_t* wl, float* y, int B, int C, int H, int W, int OC, int ks, int stride, int pad, int dil,
                             float scale, int dtype, hipStream_t st);
bool binary_conv_mfma_ok(int B, int C, int H, int W, int OC, int ks, int stride, int pad, int dil);
int binary_conv_mfma_launch(const void* x, const uint8_t* wimg, float* y, int B, int C, int H, int W, int OC, int ks, int stride, int pad, int dil,
                            float scale, int dtype, hipStream_t st);
size_t binary_fp4_image_bytes(long rows, long K);
int binary_fp4_image_launch(const uint8_t* rowpacked, uint8_t* image, long rows, long K, hipStream_t st);
int binary_fp4_image_values_launch(const void* v, const void* bias, uint8_t* image, long rows, long K, int dtype, hipStream_t st);
int binary_fp4_gemm_launch(const uint8_t* ximg, const uint8_t* wimg, void* y, long M, long N, long K, float scale, const void* sa, const void* sw, int dtype,
                           int tile, hipStream_t st);
size_t binary_conv_fp4_workspace_bytes(int B, int C, int H, int W, int ks, int stride, int pad, int dil);
int binary_conv_fp4_launch(const void* x, const uint8_t* wimg, float* y, void* ws, int B, int C, int H, int W, int OC, int ks, int stride, int pad, int dil,
                           float scale, int dtype, int tile, hipStream_t st);
int binary_matmul_batched_launch(const uint8_t* xp, const uint8_t* wp, float* y, long batch, long M, long N, long K, long stride_x,
                                 long stride_w, long stride_y, float scale, hipStream_t st);
size_t binary_conv_workspace_bytes(int B, int C, int H, int W, int OC, int ks, int stride, int pad, int dil);
int binary_conv_launch(const void* x, const uint8_t* wpacked, float* y, void* ws, int B, int C, int H, int W, int OC, int ks,
                       int stride, int pad, int dil, float scale, int dtype, hipStream_t st);
int unpack_u8_scaled_launch(const uint8_t* in, const float* scale, float* out, long n, long packed_dim, hipStream_t st);
int q4_pack_launch(const int32_t* in, int8_t* out, long n_out, hipStream_t st);
int q4_unpack_launch(const int8_t* in, int32_t* out, long n_in, hipStream_t st);
int q4_unpack_scale_launch(const int8_t* in, float* out, long n_in, float scale, hipStream_t st);
// intgemm.hip
int int_gemm_launch(int mode, const void* A, const void* W, void* y, int M, int N, int K, float sa, float sw, int dtype, int batch,
                    long strideA, long strideW, long strideY, hipStream_t st);
int q4_quantize_pack_launch(const void* x, int8_t* out, long n_out, float scale, int dtype, hipStream_t st);
size_t q4_conv2d_workspace_bytes(int B, int H, int W, int C, int OC, int KS, int stride, int pad, int dil);
int q4_conv2d_launch(const int8_t* a_packed, const int8_t* w_packed, void* y, void* workspace, int B, int H, int W, int C, int OC,
                     int KS, int stride, int pad, int dil, float sa, float sw, int dtype, hipStream_t st);
}  // namespace bie

using namespace bie;

static const int GENERIC_M_CHUNK = 32;
// The first 16 KiB (BIE_WS_HEAD_BYTES) of every workspace hold the GEMV's split-K arrival counters (zero on first use, returned to zero by
// the kernel); every other scratch user starts behind them.
static const size_t WS_HEAD = BIE_WS_HEAD_BYTES;

static int validate_mpq(const char* fn, int K, int N, int w_bit, int group_size, int dtype) {
    BIE_REQUIRE(w_bit == 1 || w_bit == 2 || w_bit == 4 || w_bit == 8, BIE_ERR_UNSUPPORTED,
                "%s: w_bit=%d is not supported (1, 2, 4, 8)", fn, w_bit);
    BIE_REQUIRE(dtype == BIE_F16 || dtype == BIE_BF16 || dtype == BIE_F32, BIE_ERR_UNSUPPORTED,
                "%s: dtype=%d is not supported (0=f16, 1=bf16, 2=f32)", fn, dtype);
    BIE_REQUIRE(K > 0 && N > 0 && group_size > 0, BIE_ERR_INVALID_ARG, "%s: K=%d N=%d group_size=%d must be positive", fn, K, N, group_size);
    BIE_REQUIRE(K % (32 / w_bit) == 0, BIE_ERR_INVALID_ARG, "%s: K=%d must be a multiple of %d for w_bit=%d", fn, K, 32 / w_bit, w_bit);
    return BIE_OK;
}

extern "C" {

int bie_version(void) { return BIE_VERSION; }
const char* bie_last_error(void) { return bie::get_error(); }

int bie_status_init(void) { return status_init(); }
unsigned bie_device_status(int clear) { return status_read(clear != 0); }
// include/bie_hip_testing.h: fault injection for the fail-loud tests (not part of the drop-in ABI)
void bie_test_forge_reducer(unsigned tag_skew, int spin_limit) { test_forge_set(tag_skew, spin_limit); }
void bie_test_forge_dependency(int extra) { test_forge_dep_set(extra); }

size_t bie_mpq_list_device_bytes(int n_entries, const bie_mpq_list_entry* entries, int M, int w_bit, int group_size) {
    return mpq_list_device_bytes(n_entries, entries, M, w_bit, group_size);
}
int bie_mpq_list_create(bie_mpq_list_t** plan, int n_entries, const bie_mpq_list_entry* entries, int M, int w_bit, int group_size,
                        int asym, int dtype, void* device_mem, size_t device_bytes) {
    return mpq_list_create(reinterpret_cast<MpqList**>(plan), n_entries, entries, M, w_bit, group_size, asym, dtype, device_mem, device_bytes);
}
int bie_mpq_list_forward(bie_mpq_list_t* plan, void* stream) { return mpq_list_forward(reinterpret_cast<MpqList*>(plan), as_stream(stream)); }
int bie_mpq_list_launches(const bie_mpq_list_t* plan) { return mpq_list_launches(reinterpret_cast<const MpqList*>(plan)); }
int bie_mpq_list_form(const bie_mpq_list_t* plan) { return mpq_list_form(reinterpret_cast<const MpqList*>(plan)); }
void bie_mpq_list_destroy(bie_mpq_list_t* plan) { mpq_list_destroy(reinterpret_cast<MpqList*>(plan)); }

// explicit g_idx that is not a permutation of k // group_size, prefill: per-k dequantise into the fragment image + the dense GEMM (mpq_dense.hip)
static bool gidx_dense_ok(int M, int K, int N, int dtype) { return M > 32 && (dtype == BIE_F16 || dtype == BIE_BF16) && mpq_dense_shape_ok(K, N); }

int bie_mpq_prefill_form(int M, int K, int N) { return (M > 32 && mpq_dense_ok(M, K, N)) ? 1 : 0; }
int bie_mpq_grouped_max_rows(int K, long n_total, int w_bit, int dtype) {
    if (w_bit == 2) return 2;
    if (w_bit != 4) return 0;
    return mpq_lut_rb2_grouped_ok(32, K, n_total, dtype) ? 32 : 16;
}
int bie_mpq_rows_form(int M, int K, int N, int w_bit, int group_size, int dtype) {
    static const int lut_max_m = []() { const char* e = getenv("BIE_LUT_MAX_M"); return e ? atoi(e) : 16; }();
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    if (M <= (lut_max_m >= 16 ? 32 : lut_max_m) && cdiv(N, 64) <= BIE_WS_COUNTERS && (N & 3) == 0 && mpq_gemv_lut_ok(M, K, w_bit, group_size, dtype, false, N)) return 2;
    return bie_mpq_prefill_form(M, K, N);
}
size_t bie_mpq_workspace_bytes_gidx(int M, int K, int N, int w_bit) {
    const size_t base = bie_mpq_workspace_bytes(M, K, N, w_bit);
    if (base == 0 || !gidx_dense_ok(M, K, N, BIE_F16)) return base;
    const size_t img = WS_HEAD + mpq_dense_workspace_bytes(K, N);
    return img > base ? img : base;
}

size_t bie_mpq_workspace_bytes(int M, int K, int N, int w_bit) {
    if (M <= 0 || K <= 0 || N <= 0 || !(w_bit == 1 || w_bit == 2 || w_bit == 4 || w_bit == 8)) return 0;
    size_t a = M <= 32 ? mpq_gemv_workspace_bytes(M, K, N, w_bit) : 0;
    size_t b = mpq_gemm_workspace_bytes(M, K, N);
    const int mc = M < GENERIC_M_CHUNK ? M : GENERIC_M_CHUNK;
    size_t c = (size_t)cdiv(K, 512) * mc * N * sizeof(float);
    size_t r = a > b ? a : b;
    return WS_HEAD + (r > c ? r : c);
}

int bie_mpq_forward(const void* x, const int32_t* qweight, const void* scales, const void* zeros,
                    const int32_t* g_idx, const void* bias, void* y, void* workspace,
                    size_t workspace_bytes, int M, int K, int N, int w_bit, int group_size, int asym,
                    int dtype, void* stream) {
    int rc = validate_mpq("bie_mpq_forward", K, N, w_bit, group_size, dtype);
    if (rc) return rc;
    BIE_REQUIRE(x && qweight && scales && zeros && y, BIE_ERR_INVALID_ARG, "bie_mpq_forward: NULL tensor pointer");
    BIE_REQUIRE(M > 0, BIE_ERR_INVALID_ARG, "bie_mpq_forward: M=%d must be positive", M);
    if (asym) BIE_REQUIRE(N % (32 / w_bit) == 0, BIE_ERR_INVALID_ARG, "bie_mpq_forward: asym needs N %% %d == 0", 32 / w_bit);
    const size_t need = bie_mpq_workspace_bytes(M, K, N, w_bit);
    BIE_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), BIE_ERR_WORKSPACE,
                "bie_mpq_forward: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    rc = status_report("bie_mpq_forward");
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    float* head = reinterpret_cast<float*>(workspace);
    float* part = head + WS_HEAD / sizeof(float);
    const bool has_gidx = g_idx != nullptr;
    // M <= 2: the dot2 GEMV; 3 <= M: the MFMA kernel (its dequant cost does not grow with M; measured faster from M = 3).
    // The GEMV also serves M <= 8 for shapes the MFMA tiling cannot take.
    const bool gemm_ok = mpq_gemm_ok(M, K, N, w_bit, group_size, dtype, has_gidx);
    static const int gemv_max_m = []() { const char* e = getenv("BIE_GEMV_MAX_M"); return e ? atoi(e) : 2; }();  // tuning knob
    // W4 decode and small batches: the table-lookup kernels (mpq_gemv_lut_ok says which M each form takes)
    static const int lut_max_m = []() { const char* e = getenv("BIE_LUT_MAX_M"); return e ? atoi(e) : 16; }();
    if (M <= (lut_max_m >= 16 ? 32 : lut_max_m) && cdiv(N, 64) <= BIE_WS_COUNTERS && (N & 3) == 0 && mpq_gemv_lut_ok(M, K, w_bit, group_size, dtype, has_gidx, N))
        return mpq_gemv_launch(x, qweight, scales, zeros, bias, y, head, M, K, N, w_bit, group_size, asym ? 1 : 0, dtype, nullptr, st);
    if (M <= 8 && (M <= gemv_max_m || !gemm_ok) && mpq_gemv_fast_ok(M, K, N, w_bit, group_size, dtype, has_gidx))
        return mpq_gemv_launch(x, qweight, scales, zeros, bias, y, head, M, K, N, w_bit, group_size, asym ? 1 : 0, dtype, nullptr, st);
    if (gemm_ok)
        return mpq_gemm_launch(x, qweight, scales, zeros, bias, y, part, M, K, N, w_bit, group_size, asym ? 1 : 0, dtype, nullptr, st);
    // explicit irregular g_idx, prefill: the reference materialises the dense weight and calls cuBLAS (mpq_layer.py:59-63); here the
    // per-k dequantise writes the MFMA fragment image and the dense kernel multiplies -- when the caller sized the workspace for it
    // (bie_mpq_workspace_bytes_gidx); a smaller workspace keeps the generic kernel below
    if (has_gidx && gidx_dense_ok(M, K, N, dtype) && workspace_bytes >= WS_HEAD + mpq_dense_workspace_bytes(K, N))
        return mpq_dense_gidx_launch(x, qweight, scales, zeros, g_idx, bias, y, part, M, K, N, w_bit, asym ? 1 : 0, dtype, st);
    // generic path (explicit g_idx / odd shapes / fp32), GENERIC_M_CHUNK rows at a time
    const size_t esz = dtype == BIE_F32 ? 4 : 2;
    for (int m0 = 0; m0 < M; m0 += GENERIC_M_CHUNK) {
        const int mc = (M - m0) < GENERIC_M_CHUNK ? (M - m0) : GENERIC_M_CHUNK;
        rc = mpq_gemv_generic_launch((const char*)x + (size_t)m0 * K * esz, qweight, scales, zeros, g_idx, bias,
                                     (char*)y + (size_t)m0 * N * esz, part, mc, K, N, w_bit, group_size, asym, dtype, st);
        if (rc) return rc;
    }
    return BIE_OK;
}

// y is a column range of a wider row-major destination: row m of the result goes to y + m * ldy (elements).  Served by the MFMA GEMM range
// (the prefill shapes a column-sharded layer runs); anything else reports BIE_ERR_UNSUPPORTED and the caller keeps its copy.
int bie_mpq_forward_pitched(const void* x, const int32_t* qweight, const void* scales, const void* zeros, const void* bias, void* y, int ldy,
                            void* workspace, size_t workspace_bytes, int M, int K, int N, int w_bit, int group_size, int asym, int dtype,
                            void* stream) {
    if (ldy == N)
        return bie_mpq_forward(x, qweight, scales, zeros, nullptr, bias, y, workspace, workspace_bytes, M, K, N, w_bit, group_size, asym, dtype, stream);
    int rc = validate_mpq("bie_mpq_forward_pitched", K, N, w_bit, group_size, dtype);
    if (rc) return rc;
    BIE_REQUIRE(x && qweight && scales && zeros && y && M > 0, BIE_ERR_INVALID_ARG, "bie_mpq_forward_pitched: bad argument");
    if (asym) BIE_REQUIRE(N % (32 / w_bit) == 0, BIE_ERR_INVALID_ARG, "bie_mpq_forward_pitched: asym needs N %% %d == 0", 32 / w_bit);
    static const int lut_max_m = []() { const char* e = getenv("BIE_LUT_MAX_M"); return e ? atoi(e) : 16; }();
    BIE_REQUIRE(M > lut_max_m && M > 8 && mpq_gemm_ok(M, K, N, w_bit, group_size, dtype, false) && mpq_gemm_pitch_ok(M, K, N, ldy) &&
                    (reinterpret_cast<uintptr_t>(y) & 7) == 0,
                BIE_ERR_UNSUPPORTED, "bie_mpq_forward_pitched: M=%d K=%d N=%d ldy=%d is outside the range of the MFMA GEMM's pitched epilogue (M > %d, ldy %% 4 == 0, ldy >= N, no split-K plan, y 8-byte aligned)",
                M, K, N, ldy, lut_max_m > 8 ? lut_max_m : 8);
    const size_t need = bie_mpq_workspace_bytes(M, K, N, w_bit);
    BIE_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), BIE_ERR_WORKSPACE, "bie_mpq_forward_pitched: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    rc = status_report("bie_mpq_forward_pitched");
    if (rc) return rc;
    float* part = reinterpret_cast<float*>(workspace) + WS_HEAD / sizeof(float);
    return mpq_gemm_launch_ld(x, qweight, scales, zeros, bias, y, part, M, K, N, w_bit, group_size, asym ? 1 : 0, dtype, nullptr, as_stream(stream), ldy);
}

int bie_workspace_init(void* workspace, size_t workspace_bytes, void* stream) {
    BIE_REQUIRE(workspace && workspace_bytes >= WS_HEAD, BIE_ERR_WORKSPACE, "bie_workspace_init: a workspace holds at least %zu bytes", WS_HEAD);
    const hipError_t e = hipMemsetAsync(workspace, 0, WS_HEAD, as_stream(stream));
    BIE_REQUIRE(e == hipSuccess, BIE_ERR_HIP, "bie_workspace_init: hipMemsetAsync: %s", hipGetErrorString(e));
    return BIE_OK;
}

static int grouped_tiles(int n_sets, const int* N) {
    int tiles = 0;
    for (int i = 0; i < n_sets; i++) tiles += cdiv(N[i], 64);
    return tiles;
}

size_t bie_mpq_grouped_workspace_bytes(int n_sets, const int* N, int M, int K, int w_bit) {
    if (n_sets <= 0 || n_sets > BIE_MAX_GROUPED_SETS || !N || M <= 0 || K <= 0) return 0;
    size_t need = 0;
    for (int i = 0; i < n_sets; i++) {
        const size_t b = bie_mpq_workspace_bytes(M, K, N[i], w_bit);
        if (b > need) need = b;
    }
    if ((w_bit == 4 && M <= 32) || (w_bit == 2 && M <= 2)) {  // 17 .. 32 rows: the two-row-block instance (taken on measured sets only, sized always)
        const int tiles = grouped_tiles(n_sets, N);
        for (int gs = 32; gs <= 256; gs *= 2)
            if (K % gs == 0) {
                const size_t b = WS_HEAD + mpq_gemv_lut_part_floats(M, K, gs, tiles, w_bit) * sizeof(float);
                if (b > need) need = b;
            }
    }
    return need;
}

int bie_mpq_forward_grouped(const void* x, int n_sets, const int32_t* const* qweight, const void* const* scales,
                            const void* const* zeros, const void* const* bias, void* const* y, const int* N,
                            void* workspace, size_t workspace_bytes, int M, int K, int w_bit, int group_size,
                            int asym, int dtype, void* stream) {
    BIE_REQUIRE(n_sets > 0 && n_sets <= BIE_MAX_GROUPED_SETS, BIE_ERR_INVALID_ARG, "bie_mpq_forward_grouped: n_sets=%d must be 1..%d", n_sets, BIE_MAX_GROUPED_SETS);
    BIE_REQUIRE(x && qweight && scales && zeros && y && N, BIE_ERR_INVALID_ARG, "bie_mpq_forward_grouped: NULL argument");
    for (int i = 0; i < n_sets; i++) {
        int rc = validate_mpq("bie_mpq_forward_grouped", K, N[i], w_bit, group_size, dtype);
        if (rc) return rc;
        BIE_REQUIRE(qweight[i] && scales[i] && zeros[i] && y[i], BIE_ERR_INVALID_ARG, "bie_mpq_forward_grouped: NULL tensor pointer in set %d", i);
        if (asym) BIE_REQUIRE(N[i] % (32 / w_bit) == 0, BIE_ERR_INVALID_ARG, "bie_mpq_forward_grouped: asym needs N %% %d == 0", 32 / w_bit);
    }
    BIE_REQUIRE(M > 0, BIE_ERR_INVALID_ARG, "bie_mpq_forward_grouped: M=%d must be positive", M);
    const size_t need = bie_mpq_grouped_workspace_bytes(n_sets, N, M, K, w_bit);
    BIE_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), BIE_ERR_WORKSPACE,
                "bie_mpq_forward_grouped: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    int src = status_report("bie_mpq_forward_grouped");
    if (src) return src;
    const int tiles = grouped_tiles(n_sets, N);
    bool n4 = true;  // the matrix-pipe form loads four adjacent columns with one 16-byte load
    for (int i = 0; i < n_sets; i++) n4 = n4 && (N[i] & 3) == 0;
    long n_total = 0;
    for (int i = 0; i < n_sets; i++) n_total += N[i];
    // 17 .. 32 rows: the two-row-block instance where it measured ahead of the members' own calls (mpq_lut_rb2_grouped_ok); the shape checks are those of a 16-row call
    const bool rb2 = w_bit == 4 && M > 16 && M <= 32 && n_sets > 1 && mpq_lut_rb2_grouped_ok(M, K, n_total, dtype) && mpq_gemv_lut_ok(16, K, w_bit, group_size, dtype, false);
    if (n4 && tiles <= BIE_WS_COUNTERS && (rb2 || mpq_gemv_lut_ok(M, K, w_bit, group_size, dtype, false))) {
        float* head = reinterpret_cast<float*>(workspace);
        return mpq_gemv_lut_launch(n_sets, qweight, scales, zeros, bias, y, N, x, reinterpret_cast<unsigned*>(head) + BIE_WS_GEN_OFFSET,
                                   head + WS_HEAD / sizeof(float), M, K, group_size, asym ? 1 : 0, dtype, as_stream(stream), w_bit);
    }
    for (int i = 0; i < n_sets; i++) {  // every other case: one launch per set (same results)
        int rc = bie_mpq_forward(x, qweight[i], scales[i], zeros[i], nullptr, bias ? bias[i] : nullptr, y[i], workspace,
                                 workspace_bytes, M, K, N[i], w_bit, group_size, asym, dtype, stream);
        if (rc) return rc;
    }
    return BIE_OK;
}

int bie_mpq_dequant(const int32_t* qweight, const void* scales, const void* zeros, const int32_t* g_idx, void* out,
                    int K, int N, int w_bit, int group_size, int asym, int dtype, void* stream) {
    int rc = validate_mpq("bie_mpq_dequant", K, N, w_bit, group_size, dtype);
    if (rc) return rc;
    BIE_REQUIRE(qweight && scales && zeros && out, BIE_ERR_INVALID_ARG, "bie_mpq_dequant: NULL tensor pointer");
    return mpq_dequant_launch(qweight, scales, zeros, g_idx, out, K, N, w_bit, group_size, asym, dtype, as_stream(stream));
}

int bie_mpq_pack(const void* weight, const void* scales, const void* zeros, const int32_t* g_idx, int32_t* out, int K,
                 int N, int w_bit, int group_size, int asym, int dtype, void* stream) {
    int rc = validate_mpq("bie_mpq_pack", K, N, w_bit, group_size, dtype);
    if (rc) return rc;
    BIE_REQUIRE(weight && scales && zeros && out, BIE_ERR_INVALID_ARG, "bie_mpq_pack: NULL tensor pointer");
    return mpq_pack_launch(weight, scales, zeros, g_idx, out, K, N, w_bit, group_size, asym, dtype, as_stream(stream));
}

int bie_mpq_grad_input(const void* grad_y, const int32_t* qweight, const void* scales, const void* zeros,
                       const int32_t* g_idx, void* grad_x, int M, int K, int N, int w_bit, int group_size, int asym,
                       int dtype, void* stream) {
    int rc = validate_mpq("bie_mpq_grad_input", K, N, w_bit, group_size, dtype);
    if (rc) return rc;
    BIE_REQUIRE(grad_y && qweight && scales && zeros && grad_x && M > 0, BIE_ERR_INVALID_ARG, "bie_mpq_grad_input: bad argument");
    return mpq_grad_input_launch(grad_y, qweight, scales, zeros, g_idx, grad_x, M, K, N, w_bit, group_size, asym, dtype, as_stream(stream));
}

int bie_mpq_sort_rows(const int32_t* qweight, const int32_t* perm, int32_t* out, int K, int N, int w_bit, void* stream) {
    BIE_REQUIRE(w_bit == 1 || w_bit == 2 || w_bit == 4 || w_bit == 8, BIE_ERR_UNSUPPORTED, "bie_mpq_sort_rows: w_bit %d not supported", w_bit);
    BIE_REQUIRE(K > 0 && N > 0 && K % (32 / w_bit) == 0, BIE_ERR_INVALID_ARG, "bie_mpq_sort_rows: K=%d must be a positive multiple of %d, N=%d positive", K, 32 / w_bit, N);
    BIE_REQUIRE(qweight && perm && out && qweight != out, BIE_ERR_INVALID_ARG, "bie_mpq_sort_rows: NULL or aliased tensor pointer");
    return mpq_sort_rows_launch(qweight, perm, out, K, N, w_bit, as_stream(stream));
}

int bie_gather_cols(const void* x, const int32_t* perm, void* out, int M, int K, int dtype, void* stream) {
    BIE_REQUIRE(dtype == BIE_F16 || dtype == BIE_BF16 || dtype == BIE_F32, BIE_ERR_UNSUPPORTED, "bie_gather_cols: dtype %d not supported", dtype);
    BIE_REQUIRE(M >= 0 && K > 0, BIE_ERR_INVALID_ARG, "bie_gather_cols: bad shape M=%d K=%d", M, K);
    if (M == 0) return BIE_OK;
    BIE_REQUIRE(x && perm && out && x != out, BIE_ERR_INVALID_ARG, "bie_gather_cols: NULL or aliased tensor pointer");
    return gather_cols_launch(x, perm, out, M, K, dtype == BIE_F32 ? 4 : 2, as_stream(stream));
}


// ---------------------------------------------------------------------------------------------- MBWQ
int bie_mbwq_rows(const int16_t* q_groups_host, int groups, int K, int* rows7_host) {
    BIE_REQUIRE(q_groups_host && rows7_host && groups > 0 && K > 0, BIE_ERR_INVALID_ARG, "bie_mbwq_rows: bad argument");
    int r[6] = {0, 0, 0, 0, 0, 0}, kp = 0, row = 0;
    for (int i = 0; i < groups; i++) {
        const int bits = (uint16_t)q_groups_host[2 * i];
        int band;
        switch (bits) {
            case 8: band = 0; break;
            case 6: band = 1; break;
            case 5: band = 2; break;
            case 4: band = 3; break;
            case 3: band = 4; break;
            case 2: band = 5; break;
            default: set_error("bie_mbwq_rows: group %d has unsupported bit width %d", i, bits); return BIE_ERR_UNSUPPORTED;
        }
        kp |= 1 << (bits - 1);
        int rows;
        if (i < groups - 1) rows = ((uint16_t)q_groups_host[2 * i + 3] - (uint16_t)q_groups_host[2 * i + 1]) * 32 / bits;
        else rows = K - row;
        r[band] += rows;
        row += rows;
    }
    for (int b = 1; b < 6; b++) r[b] += r[b - 1];
    for (int b = 0; b < 6; b++) rows7_host[b] = r[b];
    rows7_host[6] = kp;
    return BIE_OK;
}

size_t bie_mbwq_workspace_bytes(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    return WS_HEAD + mbwq_workspace_bytes(M, K, N, true);
}

size_t bie_mbwq_q4_workspace_bytes(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    return WS_HEAD + mbwq_workspace_bytes(M, K, N, false);
}

int bie_mbwq_q4_dequant(const int32_t* qweight, const void* scales, const void* zeros, const int16_t* q_perm, void* out, int K,
                        int N, int bits, int group_size, void* stream) {
    BIE_REQUIRE(qweight && scales && zeros && out, BIE_ERR_INVALID_ARG, "bie_mbwq_q4_dequant: NULL tensor pointer");
    BIE_REQUIRE(bits == 2 || bits == 4, BIE_ERR_UNSUPPORTED, "bie_mbwq_q4_dequant: weight bit width %d has not been supported yet", bits);
    BIE_REQUIRE(K > 0 && N > 0 && group_size > 0 && K % (32 / bits) == 0, BIE_ERR_INVALID_ARG, "bie_mbwq_q4_dequant: bad shape");
    return mbwq_q4_dequant_launch(qweight, scales, zeros, q_perm, out, K, N, bits, group_size, as_stream(stream));
}

// The extended table: band ends + flags as bie_mbwq_rows, then per band the first group and log2(chunks per group).  REGULAR when
// every band's groups hold the same power-of-two number of whole chunks (a shorter last group is fine: the index is a shift).
int bie_mbwq_exl2_table(const int16_t* q_groups_host, int groups, int K, int* rows_host) {  // host only: the table bie_mbwq_exl2_shuffle returns
    BIE_REQUIRE(rows_host, BIE_ERR_INVALID_ARG, "bie_mbwq_exl2_table: bad argument");
    int rc = bie_mbwq_rows(q_groups_host, groups, K, rows_host);
    if (rc) return rc;
    BIE_REQUIRE(K % 32 == 0, BIE_ERR_UNSUPPORTED, "bie_mbwq_exl2_table: K=%d must be a multiple of 32", K);
    for (int b = 0; b < 6; b++)
        BIE_REQUIRE(rows_host[b] % 32 == 0, BIE_ERR_UNSUPPORTED, "bie_mbwq_exl2_table: band boundary rows[%d]=%d must be a multiple of 32", b, rows_host[b]);
    bool regular = true;
    int gfirst[6] = {0, 0, 0, 0, 0, 0}, glog[6] = {0, 0, 0, 0, 0, 0}, gk[6] = {0, 0, 0, 0, 0, 0}, seen[6] = {0, 0, 0, 0, 0, 0};
    int row = 0;
    for (int i = 0; i < groups; i++) {
        const int bits = (uint16_t)q_groups_host[2 * i];
        const int band = bits == 8 ? 0 : (bits == 6 ? 1 : (bits == 5 ? 2 : (bits == 4 ? 3 : (bits == 3 ? 4 : 5))));
        const int k = i < groups - 1 ? ((uint16_t)q_groups_host[2 * i + 3] - (uint16_t)q_groups_host[2 * i + 1]) * 32 / bits : K - row;
        const int band_end = rows_host[band];
        const bool last_of_band = row + k == band_end;
        if (!seen[band]) {
            seen[band] = 1;
            gfirst[band] = i;
            gk[band] = k;
            int lg = 0;
            while ((32 << lg) < k) lg++;
            glog[band] = lg;
            if ((32 << lg) != k && !last_of_band) regular = false;  // a lone (last) group of any length is fine: every chunk maps to it
            if ((32 << lg) != k && last_of_band) glog[band] = 30;  // one group in the band: (c - cbs) >> 30 == 0
        } else if (k != gk[band] && !(last_of_band && k < gk[band])) {
            regular = false;
        }
        if (k % 32 != 0 || k <= 0) regular = false;
        row += k;
    }
    for (int b = 0; b < 6; b++) {
        rows_host[BIE_EXL2_ROWS_GFIRST + b] = gfirst[b];
        rows_host[BIE_EXL2_ROWS_GLOG + b] = glog[b];
    }
    rows_host[6] |= BIE_EXL2_ROWS_SHUFFLED | (regular ? BIE_EXL2_ROWS_REGULAR : 0);
    rows_host[BIE_EXL2_ROWS_LEN - 1] = BIE_EXL2_ROWS_TAG;
    return BIE_OK;
}

int bie_mbwq_exl2_shuffle(int32_t* qweight, const int16_t* q_groups_host, int groups, int K, int N, int* rows_host, void* stream) {
    BIE_REQUIRE(qweight && rows_host && N > 0, BIE_ERR_INVALID_ARG, "bie_mbwq_exl2_shuffle: bad argument");
    int rc = bie_mbwq_exl2_table(q_groups_host, groups, K, rows_host);
    if (rc) return rc;
    return mbwq_exl2_shuffle_launch(qweight, rows_host, K, N, as_stream(stream), false);
}

static int check_rows(const char* fn, const int* rows7, int K) {
    BIE_REQUIRE(rows7, BIE_ERR_INVALID_ARG, "%s: rows table is NULL", fn);
    BIE_REQUIRE((rows7[6] & BIE_EXL2_ROWS_SHUFFLED), BIE_ERR_INVALID_ARG,
                "%s: the band table does not carry the SHUFFLED mark: qweight must pass through bie_mbwq_exl2_shuffle (q_linear_cuda.mbwq_trans_qweight) once, and its table be used", fn);
    BIE_REQUIRE(rows7[BIE_EXL2_ROWS_LEN - 1] == BIE_EXL2_ROWS_TAG, BIE_ERR_INVALID_ARG, "%s: the band table is not the %d-int table of bie_mbwq_exl2_shuffle", fn, BIE_EXL2_ROWS_LEN);
    int prev = 0;
    for (int b = 0; b < 6; b++) {
        BIE_REQUIRE(rows7[b] >= prev && rows7[b] % 32 == 0, BIE_ERR_UNSUPPORTED,
                    "%s: band boundary rows[%d]=%d must be a non-decreasing multiple of 32", fn, b, rows7[b]);
        prev = rows7[b];
    }
    BIE_REQUIRE(rows7[5] == K, BIE_ERR_INVALID_ARG, "%s: rows[5]=%d must equal K=%d", fn, rows7[5], K);
    return BIE_OK;
}

int bie_mbwq_exl2_unshuffle(int32_t* qweight, const int* rows_host, int K, int N, void* stream) {
    BIE_REQUIRE(qweight && N > 0 && K > 0, BIE_ERR_INVALID_ARG, "bie_mbwq_exl2_unshuffle: bad argument");
    int rc = check_rows("bie_mbwq_exl2_unshuffle", rows_host, K);
    if (rc) return rc;
    return mbwq_exl2_shuffle_launch(qweight, rows_host, K, N, as_stream(stream), true);
}

int bie_mbwq_exl2_dequant(const int32_t* qweight, const void* scales, const void* zeros, const int16_t* q_perm,
                          const int16_t* q_group_map, const int* rows7_host, void* out, int K, int N, int groups,
                          void* stream) {
    BIE_REQUIRE(qweight && scales && zeros && q_group_map && out && K > 0 && N > 0, BIE_ERR_INVALID_ARG, "bie_mbwq_exl2_dequant: bad argument");
    int rc = check_rows("bie_mbwq_exl2_dequant", rows7_host, K);
    if (rc) return rc;
    return mbwq_exl2_dequant_launch(qweight, scales, zeros, q_perm, q_group_map, rows7_host, out, K, N, as_stream(stream));
}

int bie_mbwq_q4_forward(const void* x, const int32_t* qweight, const void* scales, const void* zeros, const int16_t* q_perm,
                        void* y, void* workspace, size_t workspace_bytes, int M, int K, int N, int bits, int group_size,
                        void* stream) {
    BIE_REQUIRE(x && qweight && scales && zeros && y && M > 0 && K > 0 && N > 0 && group_size > 0, BIE_ERR_INVALID_ARG, "bie_mbwq_q4_forward: bad argument");
    BIE_REQUIRE(bits == 2 || bits == 4, BIE_ERR_UNSUPPORTED, "bie_mbwq_q4_forward: weight bit width %d has not been supported yet", bits);
    const size_t need = WS_HEAD + mbwq_workspace_bytes(M, K, N, false);
    BIE_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), BIE_ERR_WORKSPACE, "bie_mbwq_q4_forward: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return mbwq_q4_forward_launch(x, qweight, scales, zeros, q_perm, y, (float*)workspace, M, K, N, bits, group_size, as_stream(stream));
}

int bie_mbwq_exl2_forward(const void* x, const int32_t* qweight, const void* scales, const void* zeros, const int16_t* q_perm,
                          const int16_t* q_group_map, const int* rows7_host, void* y, void* workspace, size_t workspace_bytes,
                          int M, int K, int N, int groups, void* stream) {
    BIE_REQUIRE(x && qweight && scales && zeros && q_group_map && y && M > 0 && K > 0 && N > 0, BIE_ERR_INVALID_ARG, "bie_mbwq_exl2_forward: bad argument");
    int rc = check_rows("bie_mbwq_exl2_forward", rows7_host, K);
    if (rc) return rc;
    const size_t need = WS_HEAD + mbwq_workspace_bytes(M, K, N, true);
    BIE_REQUIRE(workspace && workspace_bytes >= need, BIE_ERR_WORKSPACE, "bie_mbwq_exl2_forward: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    rc = status_report("bie_mbwq_exl2_forward");
    if (rc) return rc;
    return mbwq_exl2_forward_launch(x, qweight, scales, zeros, q_perm, q_group_map, rows7_host, y, (float*)workspace, (float*)workspace + WS_HEAD / sizeof(float), M, K, N, as_stream(stream));
}

size_t bie_mbwq_exl2_list_device_bytes(int n_entries, const bie_exl2_list_entry* entries, int M) { return exl2_list_device_bytes(n_entries, entries, M); }
int bie_mbwq_exl2_list_create(bie_exl2_list_t** plan, int n_entries, const bie_exl2_list_entry* entries, int M, void* device_mem, size_t device_bytes) {
    return exl2_list_create(reinterpret_cast<Exl2List**>(plan), n_entries, entries, M, device_mem, device_bytes);
}
int bie_mbwq_exl2_list_forward(bie_exl2_list_t* plan, void* stream) { return exl2_list_forward(reinterpret_cast<Exl2List*>(plan), as_stream(stream)); }
void bie_mbwq_exl2_list_destroy(bie_exl2_list_t* plan) { exl2_list_destroy(reinterpret_cast<Exl2List*>(plan)); }

size_t bie_mbwq_exl2_grouped_workspace_bytes(int n_members, const bie_exl2_list_entry* members, int M) {
    const size_t b = exl2_group_workspace_bytes(n_members, members, M);
    return b ? WS_HEAD + b : 0;
}
int bie_mbwq_exl2_forward_grouped(const void* x, int M, int n_members, const bie_exl2_list_entry* members, void* workspace, size_t workspace_bytes, void* stream) {
    BIE_REQUIRE(x && members && n_members >= 1, BIE_ERR_INVALID_ARG, "bie_mbwq_exl2_forward_grouped: bad argument");
    BIE_REQUIRE(exl2_group_ok(n_members, members, M), BIE_ERR_UNSUPPORTED,
                "bie_mbwq_exl2_forward_grouped: takes 1..16 rows of x and 1..8 members with K %% 32 == 0, tables from bie_mbwq_exl2_shuffle carrying the REGULAR mark, at most %d column blocks in all", BIE_WS_COUNTERS);
    for (int i = 0; i < n_members; i++)
        BIE_REQUIRE(members[i].qweight && members[i].scales && members[i].zeros && members[i].q_group_map && members[i].y, BIE_ERR_INVALID_ARG,
                    "bie_mbwq_exl2_forward_grouped: NULL tensor pointer in member %d", i);
    const size_t need = bie_mbwq_exl2_grouped_workspace_bytes(n_members, members, M);
    BIE_REQUIRE(workspace && workspace_bytes >= need, BIE_ERR_WORKSPACE, "bie_mbwq_exl2_forward_grouped: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    int rc = status_report("bie_mbwq_exl2_forward_grouped");
    if (rc) return rc;
    return exl2_group_forward(n_members, members, x, M, (float*)workspace, (char*)workspace + WS_HEAD, as_stream(stream));
}

// ---------------------------------------------------------------------------------------------- binary
int bie_binary_pack_rows_u8(const void* a, uint8_t* out, long rows, long K, int dtype, void* stream) {
    BIE_REQUIRE(a && out && rows > 0 && K > 0 && K % 8 == 0, BIE_ERR_INVALID_ARG, "bie_binary_pack_rows_u8: rows=%ld K=%ld (K %% 8 == 0 required)", rows, K);
    BIE_REQUIRE(dtype >= 0 && dtype <= 3, BIE_ERR_UNSUPPORTED, "bie_binary_pack_rows_u8: dtype %d", dtype);
    return pack_rows_launch(a, out, rows * (K / 8), dtype, as_stream(stream));
}

int bie_binary_pack_cols_u8(const void* w, uint8_t* out, long N, long K, int dtype, void* stream) {
    BIE_REQUIRE(w && out && N > 0 && K > 0 && K % 8 == 0, BIE_ERR_INVALID_ARG, "bie_binary_pack_cols_u8: N=%ld K=%ld (K %% 8 == 0 required)", N, K);
    BIE_REQUIRE(dtype >= 0 && dtype <= 3, BIE_ERR_UNSUPPORTED, "bie_binary_pack_cols_u8: dtype %d", dtype);
    return pack_cols_launch(w, out, N, K, dtype, as_stream(stream));
}

static int check_image_shape(const char* fn, long N, long K, int btc) {
    BIE_REQUIRE(N > 0 && K > 0, BIE_ERR_INVALID_ARG, "%s: N=%ld K=%ld must be positive", fn, N, K);
    if (btc) BIE_REQUIRE(K % 128 == 0 && N % 8 == 0, BIE_ERR_UNSUPPORTED, "%s: the BTC32 image needs K %% 128 == 0 and N %% 8 == 0 (K=%ld N=%ld)", fn, K, N);
    else BIE_REQUIRE(K % 32 == 0 && N % 32 == 0, BIE_ERR_UNSUPPORTED, "%s: the BSTC32 image needs K %% 32 == 0 and N %% 32 == 0 (K=%ld N=%ld)", fn, K, N);
    return BIE_OK;
}

int bie_binary_pack_btc32(const void* w, uint8_t* image, long N, long K, int dtype, void* stream) {
    int rc = check_image_shape("bie_binary_pack_btc32", N, K, 1);
    if (rc) return rc;
    BIE_REQUIRE(w && image && dtype >= 0 && dtype <= 3, BIE_ERR_INVALID_ARG, "bie_binary_pack_btc32: bad argument");
    return binary_image_pack_launch(w, image, N, K, 0, dtype, as_stream(stream));
}
int bie_binary_pack_bstc32(const void* w, uint8_t* image, long N, long K, int dtype, void* stream) {
    int rc = check_image_shape("bie_binary_pack_bstc32", N, K, 0);
    if (rc) return rc;
    BIE_REQUIRE(w && image && dtype >= 0 && dtype <= 3, BIE_ERR_INVALID_ARG, "bie_binary_pack_bstc32: bad argument");
    return binary_image_pack_launch(w, image, N, K, 1, dtype, as_stream(stream));
}
int bie_binary_unpack_btc32(const uint8_t* image, uint8_t* rowpacked, long N, long K, void* stream) {
    int rc = check_image_shape("bie_binary_unpack_btc32", N, K, 1);
    if (rc) return rc;
    BIE_REQUIRE(image && rowpacked, BIE_ERR_INVALID_ARG, "bie_binary_unpack_btc32: NULL pointer");
    return binary_image_unpack_launch(image, rowpacked, N, K, 0, as_stream(stream));
}
int bie_binary_unpack_bstc32(const uint8_t* image, uint8_t* rowpacked, long N, long K, void* stream) {
    int rc = check_image_shape("bie_binary_unpack_bstc32", N, K, 0);
    if (rc) return rc;
    BIE_REQUIRE(image && rowpacked, BIE_ERR_INVALID_ARG, "bie_binary_unpack_bstc32: NULL pointer");
    return binary_image_unpack_launch(image, rowpacked, N, K, 1, as_stream(stream));
}

int bie_binary_linear_forward(const uint8_t* xpacked, const uint8_t* wpacked, float* y, long M, long N, long K, int w_layout,
                              float scale, void* stream) {
    BIE_REQUIRE(xpacked && wpacked && y && M > 0 && N > 0 && K > 0 && K % 8 == 0, BIE_ERR_INVALID_ARG, "bie_binary_linear_forward: M=%ld N=%ld K=%ld (K %% 8 == 0 required)", M, N, K);
    BIE_REQUIRE(w_layout == 0 || w_layout == 1, BIE_ERR_INVALID_ARG, "bie_binary_linear_forward: w_layout %d", w_layout);
    return binary_linear_launch(xpacked, wpacked, y, M, N, K, w_layout, scale, as_stream(stream));
}

size_t bie_binary_fp4_image_bytes(long rows, long K) { return rows > 0 && K > 0 ? binary_fp4_image_bytes(rows, K) : 0; }

int bie_binary_fp4_image(const uint8_t* rowpacked, uint8_t* image, long rows, long K, void* stream) {
    BIE_REQUIRE(rowpacked && image && rows > 0 && K > 0 && K % 8 == 0, BIE_ERR_INVALID_ARG, "bie_binary_fp4_image: rows=%ld K=%ld (K %% 8 == 0 required)", rows, K);
    BIE_REQUIRE((reinterpret_cast<uintptr_t>(image) & 15) == 0, BIE_ERR_INVALID_ARG, "bie_binary_fp4_image: image must be 16-byte aligned");
    return binary_fp4_image_launch(rowpacked, image, rows, K, as_stream(stream));
}

int bie_binary_fp4_image_from_values(const void* values, const void* bias, uint8_t* image, long rows, long K, int dtype, void* stream) {
    BIE_REQUIRE(values && image && rows > 0 && K > 0, BIE_ERR_INVALID_ARG, "bie_binary_fp4_image_from_values: rows=%ld K=%ld", rows, K);
    BIE_REQUIRE(dtype >= 0 && dtype <= 3, BIE_ERR_UNSUPPORTED, "bie_binary_fp4_image_from_values: dtype %d", dtype);
    BIE_REQUIRE(!(bias && dtype == 3), BIE_ERR_INVALID_ARG, "bie_binary_fp4_image_from_values: int8 sign carriers take no bias");
    BIE_REQUIRE((reinterpret_cast<uintptr_t>(image) & 15) == 0, BIE_ERR_INVALID_ARG, "bie_binary_fp4_image_from_values: image must be 16-byte aligned");
    return binary_fp4_image_values_launch(values, bias, image, rows, K, dtype, as_stream(stream));
}

int bie_binary_linear_forward_fp4(const uint8_t* ximage, const uint8_t* wimage, float* y, long M, long N, long K, float scale, void* stream) {
    BIE_REQUIRE(ximage && wimage && y && M > 0 && N > 0 && K > 0, BIE_ERR_INVALID_ARG, "bie_binary_linear_forward_fp4: M=%ld N=%ld K=%ld", M, N, K);
    BIE_REQUIRE(K < (1L << 24) && M < (1L << 31) && N < (1L << 31), BIE_ERR_UNSUPPORTED, "bie_binary_linear_forward_fp4: K=%ld beyond the exact range of the fp32 accumulator (2^24)", K);
    BIE_REQUIRE(((reinterpret_cast<uintptr_t>(ximage) | reinterpret_cast<uintptr_t>(wimage)) & 15) == 0, BIE_ERR_INVALID_ARG, "bie_binary_linear_forward_fp4: images must be 16-byte aligned");
    const char* et = getenv("BIE_FP4_TILE");  // tuning / A-B only: 128 or 256
    return binary_fp4_gemm_launch(ximage, wimage, y, M, N, K, scale, nullptr, nullptr, -1, et ? atoi(et) : 0, as_stream(stream));
}

int bie_binary_linear_layer_fp4(const uint8_t* ximage, const uint8_t* wimage, const void* scale_a, const void* scale_w, void* y, long M, long N, long K,
                                int dtype, void* stream) {
    BIE_REQUIRE(ximage && wimage && y && M > 0 && N > 0 && K > 0, BIE_ERR_INVALID_ARG, "bie_binary_linear_layer_fp4: M=%ld N=%ld K=%ld", M, N, K);
    BIE_REQUIRE(dtype >= 0 && dtype <= 2, BIE_ERR_UNSUPPORTED, "bie_binary_linear_layer_fp4: dtype %d", dtype);
    BIE_REQUIRE(K < (1L << 24) && M < (1L << 31) && N < (1L << 31), BIE_ERR_UNSUPPORTED, "bie_binary_linear_layer_fp4: K=%ld beyond the exact range of the fp32 accumulator (2^24)", K);
    BIE_REQUIRE(((reinterpret_cast<uintptr_t>(ximage) | reinterpret_cast<uintptr_t>(wimage)) & 15) == 0, BIE_ERR_INVALID_ARG, "bie_binary_linear_layer_fp4: images must be 16-byte aligned");
    const char* et = getenv("BIE_FP4_TILE");
    return binary_fp4_gemm_launch(ximage, wimage, y, M, N, K, 1.0f, scale_a, scale_w, dtype, et ? atoi(et) : 0, as_stream(stream));
}

size_t bie_binary_conv2d_fp4_workspace_bytes(int B, int C, int H, int W, int ksize, int stride, int pad, int dilation) {
    if (B <= 0 || C <= 0 || C % 32 || H <= 0 || W <= 0 || ksize <= 0 || stride <= 0 || dilation <= 0 || pad < 0) return 0;
    return binary_conv_fp4_workspace_bytes(B, C, H, W, ksize, stride, pad, dilation);
}

int bie_binary_conv2d_forward_fp4(const void* x, const uint8_t* wimage, float* y, void* workspace, size_t workspace_bytes, int B, int C, int H, int W, int OC,
                                  int ksize, int stride, int pad, int dilation, float scale, int dtype, void* stream) {
    BIE_REQUIRE(x && wimage && y && workspace, BIE_ERR_INVALID_ARG, "bie_binary_conv2d_forward_fp4: NULL tensor pointer");
    BIE_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && OC > 0 && ksize > 0 && stride > 0 && dilation > 0 && pad >= 0, BIE_ERR_INVALID_ARG, "bie_binary_conv2d_forward_fp4: bad geometry");
    BIE_REQUIRE(C % 32 == 0, BIE_ERR_UNSUPPORTED, "bie_binary_conv2d_forward_fp4: C=%d must be a multiple of 32 (use bie_binary_conv2d_forward_taps)", C);
    BIE_REQUIRE(dtype >= 0 && dtype <= 2, BIE_ERR_UNSUPPORTED, "bie_binary_conv2d_forward_fp4: dtype %d", dtype);
    const size_t need = binary_conv_fp4_workspace_bytes(B, C, H, W, ksize, stride, pad, dilation);
    BIE_REQUIRE(need > 0, BIE_ERR_INVALID_ARG, "bie_binary_conv2d_forward_fp4: empty output");
    BIE_REQUIRE(workspace_bytes >= need, BIE_ERR_WORKSPACE, "bie_binary_conv2d_forward_fp4: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    BIE_REQUIRE(((reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(wimage)) & 15) == 0, BIE_ERR_INVALID_ARG, "bie_binary_conv2d_forward_fp4: workspace and image must be 16-byte aligned");
    BIE_REQUIRE((long)ksize * ksize * C < (1L << 24), BIE_ERR_UNSUPPORTED, "bie_binary_conv2d_forward_fp4: C*k*k beyond the exact range of the fp32 accumulator");
    const char* et = getenv("BIE_FP4_TILE");
    return binary_conv_fp4_launch(x, wimage, y, workspace, B, C, H, W, OC, ksize, stride, pad, dilation, scale, dtype, et ? atoi(et) : 0, as_stream(stream));
}

int bie_binary_matmul_batched(const uint8_t* xpacked, const uint8_t* wpacked, float* y, long batch, long M, long N, long K, long stride_x,
                              long stride_w, long stride_y, float scale, void* stream) {
    BIE_REQUIRE(xpacked && wpacked && y && batch > 0 && M > 0 && N > 0 && K > 0 && K % 8 == 0, BIE_ERR_INVALID_ARG,
                "bie_binary_matmul_batched: batch=%ld M=%ld N=%ld K=%ld (K %% 8 == 0 required)", batch, M, N, K);
    BIE_REQUIRE(stride_x >= M * (K / 8) && stride_w >= N * (K / 8) && stride_y >= M * N, BIE_ERR_INVALID_ARG, "bie_binary_matmul_batched: strides smaller than one matrix");
    return binary_matmul_batched_launch(xpacked, wpacked, y, batch, M, N, K, stride_x, stride_w, stride_y, scale, as_stream(stream));
}

int bie_binary_linear_fused_ok(long M, long N, long K) { return binary_linear_fused_ok(M, N, K) ? 1 : 0; }

int bie_binary_linear_fused(const void* x, const void* bias_a, const uint8_t* wpacked, const void* scale_a, const void* scale_w,
                            void* y, long M, long N, long K, int dtype, int y_f32, void* stream) {
    BIE_REQUIRE(dtype >= 0 && dtype <= 2, BIE_ERR_UNSUPPORTED, "bie_binary_linear_fused: dtype %d", dtype);
    BIE_REQUIRE(!y_f32 || (!scale_a && !scale_w), BIE_ERR_INVALID_ARG, "bie_binary_linear_fused: y_f32 (raw counts) takes no scales");
    BIE_REQUIRE(binary_linear_fused_ok(M, N, K), BIE_ERR_UNSUPPORTED,
                "bie_binary_linear_fused: M=%ld N=%ld K=%ld outside the one-launch range (1 <= M <= 64 with K %% 32 == 0, or 5 <= M <= 512 with K %% 512 == 0)", M, N, K);
    BIE_REQUIRE(x && wpacked && y, BIE_ERR_INVALID_ARG, "bie_binary_linear_fused: NULL tensor pointer");
    BIE_REQUIRE(((reinterpret_cast<uintptr_t>(wpacked) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(bias_a)) & 15) == 0, BIE_ERR_INVALID_ARG,
                "bie_binary_linear_fused: x, bias_a and the packed weights must be 16-byte aligned");
    return binary_linear_fused_launch(x, bias_a, wpacked, scale_a, scale_w, y, M, N, K, dtype, y_f32 ? 1 : 0, as_stream(stream));
}

size_t bie_binary_conv2d_workspace_bytes(int B, int C, int H, int W, int OC, int ksize, int stride, int pad, int dilation) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || OC <= 0 || ksize <= 0 || stride <= 0 || dilation <= 0 || pad < 0) return 0;
    const size_t a = binary_conv_workspace_bytes(B, C, H, W, OC, ksize, stride, pad, dilation), b = binary_conv_taps_workspace_bytes(B, C, H, W);
    return WS_HEAD + (a > b ? a : b);  // scratch starts behind the counter head; covers the im2col form and the tap form
}

int bie_binary_conv2d_taps_ok(int C, int W, int ksize) { return C > 0 && W > 0 && ksize > 0 && binary_conv_taps_lds_bytes(C, W, ksize) > 0 ? 1 : 0; }

int bie_binary_conv_weight_taps(const uint8_t* wpacked, uint32_t* wtaps, int OC, int C, int ksize, void* stream) {
    BIE_REQUIRE(wpacked && wtaps && OC > 0 && C > 0 && ksize > 0, BIE_ERR_INVALID_ARG, "bie_binary_conv_weight_taps: bad argument");
    BIE_REQUIRE((C * ksize * ksize) % 8 == 0, BIE_ERR_UNSUPPORTED, "bie_binary_conv_weight_taps: C*k*k=%d must be a multiple of 8", C * ksize * ksize);
    return binary_conv_weight_taps_launch(wpacked, wtaps, OC, C, ksize, as_stream(stream));
}

int bie_binary_conv2d_forward_taps(const void* x, const uint32_t* wtaps, float* y, void* workspace, size_t workspace_bytes, int B, int C,
                                   int H, int W, int OC, int ksize, int stride, int pad, int dilation, float scale, int dtype,
                                   void* stream) {
    BIE_REQUIRE(x && wtaps && y && workspace, BIE_ERR_INVALID_ARG, "bie_binary_conv2d_forward_taps: NULL tensor pointer");
    BIE_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && OC > 0 && ksize > 0 && stride > 0 && dilation > 0 && pad >= 0, BIE_ERR_INVALID_ARG, "bie_binary_conv2d_forward_taps: bad geometry");
    BIE_REQUIRE(dtype >= 0 && dtype <= 2, BIE_ERR_UNSUPPORTED, "bie_binary_conv2d_forward_taps: dtype %d", dtype);
    BIE_REQUIRE(binary_conv_taps_lds_bytes(C, W, ksize) > 0, BIE_ERR_UNSUPPORTED, "bie_binary_conv2d_forward_taps: k=%d rows of W=%d x C=%d do not fit the 16 KiB slab (use bie_binary_conv2d_forward)", ksize, W, C);
    BIE_REQUIRE((H + 2 * pad - dilation * (ksize - 1) - 1) / stride + 1 > 0 && (W + 2 * pad - dilation * (ksize - 1) - 1) / stride + 1 > 0, BIE_ERR_INVALID_ARG, "bie_binary_conv2d_forward_taps: empty output");
    BIE_REQUIRE((reinterpret_cast<uintptr_t>(wtaps) & 15) == 0, BIE_ERR_INVALID_ARG, "bie_binary_conv2d_forward_taps: wtaps must be 16-byte aligned");
    const size_t need = WS_HEAD + binary_conv_taps_workspace_bytes(B, C, H, W);
    BIE_REQUIRE(workspace_bytes >= need, BIE_ERR_WORKSPACE, "bie_binary_conv2d_forward_taps: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return binary_conv_taps_launch(x, wtaps, y, static_cast<char*>(workspace) + WS_HEAD, B, C, H, W, OC, ksize, stride, pad, dilation, scale, dtype, as_stream(stream));
}

int bie_binary_conv2d_fused_ok(int B, int C, int H, int W, int OC, int ksize, int stride, int pad, int dilation) {
    return binary_conv_fused_ok(B, C, H, W, OC, ksize, stride, pad, dilation) ? 1 : 0;
}

size_t bie_binary_conv_weight_lanes_bytes(int OC, int C, int ksize) {
    return (OC > 0 && C > 0 && C % 128 == 0 && ksize > 0) ? binary_conv_weight_lanes_words(OC, C, ksize) * 4 : 0;
}

int bie_binary_conv_weight_lanes(const uint32_t* wtaps, uint32_t* wlanes, int OC, int C, int ksize, void* stream) {
    BIE_REQUIRE(wtaps && wlanes && OC > 0 && C > 0 && ksize > 0, BIE_ERR_INVALID_ARG, "bie_binary_conv_weight_lanes: bad argument");
    BIE_REQUIRE(C % 128 == 0, BIE_ERR_UNSUPPORTED, "bie_binary_conv_weight_lanes: C=%d must be a multiple of 128 (four K quarters of whole channel words)", C);
    return binary_conv_weight_lanes_launch(wtaps, wlanes, OC, C, ksize, as_stream(stream));
}

int bie_binary_conv2d_forward_fused(const void* x, const uint32_t* wlanes, float* y, int B, int C, int H, int W, int OC, int ksize, int stride, int pad,
                                    int dilation, float scale, int dtype, void* stream) {
    BIE_REQUIRE(x && wlanes && y, BIE_ERR_INVALID_ARG, "bie_binary_conv2d_forward_fused: NULL tensor pointer");
    BIE_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && OC > 0 && ksize > 0 && stride > 0 && dilation > 0 && pad >= 0, BIE_ERR_INVALID_ARG, "bie_binary_conv2d_forward_fused: bad geometry");
    BIE_REQUIRE(dtype >= 0 && dtype <= 2, BIE_ERR_UNSUPPORTED, "bie_binary_conv2d_forward_fused: dtype %d", dtype);
    BIE_REQUIRE((long)B * C * H * W < (1L << 31) && (long)B * OC * H * W < (1L << 31), BIE_ERR_UNSUPPORTED, "bie_binary_conv2d_forward_fused: tensor beyond 2^31 elements");
    return binary_conv_fused_launch(x, wlanes, y, B, C, H, W, OC, ksize, stride, pad, dilation, scale, dtype, as_stream(stream));
}

int bie_binary_conv2d_mfma_ok(int B, int C, int H, int W, int OC, int ksize, int stride, int pad, int dilation) {
    return binary_conv_mfma_ok(B, C, H, W, OC, ksize, stride, pad, dilation) ? 1 : 0;
}

int bie_binary_conv2d_forward_mfma(const void* x, const uint8_t* wimage, float* y, int B, int C, int H, int W, int OC, int ksize, int stride, int pad,
                                   int dilation, float scale, int dtype, void* stream) {
    BIE_REQUIRE(x && wimage && y, BIE_ERR_INVALID_ARG, "bie_binary_conv2d_forward_mfma: NULL tensor pointer");
    BIE_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && OC > 0 && ksize > 0 && stride > 0 && dilation > 0 && pad >= 0, BIE_ERR_INVALID_ARG, "bie_binary_conv2d_forward_mfma: bad geometry");
    BIE_REQUIRE(dtype >= 0 && dtype <= 2, BIE_ERR_UNSUPPORTED, "bie_binary_conv2d_forward_mfma: dtype %d", dtype);
    BIE_REQUIRE((reinterpret_cast<uintptr_t>(wimage) & 15) == 0, BIE_ERR_INVALID_ARG, "bie_binary_conv2d_forward_mfma: the weight image must be 16-byte aligned");
    BIE_REQUIRE((long)B * C * H * W < (1L << 31) && (long)B * OC * H * W < (1L << 31), BIE_ERR_UNSUPPORTED, "bie_binary_conv2d_forward_mfma: tensor beyond 2^31 elements");
    BIE_REQUIRE((long)ksize * ksize * C < (1L << 24), BIE_ERR_UNSUPPORTED, "bie_binary_conv2d_forward_mfma: C*k*k beyond the exact range of the fp32 accumulator");
    return binary_conv_mfma_launch(x, wimage, y, B, C, H, W, OC, ksize, stride, pad, dilation, scale, dtype, as_stream(stream));
}

int bie_binary_conv2d_forward(const void* x, const uint8_t* wpacked, float* y, void* workspace, size_t workspace_bytes, int B,
                              int C, int H, int W, int OC, int ksize, int stride, int pad, int dilation, float scale, int dtype,
                              void* stream) {
    BIE_REQUIRE(x && wpacked && y && workspace, BIE_ERR_INVALID_ARG, "bie_binary_conv2d_forward: NULL tensor pointer");
    BIE_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && OC > 0 && ksize > 0 && stride > 0 && dilation > 0 && pad >= 0, BIE_ERR_INVALID_ARG, "bie_binary_conv2d_forward: bad geometry");
    BIE_REQUIRE((C * ksize * ksize) % 8 == 0, BIE_ERR_UNSUPPORTED, "bie_binary_conv2d_forward: C*k*k=%d must be a multiple of 8", C * ksize * ksize);
    BIE_REQUIRE(dtype >= 0 && dtype <= 2, BIE_ERR_UNSUPPORTED, "bie_binary_conv2d_forward: dtype %d", dtype);
    const size_t need = WS_HEAD + binary_conv_workspace_bytes(B, C, H, W, OC, ksize, stride, pad, dilation);
    BIE_REQUIRE(workspace_bytes >= need, BIE_ERR_WORKSPACE, "bie_binary_conv2d_forward: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    // a shared per-stream workspace also serves the GEMV kernels: never touch its head of split-K tickets
    return binary_conv_launch(x, wpacked, y, static_cast<char*>(workspace) + WS_HEAD, B, C, H, W, OC, ksize, stride, pad, dilation, scale, dtype, as_stream(stream));
}

// ---------------------------------------------------------------------------------------------- functions
int bie_pack_sign_u8(const void* a, uint8_t* out, long n_bytes, int dtype, void* stream) {
    BIE_REQUIRE(a && out && n_bytes > 0, BIE_ERR_INVALID_ARG, "bie_pack_sign_u8: bad argument");
    BIE_REQUIRE(dtype >= 0 && dtype <= 3, BIE_ERR_UNSUPPORTED, "bie_pack_sign_u8: dtype %d", dtype);
    return pack_rows_launch(a, out, n_bytes, dtype, as_stream(stream));
}
int bie_unpack_u8_scaled(const uint8_t* in, const float* scale, float* out, long n_bytes, long packed_dim, void* stream) {
    BIE_REQUIRE(in && scale && out && n_bytes > 0 && packed_dim > 0, BIE_ERR_INVALID_ARG, "bie_unpack_u8_scaled: bad argument");
    return unpack_u8_scaled_launch(in, scale, out, n_bytes, packed_dim, as_stream(stream));
}
int bie_q4_pack(const int32_t* in, int8_t* out, long n_out, void* stream) {
    BIE_REQUIRE(in && out && n_out > 0, BIE_ERR_INVALID_ARG, "bie_q4_pack: bad argument");
    return q4_pack_launch(in, out, n_out, as_stream(stream));
}
int bie_q4_unpack(const int8_t* in, int32_t* out, long n_in, void* stream) {
    BIE_REQUIRE(in && out && n_in > 0, BIE_ERR_INVALID_ARG, "bie_q4_unpack: bad argument");
    return q4_unpack_launch(in, out, n_in, as_stream(stream));
}
int bie_q4_unpack_scale(const int8_t* in, float* out, long n_in, float scale, void* stream) {
    BIE_REQUIRE(in && out && n_in > 0, BIE_ERR_INVALID_ARG, "bie_q4_unpack_scale: bad argument");
    return q4_unpack_scale_launch(in, out, n_in, scale, as_stream(stream));
}


// ---------------------------------------------------------------------------------------------- W4A4 / W8A8
int bie_q4_quantize_pack(const void* x, int8_t* out, long n_out, float scale, int dtype, void* stream) {
    BIE_REQUIRE(x && out && n_out > 0, BIE_ERR_INVALID_ARG, "bie_q4_quantize_pack: bad argument");
    BIE_REQUIRE(dtype >= 0 && dtype <= 2, BIE_ERR_UNSUPPORTED, "bie_q4_quantize_pack: tensor type not supported: %d", dtype);
    return q4_quantize_pack_launch(x, out, n_out, scale, dtype, as_stream(stream));
}

int bie_q4_gemm(const int8_t* a_packed, const int8_t* w_packed, void* y, int M, int N, int K, float scale_a, float scale_w,
                int dtype, int batch, long stride_a, long stride_w, long stride_y, void* stream) {
    BIE_REQUIRE(a_packed && w_packed && y && M > 0 && N > 0 && K > 0 && batch > 0, BIE_ERR_INVALID_ARG, "bie_q4_gemm: bad argument");
    BIE_REQUIRE(K % 64 == 0 && N % 4 == 0, BIE_ERR_UNSUPPORTED, "bie_q4_gemm: K=%d must be a multiple of 64 and N=%d of 4", K, N);
    BIE_REQUIRE(dtype >= 0 && dtype <= 2, BIE_ERR_UNSUPPORTED, "bie_q4_gemm: tensor type not supported: %d", dtype);
    return int_gemm_launch(0, a_packed, w_packed, y, M, N, K, scale_a, scale_w, dtype, batch, stride_a, stride_w, stride_y, as_stream(stream));
}

int bie_q8_gemm(const int8_t* a, const int8_t* w, float* y, int M, int N, int K, float scale_a, float scale_w, void* stream) {
    BIE_REQUIRE(a && w && y && M > 0 && N > 0 && K > 0, BIE_ERR_INVALID_ARG, "bie_q8_gemm: bad argument");
    BIE_REQUIRE(K % 64 == 0 && N % 4 == 0, BIE_ERR_UNSUPPORTED, "bie_q8_gemm: K=%d must be a multiple of 64 and N=%d of 4", K, N);
    return int_gemm_launch(1, a, w, y, M, N, K, scale_a, scale_w, BIE_F32, 1, 0, 0, 0, as_stream(stream));
}

int bie_int_gemm_i32(const int8_t* a, const int8_t* w, int32_t* y, int M, int N, int K, int bits, int batch, long stride_a, long stride_w,
                     long stride_y, void* stream) {
    BIE_REQUIRE(a && w && y && M > 0 && N > 0 && K > 0 && batch > 0, BIE_ERR_INVALID_ARG, "bie_int_gemm_i32: bad argument");
    BIE_REQUIRE(bits == 4 || bits == 8, BIE_ERR_UNSUPPORTED, "bie_int_gemm_i32: bits=%d (4 or 8)", bits);
    BIE_REQUIRE(K % 64 == 0 && N % 4 == 0, BIE_ERR_UNSUPPORTED, "bie_int_gemm_i32: K=%d must be a multiple of 64 and N=%d of 4", K, N);
    return int_gemm_launch(bits == 4 ? 2 : 3, a, w, y, M, N, K, 1.0f, 1.0f, BIE_F32, batch, stride_a, stride_w, stride_y, as_stream(stream));
}

size_t bie_q4_conv2d_workspace_bytes(int B, int H, int W, int C, int OC, int ksize, int stride, int pad, int dilation) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || OC <= 0 || ksize <= 0 || stride <= 0 || dilation <= 0 || pad < 0) return 0;
    return q4_conv2d_workspace_bytes(B, H, W, C, OC, ksize, stride, pad, dilation);
}

int bie_q4_conv2d_forward(const int8_t* a_packed, const int8_t* w_packed, void* y, void* workspace, size_t workspace_bytes, int B, int H,
                          int W, int C, int OC, int ksize, int stride, int pad, int dilation, float scale_a, float scale_w, int dtype,
                          void* stream) {
    BIE_REQUIRE(a_packed && w_packed && y && workspace, BIE_ERR_INVALID_ARG, "bie_q4_conv2d_forward: NULL tensor pointer");
    BIE_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && OC > 0 && ksize > 0 && stride > 0 && dilation > 0 && pad >= 0, BIE_ERR_INVALID_ARG, "bie_q4_conv2d_forward: bad geometry");
    BIE_REQUIRE(C % 8 == 0 && OC % 4 == 0, BIE_ERR_UNSUPPORTED, "bie_q4_conv2d_forward: C=%d must be a multiple of 8 and OC=%d of 4", C, OC);
    BIE_REQUIRE(dtype >= 0 && dtype <= 2, BIE_ERR_UNSUPPORTED, "bie_q4_conv2d_forward: tensor type not supported: %d", dtype);
    const size_t need = q4_conv2d_workspace_bytes(B, H, W, C, OC, ksize, stride, pad, dilation);
    BIE_REQUIRE(need > 0, BIE_ERR_INVALID_ARG, "bie_q4_conv2d_forward: empty output");
    BIE_REQUIRE(workspace_bytes >= need, BIE_ERR_WORKSPACE, "bie_q4_conv2d_forward: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    return q4_conv2d_launch(a_packed, w_packed, y, workspace, B, H, W, C, OC, ksize, stride, pad, dilation, scale_a, scale_w, dtype, as_stream(stream));
}

}  // extern "C"
