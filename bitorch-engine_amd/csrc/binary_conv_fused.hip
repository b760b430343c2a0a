// Binary conv2d (1-bit W / 1-bit A, XNOR-popcount) as ONE launch -- BASELINE.json configs[3] (ResNet 3x3x512 on 7x7 maps).
// Replaces binary_conv_cpp.forward (reference layers/qconv/binary/cpp/binary_conv.cpp: im2binary_col :319-365 + the
// XNOR GEMM :464-530) and, through the layer, binary_conv2d_cutlass.forward (binary_conv2d_cutlass_kernel.cu:122-183).
// Same integers as bie_binary_conv2d_forward / _taps / _fp4: y[b][oc][oh][ow] = (C*k*k - 2*popc(x ^ w)) * scale, padding = -1.
//
// Round 5 ran this op as three launches (sign-pack pass, FP4 image pass, GEMM) or two (sign-pack, tap kernel): 8.9 / 20.7 / 29.8 us at
// B = 1 / 32 / 128 for an op that is 0.2-6 us of VALU work: two kernel boundaries and two HBM round trips of
// intermediate images cost more than the arithmetic.  Here a workgroup does all of it for (image b, 64*G output channels, a range of
// whole output rows):
//   1. its K quarter of the weights goes into REGISTERS: lane = output channel, wave = quarter of the (tap, channel word) list, from a
//      lane-major image of the tap words (bie_binary_conv_weight_lanes, once per weight tensor) -- every wave-load is 256 contiguous bytes;
//   2. the input rows it needs are sign-packed straight from x (NCHW, any float dtype) into an LDS bit image [row][col][C/32 words] with a
//      ZERO border (a zero word = 32 times -1 = the reference's padding), so a tap of an output pixel is a constant offset from the pixel;
//   3. per output pixel and tap ONE uniform-address LDS read (a broadcast: every lane = every output channel needs the same x words)
//      feeds CWW * G x (v_xor_b32 + v_bcnt_u32_b32 accumulate) against the register-resident weights, PXB pixels at a time;
//   4. the four K quarters meet in an LDS [channel][pixel] array (ds_add_u32), which is then written out pixel-contiguous: whole 128-byte
//      lines of y instead of one float per channel row.
// No workspace, no intermediate image in HBM, no second launch.  VALU bound (2 instructions per 32 MACs per lane): B = 32 is 2.9 us of
// issue on the whole chip.
#include "mfma_pipe.cuh"
#include <type_traits>
#include <stdlib.h>

namespace bie {

namespace {

// 32 sign bits (value >= 0, NaN -> 0: binary.hip::sign_bit) of the elements base + cc * step, cc = 0 .. 31.  The loads are issued as ONE batch
// (raw bit patterns into registers, a compiler barrier, then the compares): written as `v |= sign(x[..]) << cc` hipcc emitted load, wait, compare
// 32 times in a row -- 7 us of the first version's 9 at B = 1 (profiles/r06_conv_timelines.txt).
template <int DT>
__device__ __forceinline__ uint32_t conv_sign_word(const void* x, long base, long step) {
    uint32_t raw[32];
#pragma unroll
    for (int cc = 0; cc < 32; cc++) {
        if constexpr (DT == BIE_F32) raw[cc] = reinterpret_cast<const uint32_t*>(x)[base + cc * step];
        else raw[cc] = reinterpret_cast<const uint16_t*>(x)[base + cc * step];
    }
    asm volatile("" ::: "memory");
    uint32_t v = 0;
#pragma unroll
    for (int cc = 0; cc < 32; cc++) {
        float f;
        if constexpr (DT == BIE_F32) f = __uint_as_float(raw[cc]);
        else if constexpr (DT == BIE_F16) f = f16_bits_to_f32(raw[cc]);
        else f = bf16_bits_to_f32(raw[cc]);
        v |= (uint32_t)(f >= 0.0f) << cc;
    }
    return v;
}

#ifdef BIE_CONV_LAB
// lab build (tools/build_variant.sh convlab binary_conv_fused.hip -DBIE_CONV_LAB; tools/conv_fused_timeline.py): per workgroup
// {start, weights requested, image packed (after the barrier), popcount passes done, sums complete (after the barrier), end}
__device__ unsigned long long g_conv_stamps[8192 * 8];
#define BIE_CONV_STAMP(slot) do { if (tid == 0 && blockIdx.x < 8192) { g_conv_stamps[blockIdx.x * 8 + (slot)] = wall_clock64(); \
    if ((slot) == 2) g_conv_stamps[blockIdx.x * 8 + 6] = __builtin_readcyclecounter(); if ((slot) == 3) g_conv_stamps[blockIdx.x * 8 + 7] = __builtin_readcyclecounter(); } } while (0)
#else
#define BIE_CONV_STAMP(slot)
#endif

struct ConvFusedArgs {
    const void* x;
    const uint32_t* wl;  // [ceil(OC/64)][4 quarters][T*CWW][64 lanes]
    float* y;
    int B, C, H, W, OC, OH, OW, stride, pad, dtype;
    int rpw, chunks, ocbs;  // output rows per workgroup, row chunks per image, workgroups per pixel range (64*G channels each)
    int IR, WP, PP;         // LDS image rows / columns (pixels), pitch of the [channel][pixel] sums
    float scale;
};

typedef __attribute__((address_space(3))) uint32_t lds_u32;

template <int N> struct xword;  // N words read with one LDS instruction
template <> struct xword<1> { typedef uint32_t type; };
template <> struct xword<2> { typedef uint2_t type; };
template <> struct xword<4> { typedef uint4_t type; };

template <int N>
__device__ __forceinline__ uint32_t xw_get(const typename xword<N>::type& v, int j) {
    if constexpr (N == 1) return v;
    else return v[j];
}

// KS: kernel size (dilation 1); CWW: channel words per K quarter (C = 128 * CWW); G: 64-channel groups per lane; PXB: pixels per pass
template <int KS, int CWW, int G, int PXB>
__global__ __launch_bounds__(256) void xnor_conv_fused_kernel(const ConvFusedArgs a) {
    constexpr int T = KS * KS, CW = 4 * CWW, NWW = T * CWW;
    constexpr int PS = CW + 4;  // pixel pitch of the bit image in words: 16-byte aligned, and the pack phase's stores spread over the banks
    extern __shared__ __attribute__((aligned(16))) uint32_t conv_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ocb = (int)(blockIdx.x % (unsigned)a.ocbs);
    const int chunk = (int)((blockIdx.x / (unsigned)a.ocbs) % (unsigned)a.chunks);
    const long b = blockIdx.x / ((unsigned)a.ocbs * (unsigned)a.chunks);
    const int oh0 = chunk * a.rpw;
    const int nrow = min(a.rpw, a.OH - oh0);
    const int P = nrow * a.OW;
    const int nblk = (a.OC + 63) >> 6;

    BIE_CONV_STAMP(0);
    // what the write-out needs of the kernel arguments is fetched HERE (opaque copies): read where it is used, the scalar loads of the argument
    // segment missed their cache after the last barrier (write-out of 7 pixels x 64 channels at B = 1: 2.0 -> 1.76 us)
    float* ybase = a.y + ((b * a.OC) * (long)a.OH + oh0) * a.OW;
    float yscale = a.scale;
    int yoc = a.OC, ypitch = a.PP;
    long yplane = (long)a.OH * a.OW;
    asm volatile("" : "+s"(ybase), "+s"(yscale), "+s"(yoc), "+s"(ypitch), "+s"(yplane));
    // ---- 1. this wave's K quarter of the weights (requested first: in flight under the pack phase)
    uint32_t wreg[G][NWW];
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int blk = min(ocb * G + g, nblk - 1);  // a group beyond OC re-reads the last block: valid memory, never stored
        const uint32_t* wp = a.wl + ((size_t)blk * 4 + wq) * (NWW * 64) + lane;
#pragma unroll
        for (int i = 0; i < NWW; i++) wreg[g][i] = wp[i * 64];
    }

    BIE_CONV_STAMP(1);
    // ---- 2. bit image of the input rows [ih_lo, ih_lo + IR) x columns [-pad, -pad + WP), zero outside the image; the sums' array zeroed
    uint32_t* img = conv_lds;
    int* red = reinterpret_cast<int*>(conv_lds + a.IR * a.WP * PS);
    const int ih_lo = oh0 * a.stride - a.pad;
    const int IRn = (nrow - 1) * a.stride + KS;  // rows this chunk really needs (<= a.IR)
    const int npix = IRn * a.WP;
    const long HW = (long)a.H * a.W;
    auto pack = [&](auto dt_c) {
        constexpr int DT = decltype(dt_c)::value;
        for (int idx = tid; idx < npix * CW; idx += 256) {  // pixel fastest: the 32 channel loads of a wave are coalesced along the row
            const int cwi = idx / npix, pix = idx - cwi * npix;
            const int r = pix / a.WP, c = pix - r * a.WP;
            const int ih = ih_lo + r, iw = c - a.pad;
            uint32_t v = 0;
            if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W) {
                v = conv_sign_word<DT>(a.x, ((b * a.C + cwi * 32) * a.H + ih) * a.W + iw, HW);
            }
            img[pix * PS + cwi] = v;
        }
    };
    if (a.dtype == BIE_F32) pack(std::integral_constant<int, BIE_F32>{});
    else if (a.dtype == BIE_F16) pack(std::integral_constant<int, BIE_F16>{});
    else pack(std::integral_constant<int, BIE_BF16>{});
    for (int i = tid; i < G * 64 * a.PP; i += 256) red[i] = 0;
    __syncthreads();
    BIE_CONV_STAMP(2);

    // ---- 3. XNOR-popcount: PXB output pixels per pass, taps unrolled, x words by uniform-address LDS reads
    const uint32_t img_base = (uint32_t)(uintptr_t)(lds_u32*)img + (uint32_t)wq * (CWW * 4);  // LDS byte address of (row 0, col 0, this quarter)
    const uint32_t row_bytes = (uint32_t)a.WP * (PS * 4);
    int orow = 0, ocol = 0;  // output pixel p0 = (orow, ocol) inside the chunk
    for (int p0 = 0; p0 < P; p0 += PXB) {
        uint32_t base[PXB];
        {
            int r_ = orow, c_ = ocol;
#pragma unroll
            for (int i = 0; i < PXB; i++) {
                base[i] = img_base + (uint32_t)(r_ * a.stride) * row_bytes + (uint32_t)(c_ * a.stride) * (PS * 4);
                if (p0 + i + 1 < P) {  // past the end: the last pixel again (computed, never added)
                    if (++c_ == a.OW) { c_ = 0; r_++; }
                }
            }
        }
        int acc[G][PXB];
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int i = 0; i < PXB; i++) acc[g][i] = 0;
#pragma unroll
        for (int ti = 0; ti < KS; ti++) {
            uint32_t rowa[PXB];
#pragma unroll
            for (int i = 0; i < PXB; i++) rowa[i] = base[i] + (uint32_t)ti * row_bytes;
#pragma unroll
            for (int tj = 0; tj < KS; tj++) {
                typename xword<CWW>::type xa[PXB];
#pragma unroll
                for (int i = 0; i < PXB; i++)
                    xa[i] = *reinterpret_cast<const __attribute__((address_space(3))) typename xword<CWW>::type*>(rowa[i] + (uint32_t)(tj * PS * 4));
#pragma unroll
                for (int i = 0; i < PXB; i++)
#pragma unroll
                    for (int j = 0; j < CWW; j++)
#pragma unroll
                        for (int g = 0; g < G; g++)
                            asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[g][i]) : "v"(xw_get<CWW>(xa[i], j) ^ wreg[g][(ti * KS + tj) * CWW + j]));
            }
        }
        // ---- 4a. the four K quarters add up in LDS: red[group][channel][pixel]
#pragma unroll
        for (int g = 0; g < G; g++)
#pragma unroll
            for (int i = 0; i < PXB; i++)
                if (p0 + i < P) __hip_atomic_fetch_add(red + (g * 64 + lane) * a.PP + p0 + i, acc[g][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ocol += PXB;
        while (ocol >= a.OW) { ocol -= a.OW; orow++; }
    }
    BIE_CONV_STAMP(3);
    __syncthreads();
    BIE_CONV_STAMP(4);

    // ---- 4b. write-out, pixel-contiguous: row (group, channel) of `red` = P consecutive floats of y
    const int Kc = a.C * T;
    const __attribute__((address_space(3))) int* red3 = (const __attribute__((address_space(3))) int*)red;
    for (int row0 = wq * 4; row0 < G * 64; row0 += 16) {  // four rows per pass: their LDS reads and stores overlap
        for (int p = lane; p < P; p += 64) {
            int pc[4];
#pragma unroll
            for (int r = 0; r < 4; r++) pc[r] = red3[(row0 + r) * ypitch + p];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = row0 + r, oc = (ocb * G + (row >> 6)) * 64 + (row & 63);
                if (oc < yoc) ybase[oc * yplane + p] = (float)(Kc - 2 * pc[r]) * yscale;
            }
        }
    }
    BIE_CONV_STAMP(5);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The same op on the MATRIX pipe, still ONE launch: beyond a few hundred output pixels the VALU form above is bound by v_xor + v_bcnt
// (4 cycles per wave-instruction each: 5.9 us at B = 32, 23.5 at B = 128), the FP4 MFMA is not (+-1 are exact E2M1 values and
// v_mfma_scale_f32_32x32x64_f8f6f4 accumulates K - 2*popc exactly: binary_fp4.hip).  Round 5's matrix-pipe conv was three launches: sign
// bits, an FP4 im2col image in HBM (14.5 MB written and re-read at B = 128), then the GEMM: 5.5 + 11.4 + 16.3 us.  Here the im2col matrix
// never exists: a workgroup (4 waves) = (NI whole images or one image's row range: up to 32 * NPB output pixels) x 128 output channels
//   1. sign-packs its input rows from x into a zero... no: MINUS-ONE-bordered FP4 NHWC image in LDS ([slot][row][col][C/2 bytes + 16]);
//   2. wave w owns 32 output channels: per k step (tap, 64 channels) ONE 1 KiB weight fragment from the fragment-ordered FP4 image of the
//      tap-major weights (bie_binary_fp4_image: contiguous, L2-resident, two steps ahead in registers) and NPB pixel fragments gathered
//      from the LDS image by ds_read_b128 (lane = (pixel, k half): the tap is a constant offset from the pixel) feed NPB MFMAs;
//   3. the accumulators leave as y[b][oc][pixel]: lane = pixel, 128 contiguous bytes per output channel row.
// A operand = weights (rows = output channels), B operand = pixels (columns): D[i] of lane l = channel 8*(i/4) + 4*(l/32) + i%4, pixel l%32.
typedef int conv_v8i __attribute__((ext_vector_type(8)));

struct ConvMfmaArgs {
    const void* x;
    const uint8_t* wimg;  // FP4 fragments [ceil(OC/32)][kb_per_row][64 lanes][16 B], k = (tap, channel)
    float* y;
    int B, C, H, W, OC, OH, OW, stride, pad, dtype;
    int ni, rpw, chunks, ocbs;  // whole images per workgroup (chunks == 1) or output rows per workgroup of one image (ni == 1)
    int IR, WP, kb_per_row;
    float scale;
    int pitch;  // bytes per pixel of the LDS image: C / 2 + padding (a multiple of 16)
};

// 8 sign bits -> 8 E2M1 nibbles: bit 1 (value >= 0) -> 0x2 (+1.0), bit 0 -> 0xA (-1.0)  (= binary_fp4.hip::fp4_from_bits8)
__device__ __forceinline__ uint32_t conv_fp4_from_bits8(uint32_t b) {
    uint32_t x = b & 0xffu;
    x = (x | (x << 12)) & 0x000f000fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return 0xaaaaaaaau ^ (x << 3);
}

// 32 consecutive channels of ONE pixel per lane -> that pixel's sign word, bit j = (x[channel j] >= 0).  Lane = pixel: a wave-load reads 64
// consecutive pixels of one channel plane (contiguous), the compare's carry is shifted into the word by v_addc_co_u32 (word = 2 * word + bit,
// channels 31 .. 0): two VALU per value and NO transposition -- the lane that owns the pixel builds the pixel's channel words.
// NW x 32 consecutive channels of ONE pixel per lane -> that pixel's NW sign words, bit j = (x[channel j] >= 0).  Lane = pixel: a wave-load
// reads 64 consecutive pixels of one channel plane (contiguous), and the compare's carry is shifted into the word by v_addc_co_u32
// (word = 2 * word + bit, channels 31 .. 0): two VALU per value and NO transposition -- the lane that owns the pixel builds the pixel's
// channel words.  BUFFER loads: the plane offset is a scalar register (one s_add_u32 per load) and the pixel offset one VGPR; written with
// pointers hipcc computed every one of the 128 addresses by scalar 64-bit multiplies and parked them in VGPR lanes (v_writelane / v_readlane):
// 10 us per image of address arithmetic (profiles/r06_conv_timelines.txt).  All loads of the batch are issued before the first compare.
template <int DT, int NW>
__device__ __forceinline__ void conv_pixel_words(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned plane_bytes, uint32_t (&w)[NW]) {
    uint32_t raw[NW][32];
#pragma unroll
    for (int i = 0; i < NW; i++)
#pragma unroll
        for (int j = 0; j < 32; j++) {
            if constexpr (DT == BIE_F32) raw[i][j] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0);
            else raw[i][j] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0);
            soff += plane_bytes;
            asm volatile("" : "+s"(soff));  // opaque: one running scalar offset, not 32 * NW precomputed ones parked in VGPR lanes
        }
    asm volatile("" ::: "memory");
    // NW independent compare -> carry chains side by side, each with its own SGPR pair for the carry (VOP3 forms): through VCC alone the 32 * NW
    // pairs are one dependent chain
#pragma unroll
    for (int i = 0; i < NW; i++) w[i] = 0;
#pragma unroll
    for (int j = 31; j >= 0; j--) {
        unsigned long long cy[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) {
            if constexpr (DT == BIE_F32) asm volatile("v_cmp_le_f32_e64 %0, 0, %1" : "=s"(cy[i]) : "v"(raw[i][j]));
            else if constexpr (DT == BIE_F16) asm volatile("v_cmp_le_f16_e64 %0, 0, %1" : "=s"(cy[i]) : "v"(raw[i][j]));
            else asm volatile("v_lshlrev_b32 %1, 16, %1\n\tv_cmp_le_f32_e64 %0, 0, %1" : "=s"(cy[i]), "+v"(raw[i][j]));
        }
#pragma unroll
        for (int i = 0; i < NW; i++) asm volatile("v_addc_co_u32_e64 %0, %1, %0, %0, %1" : "+v"(w[i]), "+s"(cy[i]));
    }
}

template <int KS, int NPB, int CBT>  // CBT = C / 64: k steps per tap
__global__ __launch_bounds__(256) void xnor_conv_mfma_kernel(const ConvMfmaArgs a) {
    constexpr int T = KS * KS;
    extern __shared__ __attribute__((aligned(16))) uint32_t conv_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup b runs on XCD b % 8 (observed, for speed only): the ocbs workgroups that sign-pack the SAME images get ids 8 apart, so the
    // pictures come from HBM once per XCD L2 instead of once per workgroup (B = 128: 51 MB of x reads at HBM speed were 9.7 us of pack phase).
    int ocb, grp;
    {
        const unsigned per8 = 8u * (unsigned)a.ocbs, full = (gridDim.x / per8) * per8;
        if (blockIdx.x < full) {
            const unsigned blk = blockIdx.x / per8, r = blockIdx.x - blk * per8;
            ocb = (int)(r >> 3);
            grp = (int)(blk * 8u + (r & 7u));
        } else {  // the ragged tail: plain order
            const unsigned r = blockIdx.x - full;
            ocb = (int)(r % (unsigned)a.ocbs);
            grp = (int)(full / (unsigned)a.ocbs + r / (unsigned)a.ocbs);
        }
    }
    int b0, oh0, NIh;
    if (a.chunks == 1) { b0 = grp * a.ni; oh0 = 0; NIh = min(a.ni, a.B - b0); }
    else { b0 = grp / a.chunks; oh0 = (grp - b0 * a.chunks) * a.rpw; NIh = 1; }
    const int nrow = min(a.rpw, a.OH - oh0);
    const int Pw = nrow * a.OW;                  // output pixels per image slot of this workgroup
    const int CW = a.C >> 5;                     // 32-channel words per pixel
    constexpr int CB = CBT;                      // 64-channel k steps per tap
    const uint32_t PITCH = (uint32_t)a.pitch;  // bytes per pixel of the LDS image (C / 2 + padding: consecutive pixels start in different banks)
    BIE_CONV_STAMP(0);

    // ---- weights: this wave's 32 output channels (fragment-ordered FP4 image: 1 KiB per k step, contiguous)
    const int nrb = (a.OC + 31) >> 5;
    const int rbw = min(ocb * 4 + wave, nrb - 1);
    const uint4_t* wp = reinterpret_cast<const uint4_t*>(a.wimg) + (size_t)rbw * a.kb_per_row * 64 + lane;
    BIE_CONV_STAMP(1);

    // ---- 1. FP4 image of the input rows in LDS: everything -1.0 first (the border stays), then the pixels of the picture
    const int ih_lo = oh0 * a.stride - a.pad;
    const int IRn = (nrow - 1) * a.stride + KS;
    const int slot_pix = a.IR * a.WP;
    const long HW = (long)a.H * a.W;
    unsigned char* img = reinterpret_cast<unsigned char*>(conv_lds);
    {
        const int n16 = (int)((size_t)NIh * slot_pix * PITCH >> 4);
        uint4_t* i16 = reinterpret_cast<uint4_t*>(conv_lds);
        for (int i = tid; i < n16; i += 256) i16[i] = uint4_t{0xaaaaaaaau, 0xaaaaaaaau, 0xaaaaaaaau, 0xaaaaaaaau};
    }
    __syncthreads();
    const int ia = max(ih_lo, 0), ib = min(ih_lo + IRn, a.H);  // input rows of the picture this workgroup needs
    const int run = (ib - ia) * a.W;                            // their pixels: contiguous in every channel plane
    auto pack = [&](auto dt_c, auto nw_c) {
        constexpr int DT = decltype(dt_c)::value, NW = decltype(nw_c)::value;  // wave w: channel words w*NW .. w*NW + NW - 1 per pass
        constexpr unsigned EB = DT == BIE_F32 ? 4u : 2u;
        const unsigned plane_bytes = (unsigned)HW * EB;
        for (int sl = 0; sl < NIh; sl++) {
            // one image = C * H * W values <= 2^31 bytes: a buffer descriptor over it (out-of-range offsets would read 0, none occur)
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.x)) + (size_t)(b0 + sl) * a.C * HW * EB, 0,
                                                              (int)((size_t)a.C * HW * EB), 0x00020000);
            for (int q0 = 0; q0 < run; q0 += 64) {
                const int q = q0 + lane;
                const bool live = q < run;
                const int qc = live ? q : run - 1;  // lanes past the run re-read its last pixel (valid memory) and store nothing
                const int r = qc / a.W, c = qc - r * a.W;
                const int pix = (ia - ih_lo + r) * a.WP + c + a.pad;
                // with a stride the last picture columns may lie beyond the last tap (WP < W + 2 * pad): not part of the image -- stored, they
                // would land on the next row's left border (found by tests/sweeps/fuzz_other_ops.py at stride 3)
                const bool keep = live && c + a.pad < a.WP;
                unsigned char* dst = img + (size_t)(sl * slot_pix + pix) * PITCH;
                for (int cw0 = wave * NW; cw0 < CW; cw0 += 4 * NW) {
                    uint32_t v[NW];
                    conv_pixel_words<DT, NW>(rs, (unsigned)(ia * a.W + qc) * EB, (unsigned)(cw0 * 32) * plane_bytes, plane_bytes, v);
                    if (keep) {
#pragma unroll
                        for (int i = 0; i < NW; i++)
                            *reinterpret_cast<uint4_t*>(dst + (cw0 + i) * 16) = uint4_t{conv_fp4_from_bits8(v[i]), conv_fp4_from_bits8(v[i] >> 8),
                                                                                        conv_fp4_from_bits8(v[i] >> 16), conv_fp4_from_bits8(v[i] >> 24)};
                    }
                }
            }
        }
    };
    auto pack_dt = [&](auto dt_c) {
        if (CW % 16 == 0) pack(dt_c, std::integral_constant<int, 4>{});       // 512 / 1024 channels: 128 loads in flight per lane
        else if (CW % 8 == 0) pack(dt_c, std::integral_constant<int, 2>{});
        else pack(dt_c, std::integral_constant<int, 1>{});                    // (CW % 4 != 0: waves beyond CW idle)
    };
    if (a.dtype == BIE_F32) pack_dt(std::integral_constant<int, BIE_F32>{});
    else if (a.dtype == BIE_F16) pack_dt(std::integral_constant<int, BIE_F16>{});
    else pack_dt(std::integral_constant<int, BIE_BF16>{});
    __syncthreads();
    BIE_CONV_STAMP(2);

    // ---- 2. pixel fragments' base addresses: local pixel row r = 32 * pb + (lane & 31) -> (slot, output row, output column)
    const uint32_t img_base = (uint32_t)(uintptr_t)(lds_u32*)conv_lds;
    uint32_t pixaddr[NPB];
    int pslot[NPB], ppix[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; pb++) {
        const int r = pb * 32 + (lane & 31);
        int sl = r / Pw, p = r - sl * Pw;
        if (sl >= NIh) { sl = -1; p = 0; }
        pslot[pb] = sl; ppix[pb] = p;
        const int orow = p / a.OW, ocol = p - orow * a.OW;
        pixaddr[pb] = img_base + (uint32_t)(((sl < 0 ? 0 : sl) * a.IR + orow * a.stride) * a.WP + ocol * a.stride) * PITCH + (uint32_t)(lane >> 5) * 16u;
    }
    typedef float acc_t __attribute__((ext_vector_type(16)));
    // an accumulator can take its next MFMA only when the previous one has left the pipe (16 passes): with two pixel blocks a wave alone on its
    // SIMD ran 2 MFMAs per ~216 cycles.  Even / odd k steps get their own accumulators there (summed at the end: integers, exact)
    constexpr int NACC = NPB == 2 ? 2 : 1;
    acc_t acc[NACC][NPB];
#pragma unroll
    for (int h = 0; h < NACC; h++)
#pragma unroll
        for (int pb = 0; pb < NPB; pb++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[h][pb][i] = 0.0f;

    // ---- 3. k loop, one (tap, 64 channels) step per pass, rings of depth four: the pixel fragments are read from LDS TWO steps ahead and the
    // weight fragment comes from L2 THREE steps ahead -- a workgroup is often alone on its CU (one wave per SIMD), so nothing but the wave's own
    // look-ahead hides the LDS (~120 cycles) and L2 (~500) latencies; the first version (fragments one step ahead, weights two) ran 250 cycles
    // per step of 64 (profiles/r06_conv_timelines.txt).  The waits are the compiler's (counted vmcnt / lgkmcnt), the order is fenced.
    // ---- 3. contraction.  k step = (tap, 64 channels); a CHUNK = CH steps of one tap.  Inside a chunk every LDS address is the pixel's tap
    // address plus a COMPILE-TIME offset (ds_read_b128 offset:), every weight address the chunk pointer plus a compile-time offset: no branch and
    // no address arithmetic between the MFMAs.  The pixel fragments of chunk c + 1 (LDS) and the weight fragments of chunk c + 2 (L2) are requested
    // in one burst, then the CH * NPB MFMAs of chunk c issue back to back: an instruction between two MFMAs costs a lone in-order wave far more
    // than its issue slot (MI355X_MICROARCH.md).  History of this loop: profiles/r06_conv_timelines.txt.
    typedef const __attribute__((address_space(3))) uint4_t* lds_u4p;
    constexpr int CH = CB < 4 ? CB : 4, NCH = CB / CH, NCHUNK = T * NCH;  // chunks per tap / in all
    uint4_t xr[2][CH][NPB], wr[3][CH];
    auto load_w = [&](int c, uint4_t (&dst)[CH]) {  // chunk c (clamped: the look-ahead past the end re-reads the last chunk)
        const uint4_t* src = wp + (size_t)min(c, NCHUNK - 1) * (CH * 64);
#pragma unroll
        for (int d = 0; d < CH; d++) dst[d] = src[d * 64];
    };
    // tap of chunk c -> LDS byte offset ((ti * WP + tj) * PITCH); scalar, once per chunk
    auto tap_base = [&](int c) -> uint32_t {
        const int t = min(c, NCHUNK - 1) / NCH, ti = t / KS, tj = t - ti * KS;  // NCH and KS are compile-time constants: shifts / mul-hi
        return (uint32_t)(ti * a.WP + tj) * PITCH;
    };
    auto load_x = [&](int c, auto hc, uint4_t (&dst)[CH][NPB]) {  // hc: which part of the tap (compile time)
        constexpr int H = decltype(hc)::value;
        const uint32_t tb = tap_base(c);
#pragma unroll
        for (int pb = 0; pb < NPB; pb++) {
            const uint32_t ad = pixaddr[pb] + tb;
#pragma unroll
            for (int d = 0; d < CH; d++) dst[d][pb] = *reinterpret_cast<lds_u4p>(ad + (uint32_t)((H * CH + d) * 32));
        }
    };
    auto mfmas = [&](uint4_t (&xs)[CH][NPB], uint4_t (&ws)[CH]) {
#pragma unroll
        for (int d = 0; d < CH; d++) {
            const conv_v8i wa = {(int)ws[d].x, (int)ws[d].y, (int)ws[d].z, (int)ws[d].w, 0, 0, 0, 0};
#pragma unroll
            for (int pb = 0; pb < NPB; pb++) {
                const conv_v8i xb = {(int)xs[d][pb].x, (int)xs[d][pb].y, (int)xs[d][pb].z, (int)xs[d][pb].w, 0, 0, 0, 0};
                acc[d % NACC][pb] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa, xb, acc[d % NACC][pb], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            }
        }
    };
    // chunk c: P = pixel buffer it computes from (the other one is refilled with chunk c + 1), Wb = its weight buffer, HN = part of the tap of chunk c + 1
    auto chunk = [&](int c, auto pc, auto wc, auto hn) {
        constexpr int P = decltype(pc)::value, Wb = decltype(wc)::value;
        load_w(c + 2, wr[(Wb + 2) % 3]);
        load_x(c + 1, hn, xr[P ^ 1]);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(xr[P], wr[Wb]);
        __builtin_amdgcn_sched_barrier(0);
    };
    load_w(0, wr[0]);
    load_w(1, wr[1]);
    load_x(0, ic_t<0>{}, xr[0]);
    // 6 chunks = one turn of both rings (pixel buffers alternate, weight buffers rotate by three); the part-of-tap index of chunk c + 1 is
    // (c + 1) % NCH: with NCH in {1, 2} and c a multiple of 6 at the top of the loop it is a compile-time value in every position
    int c = 0;
#pragma unroll 1
    for (; c + 6 <= NCHUNK; c += 6) {
        chunk(c, ic_t<0>{}, ic_t<0>{}, ic_t<1 % NCH>{});
        chunk(c + 1, ic_t<1>{}, ic_t<1>{}, ic_t<2 % NCH>{});
        chunk(c + 2, ic_t<0>{}, ic_t<2>{}, ic_t<3 % NCH>{});
        chunk(c + 3, ic_t<1>{}, ic_t<0>{}, ic_t<4 % NCH>{});
        chunk(c + 4, ic_t<0>{}, ic_t<1>{}, ic_t<5 % NCH>{});
        chunk(c + 5, ic_t<1>{}, ic_t<2>{}, ic_t<6 % NCH>{});
    }
    if (c < NCHUNK) chunk(c, ic_t<0>{}, ic_t<0>{}, ic_t<1 % NCH>{});
    if (c + 1 < NCHUNK) chunk(c + 1, ic_t<1>{}, ic_t<1>{}, ic_t<2 % NCH>{});
    if (c + 2 < NCHUNK) chunk(c + 2, ic_t<0>{}, ic_t<2>{}, ic_t<3 % NCH>{});
    if (c + 3 < NCHUNK) chunk(c + 3, ic_t<1>{}, ic_t<0>{}, ic_t<4 % NCH>{});
    if (c + 4 < NCHUNK) chunk(c + 4, ic_t<0>{}, ic_t<1>{}, ic_t<5 % NCH>{});
    BIE_CONV_STAMP(3);
    BIE_CONV_STAMP(4);

    // ---- 4. y[b][oc][oh0 * OW + p]: lanes 0..31 of a register = 32 consecutive pixels of one output channel
    const long plane = (long)a.OH * a.OW;
    const int oc0 = ocb * 128 + wave * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int pb = 0; pb < NPB; pb++) {
        if (pslot[pb] < 0) continue;
        float* yp = a.y + ((long)(b0 + pslot[pb]) * a.OC) * plane + (long)oh0 * a.OW + ppix[pb];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int oc = oc0 + 8 * (i >> 2) + (i & 3);
            if (oc < a.OC) yp[oc * plane] = (NACC == 2 ? acc[0][pb][i] + acc[NACC - 1][pb][i] : acc[0][pb][i]) * a.scale;
        }
    }
    BIE_CONV_STAMP(5);
}

__global__ __launch_bounds__(256) void conv_weight_lanes_kernel(const uint32_t* __restrict__ wtaps, uint32_t* __restrict__ wl, int OC, int T, int CW,
                                                                long total) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;  // [block of 64 channels][quarter][t * CWW + j][lane]
    if (idx >= total) return;
    const int CWW = CW / 4, NWW = T * CWW;
    const int lane = (int)(idx & 63);
    const long r = idx >> 6;
    const int i = (int)(r % NWW);
    const int wq = (int)((r / NWW) & 3);
    const long blk = r / ((long)NWW * 4);
    const int t = i / CWW, j = i - t * CWW;
    const long oc = min(blk * 64 + lane, (long)OC - 1);
    wl[idx] = wtaps[(oc * T + t) * CW + wq * CWW + j];
}

struct ConvFusedPlan { int G, rpw, chunks, ocbs, IR, WP, PP, pxb; size_t lds; };

bool conv_fused_plan(int B, int C, int H, int W, int OC, int ks, int stride, int pad, int dil, ConvFusedPlan* out) {
    if (!(ks == 1 || ks == 3) || dil != 1 || !(C == 128 || C == 256 || C == 512) || B < 1 || OC < 1 || stride < 1 || pad < 0) return false;
    const int OH = (H + 2 * pad - (ks - 1) - 1) / stride + 1, OW = (W + 2 * pad - (ks - 1) - 1) / stride + 1;
    if (OH < 1 || OW < 1 || OW > 64) return false;
    ConvFusedPlan p;
    const int CW = C / 32, PS = CW + 4;
    // two 64-channel groups per lane halve the LDS reads per popcount and the redundant sign-packing, but also the number of workgroups
    const int rmax = 64 / OW;  // output pixels of a workgroup live in 64 lanes at write-out
    auto wgs = [&](int G, int rpw) { return (long)B * cdiv(OC, 64 * G) * cdiv(OH, rpw); };
    p.G = (OC > 64 && wgs(2, min(rmax, OH)) >= 256) ? 2 : 1;
    p.ocbs = cdiv(OC, 64 * p.G);
    // at least ~512 workgroups when the problem has them, whole output rows per workgroup, as many rows as still give that
    int rpw = min(rmax, OH);
    while (rpw > 1 && wgs(p.G, rpw) < 512) rpw--;
    p.rpw = rpw;
    p.chunks = cdiv(OH, rpw);
    p.IR = (rpw - 1) * stride + ks;
    p.WP = (OW - 1) * stride + ks;
    p.PP = (rpw * OW) | 1;
    p.pxb = OW % 7 == 0 ? 7 : 8;
    p.lds = ((size_t)p.IR * p.WP * PS + (size_t)p.G * 64 * p.PP) * 4;
    if (p.lds > 64 * 1024) return false;
    *out = p;
    return true;
}

struct ConvMfmaPlan { int npb, ni, rpw, chunks, ocbs, IR, WP, pitch; size_t lds; };

int conv_mfma_pitch_pad() {
    static const int pad = [] { const char* e = getenv("BIE_CONV_PITCH_PAD"); const int v = e ? atoi(e) : 16; return (v < 0 || v % 16) ? 16 : v; }();
    return pad;
}

bool conv_mfma_plan(int B, int C, int H, int W, int OC, int ks, int stride, int pad, int dil, ConvMfmaPlan* out) {
    if (!(ks == 1 || ks == 3) || dil != 1 || !(C == 64 || C == 128 || C == 256 || C == 512) || B < 1 || OC < 1 || stride < 1 || pad < 0) return false;
    const int OH = (H + 2 * pad - (ks - 1) - 1) / stride + 1, OW = (W + 2 * pad - (ks - 1) - 1) / stride + 1;
    if (OH < 1 || OW < 1 || OW > 128) return false;
    ConvMfmaPlan p;
    const int P = OH * OW;
    const size_t pitch = (size_t)C / 2 + conv_mfma_pitch_pad();
    p.pitch = (int)pitch;
    p.ocbs = cdiv(OC, 128);
    p.WP = (OW - 1) * stride + ks;
    p.npb = 4;
    if (P <= 128) {
        // ONE whole image per workgroup (two pixel blocks up to 64 pixels, four up to 128).  Two images per workgroup halve the weight traffic but
        // measured slower at every batch size (B = 128: 19.2 against 16.1 us, B = 512: 67.6 against 52.4): the two-block instance needs 228
        // registers (two workgroups per CU: one packs while the other multiplies), the four-block one 308 (one workgroup per CU)
        p.chunks = 1;
        p.rpw = OH;
        p.IR = (OH - 1) * stride + ks;
        p.ni = 1;
        p.npb = P <= 64 ? 2 : 4;
    } else {
        p.ni = 1;
        p.rpw = 128 / OW;
        p.chunks = cdiv(OH, p.rpw);
        p.IR = (p.rpw - 1) * stride + ks;
    }
    p.lds = (size_t)p.ni * p.IR * p.WP * pitch;
    if (p.lds > 64 * 1024) return false;
    *out = p;
    return true;
}

}  // namespace

bool binary_conv_mfma_ok(int B, int C, int H, int W, int OC, int ks, int stride, int pad, int dil) {
    ConvMfmaPlan p;
    return conv_mfma_plan(B, C, H, W, OC, ks, stride, pad, dil, &p);
}

int binary_conv_mfma_launch(const void* x, const uint8_t* wimg, float* y, int B, int C, int H, int W, int OC, int ks, int stride, int pad, int dil,
                            float scale, int dtype, hipStream_t st) {
    ConvMfmaPlan p;
    if (!conv_mfma_plan(B, C, H, W, OC, ks, stride, pad, dil, &p)) {
        set_error("bie_binary_conv2d_forward_mfma: geometry outside the one-launch matrix-pipe form (k in {1, 3}, dilation 1, C in {64, 128, 256, 512}, OW <= 128)");
        return BIE_ERR_UNSUPPORTED;
    }
    ConvMfmaArgs a;
    a.x = x; a.wimg = wimg; a.y = y;
    a.B = B; a.C = C; a.H = H; a.W = W; a.OC = OC;
    a.OH = (H + 2 * pad - (ks - 1) - 1) / stride + 1;
    a.OW = (W + 2 * pad - (ks - 1) - 1) / stride + 1;
    a.stride = stride; a.pad = pad; a.dtype = dtype;
    a.ni = p.ni; a.rpw = p.rpw; a.chunks = p.chunks; a.ocbs = p.ocbs; a.IR = p.IR; a.WP = p.WP;
    a.kb_per_row = 2 * cdiv(ks * ks * C, 128);
    a.scale = scale;
    a.pitch = p.pitch;
    const long groups = p.chunks == 1 ? cdiv(B, p.ni) : (long)B * p.chunks;
    const unsigned grid = (unsigned)(groups * p.ocbs);
    const int cbt = C / 64;
#define BIE_CM3(KSV, NPBV, CBV) hipLaunchKernelGGL((xnor_conv_mfma_kernel<KSV, NPBV, CBV>), dim3(grid), dim3(256), p.lds, st, a)
#define BIE_CM2(KSV, NPBV) do { if (cbt == 8) BIE_CM3(KSV, NPBV, 8); else if (cbt == 4) BIE_CM3(KSV, NPBV, 4); else if (cbt == 2) BIE_CM3(KSV, NPBV, 2); else BIE_CM3(KSV, NPBV, 1); } while (0)
    if (ks == 3) { if (p.npb == 4) BIE_CM2(3, 4); else BIE_CM2(3, 2); }
    else { if (p.npb == 4) BIE_CM2(1, 4); else BIE_CM2(1, 2); }
#undef BIE_CM2
#undef BIE_CM3
    return check_launch("xnor_conv_mfma_kernel");
}

bool binary_conv_fused_ok(int B, int C, int H, int W, int OC, int ks, int stride, int pad, int dil) {
    ConvFusedPlan p;
    return conv_fused_plan(B, C, H, W, OC, ks, stride, pad, dil, &p);
}

size_t binary_conv_weight_lanes_words(int OC, int C, int ks) { return (size_t)cdiv(OC, 64) * 4 * ((size_t)ks * ks * (C / 128)) * 64; }

int binary_conv_weight_lanes_launch(const uint32_t* wtaps, uint32_t* wl, int OC, int C, int ks, hipStream_t st) {
    const long total = (long)binary_conv_weight_lanes_words(OC, C, ks);
    hipLaunchKernelGGL(conv_weight_lanes_kernel, dim3((unsigned)cdivl(total, 256)), dim3(256), 0, st, wtaps, wl, OC, ks * ks, C / 32, total);
    return check_launch("conv_weight_lanes_kernel");
}

int binary_conv_fused_launch(const void* x, const uint32_t* wl, float* y, int B, int C, int H, int W, int OC, int ks, int stride, int pad, int dil,
                             float scale, int dtype, hipStream_t st) {
    ConvFusedPlan p;
    if (!conv_fused_plan(B, C, H, W, OC, ks, stride, pad, dil, &p)) {
        set_error("bie_binary_conv2d_forward_fused: geometry outside the one-launch form (k in {1, 3}, dilation 1, C in {128, 256, 512}, OW <= 64)");
        return BIE_ERR_UNSUPPORTED;
    }
    ConvFusedArgs a;
    a.x = x; a.wl = wl; a.y = y;
    a.B = B; a.C = C; a.H = H; a.W = W; a.OC = OC;
    a.OH = (H + 2 * pad - (ks - 1) - 1) / stride + 1;
    a.OW = (W + 2 * pad - (ks - 1) - 1) / stride + 1;
    a.stride = stride; a.pad = pad; a.dtype = dtype;
    a.rpw = p.rpw; a.chunks = p.chunks; a.ocbs = p.ocbs; a.IR = p.IR; a.WP = p.WP; a.PP = p.PP;
    a.scale = scale;
    const unsigned grid = (unsigned)((long)B * p.chunks * p.ocbs);
    const int cww = C / 128;
#define BIE_CF4(KSV, CWWV, GV, PXV) hipLaunchKernelGGL((xnor_conv_fused_kernel<KSV, CWWV, GV, PXV>), dim3(grid), dim3(256), p.lds, st, a)
#define BIE_CF3(KSV, CWWV, GV) do { if (p.pxb == 7) BIE_CF4(KSV, CWWV, GV, 7); else BIE_CF4(KSV, CWWV, GV, 8); } while (0)
#define BIE_CF2(KSV, CWWV) do { if (p.G == 2) BIE_CF3(KSV, CWWV, 2); else BIE_CF3(KSV, CWWV, 1); } while (0)
#define BIE_CF1(KSV) do { if (cww == 4) BIE_CF2(KSV, 4); else if (cww == 2) BIE_CF2(KSV, 2); else BIE_CF2(KSV, 1); } while (0)
    if (ks == 3) BIE_CF1(3); else BIE_CF1(1);
#undef BIE_CF1
#undef BIE_CF2
#undef BIE_CF3
#undef BIE_CF4
    return check_launch("xnor_conv_fused_kernel");
}

#ifdef BIE_CONV_LAB
extern "C" int bie_debug_conv_stamps(unsigned long long* out, int n_wg) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_conv_stamps), (size_t)n_wg * 8 * sizeof(unsigned long long));
}
#endif

}  // namespace bie
