// fp32-grade convolution / linear kernel on the bf16 matrix pipe, WIDE form (a3d_conv_desc.precision == 2 with w_x3 given).
//
// Same arithmetic as conv_bf16x3.hip (every fp32 operand split exactly into hi | mid | lo bf16 terms, six
// v_mfma_f32_32x32x16_bf16 per 16-deep k step, fp32 accumulation) and the same per-output operation order -- the two kernels
// agree bit for bit (tests/test_gpu_parity.py), so the launcher chooses by problem size.  What differs is the data movement,
// shaped by what the split-operand Winograd GEMM taught (conv_wino.hip, "2x-wide"): these loops are bound by the issue cost of
// their vector-memory instructions (~100 cycles each inside an MFMA stream, 1 KiB per instruction) and by the VALU / VGPR -> LDS
// work of the in-kernel split, i.e. by operand bytes per MFMA, not by the matrix pipe.
//
//   conv_x3_kernel<2>: 128 px x 128 ch per 256-thread workgroup, both operands fp32 through registers, split in the kernel:
//       16 KiB-loads and 1024 split float4 per 96 MFMAs, fragment reads 0.5 per MFMA.
//   this kernel:       256 px x 256 ch per 512-thread workgroup (8 waves as 4 x 2, wave tile 64 px x 128 ch = 8 accumulators):
//       16 activation loads + 24 weight DMA pieces and 1024 split float4 per 384 MFMAs (0.10 vector-memory instructions per
//       MFMA instead of 0.17, a quarter of the split work), fragment reads 0.375 per MFMA.
//       WEIGHTS are pre-split once per layer into bf16 planes in chunk-major LDS-image order (a3d_conv_desc.w_x3, a3d_split_bf16x3_chunk
//       with chunk = 16) and go global -> LDS by LDS-DMA: no registers, no VALU, no VGPR -> LDS stores for that operand.
//
// Schedule.  One 16-deep chunk per iteration = 48 MFMAs per wave in four channel blocks n = 0..3 of 12 (six terms x two pixel
// blocks); ONE barrier per iteration, after block 1:
//   n = 0, 1 : fragments b(c) (activations, all three planes) and a(c)[n] in registers; the staged activation chunk c+1 is split
//              into X stage (c+1) % 2; a(c)[n+1] is read one block ahead
//   -- wait: weight DMA of chunk c+1 landed (counted vmcnt), own LDS traffic drained; barrier --
//   n = 2    : weight DMA of chunk c+2 into W stage (c+2) % 3 (three weight stages: the fragments a(c)[n] are read block by
//              block, so stage c % 3 stays in use to the end of the iteration), activation loads of chunk c+3 (two register
//              sets), a(c)[3] read
//   n = 3    : b(c+1) and a(c+1)[0] read from the stages the barrier has just completed
// Registers: 8 x 16 accumulators + two b sets (24 each) + two a sets (12 each) + 16 staging = 216 of the 256 that two waves per
// SIMD leave each wave.  LDS: 2 x 24 KiB (X) + 3 x 24 KiB (W) + scale | shift.
#include "conv_common.h"
#include <type_traits>

namespace {
typedef __bf16 wx_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wx_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t wx_rsrc(const void *p, unsigned bytes) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ f32x4 wx_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void wx_dma16(__amdgpu_buffer_rsrc_t r, __bf16 *lds_dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds_dst, 16, voff, soff, 0, 0);
}
// x = h + m + l exactly (round-to-nearest-even at each level): conv_bf16x3.hip's split3
__device__ __forceinline__ void wx_split3(const f32x4 v, wx_bf16x4 &h, wx_bf16x4 &m, wx_bf16x4 &l) {
    h = __builtin_convertvector(v, wx_bf16x4);
    const f32x4 r1 = v - __builtin_convertvector(h, f32x4);
    m = __builtin_convertvector(r1, wx_bf16x4);
    const f32x4 r2 = r1 - __builtin_convertvector(m, f32x4);
    l = __builtin_convertvector(r2, wx_bf16x4);
}

typedef _Float16 wx_h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 wx_h16x8 __attribute__((ext_vector_type(8)));
// fp16x2 (precision 3): x * s = h + l, conv_bf16x3.hip's split2h
__device__ __forceinline__ void wx_split2h(const f32x4 v, const float s, wx_h16x4 &h, wx_h16x4 &l) {
    const f32x4 xs = v * s;
    h = __builtin_convertvector(xs, wx_h16x4);
    l = __builtin_convertvector(xs - __builtin_convertvector(h, f32x4), wx_h16x4);
}

constexpr int XW_BM = 256, XW_BN = 256, XW_BK = 16;
// PH4: the phases (bit 2 dy + dx) whose 2x2 window inside the 3x3 neighbourhood holds tap kh = tap / 3, kw = tap % 3
constexpr unsigned xw_tap_phases(const int tap) {
    const int th = tap / 3, tw = tap - 3 * th;
    const unsigned rows = th == 0 ? 0x3u : (th == 1 ? 0xFu : 0xCu);  // dy <= kh <= dy + 1
    const unsigned cols = tw == 0 ? 0x5u : (tw == 1 ? 0xFu : 0xAu);  // dx <= kw <= dx + 1
    return rows & cols;
}
constexpr int XW_PX = XW_BM * XW_BK, XW_PW = XW_BN * XW_BK;  // one operand plane of one stage (bf16 elements; 32-byte rows)
constexpr int xw_lds_bytes(int NP) { return (2 * NP * XW_PX + 3 * NP * XW_PW) * 2 + 2 * XW_BN * 4; }
constexpr int XD_NST = 4;  // XD: stages of the dual-DMA ring (each holds one chunk of BOTH operands)
constexpr int xd_lds_bytes() { return XD_NST * (2 * XW_PX + 2 * XW_PW) * 2 + 2 * XW_BN * 4; }

// F16: the fp16x2 arithmetic (a3d_conv_desc.precision == 3) -- two operand planes, three product terms (h.h, h.l, l.h), activation rows
// scaled per image from d.in_amax, w_x3 = the filter pre-split by a3d_split_f16x2_chunk with d.w_scale.
// PH4 (fp16x2 only; a3d_conv_desc.phase == 5): ALL FOUR output phases of a 3x3 pad-1 convolution over a nearest-x2 upsampled input in
// one launch.  The GEMM is a 3x3 convolution on the source grid whose 4 Cout columns are the phases' pre-summed 2x2 filters placed
// at their positions inside the 3x3 neighbourhood (phase (dy, dx) uses taps kh - dy, kw - dx in {0, 1}; the other taps are zero).
// Columns are ordered so that every wave holds all four phases: column j = 128 g + 32 phase + c  <->  output channel 32 g + c,
// i.e. channel block n of a wave IS phase n.  A tap that lies outside a phase's window contributes nothing, so the wave skips that
// block's MFMAs (uniform branch): corner taps multiply one block, edge taps two, the centre all four -- exactly the 16 tap-phase
// products of the four-launch form, in the same order per output (bit-identical results), but every activation chunk is loaded and
// split ONCE for the phases that share it (9 tap loads instead of 16) and all waves do equal work on every tap.
// XD (fp16x2 only; a3d_conv_desc.x_h2): the ACTIVATIONS arrive pre-split as well -- [pixel][C/16][h | l][16] fp16, written by their producer
// (a3d_roi_align_fpn's out_h2, a3d_presplit_f16x2) with the bits this kernel's loader would compute -- and take the filter's road:
// global -> LDS by LDS-DMA, the image's half swizzle applied on the global side of each lane's address, zero padding by the buffer
// range check.  No activation registers, no split arithmetic, no VGPR -> LDS stores: a chunk is 2 + 2 DMA instructions per wave beside
// 24 MFMAs.  Ring of XD_NST = 4 stages (32 KiB each): chunk c lives in stage c % 4; ONE bare s_barrier per chunk, behind the last
// fragment read of the chunk (a(c)[3], issued under block 2's MFMAs), after which the DMA of chunk c + 4 goes straight into the
// stage of chunk c and has three iterations to land (fc1 streams its 3.2 GB of activations from HBM).  Same fragments, same term
// order per output as the register-staged loop: bit-identical results (tests/test_gpu_parity.py).
// (Round 5, built, bit-identical, measured and removed -- profiles/r05_x3w_pp_ab.txt: this kernel's plain fp16x2 loop in the Winograd GEMM's
// PING-PONG form -- the workgroup's halves in antiphase, a memory phase (12 fragment reads, split + LDS stores of the next chunk, filter
// DMA, global loads) alternating with 24 MFMAs back to back.  4-15 % SLOWER on all eight layers (fc1 4.445 | 4.073 ms, fc2 0.490 | 0.425,
// p2 lateral 0.825 | 0.773, 6400-row head FC 2.187 | 2.013): here the activations pass through registers, and the hand-dealt loop below
// already runs the split and the LDS stores in the MFMAs' shadows -- bunching them leaves the partner's 24 MFMAs too short a cover.)
template <bool F16, bool PH4 = false, bool XD = false>
__global__ __launch_bounds__(512, 1) void conv_x3w_kernel(const a3d_conv_desc d, const int M, const int ntiles, const int nblk) {
    static_assert(!XD || F16, "pre-split activations belong to the fp16x2 arithmetic");
    constexpr int NP = F16 ? 2 : 3;
    constexpr int XW_XST = NP * XW_PX, XW_WST = NP * XW_PW;  // one stage
    constexpr int TM = 2, TN = 4, BM = XW_BM, BN = XW_BN, BKT = XW_BK, LKB = XW_BK;
    constexpr int TPR = BKT / 4, RPP = 512 / TPR, XR = BM / RPP;  // 4 lanes x float4 per row, 128 rows per pass, 2 passes
    constexpr int PX = XW_PX, PW = XW_PW;
    static_assert(XR == 2, "the counted vmcnt waits below assume two activation loads per chunk");
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    constexpr int NXS = XD ? XD_NST : 2, NWS = XD ? XD_NST : 3;  // X / W stages
    __bf16 *const Xs = lds;                    // [NXS][NP][256][16]
    __bf16 *const Ws = lds + NXS * XW_XST;     // [NWS][NP][256][16]
    float *const ss = reinterpret_cast<float *>(lds + NXS * XW_XST + NWS * XW_WST);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int logical = a3d_xcd_remap(blockIdx.x, nblk);
    const int mt = logical / ntiles, nt = logical - mt * ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int nk_all = d.Kpad / BKT;
    // split-K (a3d_conv_desc.splitk > 1, the 50176-deep head FCs): blockIdx.y = z multiplies chunks [z * kps, (z + 1) * kps) and
    // stores its raw partial sums to workspace[z]; conv_splitk_reduce_v2_kernel adds the slices in z order and applies the epilogue
    const int kps = (nk_all + d.splitk - 1) / d.splitk;
    const int kbeg = blockIdx.y * kps;
    const int nk = min(nk_all, kbeg + kps);  // first chunk index past this slice
    const int lr = tid / TPR, lc = (tid % TPR) * 4;
    // LDS image (both operands): [plane][row][16 k] bf16, 32-byte rows; the 16-byte half holding k = 8h..8h+7 of row r sits at slot
    // h ^ ((r >> 3) & 1) -- conv_bf16x3.hip's image (conflict-free ds_read_b128 and ds_write_b64)
    const int lcs = (lc & 7) | ((((lc >> 3) ^ (lr >> 3)) & 1) << 3);
    const int cs4 = d.Cin * 4;
    const int CinT = d.Cin + d.Cin2;
    // (XD: the pre-split tensors hold the same number of bytes per pixel as the fp32 ones: Cin / 16 chunks x 64 B)
    const void *const x0p = XD ? d.x_h2 : (const void *)d.x;
    const void *const x1p = XD ? (d.x2_h2 ? d.x2_h2 : d.x_h2) : (const void *)(d.x2 ? d.x2 : d.x);
    const __amdgpu_buffer_rsrc_t rx = wx_rsrc(x0p, (unsigned)((size_t)d.B * d.H * d.W * (size_t)cs4));
    const __amdgpu_buffer_rsrc_t rx2 = wx_rsrc(x1p, (unsigned)((size_t)d.B * d.H * d.W * (size_t)cs4));
    const unsigned w3chunk = (unsigned)d.Cout * 32u * NP;  // bytes of one chunk of w_x3: NP planes x Cout rows x 32 B
    const __amdgpu_buffer_rsrc_t rw = wx_rsrc(d.w_x3, (unsigned)((size_t)nk_all * w3chunk));

    const float sw = F16 ? d.w_scale : 1.f;
    float sxr[XR];  // fp16x2: the scale of each loader row's image
    int rowoff[XR];
    unsigned vmask[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int m = m0 + lr + RPP * i;
        const bool rok = m < M;
        const int mm = rok ? m : 0;
        const int hw = d.Ho * d.Wo;
        const int b = mm / hw, r = mm - b * hw;
        const int oh = r / d.Wo, ow = r - oh * d.Wo;
        int ih0 = oh * d.stride - d.pad, iw0 = ow * d.stride - d.pad;
        if (!PH4 && d.phase) {  // 2x2 taps of output phase (dy,dx) of an upsampled 3x3 conv, on the source grid (conv_gemm_v2.hip)
            ih0 = oh - 1 + ((d.phase - 1) >> 1);
            iw0 = ow - 1 + ((d.phase - 1) & 1);
        }
        sxr[i] = (F16 && rok) ? a3d_in_scale(d, b) : 1.f;
        unsigned mask = 0;
        for (int kh = 0; kh < d.KH; ++kh)
            for (int kw = 0; kw < d.KW; ++kw)
                mask |= (rok && (unsigned)(ih0 + kh) < (unsigned)d.H && (unsigned)(iw0 + kw) < (unsigned)d.W) ? (1u << (kh * d.KW + kw)) : 0u;
        rowoff[i] = ((b * d.H + ih0) * d.W + iw0) * cs4 + lc * 4;
        vmask[i] = mask;
    }
    // position of the next activation chunk to load inside the filter (k = (kh, kw, c): chunk kbeg starts at tap kbeg*16 / CinT)
    int kc = kbeg, c0 = (kbeg * BKT) % CinT, kh = ((kbeg * BKT) / CinT) / d.KW, kw = ((kbeg * BKT) / CinT) % d.KW;
    f32x4 xsA[XR], xsB[XR];
    auto load_chunk = [&](f32x4 (&xs)[XR]) {
        const int tap = kh * d.KW + kw;
        const unsigned livebit = (kc < nk) ? 1u : 0u;
        const bool second = c0 >= d.Cin;  // channel concat: the second source supplies channels Cin .. Cin+Cin2-1 of every tap
        const __amdgpu_buffer_rsrc_t r = second ? rx2 : rx;
        const int tapoff = (kh * d.W + kw) * cs4 + (second ? c0 - d.Cin : c0) * 4;
#pragma unroll
        for (int i = 0; i < XR; ++i) xs[i] = wx_load4(r, ((vmask[i] >> (tap & 31)) & livebit) ? rowoff[i] + tapoff : -1, 0);
        ++kc;
        c0 += BKT;
        if (c0 >= CinT) {
            c0 = 0;
            if (++kw == d.KW) {
                kw = 0;
                ++kh;
            }
        }
    };
    struct Split {
        wx_bf16x4 h, m, l;  // (fp16x2: h, m hold the two fp16 planes' bits)
    };
    auto split = [&](const f32x4 v, const int i, Split &o) {
        if constexpr (F16) {
            wx_h16x4 h, l;
            wx_split2h(v, sxr[i], h, l);
            o.h = __builtin_bit_cast(wx_bf16x4, h);
            o.m = __builtin_bit_cast(wx_bf16x4, l);
        } else {
            wx_split3(v, o.h, o.m, o.l);
        }
    };
    auto put = [&](const int xst, const int i, const Split &v) {  // the planes of loader row lr + RPP i
        __bf16 *p = Xs + xst * XW_XST + (lr + RPP * i) * LKB + lcs;
        *reinterpret_cast<wx_bf16x4 *>(p) = v.h;
        *reinterpret_cast<wx_bf16x4 *>(p + PX) = v.m;
        if constexpr (!F16) *reinterpret_cast<wx_bf16x4 *>(p + 2 * PX) = v.l;
    };
    // weights: w_x3 [Kpad/16][3][Cout][16] bf16; one (chunk, plane) tile of this workgroup's 256 rows is an 8 KiB run = 8 DMA
    // wave-instructions of 32 rows.  Lane i of an instruction lands at LDS byte 16 i of its 1 KiB = row i/2, half i%2, and fetches
    // the k half that the image keeps there: half ^ ((row >> 3) & 1)  (row base is a multiple of 32).  Rows past Cout read the
    // next plane's rows or, past the end of w_x3, zeros: their accumulators are never stored.
    const int wvoff = (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4);
    // chunk and W stage of the next weight DMA (past the slice the activations are zero; past the end of w_x3 the DMA reads zeros)
    int dma_c = kbeg, dma_st = 0;
    auto dma_w = [&]() {
        __bf16 *Wt = Ws + dma_st * XW_WST;
        const int base = __builtin_amdgcn_readfirstlane(dma_c * (int)w3chunk + n0 * 32);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int j = wave * NP + i;  // 8 NP instructions per chunk: plane j/8, row group j%8
            const int p = j >> 3, g = j & 7;
            wx_dma16(rw, Wt + p * PW + g * 32 * LKB, wvoff, base + __builtin_amdgcn_readfirstlane(p * d.Cout * 32 + g * 1024));
        }
        ++dma_c;
        dma_st = dma_st == NWS - 1 ? 0 : dma_st + 1;
    };
    // XD: the activation chunk by DMA.  One (chunk, plane) tile of the workgroup's 256 rows = 8 instructions of 32 rows; wave w issues
    // instructions j = 2 w, 2 w + 1 (plane j / 8, row group j % 8): lane i -> row i / 2, LDS half i % 2, fetching the k half the image
    // keeps there.  Its two rows' pixel offsets and tap masks are per-lane constants; the chunk's tap / channel offset is uniform.
    int xd_off[2];
    unsigned xd_mask[2];
    if constexpr (XD) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int g = (wave * 2 + i) & 7;
            const int m = m0 + g * 32 + (lane >> 1);
            const bool rok = m < M;
            const int mm = rok ? m : 0;
            const int hw = d.Ho * d.Wo;
            const int b = mm / hw, r = mm - b * hw;
            const int oh = r / d.Wo, ow = r - oh * d.Wo;
            const int ih0 = oh * d.stride - d.pad, iw0 = ow * d.stride - d.pad;
            unsigned mask = 0;
            for (int kh = 0; kh < d.KH; ++kh)
                for (int kw = 0; kw < d.KW; ++kw)
                    mask |= (rok && (unsigned)(ih0 + kh) < (unsigned)d.H && (unsigned)(iw0 + kw) < (unsigned)d.W) ? (1u << (kh * d.KW + kw)) : 0u;
            xd_off[i] = ((b * d.H + ih0) * d.W + iw0) * cs4 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4);
            xd_mask[i] = mask;
        }
    }
    int xd_st = 0;
    auto dma_x = [&]() {  // next activation chunk (position kc / c0 / kh / kw, shared with load_chunk) into X stage xd_st
        const int tap = kh * d.KW + kw;
        const unsigned livebit = (kc < nk) ? 1u : 0u;
        const bool second = c0 >= d.Cin;
        const __amdgpu_buffer_rsrc_t r = second ? rx2 : rx;
        const int tapoff = __builtin_amdgcn_readfirstlane((kh * d.W + kw) * cs4 + (second ? c0 - d.Cin : c0) * 4);
        const int p = wave >> 2;  // plane of this wave's two instructions
        __bf16 *Xt = Xs + xd_st * XW_XST + p * PX;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int g = (wave * 2 + i) & 7;
            wx_dma16(r, Xt + g * 32 * LKB, ((xd_mask[i] >> (tap & 31)) & livebit) ? xd_off[i] + tapoff : -1, p * 32);
        }
        ++kc;
        c0 += BKT;
        if (c0 >= CinT) {
            c0 = 0;
            if (++kw == d.KW) {
                kw = 0;
                ++kh;
            }
        }
        xd_st = xd_st == XD_NST - 1 ? 0 : xd_st + 1;
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    a3d_stage_scale_shift(ss, d, n0, BN, tid);

    // fragments: row = lane % 32 of a 32-row block, k = 8 * (lane / 32) .. + 7
    const int frow = lane & 31;
    const int frag_off = frow * LKB + ((((lane >> 5) ^ (frow >> 3)) & 1) << 3);
    const __bf16 *const fX = Xs + (wm * TM * 32) * LKB + frag_off;
    const __bf16 *const fW = Ws + (wn * TN * 32) * LKB + frag_off;
    struct FragA {
        wx_bf16x8 p[NP];  // weights of one 32-channel block, hi | mid | lo
    };
    struct FragB {
        wx_bf16x8 p[NP][TM];  // activations of the wave's two 32-pixel blocks
    };
    auto rdA = [&](FragA &A, const int wst, const int n) {
#pragma unroll
        for (int p = 0; p < NP; ++p) A.p[p] = *reinterpret_cast<const wx_bf16x8 *>(fW + wst * XW_WST + p * PW + n * 32 * LKB);
    };
    auto rdB = [&](FragB &Bf, const int xst, const int p) {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) Bf.p[p][mi] = *reinterpret_cast<const wx_bf16x8 *>(fX + xst * XW_XST + p * PX + mi * 32 * LKB);
    };

#define XW_FENCE __builtin_amdgcn_sched_barrier(0);
#define XW_MFMA(C, A, Bv)                                                                                                              \
    if constexpr (F16) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wx_h16x8, A), __builtin_bit_cast(wx_h16x8, Bv), C, 0, 0, 0); \
    else C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bv, C, 0, 0, 0);
#define XW_TERM(N, A, Bf, PA, PB)                \
    XW_MFMA(acc[N][0], A.p[PA], Bf.p[PB][0]) \
    XW_MFMA(acc[N][1], A.p[PA], Bf.p[PB][1])
// (one MFMA, then its share of the block's other instructions: the wave issues in order, so what follows an MFMA runs in its shadow)
#define XW_MIX(NV)                                         \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
// the tail of a channel block: the four terms that carry no other work
#define XW_REST(N, A, Bf)                                                                                       \
    if constexpr (!F16) {                                                                                        \
        XW_TERM(N, A, Bf, 1, 1) XW_FENCE XW_TERM(N, A, Bf, 2 % NP, 0) XW_FENCE XW_TERM(N, A, Bf, 0, 2 % NP) XW_FENCE \
    }

    int wst = 0;  // W stage of the chunk being multiplied
    // PH4: the tap of the chunk being multiplied and the phases (= channel blocks) whose 2x2 window holds it
    const int cpt = PH4 ? CinT / BKT : 1;  // chunks per tap
    // (Measured and not taken, round 4: the filter DMA pieces of the phases a tap does not multiply -- 7.1 of 16 pieces per chunk on
    // average -- left out: 2.70 | 2.82 ms on the 120x160 stage, 1.43 | 1.40 on 60x80: the filter stream is not what bounds this loop.)
    // one iteration; xst = X stage of chunk c (compile-time), Bc / Bn = the b sets of chunk c / c+1, xs = the staged chunk c+1
    auto iteration = [&](const int xst, FragA &A0, FragA &A1, FragB &Bc, FragB &Bn, f32x4 (&xs)[XR]) {
        const int wnext = wst == 2 ? 0 : wst + 1;
        Split s;
        // ---- n = 0
        XW_TERM(0, A0, Bc, 0, 0)
        split(xs[0], 0, s);
        XW_MIX(12)
        XW_FENCE
        XW_TERM(0, A0, Bc, 0, 1)
        put(xst ^ 1, 0, s);
        XW_MIX(4)
        XW_FENCE
        XW_TERM(0, A0, Bc, 1, 0)
        rdA(A1, wst, 1);
        XW_MIX(4)
        XW_FENCE
        XW_REST(0, A0, Bc)
        // ---- n = 1
        XW_TERM(1, A1, Bc, 0, 0)
        split(xs[1], 1, s);
        XW_MIX(12)
        XW_FENCE
        XW_TERM(1, A1, Bc, 0, 1)
        put(xst ^ 1, 1, s);
        XW_MIX(4)
        XW_FENCE
        XW_TERM(1, A1, Bc, 1, 0)
        rdA(A0, wst, 2);
        XW_MIX(4)
        XW_FENCE
        XW_REST(1, A1, Bc)
        __asm__ volatile("s_waitcnt vmcnt(2)" ::: "memory");  // all but the two youngest (activation loads): the weight DMA of chunk c+1 has landed
        __syncthreads();
        // ---- n = 2
        XW_TERM(2, A0, Bc, 0, 0)
        dma_w();
        XW_MIX(6)
        XW_FENCE
        XW_TERM(2, A0, Bc, 0, 1)
        load_chunk(xs);
        XW_MIX(8)
        XW_FENCE
        XW_TERM(2, A0, Bc, 1, 0)
        rdA(A1, wst, 3);
        XW_MIX(4)
        XW_FENCE
        XW_REST(2, A0, Bc)
        // ---- n = 3
        XW_TERM(3, A1, Bc, 0, 0)
        rdB(Bn, xst ^ 1, 0);
        rdB(Bn, xst ^ 1, 1);
        XW_MIX(4)
        XW_FENCE
        XW_TERM(3, A1, Bc, 0, 1)
        if constexpr (!F16) rdB(Bn, xst ^ 1, 2);
        rdA(A0, wnext, 0);
        XW_MIX(4)
        XW_FENCE
        XW_TERM(3, A1, Bc, 1, 0)
        XW_FENCE
        XW_REST(3, A1, Bc)
        wst = wnext;
    };

    // PH4: one chunk of a tap whose phase set ACT is a compile-time constant (nine sets: four corners with one phase, four edges with
    // two, the centre with all four).  The schedule is the dense one with the loader work re-dealt over the ACTIVE blocks' terms, so
    // every MFMA pair still carries its share of the split / LDS / vector-memory instructions; A0 holds the first active block's
    // weights on entry, `nf` = first active block of the NEXT chunk (the next tap's at a tap boundary).
#define PH_T(N, A, PA, PB, ...) XW_TERM(N, A, Bc, PA, PB) __VA_ARGS__ XW_MIX(12) XW_FENCE
    auto iter_ph4 = [&](auto act_c, const int xst, FragA &A0, FragA &A1, FragB &Bc, FragB &Bn, f32x4 (&xs)[XR], const int nf) __attribute__((always_inline)) {
        constexpr unsigned ACT = decltype(act_c)::value;
        constexpr int NA = __builtin_popcount(ACT);
        constexpr int b0 = __builtin_ctz(ACT), b1 = NA > 1 ? __builtin_ctz(ACT & (ACT - 1)) : 0;
        const int wnext = wst == 2 ? 0 : wst + 1;
        Split s;
        // (bf16x3, round 5: a block's three further terms -- m.m, l.h, h.l, XW_REST -- follow its first three and carry no loader work;
        // the third activation plane of the next chunk is read with the other two.  Same term order per accumulator as conv_x3_kernel's:
        // bit-identical to the four phase launches of that arithmetic as well.)
        if constexpr (NA == 4) {
            PH_T(0, A0, 0, 0, split(xs[0], 0, s);)
            PH_T(0, A0, 0, 1, put(xst ^ 1, 0, s);)
            PH_T(0, A0, 1, 0, rdA(A1, wst, 1);)
            XW_REST(0, A0, Bc)
            PH_T(1, A1, 0, 0, split(xs[1], 1, s);)
            PH_T(1, A1, 0, 1, put(xst ^ 1, 1, s);)
            PH_T(1, A1, 1, 0, rdA(A0, wst, 2);)
            XW_REST(1, A1, Bc)
            __asm__ volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __syncthreads();
            PH_T(2, A0, 0, 0, dma_w();)
            PH_T(2, A0, 0, 1, load_chunk(xs);)
            PH_T(2, A0, 1, 0, rdA(A1, wst, 3);)
            XW_REST(2, A0, Bc)
            PH_T(3, A1, 0, 0, rdB(Bn, xst ^ 1, 0); rdB(Bn, xst ^ 1, 1);)
            PH_T(3, A1, 0, 1, if constexpr (!F16) rdB(Bn, xst ^ 1, 2); rdA(A0, wnext, nf);)
            PH_T(3, A1, 1, 0, )
            XW_REST(3, A1, Bc)
        } else if constexpr (NA == 2) {
            PH_T(b0, A0, 0, 0, split(xs[0], 0, s);)
            PH_T(b0, A0, 0, 1, put(xst ^ 1, 0, s); split(xs[1], 1, s);)
            PH_T(b0, A0, 1, 0, put(xst ^ 1, 1, s); rdA(A1, wst, b1);)
            XW_REST(b0, A0, Bc)
            __asm__ volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __syncthreads();
            PH_T(b1, A1, 0, 0, dma_w(); rdB(Bn, xst ^ 1, 0);)
            PH_T(b1, A1, 0, 1, load_chunk(xs); rdB(Bn, xst ^ 1, 1);)
            PH_T(b1, A1, 1, 0, if constexpr (!F16) rdB(Bn, xst ^ 1, 2); rdA(A0, wnext, nf);)
            XW_REST(b1, A1, Bc)
        } else {
            static_assert(NA == 1, "corner, edge or centre tap");
            PH_T(b0, A0, 0, 0, split(xs[0], 0, s); put(xst ^ 1, 0, s);)
            PH_T(b0, A0, 0, 1, split(xs[1], 1, s); put(xst ^ 1, 1, s);)
            __asm__ volatile("s_waitcnt vmcnt(2)" ::: "memory");
            __syncthreads();
            PH_T(b0, A0, 1, 0, dma_w(); XW_FENCE load_chunk(xs); rdB(Bn, xst ^ 1, 0); rdB(Bn, xst ^ 1, 1); if constexpr (!F16) rdB(Bn, xst ^ 1, 2);)  // (DMA before the loads: the counted wait)
            XW_REST(b0, A0, Bc)
            rdA(A0, wnext, nf);  // (A0 was in use until the last term)
            XW_FENCE
        }
        wst = wnext;
    };
#undef PH_T

    FragA A0, A1;
    FragB B0, B1;
    if constexpr (XD && !PH4) {
        // ---- dual-DMA ring.  An iteration = one 16-deep chunk = four channel blocks of six MFMAs.  Fragments: b(c) and a(c)[0] were
        // read during the previous iteration's last block; a(c)[n + 1] is read under block n.  Behind block 2 every wave has issued its
        // last read of stage st: own reads drained (lgkmcnt), own DMA pieces of chunk c + 1 landed (counted vmcnt: all but the two
        // youngest chunks), bare barrier -- then the DMA of chunk c + 4 into stage st, and block 3 reads b(c + 1), a(c + 1)[0].
        constexpr int OPS = 4;  // vector-memory instructions per wave and chunk (2 activation + 2 filter pieces)
        static_assert(NP == 2, "two filter DMA pieces per wave");
        int st = 0;
#pragma unroll
        for (int i = 0; i < XD_NST; ++i) {
            dma_x();
            dma_w();
        }
        __asm__ volatile("s_waitcnt vmcnt(%0)" ::"n"((XD_NST - 1) * OPS) : "memory");
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (scale / shift staged above)
        __builtin_amdgcn_s_barrier();
        rdB(B0, 0, 0);
        rdB(B0, 0, 1);
        rdA(A0, 0, 0);
        XW_FENCE
        auto iter_xd = [&](FragB &Bc, FragB &Bn) {
            const int sn = st == XD_NST - 1 ? 0 : st + 1;
            XW_TERM(0, A0, Bc, 0, 0)
            rdA(A1, st, 1);
            XW_MIX(4)
            XW_FENCE
            XW_TERM(0, A0, Bc, 0, 1)
            XW_TERM(0, A0, Bc, 1, 0)
            XW_FENCE
            XW_TERM(1, A1, Bc, 0, 0)
            rdA(A0, st, 2);
            XW_MIX(4)
            XW_FENCE
            XW_TERM(1, A1, Bc, 0, 1)
            XW_TERM(1, A1, Bc, 1, 0)
            XW_FENCE
            XW_TERM(2, A0, Bc, 0, 0)
            rdA(A1, st, 3);
            XW_MIX(4)
            XW_FENCE
            XW_TERM(2, A0, Bc, 0, 1)
            XW_TERM(2, A0, Bc, 1, 0)
            XW_FENCE
            __asm__ volatile("s_waitcnt vmcnt(%0)" ::"n"((XD_NST - 2) * OPS) : "memory");
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef A3D_ABLATIONS  // timing-only variants (developer builds, A3D_HIPCC_FLAGS=-DA3D_ABLATIONS): tune bit 10 no activation DMA, 11 no filter DMA, 12 no barrier
            if (!(d.tune & 4096)) __builtin_amdgcn_s_barrier();
            XW_TERM(3, A1, Bc, 0, 0)
            if (!(d.tune & 1024)) dma_x();
            if (!(d.tune & 2048)) dma_w();
#else
            __builtin_amdgcn_s_barrier();
            XW_TERM(3, A1, Bc, 0, 0)
            dma_x();
            dma_w();
#endif
            XW_MIX(8)
            XW_FENCE
            XW_TERM(3, A1, Bc, 0, 1)
            rdB(Bn, sn, 0);
            rdB(Bn, sn, 1);
            XW_MIX(4)
            XW_FENCE
            XW_TERM(3, A1, Bc, 1, 0)
            rdA(A0, sn, 0);
            XW_MIX(4)
            XW_FENCE
            st = sn;
        };
        for (int it = kbeg; it < nk; it += 2) {
            iter_xd(B0, B1);
            iter_xd(B1, B0);
        }
        // the DMAs past the last chunk land somewhere in the ring (never in scale | shift): drain them before the LDS is given back
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
    // prologue: W(0), W(1) by DMA; X(0) split into X stage 0; X(1), X(2) staged in registers; b(0), a(0)[0] read
    dma_w();
    dma_w();
    load_chunk(xsA);
    {
        Split s0, s1;
        split(xsA[0], 0, s0);
        split(xsA[1], 1, s1);
        put(0, 0, s0);
        put(0, 1, s1);
    }
    load_chunk(xsB);
    load_chunk(xsA);
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NP; ++p) rdB(B0, 0, p);
    rdA(A0, 0, 0);
    XW_FENCE

    if constexpr (PH4) {
        // tap-outer, straight-line over the nine taps (each with its compile-time phase set); the chunks of a tap -- an even number:
        // 32 | Cin + Cin2 -- share the set
#define PH_RUN(TAP)                                                                                              \
    {                                                                                                            \
        constexpr unsigned act = xw_tap_phases(TAP);                                                             \
        constexpr int nf_same = __builtin_ctz(act), nf_last = TAP < 8 ? __builtin_ctz(xw_tap_phases((TAP + 1) % 9)) : 0; \
        for (int c = 0; c < cpt; c += 2) {                                                                       \
            iter_ph4(std::integral_constant<unsigned, act>{}, 0, A0, A1, B0, B1, xsB, nf_same);                  \
            iter_ph4(std::integral_constant<unsigned, act>{}, 1, A0, A1, B1, B0, xsA, c + 2 < cpt ? nf_same : nf_last); \
        }                                                                                                        \
    }
        PH_RUN(0) PH_RUN(1) PH_RUN(2) PH_RUN(3) PH_RUN(4) PH_RUN(5) PH_RUN(6) PH_RUN(7) PH_RUN(8)
#undef PH_RUN
    } else {
    for (int it = kbeg; it < nk; it += 2) {  // (Kpad % 32 == 0 on every packed layer; an odd chunk count would multiply one all-zero chunk: loads / DMA past nk read 0)
        iteration(0, A0, A1, B0, B1, xsB);
        iteration(1, A0, A1, B1, B0, xsA);
    }
    }
    }
#undef XW_REST
#undef XW_MIX
#undef XW_TERM
#undef XW_FENCE

    const bool has_res = d.res != nullptr && d.splitk == 1;
    const int hwo = d.Ho * d.Wo;
    float vmaxs[TM];
    int bimgs[TM];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + (wm * TM + mi) * 32 + (lane & 31);
        const bool mok = m < M;
        const int bimg = mok ? m / hwo : 0;  // image / ROI of this lane's output pixel
        bimgs[mi] = mok ? bimg : -1;
        float &vmax = vmaxs[mi];
        vmax = 0.f;
        if (mok) {
        // (two exact factors, applied one after the other: their product can leave fp32's range for images of extreme magnitude)
        const float unx = F16 ? 1.f / a3d_in_scale(d, bimg) : 1.f, unw = F16 ? 1.f / sw : 1.f;
        size_t res_row;
        int b, oh, ow;
        out_rows(d, m, res_row, b, oh, ow);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            f32x4 rv[4];
            if (has_res) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int n = n0 + (wn * TN + ni) * 32 + rg * 8 + (lane >> 5) * 4;
                    rv[rg] = *reinterpret_cast<const f32x4 *>(d.res + res_row * (size_t)d.Cout + min(n, d.Cout - 4));
                }
            }
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int nl = (wn * TN + ni) * 32 + rg * 8 + (lane >> 5) * 4;
                const int n = n0 + nl;
                if (n >= d.Cout) continue;
                f32x4 v = {acc[ni][mi][rg * 4 + 0], acc[ni][mi][rg * 4 + 1], acc[ni][mi][rg * 4 + 2], acc[ni][mi][rg * 4 + 3]};
                if constexpr (F16) v = (v * unx) * unw;  // exact: powers of two
                if (d.splitk > 1) {
                    *reinterpret_cast<f32x4 *>(d.workspace + ((size_t)blockIdx.y * M + m) * d.Cout + n) = v;
                    continue;
                }
                v = a3d_epilogue_math(d, v, *reinterpret_cast<const f32x4 *>(ss + nl), *reinterpret_cast<const f32x4 *>(ss + BN + nl), has_res, rv[rg]);
                vmax = fmaxf(vmax, a3d_absmax4(v));
                if constexpr (PH4) {  // column 128 g + 32 phase + c -> pixel (2 oh + dy, 2 ow + dx), channel 32 g + c of y [B, 2Ho, 2Wo, Cout / 4]
                    const int co = (n >> 7) * 32 + (n & 31), co_n = d.Cout >> 2;
                    const size_t row = ((size_t)b * (2 * d.Ho) + (2 * oh + (ni >> 1))) * (size_t)(2 * d.Wo) + (2 * ow + (ni & 1));
                    *reinterpret_cast<f32x4 *>(d.y + row * co_n + co) = v;
                    continue;
                }
                store_out(d, v, m, n, b, oh, ow);
            }
        }
        }
    }
    // The maxima are recorded behind the wave's LAST store (the pre-check read of a slot waits for every store issued before it and
    // queues at the L2 behind the other workgroups' atomics: between the row blocks it held the second block's stores back).  Every
    // lane of the wave gets here; split-K: the reducer records.  Lanes l and l+32 hold the same row: one of them reports it -- with one
    // image per row, as in the FC layers, every report is a pre-checked atomic.  Row blocks of one image share one report.
    if (d.y_amax && d.splitk == 1) {
        bool merged = false;
        if constexpr (TM == 2) {
            if (__all(bimgs[0] == bimgs[1] && bimgs[0] >= 0)) {
                const float v = fmaxf(vmaxs[0], vmaxs[1]);
                a3d_note_amax(d.y_amax, bimgs[0], fmaxf(v, __shfl_xor(v, 32, 64)), lane < 32);
                merged = true;
            }
        }
        if (!merged) {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
                a3d_note_amax(d.y_amax, max(bimgs[mi], 0), fmaxf(vmaxs[mi], __shfl_xor(vmaxs[mi], 32, 64)), bimgs[mi] >= 0 && lane < 32);
        }
    }
}
}  // namespace

// Returns A3D_ERR_UNSUPPORTED when the layer should take conv_x3_kernel (conv_bf16x3.hip): no pre-split weights, a shallow reduction,
// a narrow or small problem (256-wide channel tiles mostly padding, or too few 256 x 256 tiles to fill the 256 CUs twice), 32-bit
// offset limits.
// The fused four-phase form (a3d_conv_desc.phase == 5): see conv_x3w_kernel's PH4.
static int launch_ph4(const a3d_conv_desc *d, hipStream_t s) {
    if ((d->precision != 3 && d->precision != 2) || !d->w_x3) return A3D_ERR_ARG;
    if (d->precision == 3 && (!d->in_amax || !(d->w_scale > 0.f))) return A3D_ERR_ARG;
    if (d->precision == 3) {
        const int rp = a3d_conv_launch_ph4p(d, s);  // maps that fit its 8 x 32 tiles: the patch-resident form (conv_ph4p.hip)
        if (rp != A3D_ERR_UNSUPPORTED) return rp;
    }
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->Ho != d->H || d->Wo != d->W) return A3D_ERR_ARG;
    if (d->stem || d->ups || d->m_dev || d->splitk != 1 || d->res || d->gate || d->pixshuf) return A3D_ERR_ARG;
    if (d->Cin2 && (d->Cin2 != d->Cin || !d->x2)) return A3D_ERR_ARG;
    const int CinT = d->Cin + d->Cin2;
    if ((d->Cin & 15) || (CinT & 31) || d->Kpad != 9 * CinT || (d->Cout & 127)) return A3D_ERR_ARG;  // (Cout = 4 x real channels, 32 | real channels)
    if ((size_t)d->B * d->H * d->W * d->Cin * 4 >= ((size_t)1 << 32) || (size_t)d->Cout * d->Kpad * 4 >= ((size_t)1 << 31)) return A3D_ERR_UNSUPPORTED;
    const int M = d->B * d->Ho * d->Wo;
    const int mtiles = (M + XW_BM - 1) / XW_BM, ntiles = (d->Cout + XW_BN - 1) / XW_BN;
    static a3d_attr_once attr_ph4;
    if (attr_ph4.needed()) {
        if (hipFuncSetAttribute((const void *)conv_x3w_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, xw_lds_bytes(2)) != hipSuccess ||
            hipFuncSetAttribute((const void *)conv_x3w_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, xw_lds_bytes(3)) != hipSuccess)
            return A3D_ERR_LAUNCH;
        attr_ph4.mark();
    }
    if (d->precision == 2) {  // round 5: the same loop with three operand planes and six terms (the like-for-like arithmetic)
        if ((size_t)d->Cout * d->Kpad * 6 >= ((size_t)1 << 31)) return A3D_ERR_UNSUPPORTED;
        a3d_note_variant("conv_x3w_kernel ph4");
        hipLaunchKernelGGL((conv_x3w_kernel<false, true>), dim3(mtiles * ntiles, 1), dim3(512), xw_lds_bytes(3), s, *d, M, ntiles, mtiles * ntiles);
        return a3d_check_launch();
    }
    a3d_note_variant("conv_h2w_kernel ph4");
    hipLaunchKernelGGL((conv_x3w_kernel<true, true>), dim3(mtiles * ntiles, 1), dim3(512), xw_lds_bytes(2), s, *d, M, ntiles, mtiles * ntiles);
    return a3d_check_launch();
}

// Pre-split activations (a3d_conv_desc.x_h2): both operands by LDS-DMA.  Any plain direct layer of the fp16x2 arithmetic qualifies
// (1x1 / linear, 3x3, strided; one or two equal-width sources); there is no other kernel that reads this format, so everything else
// is an argument error, not "unsupported".
static int launch_xd(const a3d_conv_desc *d, hipStream_t s) {
    if (d->precision != 3 || !d->w_x3 || !d->in_amax || !(d->w_scale > 0.f)) return A3D_ERR_ARG;
    if (d->stem || d->ups || d->m_dev || d->splitk != 1 || d->pixshuf || d->gate || d->phase) return A3D_ERR_ARG;
    if (d->Cin2 && (d->Cin2 != d->Cin || !d->x2_h2)) return A3D_ERR_ARG;
    if ((d->Cin & 15) || d->Kpad != d->KH * d->KW * (d->Cin + d->Cin2) || d->KH * d->KW > 32) return A3D_ERR_ARG;
    if ((size_t)d->B * d->H * d->W * d->Cin * 4 >= ((size_t)1 << 32) || (size_t)d->Cout * d->Kpad * 4 >= ((size_t)1 << 31)) return A3D_ERR_UNSUPPORTED;
    const int M = d->B * d->Ho * d->Wo;
    const int mtiles = (M + XW_BM - 1) / XW_BM, ntiles = (d->Cout + XW_BN - 1) / XW_BN;
    static a3d_attr_once attr_xd;
    if (attr_xd.needed()) {
        if (hipFuncSetAttribute((const void *)conv_x3w_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, xd_lds_bytes()) != hipSuccess) return A3D_ERR_LAUNCH;
        attr_xd.mark();
    }
    a3d_note_variant("conv_h2w_kernel xd");
    hipLaunchKernelGGL((conv_x3w_kernel<true, false, true>), dim3(mtiles * ntiles, 1), dim3(512), xd_lds_bytes(), s, *d, M, ntiles, mtiles * ntiles);
    return a3d_check_launch();
}

int a3d_conv_launch_bf16x3_wide(const a3d_conv_desc *d, hipStream_t s) {
    if (d->x_h2 && d->phase != 5) return launch_xd(d, s);
    if (d->phase == 5) return launch_ph4(d, s);
    if (!d->w_x3 || d->tune == 8) return A3D_ERR_UNSUPPORTED;
    if (d->stem || d->ups || d->m_dev || d->splitk < 1) return A3D_ERR_UNSUPPORTED;
    if (d->splitk > 1 && (!d->workspace || d->phase || d->gate)) return A3D_ERR_UNSUPPORTED;
    if (d->pixshuf && (d->res || d->phase || d->gate)) return A3D_ERR_UNSUPPORTED;
    if (d->phase && (d->KH != 2 || d->KW != 2 || d->stride != 1 || d->res)) return A3D_ERR_UNSUPPORTED;
    if (d->Cin2 && (d->Cin2 != d->Cin || !d->x2)) return A3D_ERR_UNSUPPORTED;
    if ((d->Cin & 15) || d->Kpad != d->KH * d->KW * (d->Cin + d->Cin2) || d->KH * d->KW > 32) return A3D_ERR_UNSUPPORTED;
    if ((size_t)d->B * d->H * d->W * d->Cin * 4 >= ((size_t)1 << 32) || (size_t)d->Cout * d->Kpad * 6 >= ((size_t)1 << 31)) return A3D_ERR_UNSUPPORTED;
    const int M = d->B * d->Ho * d->Wo;
    const int mtiles = (M + XW_BM - 1) / XW_BM, ntiles = (d->Cout + XW_BN - 1) / XW_BN;
    if (d->tune != 9) {  // (tune 9 forces this kernel: A/B runs and the bit-equality test)
        // Measured per layer shape (tools/x3_wide_check.py, 64 frames): the box head's fc1 (K = 12544) 8.32 -> 6.99 ms (235 fp32-
        // equivalent TFLOP/s); K = 2048 equal; K <= 1024 equal to 15 % slower -- with ONE workgroup per CU nothing overlaps the
        // prologue and the 256 x 256 epilogue (residual reads, stores), which a 16..64-iteration loop does not amortise, where
        // the narrow kernel's three workgroups per CU cover each other's.  Hence deep reductions only.
        // fp16x2, measured again per layer (tools/narrow_wide_ab.py, 64 frames, narrow | wide ms): shallow reductions go wide too when
        // the layer has no residual to fetch and at least ~1000 of the 256 x 256 blocks -- fc2 64000 x 1024 -> 1024 0.586 | 0.487, the
        // p2 lateral 256 -> 256 0.870 | 0.781, strided 256 -> 512 0.487 | 0.474; with a residual (256 -> 1024 0.274 | 0.310) or fewer
        // blocks (1024 -> 256 at 30x40 0.190 | 0.209) the narrow kernel's three workgroups per CU stay ahead.  Same bits either way.
        const bool shallow_ok = d->precision == 3 && !d->res && d->splitk == 1 && d->Kpad >= 256 && d->Cout >= 256 && (long)mtiles * ntiles >= 1000;
        if (d->Kpad < 4096 && !shallow_ok) return A3D_ERR_UNSUPPORTED;
        if (d->Cout < 192 || ntiles * XW_BN > d->Cout + d->Cout / 4) return A3D_ERR_UNSUPPORTED;
        if (d->splitk == 1 && (long)mtiles * ntiles < 2 * 256) return A3D_ERR_UNSUPPORTED;  // (split-K launches stream the weights: any M)
    }
    static a3d_attr_once attr_set;
    if (attr_set.needed()) {  // > 64 KiB of dynamic LDS needs the opt-in attribute (once per device)
        if (hipFuncSetAttribute((const void *)conv_x3w_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, xw_lds_bytes(3)) != hipSuccess ||
            hipFuncSetAttribute((const void *)conv_x3w_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, xw_lds_bytes(2)) != hipSuccess)
            return A3D_ERR_LAUNCH;
        attr_set.mark();
    }
    if (d->precision == 3) {
        if (!d->in_amax || !(d->w_scale > 0.f)) return A3D_ERR_ARG;
        a3d_note_variant(d->splitk > 1 ? "conv_h2w_kernel sk%d" : "conv_h2w_kernel", d->splitk);
        hipLaunchKernelGGL(conv_x3w_kernel<true>, dim3(mtiles * ntiles, d->splitk), dim3(512), xw_lds_bytes(2), s, *d, M, ntiles, mtiles * ntiles);
        if (d->splitk > 1) a3d_launch_splitk_reduce(d, M, s);
        return a3d_check_launch();
    }
    a3d_note_variant(d->splitk > 1 ? "conv_x3w_kernel sk%d" : "conv_x3w_kernel", d->splitk);
    hipLaunchKernelGGL(conv_x3w_kernel<false>, dim3(mtiles * ntiles, d->splitk), dim3(512), xw_lds_bytes(3), s, *d, M, ntiles, mtiles * ntiles);
    if (d->splitk > 1) a3d_launch_splitk_reduce(d, M, s);
    return a3d_check_launch();
}
