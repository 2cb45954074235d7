// bf16 (autocast) convolution / linear kernel of the TRAINING step with both operands by LDS-DMA -- round 5.
// (a3d_conv_desc.precision == 1 with w_bf16: BASELINE configs[4], tools/train_net.py + config/step1_bbox.yaml under bf16 autocast.)
//
// Why.  conv_bf16_kernel (conv_bf16.hip) is the kernel furthest below its roof in the repository (0.07-0.17 of the bf16 pipe): a 128 x 128 or
// 128 x 64 tile whose operands pass through registers as fp32 -- the weights are the fp32 master copy, converted by every workgroup in every
// chunk -- and whose 32-deep chunk is 8 MFMAs per wave behind 16-32 KiB of loads: with ONE product per multiply-add (the fp16x2 inference
// kernels issue three) it moves 3 x their bytes per MFMA.  Here:
//   * the filter arrives as bf16 (a3d_conv_desc.w_bf16: the trainer rounds its flat parameter buffer and the data-gradient filters once per
//     step, two launches), half the bytes and no conversion in the loop; activations stored as bf16 (the trainer's res3-res5 / FPN / RPN /
//     head tensors) go global -> LDS as they lie, fp32-stored ones (the pyramid p2-p6, the frozen res2 output) are converted on the fragment;
//   * tiles of 256 pixels x 256 or 128 channels (wave tile 64 x 128 / 64 x 64): 256 / 384 operand bytes per MFMA instead of 1024;
//   * both operands by LDS-DMA through a ring of 3-5 stages, no staging registers; the 3x3 taps are gathered by the DMA's per-lane global
//     addresses (a pixel row's 32-channel chunk of a tap is one contiguous 64- or 128-byte run; taps in the zero padding and rows past M
//     read out of the buffer's range: zeros);
//   * the 512-thread workgroup's halves in antiphase (conv_wino.hip, "PING-PONG": memory phase = fragment reads + DMA issue, compute phase =
//     the chunk's MFMAs back to back).
// Arithmetic: that of conv_bf16_kernel -- operands rounded to bf16 (nearest even), fp32 accumulation, chunks of 32 in k = (kh, kw, c) order,
// two 16-deep MFMA steps per chunk: the two kernels agree bit for bit (tests/test_gpu_training.py), so the launcher may choose by size.
#include "conv_common.h"

namespace {
typedef __bf16 bw_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bw_bf16x8 __attribute__((ext_vector_type(8)));
typedef float bw_f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int bw_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t bw_rsrc(const void *p, unsigned bytes) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ void bw_dma16(__amdgpu_buffer_rsrc_t r, void *lds_dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds_dst, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ f32x4 bw_widen4(const bw_u32x2 v) {
    f32x4 o;
    o[0] = __builtin_bit_cast(float, v[0] << 16);
    o[1] = __builtin_bit_cast(float, v[0] & 0xFFFF0000u);
    o[2] = __builtin_bit_cast(float, v[1] << 16);
    o[3] = __builtin_bit_cast(float, v[1] & 0xFFFF0000u);
    return o;
}
__device__ __forceinline__ f32x4 bw_read4(const float *base, size_t idx, bool is_bf16) {
    if (is_bf16) return bw_widen4(*reinterpret_cast<const bw_u32x2 *>(reinterpret_cast<const __bf16 *>(base) + idx));
    return *reinterpret_cast<const f32x4 *>(base + idx);
}
__device__ __forceinline__ void bw_write4(float *base, size_t idx, const f32x4 v, bool is_bf16) {
    if (is_bf16) *reinterpret_cast<bw_bf16x4 *>(reinterpret_cast<__bf16 *>(base) + idx) = __builtin_convertvector(v, bw_bf16x4);
    else *reinterpret_cast<f32x4 *>(base + idx) = v;
}

constexpr int BW_BM = 256, BW_BK = 32;
constexpr int bw_xrow(bool XB) { return XB ? 64 : 128; }  // bytes of a pixel row's 32-channel chunk
constexpr int bw_stage(int TN, bool XB) { return BW_BM * bw_xrow(XB) + 64 * TN * 64; }
constexpr int bw_stages(int TN, bool XB) { return XB ? (TN == 2 ? 5 : 4) : 3; }
constexpr int bw_lds_bytes(int TN, bool XB) { return bw_stages(TN, XB) * bw_stage(TN, XB) + 2 * 64 * TN * 4; }

// TN = 32-channel blocks per wave: the workgroup's tile is 256 pixels x (64 TN) channels, 8 waves as 4 (pixels) x 2 (channels).
// XB: the activations are stored as bf16 (a3d_conv_desc.io_bf16 bit 0).
template <int TN, bool XB>
__global__ __launch_bounds__(512, 2) void conv_bf16w_kernel(const a3d_conv_desc d, const int M, const int ntiles, const int nblk) {
    constexpr int TM = 2, BM = BW_BM, BN = 64 * TN, NST = bw_stages(TN, XB), STAGE = bw_stage(TN, XB);
    constexpr int XROW = bw_xrow(XB), XBYTES = BM * XROW, ES = XB ? 2 : 4;
    constexpr int RPP = XB ? 16 : 8;           // pixel rows per activation DMA piece (1 KiB)
    constexpr int XPW = BM / RPP / 8;          // activation pieces per wave and chunk (2 | 4)
    constexpr int WPW = BN / 16 / 8;           // filter pieces per wave and chunk: 16 rows x 64 B each (1 | 2)
    constexpr int OPS = XPW + WPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char bw_lds[];
    float *ss = reinterpret_cast<float *>(bw_lds + NST * STAGE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int logical = a3d_xcd_remap(blockIdx.x, nblk);
    const int mt = logical / ntiles, nt = logical - mt * ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int nk = d.Kpad / BW_BK;
    const int cpt = d.Cin / BW_BK;  // chunks per filter tap
    const int hwo = d.Ho * d.Wo;
    const bool yb = d.io_bf16 & 2, rb = d.io_bf16 & 4, gb = d.io_bf16 & 8;

    // ---- activation DMA.  Piece j of this wave = tile rows (wave XPW + j) RPP .. + RPP - 1; lane i -> row i / (64 / RPP), LDS slot
    // i % (64 / RPP), which keeps the row's global 16-byte slot  slot ^ swizzle(row): 64-byte rows (row >> 2) & 3, 128-byte rows
    // (row >> 1) & 7 -- the fragment reads below are conflict-free with either.
    const __amdgpu_buffer_rsrc_t rx = bw_rsrc(d.x, (unsigned)((size_t)d.B * d.H * d.W * d.Cin * ES));
    int xoff[XPW];
    unsigned xmask[XPW];
#pragma unroll
    for (int j = 0; j < XPW; ++j) {
        constexpr int LPR = 64 / RPP;  // lanes per row
        const int row = (wave * XPW + j) * RPP + lane / LPR;
        const int slot = lane % LPR;
        const int gslot = XB ? (slot ^ ((row >> 2) & 3)) : (slot ^ ((row >> 1) & 7));
        const int m = m0 + row;
        const bool rok = m < M;
        const int mm = rok ? m : 0;
        const int b = mm / hwo, r = mm - b * hwo;
        const int oh = r / d.Wo, ow = r - oh * d.Wo;
        const int ih0 = oh * d.stride - d.pad, iw0 = ow * d.stride - d.pad;
        unsigned mask = 0;
        for (int kh = 0; kh < d.KH; ++kh)
            for (int kw = 0; kw < d.KW; ++kw)
                mask |= (rok && (unsigned)(ih0 + kh) < (unsigned)d.H && (unsigned)(iw0 + kw) < (unsigned)d.W) ? (1u << (kh * d.KW + kw)) : 0u;
        xoff[j] = ((b * d.H + ih0) * d.W + iw0) * d.Cin * ES + gslot * 16;
        xmask[j] = mask;
    }
    // ---- filter DMA: w_bf16 [Cout][Kpad] bf16; piece j of this wave = tile rows (wave WPW + j) 16 .. + 15, 64 bytes of chunk c each
    const __amdgpu_buffer_rsrc_t rw = bw_rsrc(d.w_bf16, (unsigned)((size_t)d.Cout * d.Kpad * 2));
    int woff[WPW];
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
        const int row = (wave * WPW + j) * 16 + (lane >> 2);
        const int n = n0 + row;
        woff[j] = n < d.Cout ? n * d.Kpad * 2 + (((lane & 3) ^ ((row >> 2) & 3)) << 4) : -1;
    }
    int dma_c = 0, dtap = 0, dc0 = 0;  // next chunk to fetch: index, filter tap, first channel
    auto dma = [&](const int st) {
        unsigned char *X = bw_lds + st * STAGE;
        unsigned char *Wt = X + XBYTES;
        const bool live = dma_c < nk;  // (past the last chunk: zeros that nobody reads; keeps the counted waits uniform)
        const int kh = dtap / d.KW, kw = dtap - kh * d.KW;
        const int tapoff = __builtin_amdgcn_readfirstlane(((kh * d.W + kw) * d.Cin + dc0) * ES);
#pragma unroll
        for (int j = 0; j < XPW; ++j)
            bw_dma16(rx, X + (wave * XPW + j) * 1024, (live && ((xmask[j] >> dtap) & 1u)) ? xoff[j] + tapoff : -1, 0);
        const int wsoff = __builtin_amdgcn_readfirstlane(dma_c * 64);
#pragma unroll
        for (int j = 0; j < WPW; ++j) bw_dma16(rw, Wt + (wave * WPW + j) * 1024, (live && woff[j] >= 0) ? woff[j] + wsoff : -1, 0);
        ++dma_c;
        dc0 += BW_BK;
        if (dc0 >= d.Cin) {
            dc0 = 0;
            ++dtap;
        }
    };
    (void)cpt;

    // ---- fragments: row = lane % 32 of a 32-row block, k = 8 (lane / 32) .. + 7 of a 16-deep step
    const int frow = lane & 31;
    const int sw4 = (frow >> 2) & 3, sw8 = (frow >> 1) & 7;  // (block row offsets are multiples of 32: a row's swizzle is that of lane % 32)
    bw_bf16x8 fa[2][TN], fb[2][TM];
    auto rd_all = [&](const int st) {
        const unsigned char *X = bw_lds + st * STAGE;
        const unsigned char *Wt = X + XBYTES;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int g = 2 * s2 + (lane >> 5);  // the fragment's 16-byte slot of a 64-byte row
#pragma unroll
            for (int n = 0; n < TN; ++n)
                fa[s2][n] = *reinterpret_cast<const bw_bf16x8 *>(Wt + ((wn * TN + n) * 32 + frow) * 64 + ((g ^ sw4) << 4));
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const int row = (wm * TM + mi) * 32 + frow;
                if constexpr (XB) {
                    fb[s2][mi] = *reinterpret_cast<const bw_bf16x8 *>(X + row * 64 + ((g ^ sw4) << 4));
                } else {  // fp32-stored: the 8 k values are two slots of the 128-byte row; rounded to bf16 here (nearest even)
                    const f32x4 lo = *reinterpret_cast<const f32x4 *>(X + row * 128 + (((2 * g) ^ sw8) << 4));
                    const f32x4 hi = *reinterpret_cast<const f32x4 *>(X + row * 128 + (((2 * g + 1) ^ sw8) << 4));
                    const bw_f32x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    fb[s2][mi] = __builtin_convertvector(v, bw_bf16x8);
                }
            }
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][mi][r] = 0.f;
    a3d_stage_scale_shift(ss, d, n0, BN, tid);

#define BW_FENCE __builtin_amdgcn_sched_barrier(0);
    auto compute = [&]() {
        BW_FENCE
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) acc[n][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s2][n], fb[s2][mi], acc[n][mi], 0, 0, 0);
        BW_FENCE
    };

    // ---- prologue: the ring filled, chunk 0 landed
#pragma unroll
    for (int i = 0; i < NST; ++i) dma(i);
    __asm__ volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * OPS) : "memory");
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- ping-pong loop: waves 0-3 run memory, compute, BARRIER and waves 4-7 memory, BARRIER, compute.  Between BAR_c and BAR_c+1 every wave
    // reads stage(c) only; the DMA of chunk c - 1 + NST goes into stage(c - 1); chunk c + 1 has landed before BAR_c+1 (counted wait + barrier).
    const bool grpB = wave >= 4;
    int st = 0, stp = NST - 1;
    for (int it = 0; it < nk; ++it) {
        BW_FENCE
        rd_all(st);
        if (it > 0) dma(stp);
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (grpB) {
            __asm__ volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * OPS) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        compute();
        if (!grpB) {
            __asm__ volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * OPS) : "memory");
            __builtin_amdgcn_s_barrier();
        }
        stp = st;
        st = st == NST - 1 ? 0 : st + 1;
    }
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the DMAs past the last chunk: landed before the LDS is given back)
#undef BW_FENCE

    // ---- epilogue (conv_bf16_kernel's: per-lane output quads; fp32 or bf16 stores, residual, ReLU-backward gate)
    const bool has_res = d.res != nullptr;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + (wm * TM + mi) * 32 + (lane & 31);
        if (m >= M) continue;
        size_t res_row;
        int b, oh, ow;
        out_rows(d, m, res_row, b, oh, ow);
        // (residual and gate quads of two 32-channel groups at a time in front of their stores: conv_bf16.hip's epilogue)
        const bool has_gate = d.io_bf16 && d.gate != nullptr;
#pragma unroll
        for (int np = 0; np < TN; np += 2) {
            f32x4 rv[2][4], gv[2][4];
#pragma unroll
            for (int nj = 0; nj < 2; ++nj)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int n = n0 + (wn * TN + np + nj) * 32 + rg * 8 + (lane >> 5) * 4;
                    if (has_res) rv[nj][rg] = bw_read4(d.res, res_row * (size_t)d.Cout + min(n, d.Cout - 4), rb);
                    if (has_gate) gv[nj][rg] = bw_read4(d.gate, (size_t)m * d.Cout + min(n, d.Cout - 4), gb);
                }
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
                const int ni = np + nj;
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int nl = (wn * TN + ni) * 32 + rg * 8 + (lane >> 5) * 4;
                    const int n = n0 + nl;
                    if (n >= d.Cout) continue;
                    f32x4 v = {acc[ni][mi][rg * 4 + 0], acc[ni][mi][rg * 4 + 1], acc[ni][mi][rg * 4 + 2], acc[ni][mi][rg * 4 + 3]};
                    v = a3d_epilogue_math(d, v, *reinterpret_cast<const f32x4 *>(ss + nl), *reinterpret_cast<const f32x4 *>(ss + BN + nl), has_res, rv[nj][rg]);
                    if (d.io_bf16) {
                        const size_t o = (size_t)m * d.Cout + n;
                        if (has_gate) {
                            const f32x4 g = gv[nj][rg];
                            for (int i = 0; i < 4; ++i) v[i] = g[i] > 0.f ? v[i] : 0.f;
                        }
                        bw_write4(d.y, o, v, yb);
                    } else {
                        store_out(d, v, m, n, b, oh, ow);
                    }
                }
            }
        }
    }
}

template <int TN, bool XB>
int bw_launch(const a3d_conv_desc *d, hipStream_t s, int M) {
    static a3d_attr_once attr;
    if (attr.needed()) {
        if (hipFuncSetAttribute((const void *)conv_bf16w_kernel<TN, XB>, hipFuncAttributeMaxDynamicSharedMemorySize, bw_lds_bytes(TN, XB)) != hipSuccess) return A3D_ERR_LAUNCH;
        attr.mark();
    }
    const int mtiles = (M + BW_BM - 1) / BW_BM, ntiles = (d->Cout + 64 * TN - 1) / (64 * TN);
    a3d_note_variant("conv_bf16w_kernel<%d>", TN);
    hipLaunchKernelGGL((conv_bf16w_kernel<TN, XB>), dim3(mtiles * ntiles), dim3(512), bw_lds_bytes(TN, XB), s, *d, M, ntiles, mtiles * ntiles);
    return a3d_check_launch();
}
}  // namespace

// A3D_ERR_UNSUPPORTED: not a launch of this form (no bf16 filter, a kind of layer the kernel does not take, a problem too small for
// one-workgroup-per-CU tiles: conv_bf16_kernel with its 128-row tiles and split-K runs it -- the same bits).  tune 30 / 31: this kernel
// with 128 / 256 channel tiles whatever the size; tune 32: never.
int a3d_conv_launch_bf16w(const a3d_conv_desc *d, hipStream_t s) {
    if (d->precision != 1 || !d->w_bf16 || d->tune == 32 || !(d->tune == 0 || d->tune == 30 || d->tune == 31)) return A3D_ERR_UNSUPPORTED;
    if (d->stem || d->ups || d->phase || d->pixshuf || d->x2 || d->Cin2 || d->splitk != 1 || d->m_dev || d->dot_w) return A3D_ERR_UNSUPPORTED;
    if ((d->Cin & 31) || d->Kpad != d->KH * d->KW * d->Cin || d->KH * d->KW > 32 || (d->Cout & 3) || (d->io_bf16 & ~15)) return A3D_ERR_UNSUPPORTED;
    if ((size_t)d->B * d->H * d->W * d->Cin * 4 >= ((size_t)1 << 31) || (size_t)d->Cout * d->Kpad * 2 >= ((size_t)1 << 31)) return A3D_ERR_UNSUPPORTED;
    const int M = d->B * d->Ho * d->Wo;
    const bool xb = d->io_bf16 & 1;
    const long b2 = (long)((M + 255) / 256) * ((d->Cout + 127) / 128), b4 = (long)((M + 255) / 256) * ((d->Cout + 255) / 256);
    int tn;
    if (d->tune == 30) tn = 2;
    else if (d->tune == 31) tn = 4;
    else {
        // Measured per layer at 16 images per GPU (tools/bf16w_ab.py, profiles/r05_bf16w_ab.txt; conv_bf16_kernel | this kernel, ms): FPN output
        // 3x3 on p2 0.577 | 0.398, RPN conv on p2 (fp32-stored input) 0.651 | 0.488, fc1 0.364 | 0.237, res4 conv2 0.058 | 0.047, res4 conv1
        // 0.033 | 0.030; it loses where the reduction is a handful of chunks (one workgroup per CU: ring fill and epilogue are not
        // covered -- 1x1 128 -> 512 + residual 0.073 | 0.077, 256 -> 1024 0.047 | 0.054, lateral 512 -> 256 0.050 | 0.053) and under half a
        // round of the chip, where conv_bf16_kernel splits K (res5 3x3 0.067 | 0.073).  Hence: K >= 1024 and at least 128 blocks.
        if (d->Cout < 128 || d->Kpad < 1024 || b2 < 128) return A3D_ERR_UNSUPPORTED;
        // 256-channel tiles where they waste no channels and leave at least ~2 rounds of the chip; 128-channel tiles otherwise
        tn = ((d->Cout & 255) == 0 && b4 >= 448) ? 4 : 2;
    }
    if (tn == 4) return xb ? bw_launch<4, true>(d, s, M) : bw_launch<4, false>(d, s, M);
    return xb ? bw_launch<2, true>(d, s, M) : bw_launch<2, false>(d, s, M);
}
