// fp16x2 (a3d_conv_desc.precision == 3) pointwise convolution with a DEEP reduction (Cin >= 512): the 1x1 reductions of the ResNet
// bottlenecks (res3 512 -> 128, res4 1024 -> 256, res5 2048 -> 512, stride-2 entries and shortcuts included), res5's 512 -> 2048 +
// residual expansions and the FPN laterals over res3 / res4 / res5 (planercnn.py:29,150 -> detectron2 ResNet / FPN).
//
// Why another kernel.  conv_x3_kernel<2> (conv_bf16x3.hip) runs these layers at <= 0.38 of BOTH of their roofs: its activations pass
// through registers (loaded three chunks ahead, split by the loader, stored to LDS), the 168 registers that three workgroups per CU
// allow leave no room for a fourth staging set, and PMC shows its waves parked at waits 60 % of their cycles -- the layer is bound by
// the bytes a CU keeps in flight, not by a pipe.  Round 5 restructures WHO ISSUES WHAT, as in the Winograd GEMM:
//   * the RAW fp32 activation tile travels global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`), like the pre-split filter:
//     no staging registers at all, a ring of 3 - 4 stages of a 32-deep chunk each (48 / 32 KiB per stage), every operand byte in
//     flight for 2 - 3 chunk times;
//   * the split x * s = h + l happens ON THE FRAGMENT: a lane reads its 8 k values as fp32 (two ds_read_b128; 128-byte rows whose
//     16-byte slots are XOR-swizzled with (row >> 1) & 7 on the GLOBAL side of the DMA: conflict-free) and splits them in registers.
//     For a 1x1 layer every activation is multiplied by one workgroup's waves only, so splitting after the read costs what
//     splitting before the LDS store cost (the two waves that share a pixel row each split it: 2 x, all of it in a memory phase);
//   * 512 threads, one workgroup per CU, the two halves of the workgroup in ANTIPHASE (waves 4-7 run half a chunk behind waves 0-3;
//     the wave pairs of a SIMD alternate a memory phase -- fragment reads, DMA issue, the split's vector work -- with a compute phase
//     of 12 TM MFMAs back to back).  conv_wino.hip's ping-pong loop describes the barrier protocol; it is the same here.
// Per output element the operations are conv_x3_kernel's (split2h, 16-deep steps in k order, h.h + h.l + l.h per step into one fp32
// accumulator, the shared epilogue): the kernels agree BIT FOR BIT (tests/test_gpu_parity.py), so the launcher may choose by size.
#include "conv_common.h"

namespace {
typedef _Float16 dk_h16x8 __attribute__((ext_vector_type(8)));
typedef float dk_f32x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t dk_rsrc(const void *p, unsigned bytes) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ void dk_dma16(__amdgpu_buffer_rsrc_t r, void *lds_dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds_dst, 16, voff, soff, 0, 0);
}

constexpr int DK_BN = 128, DK_BK = 32;
constexpr int dk_stage_bytes(int TM) { return 128 * TM * DK_BK * 4 + 2 * 2 * DK_BN * 32; }  // x fp32 rows | w [k16][plane][row][16] fp16
constexpr int dk_stages(int TM) { return TM == 1 ? 4 : 3; }
constexpr int dk_lds_bytes(int TM) { return dk_stages(TM) * dk_stage_bytes(TM) + 2 * DK_BN * 4; }

// TM = 32-pixel tiles per wave: the workgroup's tile is (128 TM) pixels x 128 channels, 8 waves as 4 (pixels) x 2 (channels).
template <int TM>
__global__ __launch_bounds__(512, 2) void conv_dk_kernel(const a3d_conv_desc d, const int M, const int ntiles, const int nblk) {
    constexpr int TN = 2, BM = 128 * TM, BN = DK_BN, NST = dk_stages(TM), STAGE = dk_stage_bytes(TM);
    constexpr int XB = BM * DK_BK * 4;         // bytes of a stage's activation area
    constexpr int WPL = BN * 32;               // bytes of one (k16, plane) filter tile: 128 rows x 16 fp16
    constexpr int XPW = 2 * TM, WPW = 2;       // DMA pieces per wave and chunk: activations (8 rows x 128 B each) | filter (32 rows x 32 B)
    constexpr int OPS = XPW + WPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char dk_lds[];
    float *ss = reinterpret_cast<float *>(dk_lds + NST * STAGE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int logical = a3d_xcd_remap(blockIdx.x, nblk);
    const int mt = logical / ntiles, nt = logical - mt * ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int nk = d.Kpad / DK_BK;  // 32-deep chunks (Cin % 32 == 0)
    const int hwo = d.Ho * d.Wo;

    // ---- activation DMA: piece j of this wave covers tile rows wave * 8 XPW + 8 j .. + 7; lane i -> row i / 8, LDS slot i % 8, which
    // keeps global slot (i % 8) ^ ((row >> 1) & 7) of the row's 128-byte chunk
    const __amdgpu_buffer_rsrc_t rx = dk_rsrc(d.x, (unsigned)((size_t)d.B * d.H * d.W * d.Cin * 4));
    int xoff[XPW];
#pragma unroll
    for (int j = 0; j < XPW; ++j) {
        const int row = wave * 8 * XPW + 8 * j + (lane >> 3);
        const int m = m0 + row;
        int off = -1;  // (rows past M: out of the buffer's range -> zeros)
        if (m < M) {
            const int b = m / hwo, r = m - b * hwo;
            const int oh = r / d.Wo, ow = r - oh * d.Wo;
            off = ((b * d.H + oh * d.stride) * d.W + ow * d.stride) * d.Cin * 4 + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
        }
        xoff[j] = off;
    }
    // ---- filter DMA: w_x3 [Kpad / 16][2][Cout][16] fp16; a (k16, plane) tile of this workgroup's 128 rows = 4 pieces of 32 rows; lane i
    // -> row i / 2, half i % 2, which keeps global half (i % 2) ^ ((row >> 3) & 1)  (conv_x3_kernel's image).  16 pieces per chunk, 2 per wave.
    const __amdgpu_buffer_rsrc_t rw = dk_rsrc(d.w_x3, (unsigned)((size_t)(d.Kpad / 16) * d.Cout * 64));
    const int wvoff = (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4);
    int dma_c = 0;  // the next chunk to fetch
    auto dma = [&](const int st) {
        unsigned char *X = dk_lds + st * STAGE;
        unsigned char *Wt = X + XB;
        const int c = min(dma_c, nk - 1);  // (past the last chunk: a repeat that nobody reads; keeps the counted waits uniform)
        const int xs = __builtin_amdgcn_readfirstlane(c * (DK_BK * 4));
#pragma unroll
        for (int j = 0; j < XPW; ++j) dk_dma16(rx, X + (wave * XPW + j) * 1024, xoff[j], xs);
#pragma unroll
        for (int j = 0; j < WPW; ++j) {
            const int q = wave * WPW + j;          // 0 .. 15 = (k16, plane, row group of 32)
            const int kp = q >> 2, g = q & 3;      // kp = 2 * k16 + plane
            const int soff = __builtin_amdgcn_readfirstlane((2 * c + (kp >> 1)) * d.Cout * 64 + (kp & 1) * d.Cout * 32 + (n0 + g * 32) * 32);
            dk_dma16(rw, Wt + kp * WPL + g * 1024, wvoff, soff);
        }
        ++dma_c;
    };

    // ---- fragments.  A (filter): row = lane % 32 of the wave's 32-channel block, k = 8 (lane / 32) .. + 7 of a 16-deep step: ONE ds_read_b128.
    // B (activations): the same 8 k values as fp32 = 32 bytes = two slots of the row; split in registers.
    const int frow = lane & 31;
    const int aoff = frow * 32 + ((((lane >> 5) ^ (frow >> 3)) & 1) << 4);
    float sxr[TM];     // fp16x2 scale of each pixel row's image
    int brow[TM];      // byte offset of the lane's activation row inside a stage
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int row = (wm * TM + mi) * 32 + frow;
        const int m = m0 + row;
        sxr[mi] = m < M ? a3d_in_scale(d, m / hwo) : 1.f;
        brow[mi] = row * 128;
    }
    const int bsw = (frow >> 1) & 7;  // (tile rows are multiples of 32 apart: the swizzle of a row is that of lane % 32)
    dk_h16x8 fa[2][2][TN];            // [step][plane][n]
    dk_h16x8 fb[2][2][TM];            // [step][plane][m]
    dk_f32x8 raw[2][TM];
    auto rd_all = [&](const int st) {
        const unsigned char *X = dk_lds + st * STAGE;
        const unsigned char *Wt = X + XB;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    fa[s2][p][n] = *reinterpret_cast<const dk_h16x8 *>(Wt + (2 * s2 + p) * WPL + (wn * 64 + n * 32) * 32 + aoff);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const int g0 = s2 * 4 + (lane >> 5) * 2;  // global slot of k = 16 s2 + 8 (lane / 32)
                const f32x4 lo = *reinterpret_cast<const f32x4 *>(X + brow[mi] + (((g0) ^ bsw) << 4));
                const f32x4 hi = *reinterpret_cast<const f32x4 *>(X + brow[mi] + (((g0 + 1) ^ bsw) << 4));
                raw[s2][mi] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
    };
    auto split_all = [&]() {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const dk_f32x8 xs = raw[s2][mi] * sxr[mi];
                const dk_h16x8 h = __builtin_convertvector(xs, dk_h16x8);
                const dk_f32x8 r = xs - __builtin_convertvector(h, dk_f32x8);
                fb[s2][0][mi] = h;
                fb[s2][1][mi] = __builtin_convertvector(r, dk_h16x8);
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

#define DK_FENCE __builtin_amdgcn_sched_barrier(0);
#define DK_TERM(S2, PA, PB)                                                                                                          \
    _Pragma("unroll") for (int n = 0; n < TN; ++n) _Pragma("unroll") for (int mi = 0; mi < TM; ++mi)                                 \
        acc[n][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[S2][PA][n], fb[S2][PB][mi], acc[n][mi], 0, 0, 0);
    auto compute = [&]() {  // (conv_x3_kernel's order per accumulator: h.h, h.l, l.h of step 0, then of step 1; first index = filter plane)
        DK_FENCE
        DK_TERM(0, 0, 0)
        DK_TERM(0, 0, 1)
        DK_TERM(0, 1, 0)
        DK_TERM(1, 0, 0)
        DK_TERM(1, 0, 1)
        DK_TERM(1, 1, 0)
        DK_FENCE
    };

    // ---- prologue: the ring's stages filled, chunk 0 landed
#pragma unroll
    for (int i = 0; i < NST; ++i) dma(i);
    __asm__ volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * OPS) : "memory");
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // ---- ping-pong loop (conv_wino.hip, "PING-PONG"): ONE instruction stream; waves 0-3 run memory, compute, BARRIER and waves 4-7 memory,
    // BARRIER, compute, so between two barriers the former run memory(c), compute(c) and the latter compute(c - 1), memory(c).  Between
    // BAR_c and BAR_c+1 every wave reads stage(c) only; the DMA of chunk c - 1 + NST goes into stage(c - 1).
    const bool grpB = wave >= 4;
    int st = 0, stp = NST - 1;
    for (int it = 0; it < nk; ++it) {
        DK_FENCE
        rd_all(st);
        if (it > 0) dma(stp);
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        split_all();
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
    // the DMAs past the last chunk land anywhere in the ring; the epilogue's tiles go through stage 0's activation area
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#undef DK_TERM
#undef DK_FENCE

    // ---- row-major epilogue (conv_x3_kernel's: each 32 x 32 accumulator tile through 4 KiB of LDS, 8 lanes per output row; the same values
    // and operations per element): 8 waves x 4 KiB = the first 32 KiB of the ring
    static_assert(XB >= 8 * 4096 || STAGE >= 8 * 4096, "eight 4 KiB tiles");
    float *T = reinterpret_cast<float *>(dk_lds) + wave * 1024;
    const bool has_res = d.res != nullptr;
    const int pr = lane & 31, ph = lane >> 5, qr = lane >> 3, qc = lane & 7;
    const float unw = 1.f / d.w_scale;
    float vmaxs[TM][4];
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int mb = m0 + (wm * TM + mi) * 32;
        const int mp = mb + pr;
        const float unx = mp < M ? 1.f / a3d_in_scale(d, mp / hwo) : 1.f;
        float (&vmax)[4] = vmaxs[mi];
        vmax[0] = vmax[1] = vmax[2] = vmax[3] = 0.f;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                f32x4 v = {acc[ni][mi][rg * 4 + 0], acc[ni][mi][rg * 4 + 1], acc[ni][mi][rg * 4 + 2], acc[ni][mi][rg * 4 + 3]};
                v = (v * unx) * unw;  // exact: powers of two
                *reinterpret_cast<f32x4 *>(T + pr * 32 + (((rg * 2 + ph) ^ (pr & 7)) << 2)) = v;
            }
            const int nl = (wn * TN + ni) * 32 + qc * 4;
            const int n = n0 + nl;
            const bool nok = n < d.Cout;
            f32x4 rv[4], tv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = qr + 8 * j;
                tv[j] = *reinterpret_cast<const f32x4 *>(T + q * 32 + ((qc ^ (q & 7)) << 2));
                const int m = mb + q;
                if (has_res && nok && m < M) rv[j] = *reinterpret_cast<const f32x4 *>(d.res + (size_t)m * d.Cout + n);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = mb + qr + 8 * j;
                if (!nok || m >= M) continue;
                const f32x4 v = a3d_epilogue_math(d, tv[j], *reinterpret_cast<const f32x4 *>(ss + nl), *reinterpret_cast<const f32x4 *>(ss + BN + nl), has_res, rv[j]);
                vmax[j] = fmaxf(vmax[j], a3d_absmax4(v));
                *reinterpret_cast<f32x4 *>(d.y + (size_t)m * d.Cout + n) = v;
            }
        }
    }
    if (d.y_amax) {  // recorded behind the wave's last store (conv_x3_kernel: the slot's pre-check read queues at the L2)
        const int mb0 = m0 + wm * TM * 32;
        const bool whole = mb0 < M && mb0 / hwo == min(mb0 + TM * 32 - 1, M - 1) / hwo;
        if (whole) {
            float v = 0.f;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) v = fmaxf(v, fmaxf(fmaxf(vmaxs[mi][0], vmaxs[mi][1]), fmaxf(vmaxs[mi][2], vmaxs[mi][3])));
            a3d_note_amax(d.y_amax, mb0 / hwo, v, true);
        } else {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const int mb = m0 + (wm * TM + mi) * 32;
                const float (&vmax)[4] = vmaxs[mi];
                const int mlast = min(mb + 31, M - 1);
                if (mb < M && mb / hwo == mlast / hwo) {
                    a3d_note_amax(d.y_amax, mb / hwo, fmaxf(fmaxf(vmax[0], vmax[1]), fmaxf(vmax[2], vmax[3])), true);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int m = mb + qr + 8 * j;
                        float v = vmax[j];
                        v = fmaxf(v, __shfl_xor(v, 1, 64));
                        v = fmaxf(v, __shfl_xor(v, 2, 64));
                        v = fmaxf(v, __shfl_xor(v, 4, 64));
                        a3d_note_amax(d.y_amax, m < M ? m / hwo : 0, v, m < M && qc == 0);
                    }
                }
            }
        }
    }
}

template <int TM>
int dk_launch(const a3d_conv_desc *d, hipStream_t s, int M) {
    static a3d_attr_once attr;
    if (attr.needed()) {
        if (hipFuncSetAttribute((const void *)conv_dk_kernel<TM>, hipFuncAttributeMaxDynamicSharedMemorySize, dk_lds_bytes(TM)) != hipSuccess) return A3D_ERR_LAUNCH;
        attr.mark();
    }
    const int mtiles = (M + 128 * TM - 1) / (128 * TM), ntiles = (d->Cout + DK_BN - 1) / DK_BN;
    a3d_note_variant("conv_h2dk_kernel<%d>", TM);
    hipLaunchKernelGGL((conv_dk_kernel<TM>), dim3(mtiles * ntiles), dim3(512), dk_lds_bytes(TM), s, *d, M, ntiles, mtiles * ntiles);
    return a3d_check_launch();
}
}  // namespace

// A3D_ERR_UNSUPPORTED: not a layer of this form, or a problem too small for one-workgroup-per-CU tiles (the caller runs conv_x3_kernel:
// the same bits).  tune 26 / 27: this kernel with 128- / 256-pixel tiles whatever the size; tune 28: never.
int a3d_conv_launch_dk_h2(const a3d_conv_desc *d, hipStream_t s) {
    if (d->precision != 3 || d->tune == 28 || !(d->tune == 0 || d->tune == 26 || d->tune == 27)) return A3D_ERR_UNSUPPORTED;
    if (d->KH != 1 || d->KW != 1 || d->pad != 0 || (d->stride != 1 && d->stride != 2) || !d->w_x3 || !d->in_amax || !(d->w_scale > 0.f)) return A3D_ERR_UNSUPPORTED;
    if (d->x2 || d->Cin2 || d->in_amax2 || d->x_h2 || d->ups || d->phase || d->pixshuf || d->stem || d->gate || d->res_ups || d->splitk != 1 || d->m_dev || d->dot_w)
        return A3D_ERR_UNSUPPORTED;
    if ((d->Cin & 31) || d->Kpad != d->Cin || d->Cin < 512 || d->Cout < 128 || (d->Cout & 3)) return A3D_ERR_UNSUPPORTED;
    if ((size_t)d->B * d->H * d->W * d->Cin * 4 >= ((size_t)1 << 31) || (size_t)(d->Kpad / 16) * d->Cout * 64 >= ((size_t)1 << 31)) return A3D_ERR_UNSUPPORTED;
    const int M = d->B * d->Ho * d->Wo;
    const int nt = (d->Cout + DK_BN - 1) / DK_BN;
    const long b1 = (long)((M + 127) / 128) * nt, b2 = (long)((M + 255) / 256) * nt;
    if (d->tune == 26) return dk_launch<1>(d, s, M);
    if (d->tune == 27) return dk_launch<2>(d, s, M);
    if (b1 < 200) return A3D_ERR_UNSUPPORTED;  // (under a round of the chip: the narrow kernel's 128 x 64 tiles, three workgroups per CU)
    // rounds of the chip, one workgroup per CU; a 256-pixel round is ~1.75 x a 128-pixel one (measured, tools/dk_bench.py)
    const long r1 = (b1 + 255) / 256, r2 = (b2 + 255) / 256;
    return (r2 * 175 <= r1 * 100) ? dk_launch<2>(d, s, M) : dk_launch<1>(d, s, M);
}
