// fp32-MFMA implicit-GEMM convolution / linear for gfx950 (MI355X).
//
//   y[m, n] = act( scale[n] * sum_k X[m, k] * w[n, k] + shift[n] + res[m, n] )
//
// m = (b, oh, ow) over the NHWC output, n = output channel, k = (kh, kw, c).  The im2col matrix X is
// never materialised: each workgroup gathers its [BM x 32] slice of X straight from the NHWC input
// (k-chunks of 32 never straddle a filter tap because Cin % 32 == 0), stages it and the [BN x 32]
// weight slice through LDS (register-staged double buffer: global loads of chunk t+1 are in flight
// while chunk t is multiplied), and each of the 4 waves accumulates TN x TM tiles of 32x32 with
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD).
//
// MFMA operand roles: the WEIGHTS are the "A" operand (row i = output channel) and the ACTIVATIONS
// the "B" operand (column j = output pixel), so in the accumulator a lane owns one pixel and each
// group of 4 registers is 4 consecutive channels -> the fused epilogue loads scale/shift/residual and
// stores the NHWC result as float4.
//
// k order inside one 8-wide slice is permuted (lane half h multiplies k = 4h..4h+3) so that both
// operands are fetched from LDS with ONE ds_read_b128 per 4 MFMAs; A and B use the same
// permutation, so the sum over the slice is complete.  LDS rows are padded to 36 floats: 16
// consecutive rows then cover all 64 banks exactly once per b128 lane group (conflict-free).
#include "conv_common.h"
#include <stdarg.h>
#include <stdio.h>

struct RowCtx {
    int ih0, iw0, boff;  // top-left input coordinate of the receptive field, b*H*W
    bool ok;
};

template <int WAVES_M, int WAVES_N, int TM, int TN, bool STEM>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const a3d_conv_desc d, const int Mmax, const int ntiles,
                                                        const int nblk, const int kt_total, const int kt_per_split) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    const int M = d.m_dev ? min(Mmax, *d.m_dev) : Mmax;
    constexpr int XR = BM / 32, WR = BN / 32;
    constexpr int BUF = (BM + BN) * LDK;
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int logical = a3d_xcd_remap(blockIdx.x, nblk);
    const int mt = logical / ntiles, nt = logical - mt * ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    if (m0 >= M) return;  // ragged batch: whole tile past the live rows (uniform exit, before any barrier)
    const int z = blockIdx.y;
    const int kt_begin = z * kt_per_split;
    const int kt_end = min(kt_total, kt_begin + kt_per_split);

    const int lr = tid >> 3, lc = (tid & 7) * 4;
    const int CinT = d.Cin + d.Cin2;
    const int Hl = d.ups ? 2 * d.H : d.H, Wl = d.ups ? 2 * d.W : d.W;  // logical input extent

    RowCtx rc[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int m = m0 + lr + 32 * i;
        rc[i].ok = m < M;
        const int mm = rc[i].ok ? m : 0;
        const int hw = d.Ho * d.Wo;
        const int b = mm / hw;
        const int r = mm - b * hw;
        const int oh = r / d.Wo, ow = r - oh * d.Wo;
        rc[i].ih0 = oh * d.stride - d.pad;
        rc[i].iw0 = ow * d.stride - d.pad;
        rc[i].boff = b * d.H * d.W;
    }
    const float *wrow[WR];
    bool wok[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int n = n0 + lr + 32 * i;
        wok[i] = n < d.Cout;
        wrow[i] = d.w + (size_t)(wok[i] ? n : 0) * d.Kpad + lc;
    }

    f32x4 xs[XR], ws[WR];
    auto load_chunk = [&](int kc) {
        if (STEM) {
            const int j = tid & 7;
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const int ih = rc[i].ih0 + kc, iw = rc[i].iw0 + j;
                const bool ok = rc[i].ok && j < 7 && (unsigned)ih < (unsigned)d.H && (unsigned)iw < (unsigned)d.W;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = *reinterpret_cast<const f32x4 *>(d.x + ((size_t)rc[i].boff + (size_t)ih * d.W + iw) * 4);
                xs[i] = v;
            }
        } else {
            const int kk = kc * BK;
            const int tap = kk / CinT;
            const int c0 = kk - tap * CinT;
            const int kh = tap / d.KW, kw = tap - kh * d.KW;
            const bool tap_ok = tap < d.KH * d.KW;
            const bool second = c0 >= d.Cin;
            const float *src = second ? d.x2 : d.x;
            const int cs = second ? d.Cin2 : d.Cin;
            const int cc = (second ? c0 - d.Cin : c0) + lc;
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                int ih = rc[i].ih0 + kh, iw = rc[i].iw0 + kw;
                const bool ok = tap_ok && rc[i].ok && (unsigned)ih < (unsigned)Hl && (unsigned)iw < (unsigned)Wl;
                if (d.ups) {
                    ih >>= 1;
                    iw >>= 1;
                }
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = *reinterpret_cast<const f32x4 *>(src + ((size_t)rc[i].boff + (size_t)ih * d.W + iw) * cs + cc);
                xs[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (wok[i]) v = *reinterpret_cast<const f32x4 *>(wrow[i] + (size_t)kc * BK);
            ws[i] = v;
        }
    };
    auto store_chunk = [&](int buf) {
        float *X = lds + buf * BUF;
        float *Wt = X + BM * LDK;
#pragma unroll
        for (int i = 0; i < XR; ++i) *reinterpret_cast<f32x4 *>(X + (lr + 32 * i) * LDK + lc) = xs[i];
#pragma unroll
        for (int i = 0; i < WR; ++i) *reinterpret_cast<f32x4 *>(Wt + (lr + 32 * i) * LDK + lc) = ws[i];
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if (kt_begin < kt_end) {
        load_chunk(kt_begin);
        store_chunk(0);
    }
    __syncthreads();

    const int frag_off = (lane & 31) * LDK + (lane >> 5) * 4;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const bool more = kt + 1 < kt_end;
        if (more) load_chunk(kt + 1);
        const float *X = lds + cur * BUF + (wm * TM * 32) * LDK + frag_off;
        const float *Wt = lds + cur * BUF + BM * LDK + (wn * TN * 32) * LDK + frag_off;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            f32x4 a[TN], b[TM];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) a[ni] = *reinterpret_cast<const f32x4 *>(Wt + ni * 32 * LDK + q * 8);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) b[mi] = *reinterpret_cast<const f32x4 *>(X + mi * 32 * LDK + q * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ni][j], b[mi][j], acc[ni][mi], 0, 0, 0);
        }
        if (more) store_chunk(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane owns pixel m, register group rg holds channels n..n+3 -----------------
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + (wm * TM + mi) * 32 + (lane & 31);
        if (m >= M) continue;
        size_t res_row;
        int b, oh, ow;
        out_rows(d, m, res_row, b, oh, ow);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int n = n0 + (wn * TN + ni) * 32 + rg * 8 + (lane >> 5) * 4;
                if (n >= d.Cout) continue;
                f32x4 v = {acc[ni][mi][rg * 4 + 0], acc[ni][mi][rg * 4 + 1], acc[ni][mi][rg * 4 + 2],
                           acc[ni][mi][rg * 4 + 3]};
                if (d.splitk > 1) {
                    *reinterpret_cast<f32x4 *>(d.workspace + ((size_t)z * Mmax + m) * d.Cout + n) = v;
                } else {
                    v = apply_epilogue(d, v, n, res_row);
                    store_out(d, v, m, n, b, oh, ow);
                }
            }
        }
    }
}

// Split-K second pass: sum the partial slabs in slice order (bitwise reproducible) + fused epilogue.
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const a3d_conv_desc d, const int Mmax) {
    const int M = d.m_dev ? min(Mmax, *d.m_dev) : Mmax;
    const int n4 = d.Cout >> 2;
    const size_t total = (size_t)M * n4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4);
        const int n = (int)(i - (size_t)m * n4) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < d.splitk; ++z) v += *reinterpret_cast<const f32x4 *>(d.workspace + ((size_t)z * Mmax + m) * d.Cout + n);
        size_t res_row;
        int b, oh, ow;
        out_rows(d, m, res_row, b, oh, ow);
        v = apply_epilogue(d, v, n, res_row);
        store_out(d, v, m, n, b, oh, ow);
    }
}

static int conv_check(const a3d_conv_desc *d) {
    if (!d || (!d->x && !d->x_h2) || !d->w || !d->y) return A3D_ERR_ARG;
    if (d->x_h2 && d->precision != 3) return A3D_ERR_ARG;  // pre-split activations exist in the fp16x2 arithmetic only
    if (d->B <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0) return A3D_ERR_ARG;
    if ((d->Cout & 3) || (d->Kpad & 31) || d->splitk < 1) return A3D_ERR_ARG;
    if (d->stem) {
        if (d->KH != 7 || d->KW != 7 || d->stride != 2 || d->pad != 3 || d->Kpad != 224 || d->Cin2 || d->ups) return A3D_ERR_ARG;
    } else {
        if ((d->Cin & 31) || (d->Cin2 & 31) || (d->Cin2 && !d->x2 && !d->x2_h2)) return A3D_ERR_ARG;
        if (d->Kpad < d->KH * d->KW * (d->Cin + d->Cin2)) return A3D_ERR_ARG;
    }
    if (d->pixshuf && (d->Cout & 15)) return A3D_ERR_ARG;
    if (d->phase < 0 || d->phase > 5 || (d->phase == 5 && d->precision != 3)) return A3D_ERR_ARG;  // (5: the fused four-phase form, fp16x2 only)
    if (d->gate && (d->pixshuf || d->phase)) return A3D_ERR_ARG;
    if (d->io_bf16 && d->precision != 1) return A3D_ERR_ARG;  // bf16 storage belongs to the bf16 (autocast) arithmetic
    if (d->splitk > 1 && !d->workspace) return A3D_ERR_ARG;
    if ((size_t)d->B * d->H * d->W >= ((size_t)1 << 31)) return A3D_ERR_UNSUPPORTED;
    return A3D_OK;
}

static thread_local char g_last_variant[160] = "";
void a3d_note_variant(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_variant, sizeof(g_last_variant), fmt, ap);
    va_end(ap);
}
extern "C" const char *a3d_last_conv_variant(void) { return g_last_variant; }

extern "C" size_t a3d_conv_workspace_bytes(const a3d_conv_desc *d) {
    if (!d) return 0;
    if (d->tune == 0 && a3d_wino_fused_eligible(d)) return 0;
    if ((d->tune == 0 || d->tune == 7 || d->tune == 8 || d->tune >= 200) && a3d_wino_eligible(d)) return a3d_wino_workspace_bytes(d);
    if (d->splitk <= 1) return 0;
    return (size_t)d->splitk * d->B * d->Ho * d->Wo * d->Cout * sizeof(float);
}

template <int WAVES_M, int WAVES_N, int TM, int TN, bool STEM>
static int launch_cfg(const a3d_conv_desc *d, hipStream_t s) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    const int M = d->B * d->Ho * d->Wo;
    const int mtiles = (M + BM - 1) / BM, ntiles = (d->Cout + BN - 1) / BN;
    const int nblk = mtiles * ntiles;
    const int kt_total = d->Kpad / BK;
    const int kps = (kt_total + d->splitk - 1) / d->splitk;
    dim3 grid(nblk, d->splitk);
    a3d_note_variant("conv_gemm_kernel<%d,%d,%d,%d,%d>", WAVES_M, WAVES_N, TM, TN, (int)STEM);
    hipLaunchKernelGGL((conv_gemm_kernel<WAVES_M, WAVES_N, TM, TN, STEM>), grid, dim3(256), 0, s, *d, M, ntiles, nblk,
                       kt_total, kps);
    if (d->splitk > 1) {
        const size_t total = (size_t)M * (d->Cout >> 2);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, *d, M);
    }
    return a3d_check_launch();
}

extern "C" int a3d_conv2d_nhwc_f32(const a3d_conv_desc *d, void *stream) {
    const int rc = conv_check(d);
    if (rc != A3D_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    a3d_begin();
    if (d->precision == 1) return a3d_conv_launch_bf16(d, s);  // (A3D_ERR_UNSUPPORTED for layer kinds it does not cover)
    if (d->precision == 2) {  // Winograd layers keep the Winograd form (split-operand GEMM, conv_wino.hip 2x), the rest go direct
        if ((d->tune == 0 || d->tune == 8) && d->workspace && d->w_wino && d->w_wino_x3 && ((d->Cin + d->Cin2) & 31) == 0 && a3d_wino_eligible(d)) return a3d_conv_launch_wino(d, s);
        return a3d_conv_launch_bf16x3(d, s);
    }
    if (d->precision == 3 && d->x_h2) return a3d_conv_launch_bf16x3_wide(d, s);  // pre-split activations: the dual-DMA forms only
    if (d->precision == 3) {  // fp16x2 split: Winograd layers (wide kernels) when their pre-split filter is given, the rest direct
        if (d->tune == 0 && d->workspace && d->w_wino && d->w_wino_x3 && ((d->Cin + d->Cin2) & 31) == 0 && a3d_wino_eligible(d)) return a3d_conv_launch_wino(d, s);
        return a3d_conv_launch_bf16x3(d, s);
    }
    if (d->precision != 0) return A3D_ERR_ARG;
    if (d->tune == 0 && a3d_wino_fused_eligible(d)) return a3d_conv_launch_wino_fused(d, s);  // one launch, no V tensor (tune 7: the two-launch form)
    if ((d->tune == 0 || d->tune == 7 || d->tune >= 200) && d->workspace && a3d_wino_eligible(d)) return a3d_conv_launch_wino(d, s);
    if (d->tune == 0 || d->tune == 6) {  // persistent pointwise kernel for the 1x1 layers (tune 5: never, 6: whenever eligible)
        const int r1 = a3d_conv_launch_pw(d, s, d->tune == 6);
        if (r1 != A3D_ERR_UNSUPPORTED) return r1;
    }
    if (d->tune != 1) {  // tune == 1 forces the general kernel (A/B measurements, fallback)
        const int r2 = a3d_conv_launch_v2(d, s);
        if (r2 != A3D_ERR_UNSUPPORTED) return r2;
    }
    if (d->phase) return A3D_ERR_UNSUPPORTED;  // the phase form only exists in the v2 family
    if (d->stem) return launch_cfg<4, 1, 2, 2, true>(d, s);
    if (d->Cout <= 32) return launch_cfg<4, 1, 1, 1, false>(d, s);
    if (d->Cout <= 64) return launch_cfg<4, 1, 2, 2, false>(d, s);
    return launch_cfg<2, 2, 2, 2, false>(d, s);
}

extern "C" int a3d_version(void) { return 1; }
