// bf16-MFMA variant of the implicit-GEMM convolution / linear kernel (a3d_conv_desc.precision == 1).
//
// What it is for.  The reference trains under bf16 autocast (BASELINE configs[4]: tools/train_net.py with
// config/step1_bbox.yaml, SOLVER.AMP): convolutions and linear layers multiply bf16 operands and accumulate in fp32.
// Tensors stay fp32 in HBM here (master weights, activations, gradients); this kernel rounds both operands to bf16
// (round-to-nearest-even, v_cvt_pk_bf16_f32) while it stages them in LDS and multiplies with
// v_mfma_f32_32x32x16_bf16 -- 16x the matrix rate of the fp32 MFMA, so every layer of the step becomes HBM-bound.
// The default (precision 0) fp32 path is untouched: inference parity is an fp32 statement.
//
// Same decomposition as conv_gemm_v2 (MODE_GENERIC): output tile 128 x (64*TN), 4 waves as 2x2, weights are MFMA
// operand A and activations operand B (a lane owns one output pixel, register quads are 4 consecutive channels), buffer
// loads with the hardware range check for padding taps / ragged rows, single register staging set, one barrier per
// 32-deep k chunk, the LDS-staged epilogue.  LDS image: [row][32 k] bf16 with an 80-byte row pitch -- a fragment is
// ONE ds_read_b128 (8 consecutive k of one row), and 16 rows x 80 B tile the 64 banks exactly once per lane group.
#include "conv_common.h"
#include <stdlib.h>

namespace {
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 bf_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bf_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
// bf16 STORAGE (a3d_conv_desc.io_bf16): four stored bf16 values <-> f32x4.  Widening is exact; narrowing rounds to nearest even.
typedef unsigned int bf_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 bf_widen4(const bf_u32x2 v) {
    f32x4 o;
    o[0] = __builtin_bit_cast(float, v[0] << 16);
    o[1] = __builtin_bit_cast(float, v[0] & 0xFFFF0000u);
    o[2] = __builtin_bit_cast(float, v[1] << 16);
    o[3] = __builtin_bit_cast(float, v[1] & 0xFFFF0000u);
    return o;
}
__device__ __forceinline__ f32x4 bf_read4(const float *base, size_t idx, bool is_bf16) {  // element index idx (multiple of 4)
    if (is_bf16) return bf_widen4(*reinterpret_cast<const bf_u32x2 *>(reinterpret_cast<const __bf16 *>(base) + idx));
    return *reinterpret_cast<const f32x4 *>(base + idx);
}
__device__ __forceinline__ void bf_write4(float *base, size_t idx, const f32x4 v, bool is_bf16) {
    if (is_bf16) *reinterpret_cast<bf16x4 *>(reinterpret_cast<__bf16 *>(base) + idx) = __builtin_convertvector(v, bf16x4);
    else *reinterpret_cast<f32x4 *>(base + idx) = v;
}

// XB: the activations are STORED as bf16 (io_bf16 bit 0).  A lane then loads 8 channels (16 bytes) and the bits go to LDS as they are:
// half the loads of the fp32-stored form, no conversion.
template <int TN, bool XB = false>
__global__ __launch_bounds__(256, 3) void conv_bf16_kernel(const a3d_conv_desc d, const int M, const int ntiles, const int nblk) {
    constexpr int TM = 2, BKT = 32;
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
    constexpr int LKB = BKT + 8;               // bf16 elements per LDS row (80 bytes)
    constexpr int TPR = BKT / 4, RPP = 256 / TPR;  // 8 lanes x float4 per row, 32 rows per loader pass
    constexpr int XE = XB ? 8 : 4, XES = XB ? 2 : 4;  // activation elements per lane load, bytes per stored element
    constexpr int TPRX = BKT / XE, RPPX = 256 / TPRX;
    constexpr int XR = BM / RPPX, WR = BN / RPP;
    constexpr int BUF = (BM + BN) * LKB;
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * BUF];
    __shared__ __attribute__((aligned(16))) float ss[2 * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int logical = a3d_xcd_remap(blockIdx.x, nblk);
    const int mt = logical / ntiles, nt = logical - mt * ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    // split-K (a3d_conv_desc.splitk > 1, round 4): blockIdx.y owns the chunks [k_first, k_first + nk) of the reduction and stores its raw
    // accumulators to workspace [split][M][Cout]; conv_bf16_reduce_kernel adds the splits in order and applies the epilogue.  For the
    // training step at 2 images per GPU: a 3x3 256 -> 256 layer on 2 x 30 x 40 pixels is 76 workgroups walking 72 chunks one
    // memory round trip at a time.
    const int nk_all = d.Kpad / BKT;
    const int per = (nk_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int k_first = blockIdx.y * per;
    const int nk = max(0, min(nk_all, k_first + per) - k_first);
    const int lr = tid / TPR, lc = (tid % TPR) * 4;
    const int lrx = tid / TPRX, lcx = (tid % TPRX) * XE;
    const bool yb = d.io_bf16 & 2, rb = d.io_bf16 & 4, gb = d.io_bf16 & 8;  // tensors stored as bf16
    const int cs4 = d.Cin * XES;  // bytes per input pixel
    const __amdgpu_buffer_rsrc_t rx = bf_rsrc(d.x, (unsigned)((size_t)d.B * d.H * d.W * (size_t)cs4));
    const __amdgpu_buffer_rsrc_t rw = bf_rsrc(d.w, (unsigned)((size_t)d.Cout * d.Kpad * 4));

    int rowoff[XR];
    unsigned vmask[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int m = m0 + lrx + RPPX * i;
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
        rowoff[i] = ((b * d.H + ih0) * d.W + iw0) * cs4 + lcx * XES;
        vmask[i] = mask;
    }
    int woff[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int n = n0 + lr + RPP * i;
        woff[i] = n < d.Cout ? (n * d.Kpad + lc) * 4 : -1;
    }
    // position of the next chunk to load inside the filter: chunk k of the reduction = (tap k / (Cin / 32), channels 32 (k % (Cin / 32)))
    const int cpt = d.Cin / BKT;
    int kc = 0, c0 = (k_first % cpt) * BKT, kh = (k_first / cpt) / d.KW, kw = (k_first / cpt) % d.KW;
    // TWO register staging sets (round 4): the loads of chunk c are issued at iteration c - 3 and written to LDS at iteration c - 1, two
    // iterations of lead instead of one.  The training step's launches at the reference's 2 images per GPU are a handful of workgroups per
    // CU with 8-64 chunks of 4-8 MFMAs each: every iteration waited out the full latency of loads issued one short iteration earlier.
    // (A third set on the 128 x 64 tile, loads three chunks ahead: 255-256 against 258 images/s at 2 images, 826 against 838 at 16: not taken.)
    f32x4 xsA[XR], wsA[WR], xsB[XR], wsB[WR];
    auto load_chunk = [&](f32x4 (&xs)[XR], f32x4 (&ws)[WR]) {
        const int tap = kh * d.KW + kw;
        const unsigned livebit = (kc < nk) ? 1u : 0u;
        const int tapoff = (kh * d.W + kw) * cs4 + c0 * XES;
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const bool on = (vmask[i] >> (tap & 31)) & livebit;
            xs[i] = bf_load4(rx, on ? rowoff[i] + tapoff : -1, 0);  // (XB: the four dwords are 8 stored bf16 values)
        }
        const int soff = (k_first + kc) * (BKT * 4);
#pragma unroll
        for (int i = 0; i < WR; ++i) ws[i] = bf_load4(rw, livebit ? woff[i] : -1, soff);
        ++kc;
        c0 += BKT;
        if (c0 >= d.Cin) {
            c0 = 0;
            if (++kw == d.KW) {
                kw = 0;
                ++kh;
            }
        }
    };
    auto store_chunk = [&](int buf, const f32x4 (&xs)[XR], const f32x4 (&ws)[WR]) {  // fp32 -> bf16 (RNE) on the way into LDS
        __bf16 *X = lds + buf * BUF;
        __bf16 *Wt = X + BM * LKB;
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            if constexpr (XB) *reinterpret_cast<f32x4 *>(X + (lrx + RPPX * i) * LKB + lcx) = xs[i];
            else *reinterpret_cast<bf16x4 *>(X + (lrx + RPPX * i) * LKB + lcx) = __builtin_convertvector(xs[i], bf16x4);
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) *reinterpret_cast<bf16x4 *>(Wt + (lr + RPP * i) * LKB + lc) = __builtin_convertvector(ws[i], bf16x4);
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    a3d_stage_scale_shift(ss, d, n0, BN, tid);
    constexpr bool DEEP = !(TN == 2 && !XB);
    load_chunk(xsA, wsA);  // chunk 0
    store_chunk(0, xsA, wsA);
    if constexpr (DEEP) {
        load_chunk(xsB, wsB);  // chunk 1
        load_chunk(xsA, wsA);  // chunk 2
    } else {
        load_chunk(xsA, wsA);  // chunk 1
    }
    __syncthreads();

    const int frag_off = (lane & 31) * LKB + (lane >> 5) * 8;  // row = lane % 32, k = 8 * (lane / 32) .. + 7
    auto step = [&](const int cur, f32x4 (&xs)[XR], f32x4 (&ws)[WR]) {  // multiplies the chunk in LDS[cur]; xs / ws hold the next one
        const __bf16 *X = lds + cur * BUF + (wm * TM * 32) * LKB + frag_off;
        const __bf16 *Wt = lds + cur * BUF + BM * LKB + (wn * TN * 32) * LKB + frag_off;
        bf16x8 fa[2][TN], fb[2][TM];
#pragma unroll
        for (int s = 0; s < 2; ++s) {  // two 16-deep MFMA steps per chunk
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) fa[s][ni] = *reinterpret_cast<const bf16x8 *>(Wt + ni * 32 * LKB + s * 16);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) fb[s][mi] = *reinterpret_cast<const bf16x8 *>(X + mi * 32 * LKB + s * 16);
        }
        store_chunk(cur ^ 1, xs, ws);
        load_chunk(xs, ws);  // (two chunks past the one just stored)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][ni], fb[s][mi], acc[ni][mi], 0, 0, 0);
        __syncthreads();
    };
    if constexpr (DEEP) {
        for (int it = 0; it < nk; it += 2) {
            step(0, xsB, wsB);
            if (it + 1 < nk) step(1, xsA, wsA);
        }
    } else {  // (the 128 x 128 tile on fp32-stored activations: 168 registers with one staging set -- a second one spills)
        for (int it = 0; it < nk; ++it) step(it & 1, xsA, wsA);
    }

    if (gridDim.y > 1) {  // raw partial sums of this split
        float *part = d.workspace + (size_t)blockIdx.y * M * d.Cout;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int m = m0 + (wm * TM + mi) * 32 + (lane & 31);
            if (m >= M) continue;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int n = n0 + (wn * TN + ni) * 32 + rg * 8 + (lane >> 5) * 4;
                    if (n >= d.Cout) continue;
                    const f32x4 v = {acc[ni][mi][rg * 4 + 0], acc[ni][mi][rg * 4 + 1], acc[ni][mi][rg * 4 + 2], acc[ni][mi][rg * 4 + 3]};
                    *reinterpret_cast<f32x4 *>(part + (size_t)m * d.Cout + n) = v;
                }
        }
        return;
    }
    const bool has_res = d.res != nullptr;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + (wm * TM + mi) * 32 + (lane & 31);
        if (m >= M) continue;
        size_t res_row;
        int b, oh, ow;
        out_rows(d, m, res_row, b, oh, ow);
        // Residual AND gate quads of the whole output row are requested before its first store (round 5): y / res / gate may alias, so the
        // compiler keeps every load in program order with the stores -- a gate read in front of each store was one memory round trip per
        // quad, and the gated launches (every data gradient of the step) took up to twice the time of their ungated twins.
        const bool has_gate = d.io_bf16 && d.gate != nullptr;
        f32x4 rv[TN][4], gv[TN][4];
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int n = n0 + (wn * TN + ni) * 32 + rg * 8 + (lane >> 5) * 4;
                if (has_res) rv[ni][rg] = bf_read4(d.res, res_row * (size_t)d.Cout + min(n, d.Cout - 4), rb);
                if (has_gate) gv[ni][rg] = bf_read4(d.gate, (size_t)m * d.Cout + min(n, d.Cout - 4), gb);
            }
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int nl = (wn * TN + ni) * 32 + rg * 8 + (lane >> 5) * 4;
                const int n = n0 + nl;
                if (n >= d.Cout) continue;
                f32x4 v = {acc[ni][mi][rg * 4 + 0], acc[ni][mi][rg * 4 + 1], acc[ni][mi][rg * 4 + 2], acc[ni][mi][rg * 4 + 3]};
                v = a3d_epilogue_math(d, v, *reinterpret_cast<const f32x4 *>(ss + nl), *reinterpret_cast<const f32x4 *>(ss + BN + nl), has_res, rv[ni][rg]);
                if (d.io_bf16) {  // (plain output layout only: the launcher refuses pixshuf / phase with bf16 storage)
                    const size_t o = (size_t)m * d.Cout + n;
                    if (has_gate) {
                        const f32x4 g = gv[ni][rg];
                        for (int i = 0; i < 4; ++i) v[i] = g[i] > 0.f ? v[i] : 0.f;
                    }
                    bf_write4(d.y, o, v, yb);
                } else {
                    store_out(d, v, m, n, b, oh, ow);
                }
            }
        }
    }
}

// second launch of a split-K layer: the splits in order, then the epilogue of the kernel above (scale / shift, residual, activation, gate,
// fp32 or bf16 stores), per output quad.
// (Measured and removed, round 5: ONE launch -- a counter per tile, the workgroup that arrives last adds the splits and applies the
// epilogue; bit-identical, and the 51 second launches of the 2-image step were gone -- but the release / acquire pair at device scope
// is a write-back + invalidate of the XCD's whole L2 per workgroup on this chip: the step went 5.94 -> 7.87 ms, +38 us per split-K layer.)
__global__ __launch_bounds__(256) void conv_bf16_reduce_kernel(const a3d_conv_desc d, const int M) {
    const int n4 = d.Cout >> 2;
    const size_t total = (size_t)M * n4;
    const bool yb = d.io_bf16 & 2, rb = d.io_bf16 & 4, gb = d.io_bf16 & 8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / n4), n = (int)(i - (size_t)m * n4) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < d.splitk; ++z) v += *reinterpret_cast<const f32x4 *>(d.workspace + ((size_t)z * M + m) * d.Cout + n);
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, rv = {0.f, 0.f, 0.f, 0.f};
        if (d.scale) sc = *reinterpret_cast<const f32x4 *>(d.scale + n);
        if (d.shift) sh = *reinterpret_cast<const f32x4 *>(d.shift + n);
        const size_t o = (size_t)m * d.Cout + n;
        if (d.res) rv = bf_read4(d.res, o, rb);
        v = a3d_epilogue_math(d, v, sc, sh, d.res != nullptr, rv);
        if (d.gate) {
            const f32x4 g = bf_read4(d.gate, o, gb);
            for (int k = 0; k < 4; ++k) v[k] = g[k] > 0.f ? v[k] : 0.f;
        }
        bf_write4(d.y, o, v, yb);
    }
}

template <int TN>
void launch_bf16(const a3d_conv_desc *d, hipStream_t s) {
    constexpr int BM = 128, BN = 64 * TN;
    const int M = d->B * d->Ho * d->Wo;
    const int mtiles = (M + BM - 1) / BM, ntiles = (d->Cout + BN - 1) / BN;
    a3d_note_variant(d->splitk > 1 ? "conv_bf16_kernel<%d> sk%d" : "conv_bf16_kernel<%d>", TN, d->splitk);
    const dim3 grid(mtiles * ntiles, d->splitk);
    if (d->io_bf16 & 1) hipLaunchKernelGGL((conv_bf16_kernel<TN, true>), grid, dim3(256), 0, s, *d, M, ntiles, mtiles * ntiles);
    else hipLaunchKernelGGL((conv_bf16_kernel<TN, false>), grid, dim3(256), 0, s, *d, M, ntiles, mtiles * ntiles);
    if (d->splitk > 1) {
        const size_t total = (size_t)M * (d->Cout >> 2);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(conv_bf16_reduce_kernel, dim3(blocks), dim3(256), 0, s, *d, M);
    }
}
}  // namespace

int a3d_conv_launch_bf16(const a3d_conv_desc *d, hipStream_t s) {
    {
        const int rx = a3d_conv_launch_bf16xs(d, s);  // HBM-bound pointwise layers: activations stationary in registers (the same bits)
        if (rx != A3D_ERR_UNSUPPORTED) return rx;
        const int rw = a3d_conv_launch_bf16w(d, s);  // large launches with a bf16 filter: both operands by DMA, 256-pixel tiles (the same bits)
        if (rw != A3D_ERR_UNSUPPORTED) return rw;
    }
    if (d->stem || d->ups || d->phase || d->pixshuf || d->x2 || d->Cin2 || d->splitk < 1 || d->m_dev) return A3D_ERR_UNSUPPORTED;
    if (d->splitk > 1 && (!d->workspace || d->res_ups || d->splitk > d->Kpad / 32)) return A3D_ERR_ARG;  // (split-K: plain output rows only)
    if (d->io_bf16 & ~15) return A3D_ERR_ARG;
    if ((d->Cin & 31) || d->Kpad != d->KH * d->KW * d->Cin || d->KH * d->KW > 32) return A3D_ERR_UNSUPPORTED;
    if ((size_t)d->B * d->H * d->W * d->Cin * 4 >= ((size_t)1 << 32) || (size_t)d->Cout * d->Kpad * 4 >= ((size_t)1 << 32)) return A3D_ERR_UNSUPPORTED;
    const int M = d->B * d->Ho * d->Wo;
    const long n128 = (long)((M + 127) / 128) * ((d->Cout + 127) / 128);
    // (the 128 x 64 tile on small grids -- except deep reductions: the box head's fc1, K = 12544, moves 24 KiB per chunk and 128 x 64 tile
    // through L2 for 0.5 MFLOP, 383 TFLOP/s at 16 images: 842 -> 848 images/s; under 256 tiles -- 2 images -- the narrow tile
    // with split-K stays ahead; developer builds: A3D_BF16_DEEP_WIDE=0 restores the narrow tile everywhere)
    const int deep_wide = (int)a3d_dev_knob("A3D_BF16_DEEP_WIDE", 1);
    if (d->Cout <= 64 || (n128 <= 1000 && !(deep_wide && d->Kpad >= 4096 && d->Cout >= 128 && n128 >= 256))) launch_bf16<1>(d, s);
    else launch_bf16<2>(d, s);
    return a3d_check_launch();
}
