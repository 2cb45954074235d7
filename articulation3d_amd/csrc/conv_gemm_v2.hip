// conv-GEMM v2: same math, tiling, LDS image and epilogue as conv_gemm.hip (see the header comment there),
// re-built around what the v1 profile showed (MFMA pipe 76 % busy, the rest lost in a branchy 64-bit gather
// that ran BEFORE the MFMA block and a write-back + barrier that ran AFTER it):
//
//  * gather through BUFFER loads: every operand is addressed as rsrc + 32-bit byte offset; a tap that falls
//    into the zero padding (or a row past M / Cout) gets offset 0xFFFFFFFF, which the hardware range check
//    turns into a zero result -- no exec-mask branches, no 64-bit address arithmetic.  Per-row validity of
//    all KH*KW taps is one precomputed bitmask; per k-chunk the lane work is 1 add + 1 bit test + 1 select.
//  * the main loop is ONE basic block: the LDS write-back of chunk t+1 (loaded during iteration t-1) and
//    the global loads of chunk t+2 are issued in the shadow of the 64 MFMAs of chunk t (each fp32 32x32x2
//    MFMA occupies the matrix pipe for 64 cycles but only a few issue cycles), fragments are double-buffered
//    in registers, one barrier per chunk.
#include "conv_common.h"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

enum { MODE_GENERIC = 0, MODE_UPS = 1, MODE_STEM = 2 };

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

// uniform (scalar) position of the current k-chunk inside the filter
struct ChunkPos {
    int kc;       // chunk index (k = 32*kc)
    int c0;       // first channel of the chunk inside the concatenated input
    int kh, kw;   // filter tap
};

template <int WAVES_M, int WAVES_N, int TM, int TN, int MODE, int PIPE, int BKT>
// (second argument: workgroups per CU the register allocation must allow -- the LDS footprint of the BK=16 tiles fits 3)
__global__ __launch_bounds__(WAVES_M *WAVES_N * 64, (BKT == 16 && PIPE == 0) ? 3 : 1) void conv_gemm_v2_kernel(const a3d_conv_desc d, const int Mmax,
                                                                              const int ntiles, const int nblk,
                                                                              const int kt_total, const int kt_per_split) {
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    constexpr int LK = BKT + 4;   // padded LDS row (floats): 36 / 20 both tile the 64 banks exactly per b128 lane group
    constexpr int TPR = BKT / 4;  // lanes (x 16 B) per tile row
    constexpr int RPP = NT / TPR; // tile rows covered by one pass of the workgroup
    static_assert(MODE != MODE_STEM || BKT == 32, "the stem packs one 7-tap filter row into a 32-float chunk");
    constexpr int XR = BM / RPP, WR = BN / RPP;
    constexpr int BUF = (BM + BN) * LK;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the loader pass");
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
    __shared__ __attribute__((aligned(16))) float ss[2 * BN];  // folded-BN scale | shift of this N tile (epilogue)

    const int M = d.m_dev ? min(Mmax, *d.m_dev) : Mmax;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int logical = a3d_xcd_remap(blockIdx.x, nblk);
    const int mt = logical / ntiles, nt = logical - mt * ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    if (m0 >= M) return;
    const int z = blockIdx.y;
    const int kt_begin = z * kt_per_split;
    const int kt_end = min(kt_total, kt_begin + kt_per_split);
    const int nk = kt_end - kt_begin;

    const int lr = tid / TPR, lc = (tid % TPR) * 4;
    const int CinT = (MODE == MODE_STEM) ? 32 : d.Cin + d.Cin2;
    const int cs4 = (MODE == MODE_STEM) ? 16 : d.Cin * 4;  // bytes between consecutive pixels of a source
    const unsigned xbytes = (unsigned)((size_t)d.B * d.H * d.W * (size_t)cs4);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(d.x, xbytes);
    const __amdgpu_buffer_rsrc_t rx2 = make_rsrc(d.x2 ? d.x2 : d.x, xbytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(d.w, (unsigned)((size_t)d.Cout * d.Kpad * 4));

    // ---- per-row loop invariants ---------------------------------------------------------------
    int rowoff[XR];      // GENERIC/STEM: byte offset of (b, ih0, iw0[+j], lc) -- may be "negative", used mod 2^32
    unsigned vmask[XR];  // bit t set <=> filter tap t of this row reads inside the image
    int uih0[XR], uiw0[XR], uboff[XR];  // UPS only
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int m = m0 + lr + RPP * i;
        const bool rok = m < M;
        const int mm = rok ? m : 0;
        const int hw = d.Ho * d.Wo;
        const int b = mm / hw;
        const int r = mm - b * hw;
        const int oh = r / d.Wo, ow = r - oh * d.Wo;
        int ih0 = oh * d.stride - d.pad, iw0 = ow * d.stride - d.pad;
        if (d.phase) {  // 2x2 taps of output phase (dy,dx) on the source grid
            ih0 = oh - 1 + ((d.phase - 1) >> 1);
            iw0 = ow - 1 + ((d.phase - 1) & 1);
        }
        unsigned mask = 0;
        if (MODE == MODE_STEM) {
            const int j = tid & 7;
            const bool jok = rok && j < 7 && (unsigned)(iw0 + j) < (unsigned)d.W;
            for (int kh = 0; kh < 7; ++kh) mask |= (jok && (unsigned)(ih0 + kh) < (unsigned)d.H) ? (1u << kh) : 0u;
            rowoff[i] = ((b * d.H + ih0) * d.W + iw0 + j) * 16;
        } else {
            const int Hl = (MODE == MODE_UPS) ? 2 * d.H : d.H, Wl = (MODE == MODE_UPS) ? 2 * d.W : d.W;
            for (int kh = 0; kh < d.KH; ++kh)
                for (int kw = 0; kw < d.KW; ++kw)
                    mask |= (rok && (unsigned)(ih0 + kh) < (unsigned)Hl && (unsigned)(iw0 + kw) < (unsigned)Wl) ? (1u << (kh * d.KW + kw)) : 0u;
            rowoff[i] = ((b * d.H + ih0) * d.W + iw0) * cs4 + lc * 4;
        }
        vmask[i] = mask;
        uih0[i] = ih0;
        uiw0[i] = iw0;
        uboff[i] = b * d.H * d.W;
    }
    int woff[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int n = n0 + lr + RPP * i;
        woff[i] = n < d.Cout ? (n * d.Kpad + lc) * 4 : -1;
    }

    ChunkPos pos;
    pos.kc = kt_begin;
    if (MODE == MODE_STEM) {
        pos.c0 = 0;
        pos.kh = kt_begin;
        pos.kw = 0;
    } else {
        const int kk = kt_begin * BKT;
        const int tap = kk / CinT;
        pos.c0 = kk - tap * CinT;
        pos.kh = tap / d.KW;
        pos.kw = tap - pos.kh * d.KW;
    }
    auto advance = [&]() {
        ++pos.kc;
        if (MODE == MODE_STEM) {
            ++pos.kh;
        } else {
            pos.c0 += BKT;
            if (pos.c0 >= CinT) {
                pos.c0 = 0;
                if (++pos.kw == d.KW) {
                    pos.kw = 0;
                    ++pos.kh;
                }
            }
        }
    };

    // PIPE == 1: two register staging sets -- the loads of chunk c are issued at iteration c-3 into set (c & 1), written to
    // LDS at iteration c-1 and multiplied at iteration c (two full iterations in flight).  PIPE == 0: one set, one
    // iteration in flight (fewer registers).
    f32x4 xsA[XR], wsA[WR], xsB[XR], wsB[WR];
    int ld_idx = 0;  // index (relative to kt_begin) of the next chunk to load
    // issue the global loads of the chunk at `pos` (all-zero past the last chunk: branch-free tail), then advance
    auto load_chunk = [&](f32x4 (&xs)[XR], f32x4 (&ws)[WR]) {
        const bool live = ld_idx < nk;
        ++ld_idx;
        const int tap = (MODE == MODE_STEM) ? pos.kh : pos.kh * d.KW + pos.kw;
        const bool second = (MODE != MODE_STEM) && pos.c0 >= d.Cin;
        const __amdgpu_buffer_rsrc_t r = second ? rx2 : rx;
        const unsigned livebit = (live && tap < 32) ? 1u : 0u;
        if (MODE == MODE_UPS) {
            const int ccb = (second ? pos.c0 - d.Cin : pos.c0) * 4 + lc * 4;
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const int ih = (uih0[i] + pos.kh) >> 1, iw = (uiw0[i] + pos.kw) >> 1;
                const int off = (uboff[i] + ih * d.W + iw) * cs4 + ccb;
                const bool ok = ((vmask[i] >> (tap & 31)) & livebit) != 0;
                xs[i] = buf_load4(r, ok ? off : -1, 0);
            }
        } else {
            const int tapoff = (MODE == MODE_STEM) ? pos.kh * d.W * 16
                                                   : (pos.kh * d.W + pos.kw) * cs4 + (second ? pos.c0 - d.Cin : pos.c0) * 4;
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const bool ok = ((vmask[i] >> (tap & 31)) & livebit) != 0;
                xs[i] = buf_load4(r, ok ? rowoff[i] + tapoff : -1, 0);
            }
        }
        const int soff = pos.kc * (BKT * 4);
#pragma unroll
        for (int i = 0; i < WR; ++i) ws[i] = buf_load4(rw, live ? woff[i] : -1, soff);
        advance();
    };
    auto store_chunk = [&](int buf, const f32x4 (&xs)[XR], const f32x4 (&ws)[WR]) {
        float *X = lds + buf * BUF;
        float *Wt = X + BM * LK;
#pragma unroll
        for (int i = 0; i < XR; ++i) *reinterpret_cast<f32x4 *>(X + (lr + RPP * i) * LK + lc) = xs[i];
#pragma unroll
        for (int i = 0; i < WR; ++i) *reinterpret_cast<f32x4 *>(Wt + (lr + RPP * i) * LK + lc) = ws[i];
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // ---- prologue: chunk 0 -> LDS[0]; chunk 1 (and, PIPE, chunk 2) -> registers ---------------------------
    a3d_stage_scale_shift(ss, d, n0, BN, tid);  // (visible to the epilogue through the barriers of the main loop)
    load_chunk(xsA, wsA);
    store_chunk(0, xsA, wsA);
    load_chunk(xsB, wsB);
    if (PIPE) load_chunk(xsA, wsA);
    __syncthreads();

    const int frag_off = (lane & 31) * LK + (lane >> 5) * 4;
    // one iteration: multiply LDS[cur]; write the staged next chunk to LDS[cur^1]; refill that register set
    auto iteration = [&](const int cur, f32x4 (&xs)[XR], f32x4 (&ws)[WR]) {
        const float *X = lds + cur * BUF + (wm * TM * 32) * LK + frag_off;
        const float *Wt = lds + cur * BUF + BM * LK + (wn * TN * 32) * LK + frag_off;
        f32x4 fa[2][TN], fb[2][TM];
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) fa[0][ni] = *reinterpret_cast<const f32x4 *>(Wt + ni * 32 * LK);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) fb[0][mi] = *reinterpret_cast<const f32x4 *>(X + mi * 32 * LK);
        store_chunk(cur ^ 1, xs, ws);  // harmless at the tail (zeros into a buffer nobody reads again)
#pragma unroll
        for (int q = 0; q < BKT / 8; ++q) {
            const int fc = q & 1, fn = fc ^ 1;
            if (q + 1 < BKT / 8) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) fa[fn][ni] = *reinterpret_cast<const f32x4 *>(Wt + ni * 32 * LK + (q + 1) * 8);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) fb[fn][mi] = *reinterpret_cast<const f32x4 *>(X + mi * 32 * LK + (q + 1) * 8);
            }
            if (q == 0) load_chunk(xs, ws);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[fc][ni][j], fb[fc][mi][j], acc[ni][mi], 0, 0, 0);
        }
        __syncthreads();
    };
    if (PIPE) {
        for (int it = 0; it < nk; it += 2) {
            iteration(0, xsB, wsB);                 // chunk it in LDS[0]; set B = chunk it+1, refilled with it+3
            if (it + 1 < nk) iteration(1, xsA, wsA);  // chunk it+1 in LDS[1]; set A = chunk it+2, refilled with it+4
        }
    } else {
        for (int it = 0; it < nk; ++it) iteration(it & 1, xsB, wsB);
    }

    // ---- epilogue: scale / shift from LDS, all residual quads of a row fetched before its first store ----------
    const bool has_res = d.res != nullptr && d.splitk == 1;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + (wm * TM + mi) * 32 + (lane & 31);
        if (m >= M) continue;
        size_t res_row;
        int b, oh, ow;
        out_rows(d, m, res_row, b, oh, ow);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            f32x4 rv[4];  // the residual quads of this 32-channel group, all in flight before its first store
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
                f32x4 v = {acc[ni][mi][rg * 4 + 0], acc[ni][mi][rg * 4 + 1], acc[ni][mi][rg * 4 + 2],
                           acc[ni][mi][rg * 4 + 3]};
                if (d.splitk > 1) {
                    *reinterpret_cast<f32x4 *>(d.workspace + ((size_t)z * Mmax + m) * d.Cout + n) = v;
                } else {
                    v = a3d_epilogue_math(d, v, *reinterpret_cast<const f32x4 *>(ss + nl), *reinterpret_cast<const f32x4 *>(ss + BN + nl),
                                          has_res, rv[rg]);
                    store_out(d, v, m, n, b, oh, ow);
                }
            }
        }
    }
}

template <int WAVES_M, int WAVES_N, int TM, int TN, int MODE, int PIPE = 1, int BKT = 32>
static void launch_v2(const a3d_conv_desc *d, hipStream_t s) {
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    const int M = d->B * d->Ho * d->Wo;
    const int mtiles = (M + BM - 1) / BM, ntiles = (d->Cout + BN - 1) / BN;
    const int nblk = mtiles * ntiles;
    const int kt_total = d->Kpad / BKT;
    const int kps = (kt_total + d->splitk - 1) / d->splitk;
    a3d_note_variant("conv_gemm_v2_kernel<%d,%d,%d,%d,%d,%d,%d> %dx%d%s", WAVES_M, WAVES_N, TM, TN, MODE, PIPE, BKT, BM, BN,
                     MODE == MODE_STEM ? " stem" : (MODE == MODE_UPS ? " ups" : (d->phase ? " ups-phase" : "")));
    hipLaunchKernelGGL((conv_gemm_v2_kernel<WAVES_M, WAVES_N, TM, TN, MODE, PIPE, BKT>), dim3(nblk, d->splitk), dim3(WAVES_M * WAVES_N * 64), 0,
                       s, *d, M, ntiles, nblk, kt_total, kps);
}

__global__ __launch_bounds__(256) void conv_splitk_reduce_v2_kernel(const a3d_conv_desc d, const int Mmax) {
    const int M = d.m_dev ? min(Mmax, *d.m_dev) : Mmax;
    const int n4 = d.Cout >> 2;
    const size_t total = (size_t)M * n4;
    const size_t span = (size_t)gridDim.x * blockDim.x;
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x; i0 < total; i0 += span) {  // (whole waves iterate: a3d_note_amax is wave-wide)
        const size_t i = i0 + threadIdx.x;
        const bool live = i < total;
        float amax_v = 0.f;
        int amax_b = 0;
        if (live) {
        const int m = (int)(i / n4);
        const int n = (int)(i - (size_t)m * n4) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        // (eight slices requested before the first is added: the head FCs' reduce -- 276 rows x 32 slices -- is a few waves per CU, and one
        // load per add left it at the latency of 32 dependent-looking round trips, 74 us for 36 MB.  Slices are added in z order as before.)
        const float *wq = d.workspace + (size_t)m * d.Cout + n;
        const size_t zs = (size_t)Mmax * d.Cout;
        int z = 0;
        for (; z + 8 <= d.splitk; z += 8) {
            f32x4 t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = *reinterpret_cast<const f32x4 *>(wq + (size_t)(z + q) * zs);
#pragma unroll
            for (int q = 0; q < 8; ++q) v += t[q];
        }
        for (; z < d.splitk; ++z) v += *reinterpret_cast<const f32x4 *>(wq + (size_t)z * zs);
        size_t res_row;
        int b, oh, ow;
        out_rows(d, m, res_row, b, oh, ow);
        v = apply_epilogue(d, v, n, res_row);
        store_out(d, v, m, n, b, oh, ow);
        if (d.y_amax) amax_v = a3d_absmax4(v), amax_b = m / (d.Ho * d.Wo);
        }
        // (a wave's 64 quads are consecutive columns of one row unless Cout < 256: a3d_note_amax handles both)
        if (d.y_amax) a3d_note_amax(d.y_amax, amax_b, amax_v, live);
    }
}

// workspace [splitk][M][Cout] partial sums -> y: slices added in z order, then the fused epilogue (shared with conv_bf16x3_wide.hip)
void a3d_launch_splitk_reduce(const a3d_conv_desc *d, int M, hipStream_t s) {
    const size_t total = (size_t)M * (d->Cout >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(conv_splitk_reduce_v2_kernel, dim3(blocks), dim3(256), 0, s, *d, M);
}

int a3d_conv_launch_v2(const a3d_conv_desc *d, hipStream_t s) {
    const size_t lim = (size_t)1 << 32;
    const size_t cs = d->stem ? 4 : (size_t)d->Cin;
    if ((size_t)d->B * d->H * d->W * cs * 4 >= lim || (size_t)d->Cout * d->Kpad * 4 >= lim) return A3D_ERR_UNSUPPORTED;
    if (d->phase && (d->stem || d->ups || d->KH != 2 || d->KW != 2 || d->stride != 1 || d->splitk != 1 || d->res || d->pixshuf))
        return A3D_ERR_ARG;
    if (!d->stem) {
        if (d->Cin2 && d->Cin2 != d->Cin) return A3D_ERR_UNSUPPORTED;
        if (d->KH * d->KW > 32) return A3D_ERR_UNSUPPORTED;
    }
    // ---- variant selection ---------------------------------------------------------------------------
    // Tile shape never changes the summation order of an output element (k runs in chunk order inside one
    // accumulator), so it may depend on the row count; measured on MI355X (tools/conv_bench.py, one process):
    //   * BK=16 (LDS 40 KiB / 30 KiB per workgroup -> 3-5 workgroups per CU) beats BK=32 (2 per CU) by 3-25 %
    //     everywhere except the very deep 1x1 GEMMs (box-head fc1, K=12544), where fewer barriers win;
    //   * 128x64 tiles beat 128x128 whenever the 128x128 grid would be under ~4 rounds of the chip (res4/res5,
    //     p4-p6 levels, per-ROI heads) or Cout <= 64.
    const int M = d->B * d->Ho * d->Wo;
    const long n128 = (long)((M + 127) / 128) * ((d->Cout + 127) / 128);
    int cfg = d->Cout <= 32 ? 2 : ((d->Cout <= 64 || n128 <= 1000) ? 1 : 0);  // 0: 128x128, 1: 128x64, 2: 128x32
    int bk16 = !(d->KH * d->KW == 1 && d->Kpad >= 8192);
    int pipe = 0;  // measured: the second staging set costs a wave per SIMD (146 vs 90 VGPRs) and loses 3-8 % on every shape
    if (d->tune >= 100 && d->tune < 200) {  // explicit variant for A/B measurements: 100 + 10*cfg + (bk16 ? 1 : 0) + (single-set ? 5 : 0)
        cfg = (d->tune - 100) / 10;
        bk16 = (d->tune - 100) % 5;
        pipe = ((d->tune - 100) % 10) >= 5 ? 0 : 1;
        if (cfg > 2 || bk16 > 1) return A3D_ERR_ARG;
    }
    if (d->stem) {
        if (d->tune == 3) launch_v2<4, 1, 2, 2, MODE_STEM, 0, 32>(d, s);  // 256x64 tile, 1 workgroup / CU (A/B)
        else launch_v2<2, 2, 2, 1, MODE_STEM, 0, 32>(d, s);              // 128x64 tile, 2 workgroups / CU
    } else if (d->ups) {
        if (cfg == 0) launch_v2<2, 2, 2, 2, MODE_UPS, 0, 16>(d, s);
        else if (cfg == 1) launch_v2<2, 2, 2, 1, MODE_UPS, 0, 16>(d, s);
        else launch_v2<4, 1, 1, 1, MODE_UPS, 0, 32>(d, s);
    } else {
        switch ((cfg * 2 + bk16) * 2 + pipe) {
            case 0: launch_v2<2, 2, 2, 2, MODE_GENERIC, 0, 32>(d, s); break;
            case 1: launch_v2<2, 2, 2, 2, MODE_GENERIC, 1, 32>(d, s); break;
            case 2: launch_v2<2, 2, 2, 2, MODE_GENERIC, 0, 16>(d, s); break;
            case 3: launch_v2<2, 2, 2, 2, MODE_GENERIC, 1, 16>(d, s); break;
            case 4: launch_v2<2, 2, 2, 1, MODE_GENERIC, 0, 32>(d, s); break;
            case 5: launch_v2<2, 2, 2, 1, MODE_GENERIC, 1, 32>(d, s); break;
            case 6: launch_v2<2, 2, 2, 1, MODE_GENERIC, 0, 16>(d, s); break;
            case 7: launch_v2<2, 2, 2, 1, MODE_GENERIC, 1, 16>(d, s); break;
            default: launch_v2<4, 1, 1, 1, MODE_GENERIC, 0, 32>(d, s); break;  // 128x32 tile: BK=32 only
        }
    }
    if (d->splitk > 1) a3d_launch_splitk_reduce(d, M, s);
    return a3d_check_launch();
}
