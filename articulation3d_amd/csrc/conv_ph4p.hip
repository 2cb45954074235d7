// The decoder's upsampled 3x3 convolutions (all four output phases in one launch, a3d_conv_desc.phase == 5) with the INPUT PATCH RESIDENT in
// LDS -- round 4.  Replaces, on maps whose size fits its tiles, the tap-outer loop of conv_bf16x3_wide.hip's PH4 form for the layers
// pkg/modeling/depth_net/depth_head.py:40-46,58-68 (deconv2d: nearest x2 upsampling + 3x3 conv + BN + ReLU).
//
// Why.  The tap-outer form walks the nine taps of the 3x3 source neighbourhood and, per tap, streams the workgroup's 256 pixels x all
// channels through LDS again: every input element is loaded and SPLIT into its two fp16 planes nine times, and a loaded chunk feeds
// 10.7 MFMAs per wave on average (a corner tap multiplies one phase block, an edge two, the centre four).  The matrix pipe is 0.36
// busy in that loop; the loader is what it waits for.  Here the loop is chunk-outer: per 16-channel chunk the (8 + 2) x (32 + 2) pixel
// patch of the workgroup's 8 x 32 tile is loaded, split and written to LDS ONCE (1.33 x the tile instead of 9 x), and the nine taps
// multiply it as shifted views -- 96 MFMAs per wave between two patch loads.  The pre-split filter streams through a 6-stage ring by
// LDS-DMA, three taps per barrier.
//
// Arithmetic: the fp16x2 products and per-output term order (h.h, h.l, l.h) of the other split-operand kernels; the REDUCTION order over
// k is (chunk, tap) here and (tap, chunk) in the tap-outer form, so the two agree to fp32 rounding, not bit for bit (the dispatcher's
// choice between them is a function of the layer and the map size only, never of the batch: tests/test_gpu_presplit.py).
//
// Geometry: 512 threads = 8 waves as 4 (tile-row pairs) x 2 (column halves); a wave multiplies 2 rows x 32 pixels against 128 columns
// = the four phases of 32 output channels (column 128 g + 32 phase + c, the layout of ops.pack_conv_ups_fused).  A 32-pixel MFMA block is
// one contiguous tile row, so its fragment rows are 32 consecutive patch positions for every tap: the 32-byte-row image with the half
// swizzle by (position >> 3) & 1 is conflict-free at any shift.
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {
typedef __bf16 pp_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pp_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 pp_h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 pp_h16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pp_rsrc(const void *p, unsigned bytes) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ f32x4 pp_load4(__amdgpu_buffer_rsrc_t r, int voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ void pp_dma16(__amdgpu_buffer_rsrc_t r, __bf16 *lds_dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds_dst, 16, voff, soff, 0, 0);
}

// Tile shapes: 8 x 32 (a 32-pixel MFMA block = one tile row: its fragment rows are 32 consecutive patch positions at every tap, the
// conflict-free case) or 16 x 16 (a block = two tile rows: 2-way bank conflicts on some taps' activation reads, but 64 x 80 instead of
// 64 x 96 pixels cover a 60 x 80 map).  The patch buffers are sized for the larger patch (10 x 34 = 340 positions; 18 x 18 = 324).
constexpr int PP_NPOS = 340;
constexpr int PP_ITEMS = PP_NPOS * 4;                       // loader items: (position, channel quad)
constexpr int PP_LI = (PP_ITEMS + 511) / 512;               // items per thread (3)
constexpr int PP_XPL = PP_NPOS * 16, PP_XBUF = 2 * PP_XPL;  // 16-bit elements: one plane / one buffer (h | l) of the patch
constexpr int PP_NWS = 6;                                   // ring stages = two groups of three taps
// one plane / one stage of the filter ring: BN = 64 NB columns x 16 k (16 KiB per stage at 256 columns, 4 KiB at 64: the 64-column
// form then fits two workgroups per CU)
constexpr int pp_wpl(int NB) { return 64 * NB * 16; }
constexpr int pp_lds_bytes(int NB) { return (2 * PP_XBUF + PP_NWS * 2 * pp_wpl(NB)) * 2 + 2 * 256 * 4; }

// the phases (bit 2 dy + dx) whose 2x2 window inside the 3x3 neighbourhood holds tap kh = tap / 3, kw = tap % 3
constexpr unsigned pp_tap_phases(const int tap) {
    const int th = tap / 3, tw = tap - 3 * th;
    const unsigned rows = th == 0 ? 0x3u : (th == 1 ? 0xFu : 0xCu);
    const unsigned cols = tw == 0 ? 0x5u : (tw == 1 ? 0xFu : 0xAu);
    return rows & cols;
}

// live 32-column blocks of a wave at a tap: the phases whose window holds the tap (PH), or all NB blocks (a plain 3x3 convolution)
template <bool PH, int NB>
constexpr unsigned pp_act(const int tap) {
    return PH ? pp_tap_phases(tap) : ((1u << NB) - 1u);
}
// filter-fragment register set a tap starts from: A[0] holds the first block of a group's first tap and the sets alternate block by block
template <bool PH, int NB>
constexpr int pp_start(const int tap) {
    int s = 0;
    for (int t = (tap / 3) * 3; t < tap; ++t) s += __builtin_popcount(pp_act<PH, NB>(t));
    return s & 1;
}

// PH = true, NB = 4: the fused four-phase upsampled convolution (a3d_conv_desc.phase == 5).  PH = false: a plain 3x3 stride-1 pad-1
// convolution of the fp16x2 arithmetic with NB x 64 output channels per workgroup (the layers the Winograd form does not take: fewer than
// 256 input channels -- res2 / res3 conv2, pkg/modeling/meta_arch/planercnn.py:150 -> detectron2 BottleneckBlock), same loop, every block
// live at every tap.
template <bool PH, int NB, int TW, bool DOT = false>
__global__ __launch_bounds__(512, NB == 1 ? 2 : 1) void conv_ph4p_kernel(const a3d_conv_desc d, const int tiles_x, const int tiles_y, const int ntiles, const int nblk) {
    static_assert(!PH || NB == 4, "the four phases are the four blocks of a wave");
    static_assert(!DOT || PH, "the tap-product epilogue belongs to the four-phase form");
    constexpr int PP_TW = TW, PP_TH = 256 / TW, PP_PC = PP_TW + 2;
    constexpr int NPOS_T = PP_PC * (PP_TH + 2), ITEMS_T = NPOS_T * 4;  // positions / loader items of this tile shape
    static_assert(TW == 32 || TW == 16, "8 x 32 or 16 x 16 tiles");
    constexpr int WCOLS = NB * 32, BN = 2 * WCOLS;  // columns per wave / per workgroup
    constexpr int PP_WPL = pp_wpl(NB), PP_WST = 2 * PP_WPL;
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    __bf16 *const Xs = lds;                                   // [2 buffers][h | l][340 positions][16 k]
    __bf16 *const Ws = lds + 2 * PP_XBUF;                     // [6 stages][h | l][256 columns][16 k]
    float *const ss = reinterpret_cast<float *>(lds + 2 * PP_XBUF + PP_NWS * PP_WST);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int logical = a3d_xcd_remap(blockIdx.x, nblk);
    const int mt = logical / ntiles, nt = logical - mt * ntiles;
    const int tpi = tiles_x * tiles_y;
    const int b = mt / tpi, tr = mt - b * tpi;
    const int ty0 = (tr / tiles_x) * PP_TH, tx0 = (tr % tiles_x) * PP_TW;
    const int n0 = nt * BN;
    const int CinT = d.Cin + d.Cin2, cs4 = d.Cin * 4, nchunks = CinT >> 4;
    const __amdgpu_buffer_rsrc_t rx = pp_rsrc(d.x, (unsigned)((size_t)d.B * d.H * d.W * (size_t)cs4));
    const __amdgpu_buffer_rsrc_t rx2 = pp_rsrc(d.x2 ? d.x2 : d.x, (unsigned)((size_t)d.B * d.H * d.W * (size_t)cs4));
    const unsigned w3chunk = (unsigned)d.Cout * 64u;  // bytes of one 16-deep chunk of w_x3: 2 planes x Cout rows x 32 B
    const __amdgpu_buffer_rsrc_t rw = pp_rsrc(d.w_x3, (unsigned)((size_t)(d.Kpad >> 4) * w3chunk));
    const float sx = a3d_in_scale(d, b), sw = d.w_scale;

    // ---- patch loader: item j = (position j / 4, channel quad j % 4) of the 16-channel chunk; pixels outside the image read zeros
    int pixoff[PP_LI], ldsoff[PP_LI];
    unsigned inmask = 0;  // bit i: item i reads a pixel of the image (byte offsets may exceed 2^31: validity is not their sign)
#pragma unroll
    for (int i = 0; i < PP_LI; ++i) {
        const int j = tid + 512 * i;
        const int pos = min(j >> 2, NPOS_T - 1), q = j & 3;
        const int py = pos / PP_PC, px = pos - py * PP_PC;
        const int y = ty0 - 1 + py, x = tx0 - 1 + px;
        const bool inb = j < ITEMS_T && (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W;
        pixoff[i] = inb ? ((b * d.H + y) * d.W + x) * cs4 + q * 16 : -1;
        inmask |= inb ? (1u << i) : 0u;
        ldsoff[i] = j < ITEMS_T ? pos * 16 + ((((q >> 1) ^ (pos >> 3)) & 1) << 3) + (q & 1) * 4 : -1;
    }
    f32x4 xs[PP_LI];
    auto load_patch = [&](const int c) {  // chunk c of (source 0 || source 1); past the last chunk: zeros (the count of loads stays uniform)
        const int c0 = c << 4;
        const bool live = c < nchunks, second = c0 >= d.Cin;
        const __amdgpu_buffer_rsrc_t r = second ? rx2 : rx;
        const int coff = __builtin_amdgcn_readfirstlane((second ? c0 - d.Cin : c0) * 4);
#pragma unroll
        for (int i = 0; i < PP_LI; ++i) xs[i] = pp_load4(r, (live && ((inmask >> i) & 1u)) ? pixoff[i] + coff : -1);
    };
    auto store_patch = [&](const int buf) {
#pragma unroll
        for (int i = 0; i < PP_LI; ++i) {
            if (ldsoff[i] < 0) continue;
            const f32x4 v = xs[i] * sx;
            const pp_h16x4 h = __builtin_convertvector(v, pp_h16x4);
            const pp_h16x4 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), pp_h16x4);
            __bf16 *p = Xs + buf * PP_XBUF + ldsoff[i];
            *reinterpret_cast<pp_h16x4 *>(p) = h;
            *reinterpret_cast<pp_h16x4 *>(p + PP_XPL) = l;
        }
    };
    // ---- filter ring: w_x3 [Kpad / 16][h | l][Cout][16], k = (tap, c): chunk index tap * nchunks + c.  One (chunk, plane) tile of the
    // workgroup's 256 columns = 8 DMA instructions of 32 rows; wave w issues j = 2 w, 2 w + 1 (plane j / 8, row group j % 8).
    const int wvoff = (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4);
    auto dma_group = [&](const int c, const int g) {  // taps 3 g .. 3 g + 2 of chunk c into ring half (3 c + g) % 2
        const int half = (3 * c + g) & 1;
        const int cc = min(c, nchunks - 1);  // (past the last chunk: a harmless re-fetch, never read)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            __bf16 *Wt = Ws + (half * 3 + t) * PP_WST;
            const int base = __builtin_amdgcn_readfirstlane(((3 * g + t) * nchunks + cc) * (int)w3chunk + n0 * 32);
#pragma unroll
            for (int i = 0; i < 2; ++i) {  // 2 BN / 32 pieces of 32 rows: piece j = plane j / (BN / 32), row group j % (BN / 32)
                const int j = wave + 8 * i;
                const int p = j / (BN / 32), gg = j % (BN / 32);
                if (j < 2 * (BN / 32))  // (narrow tiles: fewer pieces than waves; a wave without one waits conservatively)
                    pp_dma16(rw, Wt + p * PP_WPL + gg * 32 * 16, wvoff, base + __builtin_amdgcn_readfirstlane(p * d.Cout * 32 + gg * 1024));
            }
        }
    };

    f32x16 acc[NB][2];
#pragma unroll
    for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][m][r] = 0.f;
    a3d_stage_scale_shift(ss, d, n0, BN, tid);

    // ---- fragments
    const int frow = lane & 31, khalf = lane >> 5;
    const __bf16 *const fW = Ws + (wn * WCOLS + frow) * 16 + (((khalf ^ (frow >> 3)) & 1) << 3);
    // the wave's two 32-pixel blocks: tile rows 2 wm, 2 wm + 1 (8 x 32) or row pairs 4 wm .. 4 wm + 3 (16 x 16: lane = 16 row + column)
    const int prow = TW == 32 ? 0 : frow >> 4, pcol = TW == 32 ? frow : frow & 15;
    constexpr int RPB = TW == 32 ? 1 : 2;  // tile rows per block
    const int posb0 = (2 * RPB * wm + prow) * PP_PC + pcol, posb1 = posb0 + RPB * PP_PC;
    struct FragA {
        pp_bf16x8 p[2];
    };
    struct FragB {
        pp_bf16x8 p[2][2];  // [plane][tile row of the wave]
    };
    auto rdA = [&](FragA &A, const int stage, const int n) {
#pragma unroll
        for (int p = 0; p < 2; ++p) A.p[p] = *reinterpret_cast<const pp_bf16x8 *>(fW + stage * PP_WST + p * PP_WPL + n * 32 * 16);
    };
    auto rdB = [&](FragB &Bf, const int buf, const int toff) {
        const int p0 = posb0 + toff, p1 = posb1 + toff;
        const int o0 = p0 * 16 + (((khalf ^ (p0 >> 3)) & 1) << 3), o1 = p1 * 16 + (((khalf ^ (p1 >> 3)) & 1) << 3);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            Bf.p[p][0] = *reinterpret_cast<const pp_bf16x8 *>(Xs + buf * PP_XBUF + p * PP_XPL + o0);
            Bf.p[p][1] = *reinterpret_cast<const pp_bf16x8 *>(Xs + buf * PP_XBUF + p * PP_XPL + o1);
        }
    };
#define PP_MFMA(C, A, Bv) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pp_h16x8, A), __builtin_bit_cast(pp_h16x8, Bv), C, 0, 0, 0);
#define PP_BLOCK(N, A, Bf)                   \
    PP_MFMA(acc[N][0], A.p[0], Bf.p[0][0]) \
    PP_MFMA(acc[N][1], A.p[0], Bf.p[0][1]) \
    PP_MFMA(acc[N][0], A.p[0], Bf.p[1][0]) \
    PP_MFMA(acc[N][1], A.p[0], Bf.p[1][1]) \
    PP_MFMA(acc[N][0], A.p[1], Bf.p[0][0]) \
    PP_MFMA(acc[N][1], A.p[1], Bf.p[0][1])

    FragA A[2];
    FragB Bf[2];
    // One tap: its live phase blocks (compile-time), the filter fragments ping-pong between A[S] and A[S ^ 1]; under the first block the
    // NEXT tap's activation fragments are read (the patch of the chunk -- or, behind the last tap, of the next chunk -- is complete in
    // LDS), under the last block the next tap's first filter block, unless that tap belongs to the next group (its stage may not have
    // landed: the group's barrier comes first).  Returns the register set the next tap starts from.
    auto tap = [&](auto tap_c, auto last_c, auto bi_c, const int stage, const int buf_next, const int stage_next) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tap_c)::value, BI = decltype(bi_c)::value;
        constexpr int S = pp_start<PH, NB>(TAP);
        constexpr bool LAST = decltype(last_c)::value;  // last tap of its group
        constexpr unsigned ACT = pp_act<PH, NB>(TAP);
        constexpr int NTAP = (TAP + 1) % 9;
        constexpr int ntoff = (NTAP / 3) * PP_PC + (NTAP % 3);
        constexpr int nfirst = __builtin_ctz(pp_act<PH, NB>(NTAP));
        constexpr int b0 = __builtin_ctz(ACT);
        constexpr unsigned R1 = ACT & (ACT - 1), R2 = R1 & (R1 - 1), R3 = R2 & (R2 - 1);
        // block 0 of the tap (+ the next tap's activation fragments)
        if constexpr (R1 != 0) rdA(A[(S + 1) & 1], stage, __builtin_ctz(R1 ? R1 : 1u));
        else if constexpr (!LAST) rdA(A[(S + 1) & 1], stage_next, nfirst);
        rdB(Bf[BI ^ 1], buf_next, ntoff);
        PP_BLOCK(b0, A[S & 1], Bf[BI])
        if constexpr (R1 != 0) {
            constexpr int b1 = __builtin_ctz(R1 ? R1 : 1u);
            if constexpr (R2 != 0) rdA(A[(S + 2) & 1], stage, __builtin_ctz(R2 ? R2 : 1u));
            else if constexpr (!LAST) rdA(A[(S + 2) & 1], stage_next, nfirst);
            PP_BLOCK(b1, A[(S + 1) & 1], Bf[BI])
        }
        if constexpr (R2 != 0) {
            constexpr int b2 = __builtin_ctz(R2 ? R2 : 1u);
            if constexpr (R3 != 0) rdA(A[(S + 3) & 1], stage, __builtin_ctz(R3 ? R3 : 1u));
            else if constexpr (!LAST) rdA(A[(S + 3) & 1], stage_next, nfirst);
            PP_BLOCK(b2, A[(S + 2) & 1], Bf[BI])
        }
        if constexpr (R3 != 0) {
            constexpr int b3 = __builtin_ctz(R3 ? R3 : 1u);
            if constexpr (!LAST) rdA(A[(S + 4) & 1], stage_next, nfirst);
            PP_BLOCK(b3, A[(S + 3) & 1], Bf[BI])
        }
    };

    // ---- prologue: patch of chunk 0, filter group 0
    load_patch(0);
    dma_group(0, 0);
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_patch(0);
    __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    rdB(Bf[0], 0, 0);

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using TF = std::false_type;
    using TT = std::true_type;
#define PP_T(N) std::integral_constant<int, N>{}
    // One chunk; PAR = its parity (a compile-time property: the patch buffer, the ring halves and -- nine taps being an odd number --
    // which activation fragment set tap 0 finds its fragments in).  An even number of chunks per layer (32 | Cin + Cin2).
    auto chunk = [&](auto par_c, const int c) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr int buf = PAR, h0 = PAR, h1 = PAR ^ 1;  // ring halves of groups 0 / 1 / 2: h0, h1, h0  ((3 c + g) & 1)
        using E = std::integral_constant<int, PAR>;       // fragment set of the even taps
        using O = std::integral_constant<int, PAR ^ 1>;   // ... of the odd taps
        // ---- group 0: taps 0, 1, 2 (four-phase form: one, two, one live blocks)
        dma_group(c, 1);
        load_patch(c + 1);
        rdA(A[0], h0 * 3 + 0, __builtin_ctz(pp_act<PH, NB>(0)));
        tap(PP_T(0), TF{}, E{}, h0 * 3 + 0, buf, h0 * 3 + 1);
        tap(PP_T(1), TF{}, O{}, h0 * 3 + 1, buf, h0 * 3 + 2);
        tap(PP_T(2), TT{}, E{}, h0 * 3 + 2, buf, 0);
        __asm__ volatile("s_waitcnt vmcnt(%0)" ::"n"(PP_LI) : "memory");  // filter group 1 has landed (younger: the patch loads)
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- group 1: taps 3, 4, 5 (two, four, two)
        dma_group(c, 2);
        rdA(A[0], h1 * 3 + 0, __builtin_ctz(pp_act<PH, NB>(3)));
        tap(PP_T(3), TF{}, O{}, h1 * 3 + 0, buf, h1 * 3 + 1);
        tap(PP_T(4), TF{}, E{}, h1 * 3 + 1, buf, h1 * 3 + 2);
        tap(PP_T(5), TT{}, O{}, h1 * 3 + 2, buf, 0);
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");  // filter group 2 and the next chunk's patch loads have landed
        store_patch(buf ^ 1);
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- group 2: taps 6, 7, 8 (one, two, one)
        dma_group(c + 1, 0);
        rdA(A[0], h0 * 3 + 0, __builtin_ctz(pp_act<PH, NB>(6)));
        tap(PP_T(6), TF{}, E{}, h0 * 3 + 0, buf, h0 * 3 + 1);
        tap(PP_T(7), TF{}, O{}, h0 * 3 + 1, buf, h0 * 3 + 2);
        tap(PP_T(8), TT{}, E{}, h0 * 3 + 2, buf ^ 1, 0);  // (its activation prefetch = tap 0 of the next chunk, from the next patch)
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    for (int c = 0; c < nchunks; c += 2) {
        chunk(I0{}, c);
        chunk(I1{}, c + 1);
    }
#undef PP_T
#undef PP_BLOCK
#undef PP_MFMA

    // ---- epilogue.  Four-phase form (conv_bf16x3_wide.hip's PH4 stores): column 128 g + 32 phase + c -> pixel (2 oh + dy, 2 ow + dx),
    // channel 32 g + c.  Plain form: pixel (oh, ow), channel = column.
    const float unx = 1.f / sx, unw = 1.f / sw;
    float vmax = 0.f;
    const int co_n = d.Cout >> 2;
    if constexpr (PH && DOT) {
        {
            // ---- the output feeds only a 3x3 convolution to one channel (a3d_conv_desc.dot_w): instead of the 64-channel pixels, their nine
            // tap products.  A lane holds 16 of a pixel-phase's 64 channels (rg, i; its half-wave partner 16 more, the wave with the
            // other wn the other 32): nine 16-term partial sums, + the partner's by a 32-lane shuffle, then both waves' halves meet in
            // LDS (the filter ring is idle: every wave is behind the loop's last barrier) and 512 threads store the sums.
            float *const dw = reinterpret_cast<float *>(Ws);              // [9][64] filter
            float *const part = dw + 9 * 64;                              // [wn][wm][mi][ni][9][32]
            for (int i = tid; i < 9 * 64; i += 512) dw[i] = d.dot_w[i];
            __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // (1) the shared epilogue in place: the accumulators become the output values
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NB; ++ni)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int nl = (wn * NB + ni) * 32 + rg * 8 + khalf * 4;
                        f32x4 v = {acc[ni][mi][rg * 4 + 0], acc[ni][mi][rg * 4 + 1], acc[ni][mi][rg * 4 + 2], acc[ni][mi][rg * 4 + 3]};
                        v = (v * unx) * unw;  // exact: powers of two
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        v = a3d_epilogue_math(d, v, *reinterpret_cast<const f32x4 *>(ss + nl), *reinterpret_cast<const f32x4 *>(ss + BN + nl), false, zero);
                        vmax = fmaxf(vmax, a3d_absmax4(v));
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[ni][mi][rg * 4 + i] = v[i];
                    }
            // (2) tap by tap (a ROLLED loop: with all nine taps' weights live the kernel spilled ~300 registers): the tap's 16 weights of
            // this lane's channels, the eight pixel-phases' 16-term sums, the half-wave partner's added
#pragma unroll 1
            for (int t = 0; t < 9; ++t) {
                f32x4 wq[4];
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) wq[rg] = *reinterpret_cast<const f32x4 *>(dw + t * 64 + wn * 32 + rg * 8 + khalf * 4);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NB; ++ni) {
                        float ps = 0.f;
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
                            for (int i = 0; i < 4; ++i) ps = __builtin_fmaf(acc[ni][mi][rg * 4 + i], wq[rg][i], ps);
                        const float s2 = ps + __shfl_xor(ps, 32, 64);
                        if (khalf == 0) part[((((wn * 4 + wm) * 2 + mi) * NB + ni) * 9 + t) * 32 + frow] = s2;
                    }
            }
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const size_t plane = (size_t)(2 * d.H) * (2 * d.W);
            for (int idx = tid; idx < 4 * 2 * NB * 9 * 32; idx += 512) {  // (wm, mi, ni, t, lane)
                const int fr = idx & 31, t = (idx >> 5) % 9, q = (idx >> 5) / 9;
                const int ni = q & 3, mi = (q >> 2) & 1, wmq = q >> 3;
                const int pr = TW == 32 ? 0 : fr >> 4, pc = TW == 32 ? fr : fr & 15;
                const int oh = ty0 + RPB * (2 * wmq + mi) + pr, ow = tx0 + pc;
                if (oh >= d.H || ow >= d.W) continue;
                const float s2 = part[idx] + part[idx + 4 * 2 * NB * 9 * 32];  // wn = 0 | 1: channels 0-31 | 32-63
                d.dot_y[((size_t)b * 9 + t) * plane + (size_t)(2 * oh + (ni >> 1)) * (2 * d.W) + (2 * ow + (ni & 1))] = s2;
            }
            if (d.y_amax) a3d_note_amax(d.y_amax, b, vmax, true);
            return;
        }
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int oh = ty0 + RPB * (2 * wm + mi) + prow, ow = tx0 + pcol;
        if (oh >= d.H || ow >= d.W) continue;
#pragma unroll
        for (int ni = 0; ni < NB; ++ni) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int nl = (wn * NB + ni) * 32 + rg * 8 + khalf * 4;
                const int n = n0 + nl;
                if (n >= d.Cout) continue;
                f32x4 v = {acc[ni][mi][rg * 4 + 0], acc[ni][mi][rg * 4 + 1], acc[ni][mi][rg * 4 + 2], acc[ni][mi][rg * 4 + 3]};
                v = (v * unx) * unw;  // exact: powers of two
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                v = a3d_epilogue_math(d, v, *reinterpret_cast<const f32x4 *>(ss + nl), *reinterpret_cast<const f32x4 *>(ss + BN + nl), false, zero);
                vmax = fmaxf(vmax, a3d_absmax4(v));
                if constexpr (PH) {
                    const int co = (n >> 7) * 32 + (n & 31);
                    const size_t row = ((size_t)b * (2 * d.H) + (2 * oh + (ni >> 1))) * (size_t)(2 * d.W) + (2 * ow + (ni & 1));
                    *reinterpret_cast<f32x4 *>(d.y + row * co_n + co) = v;
                } else {
                    const size_t row = ((size_t)b * d.H + oh) * (size_t)d.W + ow;
                    *reinterpret_cast<f32x4 *>(d.y + row * d.Cout + n) = v;
                }
            }
        }
    }
    if (d.y_amax) a3d_note_amax(d.y_amax, b, vmax, true);  // (every lane of the wave gets here; one image per workgroup)
}
}  // namespace

// A3D_ERR_UNSUPPORTED: tune 15 (the tap-outer form of conv_bf16x3_wide.hip: A/B runs and the bit-equality test against the four-launch
// form) or a tensor past the 32-bit offsets.  tune 16 = this kernel (the default anyway).
// tile shape by the map's size alone: the one that covers fewer pixels (ties: 8 x 32, whose fragment reads are conflict-free)
static inline long pp_covered(int H, int W, int TW) {
    const int TH = 256 / TW;
    return (long)((W + TW - 1) / TW) * TW * ((H + TH - 1) / TH) * TH;
}
static inline int pp_tile_width(int H, int W) {
    const int force = (int)a3d_dev_knob("A3D_PP_TW", 0);  // (developer builds, A/B runs: 16 | 32)
    if (force == 16 || force == 32) return force;
    return pp_covered(H, W, 16) < pp_covered(H, W, 32) ? 16 : 32;
}

template <bool PH, int NB, int TW, bool DOT = false>
static int pp_launch_t(const a3d_conv_desc *d, hipStream_t s, const char *label) {
    constexpr int TH = 256 / TW;
    const int tiles_x = (d->W + TW - 1) / TW, tiles_y = (d->H + TH - 1) / TH;
    const int ntiles = (d->Cout + 64 * NB - 1) / (64 * NB);
    const int nblk = d->B * tiles_x * tiles_y * ntiles;
    static a3d_attr_once attr;
    if (attr.needed()) {
        if (hipFuncSetAttribute((const void *)conv_ph4p_kernel<PH, NB, TW, DOT>, hipFuncAttributeMaxDynamicSharedMemorySize, pp_lds_bytes(NB)) != hipSuccess) return A3D_ERR_LAUNCH;
        attr.mark();
    }
    a3d_note_variant(DOT ? "%s%d> dot" : "%s%d>", label, TW);  // ("conv_ph4p_kernel<16>", "conv_c3p_kernel<2, 32>": the tile width is part of the name rocprofv3 shows)
    hipLaunchKernelGGL((conv_ph4p_kernel<PH, NB, TW, DOT>), dim3(nblk), dim3(512), pp_lds_bytes(NB), s, *d, tiles_x, tiles_y, ntiles, nblk);
    return a3d_check_launch();
}
template <bool PH, int NB>
static int pp_launch(const a3d_conv_desc *d, hipStream_t s, const char *label) {
    if constexpr (PH) {
        if (d->dot_w) return pp_tile_width(d->H, d->W) == 16 ? pp_launch_t<PH, NB, 16, true>(d, s, label) : pp_launch_t<PH, NB, 32, true>(d, s, label);
    }
    return pp_tile_width(d->H, d->W) == 16 ? pp_launch_t<PH, NB, 16>(d, s, label) : pp_launch_t<PH, NB, 32>(d, s, label);
}

int a3d_conv_launch_ph4p(const a3d_conv_desc *d, hipStream_t s) {
    if (d->phase != 5 || d->precision != 3 || !d->w_x3 || !d->in_amax || !(d->w_scale > 0.f)) return A3D_ERR_ARG;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->Ho != d->H || d->Wo != d->W) return A3D_ERR_ARG;
    if (d->stem || d->ups || d->m_dev || d->splitk != 1 || d->res || d->gate || d->pixshuf) return A3D_ERR_ARG;
    if (d->Cin2 && (d->Cin2 != d->Cin || !d->x2)) return A3D_ERR_ARG;
    const int CinT = d->Cin + d->Cin2;
    if ((d->Cin & 15) || (CinT & 31) || d->Kpad != 9 * CinT || (d->Cout & 127)) return A3D_ERR_ARG;
    if ((d->dot_w != nullptr) != (d->dot_y != nullptr) || (d->dot_w && d->Cout != 256)) return A3D_ERR_ARG;
    if (d->tune == 15) return d->dot_w ? A3D_ERR_ARG : A3D_ERR_UNSUPPORTED;  // (the tap-outer form has no tap-product epilogue)
    if ((size_t)d->B * d->H * d->W * d->Cin * 4 >= ((size_t)1 << 32) || (size_t)d->Cout * d->Kpad * 4 >= ((size_t)1 << 31)) return d->dot_w ? A3D_ERR_ARG : A3D_ERR_UNSUPPORTED;
    // (Measured at 64 frames, tap-outer | this kernel, tools/ups_bench.py: 8x10 0.078 | 0.060 ms, 15x20 0.142 | 0.115, 30x40 0.403 | 0.326, 60x80
    // 1.351 | 1.045, 120x160 2.825 | 1.931 -- ahead on every map of the decoder, also where its tiles cover 1.3-3 x the map: every phase-5
    // launch takes it (16 x 16 tiles where they cover fewer pixels than 8 x 32: the 30x40 and 60x80 stages).  The two forms differ in
    // their reduction order, so the choice may never depend on the batch.)
    return pp_launch<true, 4>(d, s, "conv_ph4p_kernel<");
}

// The plain 3x3 stride-1 pad-1 layers of the fp16x2 arithmetic that stay in the direct form (fewer than 256 input channels: res2 / res3
// conv2): A3D_ERR_UNSUPPORTED for anything else -- residual / gate / concat / other geometry, no pre-split filter, a map too small to
// be worth an 8 x 32 tile (a function of the layer and the map size, never of the batch: the reduction order differs from the
// tap-outer kernels'), tune != 0 (an explicit tile variant of those kernels) except 16, which forces this kernel.
int a3d_conv_launch_c3p(const a3d_conv_desc *d, hipStream_t s) {
    if (d->precision != 3 || !d->w_x3 || !d->in_amax || !(d->w_scale > 0.f) || (d->tune != 0 && d->tune != 16)) return A3D_ERR_UNSUPPORTED;
    if (d->phase || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->Ho != d->H || d->Wo != d->W) return A3D_ERR_UNSUPPORTED;
    if (d->stem || d->ups || d->m_dev || d->splitk != 1 || d->res || d->gate || d->pixshuf || d->Cin2 || d->x2) return A3D_ERR_UNSUPPORTED;
    if ((d->Cin & 31) || d->Kpad != 9 * d->Cin || (d->Cout & 3)) return A3D_ERR_UNSUPPORTED;
    if ((size_t)d->B * d->H * d->W * d->Cin * 4 >= ((size_t)1 << 32) || (size_t)d->Cout * d->Kpad * 4 >= ((size_t)1 << 31)) return A3D_ERR_UNSUPPORTED;
    // (measured at 64 frames, tap-outer | this kernel: 120x160 64 -> 64 0.465 | 0.30-0.32 ms, 60x80 128 -> 128 0.340 | 0.320, 120x160 128 -> 256
    // 2.03 | 1.81; 30x40 128 -> 128, whose tiles cover 1.28 x the map, 0.111 | 0.124-0.136: the tiles must cover at most 1.2 x the map)
    if (d->tune != 16 && pp_covered(d->H, d->W, pp_tile_width(d->H, d->W)) * 10 > (long)d->H * d->W * 12) return A3D_ERR_UNSUPPORTED;
    if (d->Cout <= 64) return pp_launch<false, 1>(d, s, "conv_c3p_kernel<1, ");
    if (d->Cout <= 128) return pp_launch<false, 2>(d, s, "conv_c3p_kernel<2, ");
    return pp_launch<false, 4>(d, s, "conv_c3p_kernel<4, ");
}
