// fp32-grade convolution / linear kernel on the bf16 matrix pipe (a3d_conv_desc.precision == 2, "bf16x3").
//
// The fp32 MFMA of gfx950 peaks at 157 TFLOP/s, the bf16 MFMA at 2.5 PFLOP/s.  An fp32 number is EXACTLY the sum of
// three bf16 numbers (x = hi + mid + lo: each split rounds to nearest even and the sign carries one bit, so 3 x 8
// significand bits cover the 24 of fp32), and a bf16 x bf16 product is exact in fp32.  So
//     a*b = hi_a*hi_b + hi_a*mid_b + mid_a*hi_b + mid_a*mid_b + hi_a*lo_b + lo_a*hi_b   (+ terms <= 2^-24 |a*b|)
// -- six bf16 MFMAs with fp32 accumulation reproduce the fp32 product to the rounding level of a single fp32 multiply
// (the dropped terms mid*lo, lo*mid, lo*lo are <= 2^-24 relative, the size of the rounding of one fp32 FMA), at
// 16/6 = 2.67x the matrix rate of the fp32 MFMA.  Tensors stay fp32 in HBM; the split happens while a chunk is staged
// into LDS (v_cvt_pk_bf16_f32 + subtract, ~22 VALU ops per float4, in the MFMAs' shadows).  Measured and rejected:
// weights pre-split once per layer into bf16 planes in HBM (halves the VALU work, same rate -- the loop is bound by the
// clock the chip holds under bf16 MFMA load, 1.5-1.7 GHz here, and by ~65 % matrix-pipe occupancy, not by the VALU).
// tools/x3_bench.py measures rate and error against float64 next to the native fp32 kernels.
//
// Decomposition: output tile 128 pixels x (64*TN) channels, 4 waves as 2x2, wave tile 64 x (32*TN); weights are MFMA
// operand A and activations operand B (a lane owns one output pixel, register quads are 4 consecutive channels).
// k chunks of 16 (= one v_mfma_f32_32x32x16_bf16 step): per chunk 3+3 operand planes in LDS ([plane][row][16 k],
// 48-byte row pitch: a fragment is ONE ds_read_b128 and 16 rows tile the 64 banks exactly once), double-buffered,
// one barrier per chunk; 6 x TN x 2 MFMAs per wave per chunk against (TN + 2) x 3 fragment reads.
#include "conv_common.h"
#include <stdlib.h>

namespace {
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 x3_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t x3_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
// fp16x2 mode (a3d_conv_desc.precision == 3): x * s = h + l with h, l fp16 and s a power of two that puts the tensor's largest
// magnitude in [2^14, 2^15): h carries 11 significant bits, l the next 11 (2^-22 relative wherever |x| >= max / 2^18, 2^-40 of the
// maximum below); h.h + h.l + l.h with fp32 accumulation drops only l.l (<= 2^-22 relative).  THREE MFMAs per k step.
__device__ __forceinline__ void split2h(const f32x4 v, const float s, h16x4 &h, h16x4 &l) {
    const f32x4 xs = v * s;
    h = __builtin_convertvector(xs, h16x4);
    const f32x4 r = xs - __builtin_convertvector(h, f32x4);
    l = __builtin_convertvector(r, h16x4);
}
// x = h + m + l exactly (round-to-nearest-even at each level)
__device__ __forceinline__ void split3(const f32x4 v, bf16x4 &h, bf16x4 &m, bf16x4 &l) {
    h = __builtin_convertvector(v, bf16x4);
    const f32x4 r1 = v - __builtin_convertvector(h, f32x4);
    m = __builtin_convertvector(r1, bf16x4);
    const f32x4 r2 = r1 - __builtin_convertvector(m, f32x4);
    l = __builtin_convertvector(r2, bf16x4);
}

// STEM: the 7x7 s2 p3 stem on the NHWC4 input (a3d_conv_desc.stem; w packed [Cout][7][8][4], Kpad = 224): a 16-deep chunk is 4
// consecutive filter columns x 4 channels of one filter row, i.e. loader lane j = tid % 4 fetches pixel (ih0 + kh, iw0 + 4 (c & 1) + j)
// whole -- the float4 it would fetch anyway -- with kh = c >> 1; validity is a row bit and a column-half bit instead of a tap bit.
// WDMA (fp16x2 only): the filter arrives pre-split and scaled (a3d_conv_desc.w_x3 = a3d_split_f16x2_chunk(w, .., 16, w_scale)) and goes
// global -> LDS by LDS-DMA instead of through registers: no weight loads into VGPRs, no split, no VGPR -> LDS stores for that operand
// (with three MFMAs per step those were 10-17 % of the compute-bound layers: res4 1x1 1024 -> 256 0.159 -> 0.135 ms without them).
template <int TN, bool STEM = false, bool F16 = false, bool WDMA = false>
__global__ __launch_bounds__(256, 3) void conv_x3_kernel(const a3d_conv_desc d, const int M, const int ntiles, const int nblk) {
    static_assert(!WDMA || F16, "pre-split weights by DMA belong to the fp16x2 form");
    constexpr int TM = 2, BKT = 16;
    constexpr int NP = F16 ? 2 : 3;  // operand planes
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
    constexpr int LKB = BKT;                       // bf16 elements per LDS row (32 bytes, no padding: the two 16-B halves of
                                                   // a row are XOR-swizzled with bit 3 of the row index instead)
    constexpr int TPR = BKT / 4, RPP = 256 / TPR;  // 4 lanes x float4 per row, 64 rows per loader pass
    constexpr int XR = BM / RPP, WR = BN / RPP;
    constexpr int PX = BM * LKB, PW = BN * LKB;    // one operand plane
    constexpr int BUF = NP * (PX + PW);
    // WDMA: THREE filter stages (the third one behind the two operand stages): the DMA of chunk c+3 is issued while chunk c is being
    // multiplied, two iterations before its fragments are read -- with one iteration of lead (~0.6 us here) the wait before the
    // barrier sat on L2 latency every time (40 % of the wave-cycles parked, PMC)
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * BUF + (WDMA ? NP * PW : 0)];
    auto wstage = [&](const int st) -> __bf16 * { return st < 2 ? lds + st * BUF + NP * PX : lds + 2 * BUF; };
    __shared__ __attribute__((aligned(16))) float ss[2 * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int logical = a3d_xcd_remap(blockIdx.x, nblk);
    const int mt = logical / ntiles, nt = logical - mt * ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int nk = d.Kpad / BKT;
    const int lr = tid / TPR, lc = (tid % TPR) * 4;
    // LDS image: [plane][row][16 k] bf16, 32-byte rows; the 16-byte half holding k = 8h..8h+7 of row r sits at slot
    // h ^ ((r >> 3) & 1).  ds_read_b128 (64-bank rule, 16-lane groups of rows {0-3,12-15,20-27} / ...) and ds_write_b64
    // (32-bank rule, 16 contiguous lanes = 4 rows) are both conflict-free with it; a padded 48-byte pitch made the
    // writes 2-way conflicted (SQ_LDS_BANK_CONFLICT = 1/3 of the LDS cycles).
    const int lcs = (lc & 7) | ((((lc >> 3) ^ (lr >> 3)) & 1) << 3);
    const int cs4 = STEM ? 16 : d.Cin * 4;
    const int CinT = d.Cin + d.Cin2;  // (a second source has the same channel count: checked by the launcher)
    const __amdgpu_buffer_rsrc_t rx = x3_rsrc(d.x, (unsigned)((size_t)d.B * d.H * d.W * (size_t)cs4));
    const __amdgpu_buffer_rsrc_t rx2 = x3_rsrc(d.x2 ? d.x2 : d.x, (unsigned)((size_t)d.B * d.H * d.W * (size_t)cs4));
    const __amdgpu_buffer_rsrc_t rw = x3_rsrc(d.w, (unsigned)((size_t)d.Cout * d.Kpad * 4));

    const float sw = F16 ? d.w_scale : 1.f;  // weight scale of the fp16x2 split; the activation rows carry their image's scale:
    float sxr[XR];
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
        if (d.phase) {  // 2x2 taps of output phase (dy,dx) of an upsampled 3x3 conv, on the source grid (conv_gemm_v2.hip)
            ih0 = oh - 1 + ((d.phase - 1) >> 1);
            iw0 = ow - 1 + ((d.phase - 1) & 1);
        }
        sxr[i] = (F16 && rok) ? a3d_in_scale(d, b) : 1.f;
        unsigned mask = 0;
        if (STEM) {
            const int j = tid & 3;
            for (int kh = 0; kh < 7; ++kh) mask |= (rok && (unsigned)(ih0 + kh) < (unsigned)d.H) ? (1u << kh) : 0u;
            mask |= ((unsigned)(iw0 + j) < (unsigned)d.W) ? (1u << 8) : 0u;               // filter columns 0..3
            mask |= (j < 3 && (unsigned)(iw0 + 4 + j) < (unsigned)d.W) ? (1u << 9) : 0u;  // filter columns 4..6 (7 is padding)
            rowoff[i] = ((b * d.H + ih0) * d.W + iw0 + j) * 16;
        } else {
            for (int kh = 0; kh < d.KH; ++kh)
                for (int kw = 0; kw < d.KW; ++kw)
                    mask |= (rok && (unsigned)(ih0 + kh) < (unsigned)d.H && (unsigned)(iw0 + kw) < (unsigned)d.W) ? (1u << (kh * d.KW + kw)) : 0u;
            rowoff[i] = ((b * d.H + ih0) * d.W + iw0) * cs4 + lc * 4;
        }
        vmask[i] = mask;
    }
    int woff[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int n = n0 + lr + RPP * i;
        woff[i] = n < d.Cout ? (n * d.Kpad + lc) * 4 : -1;
    }
    // WDMA: w_x3 [Kpad/16][2][Cout][16] fp16; one (chunk, plane) tile of this workgroup's BN rows is a contiguous run = BN / 32 DMA
    // wave-instructions of 32 rows.  Lane i lands at LDS byte 16 i of its 1 KiB = row i/2, half i%2, and fetches the k half the image
    // keeps there: half ^ ((row >> 3) & 1).  Rows past Cout read the next plane's rows / zeros: their accumulators are never stored.
    const __amdgpu_buffer_rsrc_t rw2 = x3_rsrc(WDMA ? d.w_x3 : d.w, WDMA ? (unsigned)((size_t)nk * d.Cout * 64) : 16u);
    const int wvoff = (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4);
    int dma_c = 0;
    constexpr int NPC = 2 * (BN / 32);      // DMA pieces per chunk
    constexpr int DPWN = (NPC + 3) / 4;      // ... and per wave
    int wst_dma = 0, wst_rd = 0;             // filter stage the next DMA goes to / the next fragment read comes from (chunk % 3)
    auto dma_w = [&]() {
        __bf16 *Wt = wstage(wst_dma);
        wst_dma = wst_dma == 2 ? 0 : wst_dma + 1;
        const int uw = __builtin_amdgcn_readfirstlane(wave);
        const int base = __builtin_amdgcn_readfirstlane(dma_c * d.Cout * 64 + n0 * 32);
#pragma unroll
        for (int i = 0; i < DPWN; ++i) {
            const int j = uw * DPWN + i;
            if (j < NPC) {
                const int p = j / (BN / 32), g = j % (BN / 32);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw2, (__attribute__((address_space(3))) void *)(Wt + p * PW + g * 32 * LKB), 16, wvoff,
                                                         base + __builtin_amdgcn_readfirstlane(p * d.Cout * 32 + g * 1024), 0, 0);
            }
        }
        ++dma_c;
    };
    int kc = 0, c0 = 0, kh = 0, kw = 0;  // position of the next chunk to load inside the filter
    // Two register staging sets: the loads of chunk c are issued at iteration c-3, split into LDS at iteration c-1 and
    // multiplied at iteration c, so a buffer load has two full iterations (2 x 6*TN*2 MFMAs per wave) to land.
    f32x4 xsA[XR], wsA[WR], xsB[XR], wsB[WR];
    // fp16x2 with the filter by DMA: a THIRD activation staging set -- loads are issued three chunks ahead instead of two.  These
    // layers are bound by the bytes a CU keeps in flight (PMC: waves parked at s_waitcnt 60 % of their cycles on the K = 256 layers,
    // DESIGN.md section 5a); the kernel has 155 of the 168 VGPRs that three waves per SIMD allow, i.e. room for exactly one more set.
    constexpr bool X3SETS = F16 && WDMA;
    f32x4 xsC[XR];
    auto load_chunk = [&](f32x4 (&xs)[XR], f32x4 (&ws)[WR]) {
        if (STEM) {
            const unsigned livebit = (kc < nk) ? 1u : 0u;
            const int skh = kc >> 1, half = kc & 1;
            const int tapoff = (skh * d.W + half * 4) * 16;
#pragma unroll
            for (int i = 0; i < XR; ++i)
                xs[i] = x3_load4(rx, ((vmask[i] >> skh) & (vmask[i] >> (8 + half)) & livebit) ? rowoff[i] + tapoff : -1, 0);
            const int soff = kc * (BKT * 4);
            if constexpr (!WDMA) {
#pragma unroll
                for (int i = 0; i < WR; ++i) ws[i] = x3_load4(rw, livebit ? woff[i] : -1, soff);
            }
            ++kc;
            return;
        }
        const int tap = kh * d.KW + kw;
        const unsigned livebit = (kc < nk) ? 1u : 0u;
        const bool second = c0 >= d.Cin;  // channel concat: the second source supplies channels Cin .. Cin+Cin2-1 of every tap
        const __amdgpu_buffer_rsrc_t r = second ? rx2 : rx;
        const int tapoff = (kh * d.W + kw) * cs4 + (second ? c0 - d.Cin : c0) * 4;
#pragma unroll
        for (int i = 0; i < XR; ++i) xs[i] = x3_load4(r, ((vmask[i] >> (tap & 31)) & livebit) ? rowoff[i] + tapoff : -1, 0);
        const int soff = kc * (BKT * 4);
        if constexpr (!WDMA) {
#pragma unroll
            for (int i = 0; i < WR; ++i) ws[i] = x3_load4(rw, livebit ? woff[i] : -1, soff);
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
    };
    auto store_chunk = [&](int buf, const f32x4 (&xs)[XR], const f32x4 (&ws)[WR]) {  // fp32 -> hi | mid | lo planes on the way into LDS
        __bf16 *X = lds + buf * BUF;
        __bf16 *Wt = X + NP * PX;
        if constexpr (F16) {
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                h16x4 h, l;
                split2h(xs[i], sxr[i], h, l);
                _Float16 *p = reinterpret_cast<_Float16 *>(X) + (lr + RPP * i) * LKB + lcs;
                *reinterpret_cast<h16x4 *>(p) = h;
                *reinterpret_cast<h16x4 *>(p + PX) = l;
            }
            if constexpr (!WDMA) {
#pragma unroll
                for (int i = 0; i < WR; ++i) {
                    h16x4 h, l;
                    split2h(ws[i], sw, h, l);
                    _Float16 *p = reinterpret_cast<_Float16 *>(Wt) + (lr + RPP * i) * LKB + lcs;
                    *reinterpret_cast<h16x4 *>(p) = h;
                    *reinterpret_cast<h16x4 *>(p + PW) = l;
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            bf16x4 h, m, l;
            split3(xs[i], h, m, l);
            __bf16 *p = X + (lr + RPP * i) * LKB + lcs;
            *reinterpret_cast<bf16x4 *>(p) = h;
            *reinterpret_cast<bf16x4 *>(p + PX) = m;
            *reinterpret_cast<bf16x4 *>(p + 2 * PX) = l;
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            bf16x4 h, m, l;
            split3(ws[i], h, m, l);
            __bf16 *p = Wt + (lr + RPP * i) * LKB + lcs;
            *reinterpret_cast<bf16x4 *>(p) = h;
            *reinterpret_cast<bf16x4 *>(p + PW) = m;
            *reinterpret_cast<bf16x4 *>(p + 2 * PW) = l;
        }
    };

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    a3d_stage_scale_shift(ss, d, n0, BN, tid);
    if constexpr (WDMA) {
        dma_w();  // chunks 0, 1 and 2 of the filter
        dma_w();
        dma_w();
    }
    load_chunk(xsA, wsA);  // chunk 0
    store_chunk(0, xsA, wsA);
    load_chunk(xsB, wsB);  // chunk 1
    load_chunk(xsA, wsA);  // chunk 2
    if constexpr (WDMA) __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (X3SETS) load_chunk(xsC, wsA);  // chunk 3 (behind the wait: it stays in flight)
    __syncthreads();

    const int frow = lane & 31;
    const int frag_off = frow * LKB + ((((lane >> 5) ^ (frow >> 3)) & 1) << 3);  // row = lane % 32, k = 8 * (lane / 32) .. + 7
    // [hi | mid | lo] fragments: set 0 holds the chunk being multiplied in even iterations, set 1 in odd ones; the other
    // set is filled (behind the second half of the MFMAs) with the next chunk.
    bf16x8 fa0[NP][TN], fb0[NP][TM], fa1[NP][TN], fb1[NP][TM];
    auto read_frags = [&](int buf, bf16x8 (&fa)[NP][TN], bf16x8 (&fb)[NP][TM]) {
        const __bf16 *X = lds + buf * BUF + (wm * TM * 32) * LKB + frag_off;
        const __bf16 *Wt = (WDMA ? wstage(wst_rd) : lds + buf * BUF + NP * PX) + (wn * TN * 32) * LKB + frag_off;
        if constexpr (WDMA) wst_rd = wst_rd == 2 ? 0 : wst_rd + 1;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) fa[p][ni] = *reinterpret_cast<const bf16x8 *>(Wt + p * PW + ni * 32 * LKB);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) fb[p][mi] = *reinterpret_cast<const bf16x8 *>(X + p * PX + mi * 32 * LKB);
        }
    };
    read_frags(0, fa0, fb0);
#define X3_TERM(PA, PB)                                                                                                       \
    _Pragma("unroll") for (int ni = 0; ni < TN; ++ni) _Pragma("unroll") for (int mi = 0; mi < TM; ++mi) {                    \
        if constexpr (F16)                                                                                                    \
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, fa[PA][ni]),                      \
                                                                 __builtin_bit_cast(h16x8, fb[PB][mi]), acc[ni][mi], 0, 0, 0); \
        else                                                                                                                  \
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][ni], fb[PB][mi], acc[ni][mi], 0, 0, 0);              \
    }
    // One chunk.  A wave issues in order and an MFMA holds the matrix pipe for 8 passes, so everything else is placed in
    // the MFMAs' shadows: first half = 3 product terms + the split of the staged chunk it+1 into LDS[cur^1]; ONE barrier;
    // second half = 3 terms + the fragment reads of chunk it+1 + the buffer loads of chunk it+3.  (With the reads behind
    // the barrier and outside the MFMA stream, the two co-resident workgroups fall into lockstep and the pipe idles
    // during both their read phases: 58 % MFMA-busy measured.)
    auto step = [&](const int cur, f32x4 (&xs)[XR], f32x4 (&ws)[WR], bf16x8 (&fa)[NP][TN], bf16x8 (&fb)[NP][TM],
                    bf16x8 (&fan)[NP][TN], bf16x8 (&fbn)[NP][TM]) {
        if constexpr (F16) {  // three terms in the wide kernel's order (h.h, h.l, l.h per accumulator: the two kernels agree bit for bit)
            X3_TERM(0, 0)
            X3_TERM(0, 1)
            store_chunk(cur ^ 1, xs, ws);
#pragma unroll
            for (int g = 0; g < 2 * TN * TM; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);  // <= 8 VALU
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // <= 1 LDS write
            }
            // (WDMA: the filter DMA of chunk it+1 has landed -- issued two iterations ago; younger than it are that iteration's XR
            // activation loads and the last iteration's DMA + loads)
            if constexpr (WDMA) {
                if constexpr (DPWN == 2) __asm__ volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else __asm__ volatile("s_waitcnt vmcnt(5)" ::: "memory");
                // A BARE barrier.  __syncthreads() carries a workgroup fence, and in front of a fence the compiler completes every
                // LDS-DMA it has seen issued (it emitted vmcnt(2) here: the DMA of chunk it+2, one step old, had to land as well --
                // an L2 round trip exposed in every chunk, the 60 % of parked wave-cycles PMC showed on these layers).
                __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            } else {
                __syncthreads();
            }
            X3_TERM(1, 0)
            read_frags(cur ^ 1, fan, fbn);
            if constexpr (WDMA) {
                dma_w();  // chunk it+3 into the stage whose fragments were read an iteration ago
                __builtin_amdgcn_sched_barrier(0);
            }
            load_chunk(xs, ws);
#pragma unroll
            for (int g = 0; g < TN * TM; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);  // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 1);  // <= 2 LDS reads
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);  // <= 1 buffer load
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 1);  // <= 4 VALU (addresses)
            }
            return;
        }
        X3_TERM(0, 0)
        X3_TERM(0, 1)
        X3_TERM(1, 0)
        store_chunk(cur ^ 1, xs, ws);
#pragma unroll
        for (int g = 0; g < 3 * TN * TM; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);  // <= 8 VALU
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // <= 1 LDS write
        }
        __syncthreads();
        X3_TERM(1, 1)
        X3_TERM(2, 0)
        X3_TERM(0, 2)
        read_frags(cur ^ 1, fan, fbn);
        load_chunk(xs, ws);
#pragma unroll
        for (int g = 0; g < 3 * TN * TM; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);  // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);  // <= 1 LDS read
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 1);  // <= 1 buffer load
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 1);  // <= 3 VALU (addresses)
        }
    };
    if constexpr (X3SETS) {
        // chunk c+1 waits in set (c+1) % 3 = B, A, C, B, A, C ..; the LDS stage alternates: the pattern repeats every six chunks.
        // (the counted vmcnt of a step does not change: every step still issues XR loads and one filter DMA group)
        int it = 0;
        for (; it + 6 <= nk; it += 6) {
            step(0, xsB, wsB, fa0, fb0, fa1, fb1);
            step(1, xsA, wsA, fa1, fb1, fa0, fb0);
            step(0, xsC, wsA, fa0, fb0, fa1, fb1);
            step(1, xsB, wsB, fa1, fb1, fa0, fb0);
            step(0, xsA, wsA, fa0, fb0, fa1, fb1);
            step(1, xsC, wsA, fa1, fb1, fa0, fb0);
        }
        const int rem = nk - it;  // 0 .. 5; an odd chunk count multiplies one all-zero chunk (loads past nk read as 0)
        if (rem > 0) {
            step(0, xsB, wsB, fa0, fb0, fa1, fb1);
            step(1, xsA, wsA, fa1, fb1, fa0, fb0);
        }
        if (rem > 2) {
            step(0, xsC, wsA, fa0, fb0, fa1, fb1);
            step(1, xsB, wsB, fa1, fb1, fa0, fb0);
        }
        if (rem > 4) {
            step(0, xsA, wsA, fa0, fb0, fa1, fb1);
            step(1, xsC, wsA, fa1, fb1, fa0, fb0);
        }
    } else {
    for (int it = 0; it < nk; it += 2) {  // (an odd chunk count multiplies one all-zero chunk: loads past nk read as 0)
        step(0, xsB, wsB, fa0, fb0, fa1, fb1);
        step(1, xsA, wsA, fa1, fb1, fa0, fb0);
    }
    }
#undef X3_TERM

    const bool has_res = d.res != nullptr;
    const int hwo = d.Ho * d.Wo;
    // Row-major epilogue.  The MFMA leaves a lane with ONE pixel and register quads of 4 channels: stored as they are, a wave
    // instruction touches 32 rows x 32 bytes (64 separate 16-byte requests, and the same again for the residual).  Each 32 x 32 tile
    // therefore goes through 4 KiB of LDS (quad index XOR-swizzled with the row: conflict-free both ways) and comes back with 8 lanes
    // per row: a wave instruction then covers 8 rows x 128 contiguous bytes (res2 1x1 64 -> 256 + residual: 0.79 -> 0.61 ms).  Same
    // values, same operations per element: the stored bits do not change.  No barrier: behind the last chunk's barrier every
    // fragment a wave still multiplies is in registers, and the only later LDS traffic (fragment reads of a chunk that does not
    // exist, filter DMAs past the last chunk) reads garbage nobody uses or lands in the FILTER areas -- the tiles go through the two
    // ACTIVATION areas.  Pixel-shuffle and gated stores keep the direct form; tune 12 forces it (A/B, bit-equality test).
    if (!(d.pixshuf || d.gate) && d.tune != 12) {
        static_assert(NP * PX * 2 >= 8192, "two 4 KiB tiles per activation area");
        float *T = reinterpret_cast<float *>(lds + (wave >> 1) * BUF) + (wave & 1) * 1024;
        const int pr = lane & 31, ph = lane >> 5;
        const int qr = lane >> 3, qc = lane & 7;
        float vmaxs[TM][4];
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int mb = m0 + (wm * TM + mi) * 32;
            const int mp = mb + pr;
            const float unx = (F16 && mp < M) ? 1.f / a3d_in_scale(d, mp / hwo) : 1.f, unw = F16 ? 1.f / sw : 1.f;
            float (&vmax)[4] = vmaxs[mi];
            vmax[0] = vmax[1] = vmax[2] = vmax[3] = 0.f;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    f32x4 v = {acc[ni][mi][rg * 4 + 0], acc[ni][mi][rg * 4 + 1], acc[ni][mi][rg * 4 + 2], acc[ni][mi][rg * 4 + 3]};
                    if constexpr (F16) v = (v * unx) * unw;  // exact: powers of two
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
                    if (has_res && nok && m < M) {
                        size_t res_row;
                        int b, oh, ow;
                        out_rows(d, m, res_row, b, oh, ow);
                        rv[j] = *reinterpret_cast<const f32x4 *>(d.res + res_row * (size_t)d.Cout + n);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = mb + qr + 8 * j;
                    if (!nok || m >= M) continue;
                    const f32x4 v = a3d_epilogue_math(d, tv[j], *reinterpret_cast<const f32x4 *>(ss + nl), *reinterpret_cast<const f32x4 *>(ss + BN + nl), has_res, rv[j]);
                    vmax[j] = fmaxf(vmax[j], a3d_absmax4(v));
                    size_t orow = (size_t)m;
                    if (d.phase) {  // output phase (dy, dx) of an upsampled 3x3 conv: pixel (2 oh + dy, 2 ow + dx) of the 2x map
                        const int b = m / hwo, r = m - b * hwo;
                        const int oh = r / d.Wo, ow = r - oh * d.Wo;
                        orow = ((size_t)b * (2 * d.Ho) + (2 * oh + ((d.phase - 1) >> 1))) * (size_t)(2 * d.Wo) + (2 * ow + ((d.phase - 1) & 1));
                    }
                    *reinterpret_cast<f32x4 *>(d.y + orow * d.Cout + n) = v;
                }
            }
        }
        // The maxima are recorded behind the wave's LAST store: the pre-check read of an image's slot queues at the L2 behind the
        // atomics of every other workgroup, and in front of the second row block it held that block's stores back.
        const int mb0 = m0 + wm * TM * 32;
        const bool whole = d.y_amax && mb0 < M && mb0 / hwo == min(mb0 + TM * 32 - 1, M - 1) / hwo;  // all the wave's rows in one image: ONE note
        if (whole) {
            float v = 0.f;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) v = fmaxf(v, fmaxf(fmaxf(vmaxs[mi][0], vmaxs[mi][1]), fmaxf(vmaxs[mi][2], vmaxs[mi][3])));
            a3d_note_amax(d.y_amax, mb0 / hwo, v, true);
        }
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int mb = m0 + (wm * TM + mi) * 32;
            const float (&vmax)[4] = vmaxs[mi];
            if (d.y_amax && !whole) {
                const int mlast = min(mb + 31, M - 1);
                if (mb < M && mb / hwo == mlast / hwo) {  // the tile's rows belong to one image (uniform per wave): one reduction
                    a3d_note_amax(d.y_amax, mb / hwo, fmaxf(fmaxf(vmax[0], vmax[1]), fmaxf(vmax[2], vmax[3])), true);
                } else {
                    // (rows of several images, e.g. the FC layers where every ROI is one: the 8 lanes of a row reduce first, so a row
                    // costs one pre-checked atomic per wave instead of eight -- fc2 at 64000 rows 1.03 -> see DESIGN 5a)
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
        return;
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = m0 + (wm * TM + mi) * 32 + (lane & 31);
        const bool mok = m < M;
        const int bimg = mok ? m / hwo : 0;  // image / ROI of this lane's output pixel
        float vmax = 0.f;
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
                v = a3d_epilogue_math(d, v, *reinterpret_cast<const f32x4 *>(ss + nl), *reinterpret_cast<const f32x4 *>(ss + BN + nl), has_res, rv[rg]);
                vmax = fmaxf(vmax, a3d_absmax4(v));
                store_out(d, v, m, n, b, oh, ow);
            }
        }
        }
        if (d.y_amax) a3d_note_amax(d.y_amax, bimg, vmax, mok);  // (every lane of the wave gets here)
    }
}

template <int TN, bool STEM = false>
void launch_x3(const a3d_conv_desc *d, hipStream_t s) {
    constexpr int BM = 128, BN = 64 * TN;
    const int M = d->B * d->Ho * d->Wo;
    const int mtiles = (M + BM - 1) / BM, ntiles = (d->Cout + BN - 1) / BN;
    if (d->precision == 3) {
        a3d_note_variant(STEM ? "conv_h2_kernel<%d> stem" : "conv_h2_kernel<%d>", TN);
        if (d->w_x3 && (size_t)d->Cout * d->Kpad * 4 < ((size_t)1 << 31))
            hipLaunchKernelGGL((conv_x3_kernel<TN, STEM, true, true>), dim3(mtiles * ntiles), dim3(256), 0, s, *d, M, ntiles, mtiles * ntiles);
        else
            hipLaunchKernelGGL((conv_x3_kernel<TN, STEM, true, false>), dim3(mtiles * ntiles), dim3(256), 0, s, *d, M, ntiles, mtiles * ntiles);
        return;
    }
    a3d_note_variant(STEM ? "conv_x3_kernel<%d> stem" : "conv_x3_kernel<%d>", TN);
    hipLaunchKernelGGL((conv_x3_kernel<TN, STEM>), dim3(mtiles * ntiles), dim3(256), 0, s, *d, M, ntiles, mtiles * ntiles);
}
}  // namespace

int a3d_conv_launch_bf16x3(const a3d_conv_desc *d0, hipStream_t s) {
    const bool direct_epilogue = a3d_dev_knob("A3D_X3_DIRECT_EPILOGUE", 0) == 1;  // (developer builds, A/B runs; tune 12 selects the same form per launch)
    a3d_conv_desc dd;
    const a3d_conv_desc *d = d0;
    if (direct_epilogue && d0->tune == 0) {
        dd = *d0;
        dd.tune = 12;
        d = &dd;
    }
    if (d->precision == 3 && (!d->in_amax || !(d->w_scale > 0.f))) return A3D_ERR_ARG;
    if (d->precision == 3 && (d->tune == 0 || d->tune == 17)) {  // small grids (single frames): a wave per 32 x 32 tile, no LDS (bit-identical results)
        const int rs = a3d_conv_launch_sg_h2(d, s);
        if (rs != A3D_ERR_UNSUPPORTED) return rs;
    }
    if (d->precision == 3 && (d->tune == 0 || d->tune == 13)) {  // HBM-bound 1x1 layers, Cin <= 256: x read once (bit-identical results)
        const int rx = a3d_conv_launch_xs_h2(d, s);
        if (rx != A3D_ERR_UNSUPPORTED) return rx;
    }
    // (Round 5, built, bit-identical, measured and removed -- tools/probes/rejected/conv_dk_h2.hip, profiles/r05_dk_bench.txt: the deep 1x1
    // reductions (Cin >= 512) with BOTH operands by LDS-DMA -- the raw fp32 activation tile through a 3 - 4 stage ring, split on the fragment --
    // in the Winograd GEMM's 512-thread ping-pong form: 10 - 15 % SLOWER than conv_x3_kernel<2> on all thirteen such layers of the trunk
    // (2.96 - 3.04 ms against 2.66 per 64 frames).  Three independent 256-thread workgroups per CU already hide what the antiphase halves
    // hide, and the one-workgroup-per-CU form exposes its ring fill and epilogue on a k loop of 16 - 64 chunks.)
    if (d->precision == 3 && !d->x_h2) {  // plain 3x3 s1 p1 layers: the patch-resident kernel (another reduction order; chosen by layer and map size)
        const int rc3 = a3d_conv_launch_c3p(d, s);
        if (rc3 != A3D_ERR_UNSUPPORTED) return rc3;
    }
    const int rw = a3d_conv_launch_bf16x3_wide(d, s);  // wide and large layers with pre-split weights (bit-identical results)
    if (rw != A3D_ERR_UNSUPPORTED) return rw;
    if (d->stem) {  // the 7x7 stem (x is [B,H,W,4]): its own loader, 128 x 64 tiles
        if (d->KH != 7 || d->KW != 7 || d->stride != 2 || d->pad != 3 || d->Kpad != 224 || d->x2 || d->res || d->ups || d->pixshuf || d->phase ||
            d->splitk != 1 || d->m_dev || (size_t)d->B * d->H * d->W * 16 >= ((size_t)1 << 32))
            return A3D_ERR_UNSUPPORTED;
        if (d->Cout <= 64) launch_x3<1, true>(d, s);
        else launch_x3<2, true>(d, s);
        return a3d_check_launch();
    }
    if (d->ups || d->splitk != 1 || d->m_dev) return A3D_ERR_UNSUPPORTED;
    if (d->pixshuf && (d->res || d->phase || d->gate)) return A3D_ERR_UNSUPPORTED;  // (ConvTranspose2d k2 s2: scatter in store_out only)
    if (d->phase && (d->KH != 2 || d->KW != 2 || d->stride != 1 || d->res)) return A3D_ERR_UNSUPPORTED;
    if (d->Cin2 && (d->Cin2 != d->Cin || !d->x2)) return A3D_ERR_UNSUPPORTED;
    if ((d->Cin & 15) || d->Kpad != d->KH * d->KW * (d->Cin + d->Cin2) || d->KH * d->KW > 32) return A3D_ERR_UNSUPPORTED;
    if ((size_t)d->B * d->H * d->W * d->Cin * 4 >= ((size_t)1 << 32) || (size_t)d->Cout * d->Kpad * 4 >= ((size_t)1 << 32)) return A3D_ERR_UNSUPPORTED;
    const int M = d->B * d->Ho * d->Wo;
    const long n128 = (long)((M + 127) / 128) * ((d->Cout + 127) / 128);
    if (d->tune == 10) launch_x3<1>(d, s);  // (tune 10 / 11: explicit tile width, A/B runs)
    else if (d->tune == 11) launch_x3<2>(d, s);
    else if (d->Cout <= 64 || (d->precision == 2 && n128 <= 500)) launch_x3<1>(d, s);  // (fp16x2: the 128-wide tile also on small grids,
    // measured 10-30 % faster there; same bits) -- except tiny grids with deep reductions (single frames: the launch is one
    // partial round whose length is the k loop's, and the 64-wide tile's iteration is shorter: 1x30x40 1024 -> 256 43 -> 34 us,
    // 1x15x20 2048 -> 512 84 -> 69 us, 4x30x40 1024 -> 256 48 -> 43 us; shallow layers stay: 1x60x80 128 -> 512 23 | 26 us)
    else if (d->precision == 3 && n128 <= 96 && d->Kpad >= 512) launch_x3<1>(d, s);
    else launch_x3<2>(d, s);
    return a3d_check_launch();
}
