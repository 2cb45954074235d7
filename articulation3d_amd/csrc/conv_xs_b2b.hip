// Back-to-back pointwise pair ACROSS a bottleneck boundary (round 6): conv3 (+ FrozenBN + residual + ReLU) of block i and conv1 (+ FrozenBN +
// ReLU) of block i + 1 in ONE launch (pkg/modeling/meta_arch/planercnn.py:29,150 -> detectron2 BottleneckBlock).
//
// What it is for.  The 1x1 layers of res2 / res3 are HBM-bound: a block output y (256 / 512 channels per pixel) is written by conv3 and
// read straight back by the next block's conv1, which squeezes it to 64 / 128 channels.  Here the workgroup that produces a pixel tile of
// y keeps going: every 32-channel group of y, the moment its epilogue has produced it, is multiplied into the NEXT layer's accumulators
// while it is still in LDS.  y is stored once (the next residual and the FPN need it) and never re-read by this pair: one full read of
// every res2 / res3 block output (1.26 / 0.63 GB at 64 frames) and one launch per block are gone.
//
// First GEMM: conv_xs_kernel's loop (conv_xs_h2.hip), unchanged in its arithmetic -- the wave's 32 pixels stationary in registers as fp16x2
// fragments, the pre-split filter streamed global -> LDS through a ring of 8 KiB stages -- so y and its recorded maxima are bit for bit
// what the single launch stores.
// Second GEMM: its reduction runs over y's channels, i.e. over this kernel's N steps.  The fp16x2 split of an activation needs the
// per-image maximum of the WHOLE tensor, which no workgroup knows before the launch has finished; the exact three-way bf16 split needs no
// scale at all (x = h + m + l exactly, conv_bf16x3.hip).  So the second layer runs in the bf16x3 arithmetic (a3d_conv_desc.precision 2:
// six bf16 MFMAs per 16-deep chunk, full 24-bit operands -- two bits MORE than the surrounding fp16x2 layers carry), on a layer that is
// HBM-bound with room to spare: per pixel 64 x 256 x 3 + 256 x 64 x 6 MFMA products against 2.5 KiB of HBM traffic.  Per output element
// the operations are those of conv_x3_kernel's bf16x3 loop on the stored y -- same split, same chunk order, same six terms per chunk into
// one fp32 accumulator, same epilogue --: the launch agrees BIT FOR BIT with the two launches conv_xs (fp16x2) -> conv_x3 (bf16x3).
//
// Stream of ring stages per N step (BN = 32 NG channels of y): SPT stages of the first filter (as conv_xs), then, per 32-channel group g
// behind its epilogue, the second filter's slices for the group's two 16-deep chunks: one stage per (chunk, pair of 32-row output tiles) =
// 3 planes x 2 tiles x 1 KiB (6 of the stage's 8 KiB; the other two DMA slots fetch out of range, so every step issues the same number of
// vector-memory operations and the counted waits stay static).
#include "conv_common.h"

namespace {
typedef _Float16 bb_h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 bb_h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bb_b16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bb_b16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t bb_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void bb_split2h(const f32x4 v, const float s, bb_h16x4 &h, bb_h16x4 &l) {  // conv_xs_h2.hip xs_split
    const f32x4 xs = v * s;
    h = __builtin_convertvector(xs, bb_h16x4);
    const f32x4 r = xs - __builtin_convertvector(h, f32x4);
    l = __builtin_convertvector(r, bb_h16x4);
}
__device__ __forceinline__ void bb_split3(const f32x4 v, bb_b16x4 &h, bb_b16x4 &m, bb_b16x4 &l) {  // conv_bf16x3.hip split3
    h = __builtin_convertvector(v, bb_b16x4);
    const f32x4 r1 = v - __builtin_convertvector(h, f32x4);
    m = __builtin_convertvector(r1, bb_b16x4);
    const f32x4 r2 = r1 - __builtin_convertvector(m, f32x4);
    l = __builtin_convertvector(r2, bb_b16x4);
}
template <int N>
__device__ __forceinline__ void bb_wait_vm() {
    __asm__ volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int BB_NST = 7;        // ring stages of 8 KiB
constexpr int BB_D = BB_NST - 1;  // a stage's DMA is issued this many steps before its fragments are read
constexpr int BB_STAGE = 8192;
constexpr int bb_lds_bytes(int bn, int n2) { return BB_NST * BB_STAGE + 4 * 4096 + 2 * 2 * bn * 4 + 2 * n2 * 4; }

struct B2bArgs {
    const void *w2;       // second filter, bf16x3 planes [K2/16][3][Cout2][16] (a3d_conv_desc.w_x3 at precision 2), K2 = the first layer's Cout
    const float *scale2;  // [Cout2] folded BN (or NULL)
    const float *shift2;  // [Cout2] (or NULL)
    float *z;             // [M][Cout2]
    float *z_amax;        // [B] or NULL
    int Cout2, act2;
};

// KC = Cin / 16 of the first layer, NG = 32-channel groups per N step, KS = 16-deep chunks per ring stage (NG * KS = 4), N2T = Cout2 / 32.
// PREF: the residual rows are requested one N step ahead (two register sets; the Cin 128 pair has no room for the second one).
template <int KC, int NG, int KS, int N2T, bool PREF>
__global__ __launch_bounds__(256, 2) void conv_xs_b2b_kernel(const a3d_conv_desc d, const B2bArgs e, const int M) {
    static_assert(NG * KS == 4 && KC % KS == 0 && N2T % 2 == 0, "ring stage = 8 KiB; the second layer's tiles go in pairs");
    constexpr int BN = 32 * NG;
    constexpr int SPT = KC / KS;    // ring steps of the first GEMM per N step
    constexpr int R = 4 * NG;       // residual loads of an N step
    constexpr int TP = N2T / 2;     // tile pairs of the second layer
    constexpr int G2G = 2 * TP;     // second-GEMM steps per 32-channel group (2 chunks x tile pairs)
    constexpr int S = SPT + NG * G2G;  // ring steps per N step
    static_assert(S >= BB_D, "a wait's look-back (BB_D steps) reaches at most into the previous N step");
    extern __shared__ __attribute__((aligned(16))) unsigned char bb_lds[];
    unsigned char *ring = bb_lds;
    float *Tall = reinterpret_cast<float *>(bb_lds + BB_NST * BB_STAGE);
    float *ssall = Tall + 4 * 1024;      // [2][2 * BN]: scale | shift of the N step, double-buffered
    float *ss2 = ssall + 2 * 2 * BN;     // [2][32 * N2T]: scale | shift of the second layer

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 128 + wave * 32;  // this wave's 32 pixels
    const int hwo = d.Ho * d.Wo;
    const int nsteps = d.Cout / BN;
    const int Q = nsteps * S;
    const int Cout2 = 32 * N2T;

    const __amdgpu_buffer_rsrc_t rw = bb_rsrc(d.w_x3, (unsigned)((size_t)KC * d.Cout * 64));
    const __amdgpu_buffer_rsrc_t rw2 = bb_rsrc(e.w2, (unsigned)((size_t)(d.Cout / 16) * 3 * Cout2 * 32));
    const int wvoff = (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4);
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    int dma_q = 0, dma_st = 0, rd_st = 0;
    auto dma = [&]() {
        const int q = dma_q++;
        const int ns = q / S, sp = q - ns * S;
        unsigned char *st = ring + dma_st * BB_STAGE;
        dma_st = dma_st == BB_NST - 1 ? 0 : dma_st + 1;
        const bool live = q < Q;  // (past the last step: out of range -- zeros into a stage nobody reads; the op count per step stays fixed)
        if (sp < SPT) {  // a stage of the first filter: w_x3 [Cin/16][2][Cout][16] fp16, pieces [chunk-in-stage][plane][group]
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = uw * 2 + i;
                const int kl = j / (2 * NG), p = (j / NG) & 1, g = j % NG;
                const int c = sp * KS + kl;
                const int soff = live ? ((c * 2 + p) * d.Cout + ns * BN + g * 32) * 32 : 0;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void *)(st + j * 1024), 16, live ? wvoff : -1,
                                                         __builtin_amdgcn_readfirstlane(soff), 0, 0);
            }
        } else {  // a stage of the second filter: chunk (ns * BN) / 16 + 2 g + ci, tile pair tp; pieces [plane][tile of the pair]
            const int i2 = sp - SPT;
            const int g = i2 / G2G, r2 = i2 - g * G2G;
            const int ci = r2 / TP, tp = r2 - ci * TP;
            const int c = (ns * BN) / 16 + 2 * g + ci;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = uw * 2 + i;
                const int p = j >> 1, tl = j & 1;
                const bool on = live && j < 6;
                const int soff = on ? ((c * 3 + p) * Cout2 + (tp * 2 + tl) * 32) * 32 : 0;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw2, (__attribute__((address_space(3))) void *)(st + j * 1024), 16, on ? wvoff : -1,
                                                         __builtin_amdgcn_readfirstlane(soff), 0, 0);
            }
        }
    };
#pragma unroll
    for (int i = 0; i < BB_D; ++i) dma();

    if (tid < Cout2) {  // the second layer's folded BN (read at the very end; the first barrier of the loop publishes it)
        ss2[tid] = e.scale2 ? e.scale2[tid] : 1.f;
        ss2[Cout2 + tid] = e.shift2 ? e.shift2[tid] : 0.f;
    }

    // ---- the wave's activations (conv_xs_kernel): lane (pixel lane % 32, k group lane / 32) holds channels 16 c + 8 (lane / 32) .. + 7
    const int mp = m0 + (lane & 31);
    const bool mok = mp < M;
    const float sx = mok ? a3d_in_scale(d, mp / hwo) : 1.f;
    bb_h16x8 xh[KC], xl[KC];
    {
        const __amdgpu_buffer_rsrc_t rx = bb_rsrc(d.x, (unsigned)((size_t)M * d.Cin * 4));
        const int voff = mok ? (mp * d.Cin + (lane >> 5) * 8) * 4 : -1;
        f32x4 raw[KC][2];
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            raw[c][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, c * 64, 0));
            raw[c][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, c * 64 + 16, 0));
        }
        bb_wait_vm<0>();
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            bb_h16x4 h0, l0, h1, l1;
            bb_split2h(raw[c][0], sx, h0, l0);
            bb_split2h(raw[c][1], sx, h1, l1);
            xh[c] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            xl[c] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }

    const int frow = lane & 31;
    const int frag_off = (frow * 16 + ((((lane >> 5) ^ (frow >> 3)) & 1) << 3)) * 2;  // bytes inside a 1 KiB piece
    const int pr = lane & 31, ph = lane >> 5;  // accumulator layout: pixel, channel quad half
    const int qr = lane >> 3, qc = lane & 7;   // row-major epilogue: row (+ 8 j), channel quad
    float *T = Tall + wave * 1024;
    const float unx = 1.f / sx, unw = 1.f / d.w_scale;
    const bool one_image = m0 < M && m0 / hwo == min(m0 + 31, M - 1) / hwo;
    float vmax[4] = {0.f, 0.f, 0.f, 0.f};

    f32x16 z[N2T];
#pragma unroll
    for (int t = 0; t < N2T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) z[t][r] = 0.f;

    // Counted waits.  The DMA a step waits for was issued BB_D steps earlier, behind that step's barrier; younger than it are the
    // 2 (BB_D - 1) DMA pieces of the steps since and -- when their issue point lies inside that window -- the R residual loads of the NEXT
    // N step, which go out in front of the first second-GEMM step (position SPT of every N step; the first N step's go out in the
    // prologue, in front of every DMA a loop step can still be waiting for).  The y stores of the window are NOT counted: the wait is
    // then stricter than it has to be (a few of the youngest DMAs must land early), never too weak.
    auto enter = [&](const int s) -> const unsigned char * {  // s = the step's position in the N step (a constant after unrolling)
        // (without PREF the N step's own rows go out at its start, position 0)
        if (((s - (PREF ? SPT : 0)) % S + S) % S < BB_D) bb_wait_vm<2 * (BB_D - 1) + R>();
        else bb_wait_vm<2 * (BB_D - 1)>();
        // (a bare barrier: __syncthreads() carries a fence, and the compiler completes every LDS-DMA in flight in front of a fence)
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned char *st = ring + rd_st * BB_STAGE + frag_off;
        rd_st = rd_st == BB_NST - 1 ? 0 : rd_st + 1;
        dma();  // the stage BB_D steps ahead, into the stage read one step ago (every wave is past that read: the barrier above)
        return st;
    };
    // The residual rows of an N step, in the row-major form the epilogue stores, are requested one N step AHEAD (in front of the previous
    // step's second GEMM): a wave then has loads in flight through both GEMMs, not only through the first -- the launch is HBM-bound and
    // what it lacks is bytes in flight.  Buffer loads: the request past the last N step is out of range (no traffic, zeros nobody reads),
    // so every N step issues the same operations.
    const __amdgpu_buffer_rsrc_t rres = bb_rsrc(d.res, (unsigned)((size_t)M * d.Cout * 4));
    f32x4 rvA[NG][4], rvB[NG][4];  // the residual rows of even | odd N steps (two named sets: a copy would wait for the loads where it stands)
    auto load_res = [&](f32x4 (&rvn)[NG][4], const int ns) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int voff = ns < nsteps ? (min(m0 + qr + 8 * j, M - 1) * d.Cout + ns * BN + qc * 4) * 4 : -1;
#pragma unroll
            for (int g = 0; g < NG; ++g) rvn[g][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rres, voff, g * 128, 0));
        }
    };
    if constexpr (PREF) load_res(rvA, 0);

    auto nstep = [&](const int ns, f32x4 (&rv)[NG][4], f32x4 (&rvn)[NG][4]) {
        const int n0 = ns * BN;
        if constexpr (!PREF) load_res(rv, ns);  // (position 0 of the N step)
        float *ss = ssall + (ns & 1) * 2 * BN;
        if (tid < BN) {
            ss[tid] = d.scale ? d.scale[n0 + tid] : 1.f;
            ss[BN + tid] = d.shift ? d.shift[n0 + tid] : 0.f;
        }
        f32x16 acc[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

        // ---- first GEMM: the N step's SPT stages (conv_xs_kernel's loop body)
#pragma unroll
        for (int t = 0; t < SPT; ++t) {
            const unsigned char *st = enter(t);
#pragma unroll
            for (int kl = 0; kl < KS; ++kl) {
                const int c = t * KS + kl;
                bb_h16x8 fa[2][NG];
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int g = 0; g < NG; ++g) fa[p][g] = *reinterpret_cast<const bb_h16x8 *>(st + ((kl * 2 + p) * NG + g) * 1024);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][g], xh[c], acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][g], xl[c], acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][g], xh[c], acc[g], 0, 0, 0);
            }
        }
        if constexpr (PREF) load_res(rvn, ns + 1);  // (position SPT of the N step: see enter)

        // ---- per 32-channel group: epilogue (conv_xs_kernel's; the finished rows ALSO go back into T), then the group's slice of the second GEMM
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                f32x4 v = {acc[g][rg * 4 + 0], acc[g][rg * 4 + 1], acc[g][rg * 4 + 2], acc[g][rg * 4 + 3]};
                v = (v * unx) * unw;  // exact: powers of two
                *reinterpret_cast<f32x4 *>(T + pr * 32 + (((rg * 2 + ph) ^ (pr & 7)) << 2)) = v;
            }
            const int nl = g * 32 + qc * 4;
            f32x4 tv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int qq = qr + 8 * j;
                tv[j] = *reinterpret_cast<const f32x4 *>(T + qq * 32 + ((qc ^ (qq & 7)) << 2));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int qq = qr + 8 * j;
                const int m = m0 + qq;
                const f32x4 v = a3d_epilogue_math(d, tv[j], *reinterpret_cast<const f32x4 *>(ss + nl), *reinterpret_cast<const f32x4 *>(ss + BN + nl), true, rv[g][j]);
                // (rows past M: zero inputs + the clamped residual row -- finite values nobody stores; a pixel's column of the second GEMM is its own)
                *reinterpret_cast<f32x4 *>(T + qq * 32 + ((qc ^ (qq & 7)) << 2)) = v;
                if (m < M) {
                    vmax[j] = fmaxf(vmax[j], a3d_absmax4(v));
                    *reinterpret_cast<f32x4 *>(d.y + (size_t)m * d.Cout + n0 + nl) = v;
                }
            }
            // second GEMM on the group's two 16-deep chunks: B fragment = 8 consecutive channels of the lane's pixel, split exactly three ways
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                bb_b16x8 yb[3];
                {
                    const f32x4 a0 = *reinterpret_cast<const f32x4 *>(T + pr * 32 + (((ci * 4 + ph * 2 + 0) ^ (pr & 7)) << 2));
                    const f32x4 a1 = *reinterpret_cast<const f32x4 *>(T + pr * 32 + (((ci * 4 + ph * 2 + 1) ^ (pr & 7)) << 2));
                    bb_b16x4 h0, m0_, l0, h1, m1_, l1;
                    bb_split3(a0, h0, m0_, l0);
                    bb_split3(a1, h1, m1_, l1);
                    yb[0] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                    yb[1] = __builtin_shufflevector(m0_, m1_, 0, 1, 2, 3, 4, 5, 6, 7);
                    yb[2] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int tp = 0; tp < TP; ++tp) {
                    const unsigned char *st = enter(SPT + g * G2G + ci * TP + tp);
                    bb_b16x8 fa[3][2];
#pragma unroll
                    for (int p = 0; p < 3; ++p)
#pragma unroll
                        for (int tl = 0; tl < 2; ++tl) fa[p][tl] = *reinterpret_cast<const bb_b16x8 *>(st + (p * 2 + tl) * 1024);
                    // conv_x3_kernel's six terms, in its order: (filter plane, activation plane) = (0,0) (0,1) (1,0) (1,1) (2,0) (0,2)
#define BB_TERM(PA, PB)                                                                                                          \
    _Pragma("unroll") for (int tl = 0; tl < 2; ++tl) z[tp * 2 + tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA][tl], yb[PB], z[tp * 2 + tl], 0, 0, 0);
                    BB_TERM(0, 0)
                    BB_TERM(0, 1)
                    BB_TERM(1, 0)
                    BB_TERM(1, 1)
                    BB_TERM(2, 0)
                    BB_TERM(0, 2)
#undef BB_TERM
                }
            }
        }
    };
    if constexpr (PREF) {
        for (int ns = 0; ns < nsteps; ns += 2) {  // (the launcher guarantees an even number of N steps)
            nstep(ns, rvA, rvB);
            nstep(ns + 1, rvB, rvA);
        }
    } else {
        for (int ns = 0; ns < nsteps; ++ns) nstep(ns, rvA, rvA);
    }

    // ---- the second layer's epilogue: z tiles through T (row-major), folded BN + activation, stores, maxima
    a3d_conv_desc d2 = d;
    d2.act = e.act2;
    float zmax[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < N2T; ++t) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const f32x4 v = {z[t][rg * 4 + 0], z[t][rg * 4 + 1], z[t][rg * 4 + 2], z[t][rg * 4 + 3]};
            *reinterpret_cast<f32x4 *>(T + pr * 32 + (((rg * 2 + ph) ^ (pr & 7)) << 2)) = v;
        }
        const int nl = t * 32 + qc * 4;
        f32x4 tv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int qq = qr + 8 * j;
            tv[j] = *reinterpret_cast<const f32x4 *>(T + qq * 32 + ((qc ^ (qq & 7)) << 2));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = m0 + qr + 8 * j;
            if (m >= M) continue;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4 v = a3d_epilogue_math(d2, tv[j], *reinterpret_cast<const f32x4 *>(ss2 + nl), *reinterpret_cast<const f32x4 *>(ss2 + Cout2 + nl), false, zero);
            zmax[j] = fmaxf(zmax[j], a3d_absmax4(v));
            *reinterpret_cast<f32x4 *>(e.z + (size_t)m * Cout2 + nl) = v;
        }
    }
    // maxima of both tensors, once per wave behind its last store (conv_xs_kernel)
    auto note = [&](float *slot, const float (&vm)[4]) {
        if (!slot) return;
        if (one_image) {
            a3d_note_amax(slot, m0 / hwo, fmaxf(fmaxf(vm[0], vm[1]), fmaxf(vm[2], vm[3])), true);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + qr + 8 * j;
                float v = vm[j];
                v = fmaxf(v, __shfl_xor(v, 1, 64));
                v = fmaxf(v, __shfl_xor(v, 2, 64));
                v = fmaxf(v, __shfl_xor(v, 4, 64));
                a3d_note_amax(slot, m < M ? m / hwo : 0, v, m < M && qc == 0);
            }
        }
    };
    note(d.y_amax, vmax);
    note(e.z_amax, zmax);
    bb_wait_vm<0>();  // the DMAs issued past the last step must not land in the LDS of the next workgroup
}
template <int KC, int NG, int KS, int N2T, bool PREF>
int launch_b2b(const a3d_conv_desc *d, const B2bArgs &e, hipStream_t s) {
    constexpr int BN = 32 * NG;
    const int M = d->B * d->Ho * d->Wo;
    const int mtiles = (M + 127) / 128;
    constexpr int lds = bb_lds_bytes(BN, 32 * N2T);
    static a3d_attr_once attr;
    if (attr.needed()) {
        if (hipFuncSetAttribute((const void *)conv_xs_b2b_kernel<KC, NG, KS, N2T, PREF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return A3D_ERR_LAUNCH;
        attr.mark();
    }
    a3d_note_variant("conv_h2xs_b2b_kernel<%d,%d>", 16 * KC, 32 * N2T);
    hipLaunchKernelGGL((conv_xs_b2b_kernel<KC, NG, KS, N2T, PREF>), dim3(mtiles), dim3(256), lds, s, *d, e, M);
    return a3d_check_launch();
}
}  // namespace

// include/a3d.h: a3d_conv_b2b.  d1 = the first layer exactly as a3d_conv2d_nhwc_f32 takes it on the activation-stationary fp16x2 kernel
// (1x1 stride 1, precision 3, w_x3 / in_amax / w_scale, a residual); d2 = the second layer (1x1 stride 1 over d1's output, precision 2 with
// w_x3 = its bf16x3 planes, no residual).  A3D_ERR_UNSUPPORTED: not such a pair (the caller issues the two launches).
extern "C" int a3d_conv_b2b(const a3d_conv_desc *d1, const a3d_conv_desc *d2, void *stream) {
    if (!d1 || !d2 || !d1->x || !d1->y || !d2->y) return A3D_ERR_ARG;
    const a3d_conv_desc *d = d1;
    if (d->precision != 3 || !d->w_x3 || !d->in_amax || d->in_amax2 || !(d->w_scale > 0.f) || !d->res) return A3D_ERR_UNSUPPORTED;
    if (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0 || d->Kpad != d->Cin) return A3D_ERR_UNSUPPORTED;
    if (d->stem || d->ups || d->phase || d->pixshuf || d->gate || d->x2 || d->Cin2 || d->splitk != 1 || d->m_dev || d->res_ups || d->io_bf16) return A3D_ERR_UNSUPPORTED;
    if (d2->precision != 2 || !d2->w_x3 || d2->KH != 1 || d2->KW != 1 || d2->stride != 1 || d2->pad != 0 || d2->Kpad != d2->Cin) return A3D_ERR_UNSUPPORTED;
    if (d2->stem || d2->ups || d2->phase || d2->pixshuf || d2->gate || d2->x2 || d2->Cin2 || d2->splitk != 1 || d2->m_dev || d2->res || d2->io_bf16) return A3D_ERR_UNSUPPORTED;
    if (d2->Cin != d->Cout || d2->B != d->B || d2->H != d->Ho || d2->W != d->Wo || d2->Ho != d->Ho || d2->Wo != d->Wo) return A3D_ERR_ARG;
    if (d2->x && d2->x != d->y) return A3D_ERR_ARG;  // (the second layer reads what the first one stores)
    if (d->Cout % 128) return A3D_ERR_UNSUPPORTED;  // (an even number of 64-channel N steps)
    const size_t M = (size_t)d->B * d->Ho * d->Wo;
    if (M * d->Cout * 4 >= ((size_t)1 << 31) || M * d->Cin * 4 >= ((size_t)1 << 31) || (size_t)d->Cout * d->Cin * 4 >= ((size_t)1 << 31)) return A3D_ERR_UNSUPPORTED;
    B2bArgs e;
    e.w2 = d2->w_x3;
    e.scale2 = d2->scale;
    e.shift2 = d2->shift;
    e.z = d2->y;
    e.z_amax = d2->y_amax;
    e.Cout2 = d2->Cout;
    e.act2 = d2->act;
    a3d_begin();
    if (d->Cin == 64 && d2->Cout == 64) return launch_b2b<4, 2, 2, 2, true>(d, e, (hipStream_t)stream);
    if (d->Cin == 128 && d2->Cout == 128) return launch_b2b<8, 2, 2, 4, false>(d, e, (hipStream_t)stream);
    return A3D_ERR_UNSUPPORTED;
}
