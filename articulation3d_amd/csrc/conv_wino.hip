// 3x3 stride-1 pad-1 convolutions as Winograd F(2x2, 3x3) on the fp32 MFMA pipe (gfx950).
//
// The direct implicit GEMM spends 9*C multiply-adds per output; F(2x2,3x3) needs 16*C per 2x2 output tile
// = 4*C per output: 2.25x fewer MFMA cycles on layers that are MFMA-bound (3x3 convs are ~70 % of the
// detector's FLOPs).  fp32 error vs float64 is ~2x the direct form's (6e-7 vs 3e-7 relative on a 256-channel
// layer), well inside the path's tolerances.  Y = A^T [ (G g G^T) . (B^T d B) ] A with
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1].
//
// Two launches per layer:
//  1. wino_input_kernel  (HBM-bound): d -> V[f][tile][c], f = 4u+v, 16 planes.  The nearest-x2 upsampling and the
//     2-source channel concat of the depth decoder are folded into its gather, like in the direct kernel.
//  2. wino_gemm_kernel   (MFMA-bound): per workgroup 64 tiles x (32*TN*2) channels; for each of the 16 frequency
//     planes a plain GEMM over C (operands streamed with buffer loads through a double-buffered LDS image,
//     same fragment layout / k-permutation / bank padding as conv_gemm_v2), whose accumulator M_f is folded into
//     the four output accumulators Y_ij += A^T[i][u] A^T[j][v] M_f (coefficients 0/+-1) -- the 16 products are
//     never written to memory.  Epilogue: folded BN scale/shift + activation, float4 NHWC stores.
// Weights are transformed once on the host side at pack time (U = G g G^T, [16][Cout][Cin]).
#include "conv_common.h"
#include <stdlib.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 wbuf_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wmake_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

// ------------------------------------------------------------------------------------------------
// 1. input transform.  One thread = (tile, channel quad); 16 guarded float4 loads, 32+32 adds, 16 stores.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wino_input_kernel(const float *__restrict__ x, const float *__restrict__ x2,
                                                         float *__restrict__ V, int B, int H, int W, int Cin, int Cin2,
                                                         int ups, int Ty, int Tx) {
    const int C = Cin + Cin2, C4 = C >> 2;
    const int Hl = ups ? 2 * H : H, Wl = ups ? 2 * W : W;
    const size_t T = (size_t)B * Ty * Tx;
    const size_t total = T * C4;
    // (plain round-robin block placement on purpose: giving every XCD a contiguous band of tile rows through a3d_xcd_remap -- so that
    // the overlap of vertically adjacent patches is re-read from one L2 -- measured SLOWER, 0.32 -> 0.42 ms on the 60x80x256 level)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const size_t t = i / C4;
        const int tx = (int)(t % Tx);
        const size_t r = t / Tx;
        const int ty = (int)(r % Ty);
        const int b = (int)(r / Ty);
        const bool second = c >= Cin;
        const float *src = second ? x2 : x;
        const int cs = second ? Cin2 : Cin, cc = second ? c - Cin : c;
        f32x4 d[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int ly = 2 * ty - 1 + p;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int lx = 2 * tx - 1 + q;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)ly < (unsigned)Hl && (unsigned)lx < (unsigned)Wl) {
                    const int sy = ups ? ly >> 1 : ly, sx = ups ? lx >> 1 : lx;
                    v = *reinterpret_cast<const f32x4 *>(src + (((size_t)b * H + sy) * W + sx) * cs + cc);
                }
                d[p][q] = v;
            }
        }
        f32x4 m[4][4];  // B^T d
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            m[0][q] = d[0][q] - d[2][q];
            m[1][q] = d[1][q] + d[2][q];
            m[2][q] = d[2][q] - d[1][q];
            m[3][q] = d[1][q] - d[3][q];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // (B^T d) B
            const f32x4 v0 = m[u][0] - m[u][2], v1 = m[u][1] + m[u][2], v2 = m[u][2] - m[u][1], v3 = m[u][1] - m[u][3];
            float *o = V + ((size_t)(u * 4) * T + t) * C + c;
            *reinterpret_cast<f32x4 *>(o) = v0;
            *reinterpret_cast<f32x4 *>(o + T * C) = v1;
            *reinterpret_cast<f32x4 *>(o + 2 * T * C) = v2;
            *reinterpret_cast<f32x4 *>(o + 3 * T * C) = v3;
        }
    }
}

// fp16x2 arithmetic (a3d_conv_desc.precision == 3): the same transform, but V leaves the kernel ALREADY SPLIT into the two fp16 planes
// the matrix pipe multiplies -- (B^T d B) * s = h + l with s = wino_v_scale of the tile's image (a power of two from the recorded input
// maxima: the split the GEMM used to perform in its loop, element for element the same bits) -- in the chunk-major layout the GEMM's
// LDS-DMA wants: Vs [16 planes][C/32 chunks][h | l][T tiles][32 k] fp16.  The bytes are those of the fp32 tensor (2 + 2 per element).
// What it buys: wino_gemm_x3w_kernel<.., F16> moves V global -> LDS without registers, without the
// 2.7 vector instructions per MFMA the split cost there (PMC), without VGPR -> LDS stores.
typedef _Float16 wi_h16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int wi_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int wi_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void wino_input_h2_kernel(const float *__restrict__ x, const float *__restrict__ x2, unsigned char *__restrict__ Vs,
                                                            const float *__restrict__ in_amax, const float *__restrict__ in_amax2, int B, int H, int W,
                                                            int Cin, int Cin2, int ups, int Ty, int Tx, int t_off, int t_total) {
    const int C = Cin + Cin2;
    const int Hl = ups ? 2 * H : H, Wl = ups ? 2 * W : W;
    const size_t T = (size_t)B * Ty * Tx;
    // (a3d_conv_desc.wino_t_off / wino_t_total: this layer's tiles are the slice [t_off, t_off + T) of a buffer of t_total tiles per run)
    const size_t tile = (size_t)t_total * 64;  // bytes of one (plane, chunk, h | l) tile
    const int KC = C >> 5;
    // A wave = 8 consecutive tiles x one 32-channel chunk (lane = 8 * tile + channel quad): a store instruction then writes two runs of
    // 512 B (the h rows and the l rows of the 8 tiles); with a wave = one tile x all channels it wrote 64-byte pieces a whole tile
    // apart, and the kernel fell from 5.3 to 3.0 TB/s.  Loads: 128 contiguous bytes per pixel.
    const size_t witems = ((T + 7) / 8) * KC;
    const int lane = threadIdx.x & 63;
    for (size_t w = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); w < witems; w += (size_t)gridDim.x * (blockDim.x >> 6)) {
        const int kcw = (int)(w % KC);
        const size_t t = (w / KC) * 8 + (lane >> 3);
        if (t >= T) continue;  // (whole tiles drop out: the two lanes of a k slot stay together)
        const int c = kcw * 32 + (lane & 7) * 4;
        const int tx = (int)(t % Tx);
        const size_t r = t / Tx;
        const int ty = (int)(r % Ty);
        const int b = (int)(r / Ty);
        const bool second = c >= Cin;
        const float *src = second ? x2 : x;
        const int cs = second ? Cin2 : Cin, cc = second ? c - Cin : c;
        float am = in_amax[b];
        if (in_amax2) am = fmaxf(am, in_amax2[b]);
        const float sv = 0.25f * a3d_pow2_scale(am);  // wino_v_scale
        f32x4 d[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int ly = 2 * ty - 1 + p;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int lx = 2 * tx - 1 + q;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)ly < (unsigned)Hl && (unsigned)lx < (unsigned)Wl) {
                    const int sy = ups ? ly >> 1 : ly, sx = ups ? lx >> 1 : lx;
                    v = *reinterpret_cast<const f32x4 *>(src + (((size_t)b * H + sy) * W + sx) * cs + cc);
                }
                d[p][q] = v;
            }
        }
        f32x4 m[4][4];  // B^T d
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            m[0][q] = d[0][q] - d[2][q];
            m[1][q] = d[1][q] + d[2][q];
            m[2][q] = d[2][q] - d[1][q];
            m[3][q] = d[1][q] - d[3][q];
        }
        // a thread stores its 4 channels' h values (8 B) into the h rows and the l values into the l rows: the 8 lanes of a tile fill one
        // 64-byte row, the 8 tiles of the wave one 512-byte run per instruction.  (Pairing lanes into 16-byte stores cost 32 LDS
        // permutes and ~140 selects per thread and was slower.)
        const int kc = c >> 5;
        unsigned char *oh = Vs + (size_t)kc * 2 * tile + ((size_t)t_off + t) * 64 + (c & 31) * 2;
        const size_t pstride = (size_t)KC * 2 * tile;  // one Winograd plane
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // (B^T d) B
            const f32x4 vv[4] = {m[u][0] - m[u][2], m[u][1] + m[u][2], m[u][2] - m[u][1], m[u][1] - m[u][3]};
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4 xs = vv[v] * sv;
                const wi_h16x4 h = __builtin_convertvector(xs, wi_h16x4);
                const wi_h16x4 l = __builtin_convertvector(xs - __builtin_convertvector(h, f32x4), wi_h16x4);
                *reinterpret_cast<wi_h16x4 *>(oh) = h;
                *reinterpret_cast<wi_h16x4 *>(oh + tile) = l;
                oh += pstride;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 2. 16-plane GEMM with the output transform folded into the accumulators.
// ------------------------------------------------------------------------------------------------
struct WinoArgs {
    const float *V;      // [16][T][C]
    const float *U;      // [16][Cout][C]
    const float *scale, *shift;
    const float *gate;   // optional [B, Hl, Wl, Cout]: zero the output where gate <= 0
    float *y;            // [B, Hl, Wl, Cout]
    int T, C, Cout, B, Hl, Wl, Ty, Tx, act;
    const __bf16 *U3;    // precision 2: U split into bf16 planes, chunk-major [16][C/32][3][Cout][32]
    float *y_amax;       // optional [B]: raised to max |y[b]| (a3d_conv_desc.y_amax)
    const float *in_amax, *in_amax2;  // precision 3: per-image maxima of the conv input(s)
    float w_scale;                    // precision 3: scale of the pre-split filter planes in U3
    float *M;                         // plane-split form: [16][T][Cout] per-plane products (a3d_conv_desc.wino_m)
    int abl;                          // developer builds (-DA3D_ABLATIONS, env A3D_WINO_ABL): timing-only variants of the ring loop; 0 otherwise
    int Toff, Ttot;                   // precision 3: the layer's tiles are the slice [Toff, Toff + T) of a V buffer of Ttot tiles per run
    // multi-level launch (a3d_wino_gemm_levels; the kernel's ML form): T = Ttot = all levels' tiles, level k owns [t0[k], t0[k + 1])
    struct Levels {
        int n;
        int t0[6];
        int Ty[5], Tx[5], Hl[5], Wl[5];
        float *y[5];
        const float *in_amax[5];
        float *y_amax[5];
    } lv;
};
// level of tile t and its index inside the level (n <= 5: four compares)
__device__ __forceinline__ void wino_level_of(const WinoArgs::Levels &L, const int t, int &l, int &tl) {
    l = 0;
#pragma unroll
    for (int k = 1; k < 5; ++k) l = (k < L.n && t >= L.t0[k]) ? k : l;
    int base = L.t0[0];
#pragma unroll
    for (int k = 1; k < 5; ++k) base = l == k ? L.t0[k] : base;
    tl = t - base;
}
template <typename V>
__device__ __forceinline__ V wino_level_sel(const V (&arr)[5], const int l) {
    V v = arr[0];
#pragma unroll
    for (int k = 1; k < 5; ++k) v = l == k ? arr[k] : v;
    return v;
}

typedef _Float16 wh16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 wh16x8 __attribute__((ext_vector_type(8)));
// power-of-two scale of the V rows of image b: |B^T d B| <= 4 max |d|
__device__ __forceinline__ float wino_v_scale(const WinoArgs &a, const int b) {
    float m = a.in_amax[b];
    if (a.in_amax2) m = fmaxf(m, a.in_amax2[b]);
    return 0.25f * a3d_pow2_scale(m);  // (4 m itself could overflow)
}

template <int TN, int BKT>
__global__ __launch_bounds__(256, TN == 1 ? 3 : 1) void wino_gemm_kernel(const WinoArgs a, const int ntiles, const int nblk) {
    constexpr int BM = 64, BN = 2 * TN * 32;  // 64 Winograd tiles x BN channels; waves 2 (tiles) x 2 (channels)
    constexpr int LK = BKT + 4, TPR = BKT / 4, RPP = 256 / TPR;
    constexpr int XR = (BM + RPP - 1) / RPP, WR = BN / RPP;
    constexpr int BUF = (BM + BN) * LK;
    static_assert(BN % RPP == 0 && (BM % RPP == 0 || BM < RPP), "loader pass must tile the operand rows");
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
    __shared__ __attribute__((aligned(16))) float ss[2 * BN];  // scale | shift of this N tile, staged once (epilogue)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int logical = a3d_xcd_remap(blockIdx.x, nblk);
    const int mt = logical / ntiles, nt = logical - mt * ntiles;
    const int t0 = mt * BM, n0 = nt * BN;
    const int lr = tid / TPR, lc = (tid % TPR) * 4;
    const size_t vplane = (size_t)a.T * a.C, uplane = (size_t)a.Cout * a.C;
    const unsigned vbytes = (unsigned)(vplane * 4), ubytes = (unsigned)(uplane * 4);
    const int KC = a.C / BKT;      // k-chunks per frequency plane
    const int NIT = 16 * KC;       // flat (f, kc) iteration space

    // loop-invariant row offsets; rows past T / Cout (and loader lanes past the 64 tile rows) read as zero
    int xoff[XR], woff[WR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int rrow = lr + RPP * i;
        const int t = t0 + rrow;
        xoff[i] = (rrow < BM && t < a.T) ? (t * a.C + lc) * 4 : -1;
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int n = n0 + lr + RPP * i;
        woff[i] = n < a.Cout ? (n * a.C + lc) * 4 : -1;
    }
    // Two register staging sets: the loads of chunk c are issued at iteration c-3 into set (c & 1), written to LDS at
    // iteration c-1 and multiplied at iteration c, so every buffer load has two full iterations (32 MFMAs) to land.
    f32x4 xsA[XR], wsA[WR], xsB[XR], wsB[WR];
    int ld_f = 0, ld_kc = 0;  // (frequency plane, k-chunk) of the next chunk to load: incremental, no divisions
    auto load_chunk = [&](f32x4 (&xs)[XR], f32x4 (&ws)[WR]) {
        // (the two look-ahead loads past the last chunk re-read plane 15: in range, never consumed; keeping them
        // unconditional keeps the loop body free of branches)
        const int f = min(ld_f, 15);
        const __amdgpu_buffer_rsrc_t rv = wmake_rsrc(a.V + (size_t)f * vplane, vbytes);
        const __amdgpu_buffer_rsrc_t ru = wmake_rsrc(a.U + (size_t)f * uplane, ubytes);
        const int soff = ld_kc * BKT * 4;
#pragma unroll
        for (int i = 0; i < XR; ++i) xs[i] = wbuf_load4(rv, xoff[i], soff);
#pragma unroll
        for (int i = 0; i < WR; ++i) ws[i] = wbuf_load4(ru, woff[i], soff);
        if (++ld_kc == KC) {
            ld_kc = 0;
            ++ld_f;
        }
    };
    auto store_chunk = [&](int buf, const f32x4 (&xs)[XR], const f32x4 (&ws)[WR]) {
        float *X = lds + buf * BUF;
        float *Wt = X + BM * LK;
#pragma unroll
        for (int i = 0; i < XR; ++i)
            if (BM >= RPP || lr < BM) *reinterpret_cast<f32x4 *>(X + (lr + RPP * i) * LK + lc) = xs[i];
#pragma unroll
        for (int i = 0; i < WR; ++i) *reinterpret_cast<f32x4 *>(Wt + (lr + RPP * i) * LK + lc) = ws[i];
    };

    f32x16 mf[TN];        // M_f of the current frequency plane
    f32x16 yy[4][TN];     // the four outputs of each 2x2 tile: index 2*i + j
#pragma unroll
    for (int n = 0; n < TN; ++n) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            mf[n][r] = 0.f;
            yy[0][n][r] = yy[1][n][r] = yy[2][n][r] = yy[3][n][r] = 0.f;
        }
    }

    if (tid < BN) {  // epilogue vectors of this N tile -> LDS (visible through the barriers of the main loop)
        const int n = n0 + tid;
        ss[tid] = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
        ss[BN + tid] = (a.shift && n < a.Cout) ? a.shift[n] : 0.f;
    }
    load_chunk(xsA, wsA);  // chunk 0
    store_chunk(0, xsA, wsA);
    load_chunk(xsB, wsB);  // chunk 1
    load_chunk(xsA, wsA);  // chunk 2
    __syncthreads();

    const int frag_off = (lane & 31) * LK + (lane >> 5) * 4;
    // fold the finished plane f into the outputs with the 0/+-1 coefficients of A^T (x) A^T, and clear it.
    // The coefficients are wave-uniform and the unconditional FMAs exact, so this is one basic block.
    auto fold = [&](f32x16 (&m)[TN], const int f) {
        const int u = f >> 2, v = f & 3;
        const float au0 = (u < 3) ? 1.f : 0.f, au1 = (u == 0) ? 0.f : ((u == 1) ? 1.f : -1.f);
        const float av0 = (v < 3) ? 1.f : 0.f, av1 = (v == 0) ? 0.f : ((v == 1) ? 1.f : -1.f);
        const float c00 = au0 * av0, c01 = au0 * av1, c10 = au1 * av0, c11 = au1 * av1;
#pragma unroll
        for (int n = 0; n < TN; ++n) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = m[n][r];
                yy[0][n][r] = __builtin_fmaf(c00, x, yy[0][n][r]);
                yy[1][n][r] = __builtin_fmaf(c01, x, yy[1][n][r]);
                yy[2][n][r] = __builtin_fmaf(c10, x, yy[2][n][r]);
                yy[3][n][r] = __builtin_fmaf(c11, x, yy[3][n][r]);
                m[n][r] = 0.f;
            }
        }
    };
    // one chunk: multiply LDS[cur] into `acc`; write the staged chunk it+1 to LDS[cur^1]; refill that set with chunk it+3.
    auto mma = [&](f32x16 (&acc)[TN], const int cur, f32x4 (&xs)[XR], f32x4 (&ws)[WR]) {
        const float *X = lds + cur * BUF + (wm * 32) * LK + frag_off;
        const float *Wt = lds + cur * BUF + BM * LK + (wn * TN * 32) * LK + frag_off;
        f32x4 fa[2][TN], fb[2];
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) fa[0][ni] = *reinterpret_cast<const f32x4 *>(Wt + ni * 32 * LK);
        fb[0] = *reinterpret_cast<const f32x4 *>(X);
        store_chunk(cur ^ 1, xs, ws);
#pragma unroll
        for (int q = 0; q < BKT / 8; ++q) {
            const int fc = q & 1, fn = fc ^ 1;
            if (q + 1 < BKT / 8) {
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) fa[fn][ni] = *reinterpret_cast<const f32x4 *>(Wt + ni * 32 * LK + (q + 1) * 8);
                fb[fn] = *reinterpret_cast<const f32x4 *>(X + (q + 1) * 8);
            }
            if (q == 0) load_chunk(xs, ws);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[fc][ni][j], fb[fc][j], acc[ni], 0, 0, 0);
        }
    };
    // Measured and rejected here: a separable split of the input transform (the input kernel writes only the x part, R = d B per
    // pixel row, 2x the input instead of 4x; this loader finishes V = R[ra] +- R[rb] with two loads and one FMA per float4:
    // same accuracy, input kernel 40 % cheaper, but the GEMM loses 15 % on 256-channel layers to the doubled L2 -> L1 / HBM
    // reads -- net +2 % on p2 256->256, -13 % on res2 64->64, i.e. 0.4 ms per step: not worth a second code path); a persistent form (3 workgroups per CU walking tiles, chunk stream continuing across tile
    // boundaries: bit-identical, 7-14 % slower -- the tile switch inside the loader disturbs this loop); a second (ping-pong) plane accumulator that folds plane f behind the MFMAs of plane
    // f+1 (2-8 % slower: +16 registers, no gain -- the fold is not what idles the pipe); one staging set at 4 waves / SIMD
    // (__launch_bounds__(256, 4) fits 116 VGPRs: ties with this form).
    int cf = 0, ckc = 0;  // (plane, chunk) being multiplied
    for (int it = 0; it < NIT; it += 2) {  // NIT = 16*KC is even
        mma(mf, 0, xsB, wsB);  // chunk it   in LDS[0]; set B holds chunk it+1, is refilled with chunk it+3
        if (++ckc == KC) {
            ckc = 0;
            fold(mf, cf++);
        }
        __syncthreads();
        mma(mf, 1, xsA, wsA);  // chunk it+1 in LDS[1]; set A holds chunk it+2, is refilled with chunk it+4
        if (++ckc == KC) {
            ckc = 0;
            fold(mf, cf++);
        }
        __syncthreads();
    }

    // ---- epilogue: lane owns tile t, register quad = 4 consecutive channels -------------------------
    const int t = t0 + wm * 32 + (lane & 31);
    if (t >= a.T) return;
    const int tx = t % a.Tx;
    const int r = t / a.Tx;
    const int ty = r % a.Ty, b = r / a.Ty;
#pragma unroll
    for (int ij = 0; ij < 4; ++ij) {
        const int oy = 2 * ty + (ij >> 1), ox = 2 * tx + (ij & 1);
        if (oy >= a.Hl || ox >= a.Wl) continue;
        const size_t ooff = (((size_t)b * a.Hl + oy) * a.Wl + ox) * a.Cout;
        float *orow = a.y + ooff;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int nl = (wn * TN + ni) * 32 + rg * 8 + (lane >> 5) * 4;
                const int n = n0 + nl;
                if (n >= a.Cout) continue;
                f32x4 v = {yy[ij][ni][rg * 4 + 0], yy[ij][ni][rg * 4 + 1], yy[ij][ni][rg * 4 + 2], yy[ij][ni][rg * 4 + 3]};
                // scale / shift from LDS: no global load (and no wait on the stores before it) between two output quads;
                // one fused multiply-add, the same rounding as the direct kernels' epilogue (a3d_epilogue_math)
                const f32x4 sc = *reinterpret_cast<const f32x4 *>(ss + nl), sh = *reinterpret_cast<const f32x4 *>(ss + BN + nl);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = __builtin_fmaf(v[k], sc[k], sh[k]);
                if (a.act == A3D_ACT_RELU) {
                    for (int k = 0; k < 4; ++k) v[k] = v[k] <= 0.f ? 0.f : v[k];
                } else if (a.act == A3D_ACT_LEAKY) {
                    for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.01f * v[k];
                }
                if (a.gate) {
                    const f32x4 g = *reinterpret_cast<const f32x4 *>(a.gate + ooff + n);
                    for (int k = 0; k < 4; ++k) v[k] = g[k] > 0.f ? v[k] : 0.f;
                }
                *reinterpret_cast<f32x4 *>(orow + n) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 2x. The same 16-plane GEMM with fp32-grade products on the bf16 matrix pipe (a3d_conv_desc.precision == 2; the arithmetic
// is conv_bf16x3.hip's: every fp32 operand split exactly into hi | mid | lo bf16 terms while it is staged in LDS, six
// v_mfma_f32_32x32x16_bf16 per 16-deep k step, fp32 accumulation).  64 tiles x 64 channels per workgroup, 32-deep chunks,
// the fold and the epilogue of wino_gemm_kernel unchanged (the 32x32 accumulator layout is the same).  LDS image per
// operand plane: [64 rows][32 k] bf16, 64-byte rows whose four 16-byte slots are XOR-swizzled with bits 2..3 of the row
// index (conflict-free ds_read_b128 under the 64-bank / 16-lane-group rule, conflict-free ds_write_b64).
// ------------------------------------------------------------------------------------------------
typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void wsplit3(const f32x4 v, wbf16x4 &h, wbf16x4 &m, wbf16x4 &l) {
    h = __builtin_convertvector(v, wbf16x4);
    const f32x4 r1 = v - __builtin_convertvector(h, f32x4);
    m = __builtin_convertvector(r1, wbf16x4);
    const f32x4 r2 = r1 - __builtin_convertvector(m, f32x4);
    l = __builtin_convertvector(r2, wbf16x4);
}

__global__ __launch_bounds__(256, 3) void wino_gemm_x3_kernel(const WinoArgs a, const int ntiles, const int nblk) {
    constexpr int TN = 1, BKT = 32;
    constexpr int BM = 64, BN = 64;
    constexpr int LKB = BKT;                      // bf16 elements per LDS row (64 bytes, swizzled, no padding)
    constexpr int TPR = BKT / 4, RPP = 256 / TPR;  // 8 lanes x float4 per row, 32 rows per loader pass
    constexpr int XR = BM / RPP;
    constexpr int PL = 64 * LKB;                  // one operand plane
    constexpr int BUF = 6 * PL;                   // X hi|mid|lo, W hi|mid|lo
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * BUF];
    __shared__ __attribute__((aligned(16))) float ss[2 * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int logical = a3d_xcd_remap(blockIdx.x, nblk);
    const int mt = logical / ntiles, nt = logical - mt * ntiles;
    const int t0 = mt * BM, n0 = nt * BN;
    const int lr = tid / TPR, lc = (tid % TPR) * 4;
    const size_t vplane = (size_t)a.T * a.C, uplane = (size_t)a.Cout * a.C;
    const unsigned vbytes = (unsigned)(vplane * 4), ubytes = (unsigned)(uplane * 4);
    const int KC = a.C / BKT;
    const int NIT = 16 * KC;

    int xoff[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int t = t0 + lr + RPP * i;
        xoff[i] = t < a.T ? (t * a.C + lc) * 4 : -1;
    }
    // weights: pre-split planes U3 [16][C/32][3][Cout][32] bf16 (a3d_split_bf16x3 at pack time); a lane moves one 16-byte
    // slot (8 k) of one row per plane: row = tid / 4, slot = tid % 4 -- the 64 rows of a plane are 4 KB contiguous
    const int wrow = tid >> 2, wq = tid & 3;
    const int woff = (n0 + wrow < a.Cout) ? (n0 + wrow) * 64 + wq * 16 : -1;
    const int wlds = wrow * LKB + (((wq ^ (wrow >> 2)) & 3) << 3);
    const unsigned u3tile = (unsigned)a.Cout * 64u;  // bytes of one (f, chunk, plane) tile
    // slot swizzle: k = 8q .. 8q+7 of row r lives in slot q ^ ((r >> 2) & 3); the loader rows lr + 32 i share (r >> 2) & 3
    const int lcs = ((((lc >> 3) ^ (lr >> 2)) & 3) << 3) | (lc & 7);
    f32x4 xsA[XR], wsA[3], xsB[XR], wsB[3];
    int ld_f = 0, ld_kc = 0;
    auto load_chunk = [&](f32x4 (&xs)[XR], f32x4 (&ws)[3]) {
        const int f = min(ld_f, 15);
        const __amdgpu_buffer_rsrc_t rv = wmake_rsrc(a.V + (size_t)f * vplane, vbytes);
        const __amdgpu_buffer_rsrc_t ru = wmake_rsrc(reinterpret_cast<const float *>(a.U3 + (size_t)f * 3 * uplane), (unsigned)(uplane * 6));
        const int soff = ld_kc * BKT * 4;
#pragma unroll
        for (int i = 0; i < XR; ++i) xs[i] = wbuf_load4(rv, xoff[i], soff);
#pragma unroll
        for (int p = 0; p < 3; ++p) ws[p] = wbuf_load4(ru, woff, (ld_kc * 3 + p) * (int)u3tile);
        if (++ld_kc == KC) {
            ld_kc = 0;
            ++ld_f;
        }
    };
    auto store_chunk = [&](int buf, const f32x4 (&xs)[XR], const f32x4 (&ws)[3]) {
        __bf16 *X = lds + buf * BUF;
        __bf16 *Wt = X + 3 * PL;
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            wbf16x4 h, m, l;
            wsplit3(xs[i], h, m, l);
            __bf16 *p = X + (lr + RPP * i) * LKB + lcs;
            *reinterpret_cast<wbf16x4 *>(p) = h;
            *reinterpret_cast<wbf16x4 *>(p + PL) = m;
            *reinterpret_cast<wbf16x4 *>(p + 2 * PL) = l;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<f32x4 *>(Wt + p * PL + wlds) = ws[p];
    };

    f32x16 mf[TN];
    f32x16 yy[4][TN];
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            mf[n][r] = 0.f;
            yy[0][n][r] = yy[1][n][r] = yy[2][n][r] = yy[3][n][r] = 0.f;
        }
    if (tid < BN) {
        const int n = n0 + tid;
        ss[tid] = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
        ss[BN + tid] = (a.shift && n < a.Cout) ? a.shift[n] : 0.f;
    }
    load_chunk(xsA, wsA);
    store_chunk(0, xsA, wsA);
    load_chunk(xsB, wsB);
    load_chunk(xsA, wsA);
    __syncthreads();

    const int frow = lane & 31;
    const int fsw = (frow >> 2) & 3;
    auto fold = [&](f32x16 (&m)[TN], const int f) {
        const int u = f >> 2, v = f & 3;
        const float au0 = (u < 3) ? 1.f : 0.f, au1 = (u == 0) ? 0.f : ((u == 1) ? 1.f : -1.f);
        const float av0 = (v < 3) ? 1.f : 0.f, av1 = (v == 0) ? 0.f : ((v == 1) ? 1.f : -1.f);
        const float c00 = au0 * av0, c01 = au0 * av1, c10 = au1 * av0, c11 = au1 * av1;
#pragma unroll
        for (int n = 0; n < TN; ++n) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = m[n][r];
                yy[0][n][r] = __builtin_fmaf(c00, x, yy[0][n][r]);
                yy[1][n][r] = __builtin_fmaf(c01, x, yy[1][n][r]);
                yy[2][n][r] = __builtin_fmaf(c10, x, yy[2][n][r]);
                yy[3][n][r] = __builtin_fmaf(c11, x, yy[3][n][r]);
                m[n][r] = 0.f;
            }
        }
    };
    // one chunk: 2 k steps x 6 product terms from LDS[cur]; the staged chunk it+1 is split into LDS[cur^1]; its set is refilled
    auto mma = [&](f32x16 (&acc)[TN], const int cur, f32x4 (&xs)[XR], f32x4 (&ws)[3]) {
        const __bf16 *X = lds + cur * BUF + (wm * 32 + frow) * LKB;
        const __bf16 *Wt = lds + cur * BUF + 3 * PL + (wn * 32 + frow) * LKB;
        // fragments of ONE 16-deep step at a time (24 registers; both steps at once cost 48 and spill at 3 workgroups per CU)
        wbf16x8 fa[3], fb[3];
        auto frags = [&](const int st) {
            const int slot = (((2 * st + (lane >> 5)) ^ fsw) & 3) << 3;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                fa[p] = *reinterpret_cast<const wbf16x8 *>(Wt + p * PL + slot);
                fb[p] = *reinterpret_cast<const wbf16x8 *>(X + p * PL + slot);
            }
        };
#define WX3(PA, PB) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA], fb[PB], acc[0], 0, 0, 0);
#define WX3_STEP WX3(0, 0) WX3(0, 1) WX3(1, 0) WX3(1, 1) WX3(2, 0) WX3(0, 2)
        frags(0);
        store_chunk(cur ^ 1, xs, ws);
        load_chunk(xs, ws);
        WX3_STEP
        frags(1);
        WX3_STEP
#undef WX3_STEP
#undef WX3
        // the split's VALU work, the LDS writes and the buffer loads go into the shadows of the 12 MFMAs
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
    };
    int cf = 0, ckc = 0;
    for (int it = 0; it < NIT; it += 2) {
        mma(mf, 0, xsB, wsB);
        if (++ckc == KC) {
            ckc = 0;
            fold(mf, cf++);
        }
        __syncthreads();
        mma(mf, 1, xsA, wsA);
        if (++ckc == KC) {
            ckc = 0;
            fold(mf, cf++);
        }
        __syncthreads();
    }

    const int t = t0 + wm * 32 + (lane & 31);
    const bool tok = t < a.T;
    const int bimg = tok ? t / (a.Ty * a.Tx) : 0;  // all four outputs of a tile belong to one image
    float vmax = 0.f;
    if (tok) {
    const int tx = t % a.Tx;
    const int r = t / a.Tx;
    const int ty = r % a.Ty, b = r / a.Ty;
#pragma unroll
    for (int ij = 0; ij < 4; ++ij) {
        const int oy = 2 * ty + (ij >> 1), ox = 2 * tx + (ij & 1);
        if (oy >= a.Hl || ox >= a.Wl) continue;
        const size_t ooff = (((size_t)b * a.Hl + oy) * a.Wl + ox) * a.Cout;
        float *orow = a.y + ooff;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int nl = wn * 32 + rg * 8 + (lane >> 5) * 4;
            const int n = n0 + nl;
            if (n >= a.Cout) continue;
            f32x4 v = {yy[ij][0][rg * 4 + 0], yy[ij][0][rg * 4 + 1], yy[ij][0][rg * 4 + 2], yy[ij][0][rg * 4 + 3]};
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(ss + nl), sh = *reinterpret_cast<const f32x4 *>(ss + BN + nl);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = __builtin_fmaf(v[k], sc[k], sh[k]);
            if (a.act == A3D_ACT_RELU) {
                for (int k = 0; k < 4; ++k) v[k] = v[k] <= 0.f ? 0.f : v[k];
            } else if (a.act == A3D_ACT_LEAKY) {
                for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.01f * v[k];
            }
            if (a.gate) {
                const f32x4 g = *reinterpret_cast<const f32x4 *>(a.gate + ooff + n);
                for (int k = 0; k < 4; ++k) v[k] = g[k] > 0.f ? v[k] : 0.f;
            }
            vmax = fmaxf(vmax, a3d_absmax4(v));
                *reinterpret_cast<f32x4 *>(orow + n) = v;
        }
    }
    }
    if (a.y_amax) a3d_note_amax(a.y_amax, bimg, vmax, tok);  // (every lane of the wave gets here)
}
// ------------------------------------------------------------------------------------------------
// 2x-wide.  wino_gemm_x3_kernel above moves too many operand bytes per MFMA and waits for its fragments: every 64-tile block
// streams all of U3 (6 B per weight) from L2 -- 30 GB + 20 GB of V per p2 layer, ~15 TB/s, the rate the L2s deliver -- and reads
// one ds_read_b128 per MFMA into a single fragment set (0.45 matrix-pipe occupancy).  This form:
//   * 128 tiles x 128 channels per workgroup of 512 threads (8 waves as 4 x 2, wave tile 32 tiles x 64 channels = 24 MFMAs per
//     wave and 32-deep chunk): L2 -> CU traffic 50 -> 25 GB per p2 layer, fragment reads per MFMA 1 -> 0.75, V is split by Cout/128
//     workgroups instead of Cout/64.  160 accumulator registers per wave (mf[2] + the 4 x 2 folded outputs) -> 2 waves per SIMD =
//     one workgroup per CU, 96 KiB of LDS (two stages of 3 x (128 + 128) rows x 64 B).
//   * the pre-split weight planes go global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`: no VGPRs, no VGPR -> LDS store path,
//     the XOR slot swizzle applied on the global side of each lane's address); V still passes through registers (it is split there).
//   * ONE barrier per chunk, between its two 16-deep steps, and fragment reads issued a whole step (12 MFMAs) ahead of their use
//     -- see "Schedule" in the kernel.
// Timing-only ablations of the final loop on the p2 256 -> 256 layer (64 frames, 3.50 ms; MFMA time alone 2.1 ms at the 1.75 GHz the
// chip holds): without the barrier and the DMA wait -2 %; without the weight DMA -7 %; without the V loads -14 % (also when every
// workgroup reads the same, L2-resident V rows, so it is not HBM latency); one extra 4-byte DMA per wave and chunk as an L2
// prefetch +7 % -- i.e. the residual is the issue cost of the vector-memory instructions themselves (~100 cycles each inside the
// MFMA stream; MI355X_MICROARCH.md quotes 60-185 for an LDS-DMA piece), 5 per wave and chunk, fixed by bytes per MFMA at
// 1 KiB per instruction.  Staggering them between the two waves of a SIMD (scalar branches) was 19 % slower; the barrier two terms
// later 5 % slower.
// The per-output operation order (planes, k chunks, 16-deep steps, the six product terms) is that of wino_gemm_x3_kernel: the two
// kernels agree bit for bit (tests/test_gpu_parity.py), so the launcher chooses by problem size.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wuni_rsrc(const void *p, unsigned bytes) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ void wdma16(__amdgpu_buffer_rsrc_t r, __bf16 *lds_dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds_dst, 16, voff, soff, 0, 0);
}

constexpr int X3W_BN = 128, X3W_LKB = 32;
constexpr int x3w_buf(int WM, int NP) { return NP * (32 * WM + X3W_BN) * X3W_LKB; }  // 16-bit elements of one stage: X and W, NP planes each
// stages of the operand ring: the bf16x3 form double-buffers (V passes through registers); the fp16x2 form receives BOTH operands
// pre-split by LDS-DMA and keeps 3 (two workgroups per CU) or 4 (one) stages in flight
constexpr int x3w_stages(int WM, int NP) { return NP == 2 ? (WM == 2 ? 3 : 4) : 2; }
constexpr int x3w_lds_bytes(int WM, int NP) { return x3w_stages(WM, NP) * x3w_buf(WM, NP) * 2 + 2 * X3W_BN * 4; }

// WM = wave rows: 2 -> 64 tiles x 128 channels, 256 threads, two workgroups per CU; 4 -> 128 tiles x 128 channels, 512 threads, one
// workgroup per CU (every 64-tile block streams all of U3 -- 6 B per weight -- from L2: 30 GB per p2 layer; 128-tile blocks halve it)
// F16: the fp16x2 arithmetic (a3d_conv_desc.precision == 3): two operand planes, three product terms per step; V rows are scaled by the
// power of two of 4 x their image's input maximum (|B^T d B| <= 4 max |d|), U3 holds the filter pre-split by a3d_split_f16x2_chunk.
// PS (plane-split, small problems): blockIdx.y = Winograd plane; the workgroup runs that plane's k loop only and stores the raw product
// tile to a.M [16][T][Cout]; wino_fold_kernel then folds the 16 planes in the same order and applies the same epilogue.  A problem of
// a few tile blocks otherwise occupies a few CUs for 16 x C/32 latency-bound iterations (a single 30x40 frame: 6 workgroups, 116 us).
// The ping-pong form's epilogue is TABLE-DRIVEN (round 5; WinoArgs::lv): a launch may cover the tiles of several maps that share the filter
// (a3d_wino_gemm_levels: the RPN conv over the pyramid levels; a single layer is a table of one) -- the loop walks the tiles of the
// concatenated V, the epilogue looks every tile's map up (output tensor, per-image scale, recorded maxima).
template <int WM, bool F16 = false, bool PS = false, int PP = 0>
__global__ __launch_bounds__(128 * WM, WM == 2 ? 2 : 1) void wino_gemm_x3w_kernel(const WinoArgs a, const int ntiles, const int nblk) {
    static_assert(PP == 0 || (F16 && !PS && WM == 4), "ping-pong: the 512-thread fp16x2 form (two waves per SIMD)");
    constexpr bool ML = PP > 0;
    constexpr bool DMAV = F16;  // fp16x2: V arrives pre-split and pre-scaled, global -> LDS by DMA (every F16 schedule below)
    constexpr int NP = F16 ? 2 : 3;
    constexpr int TN = 2, BKT = 32, BM = 32 * WM, BN = X3W_BN, LKB = X3W_LKB, NT = 128 * WM, NW = 2 * WM;
    constexpr int TPR = BKT / 4, RPP = NT / TPR, XR = BM / RPP;  // 8 lanes x float4 per row, BM/2 rows per pass, 2 passes
    constexpr int PLX = BM * LKB, PLW = BN * LKB, BUF = x3w_buf(WM, NP);
    constexpr int DPW = 8 * NP / NW;  // weight DMA instructions per wave and chunk
    static_assert(XR == 2, "the counted vmcnt waits below assume two V loads per chunk");
    constexpr int NST = x3w_stages(WM, NP);
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    float *ss = reinterpret_cast<float *>(lds + NST * BUF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int logical = a3d_xcd_remap(blockIdx.x, nblk);
    const int mt = logical / ntiles, nt = logical - mt * ntiles;
    const int t0 = mt * BM, n0 = nt * BN;
    const int lr = tid / TPR, lc = (tid % TPR) * 4;
    const size_t vplane = (size_t)a.T * a.C;
    const unsigned vbytes = (unsigned)(vplane * 4);
    const int KC = a.C / BKT;
    const int NIT = 16 * KC;
    const int pf = PS ? (int)blockIdx.y : 0;   // plane of a plane-split workgroup
#ifdef A3D_ABLATIONS
    if (a.abl & 128) return;                    // timing-only: the launch alone
    const int nit = (a.abl & 256) ? 4 : (PS ? KC : NIT);  // timing-only: prologue + four chunks
#else
    const int nit = PS ? KC : NIT;             // chunks this workgroup multiplies
#endif

    int xoff[XR];
    float sxr[XR];  // fp16x2: scale of each loader row (tile) = that of its image
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int t = t0 + lr + RPP * i;
        xoff[i] = t < a.T ? (t * a.C + lc) * 4 : -1;
        // (F16 launches that move V by LDS-DMA never scale a loader row -- V is stored already scaled -- and the multi-level launcher hands
        // them no in_amax / Ty / Tx at all: the read must not exist there, not merely be dead)
        sxr[i] = (F16 && !DMAV && t < a.T) ? wino_v_scale(a, t / (a.Ty * a.Tx)) : 1.f;
    }
    const int lcs = ((((lc >> 3) ^ (lr >> 2)) & 3) << 3) | (lc & 7);  // (RPP is a multiple of 16: both passes share the swizzle)
    // weights: U3 [16][C/32][3][Cout][32] bf16; one (f, chunk, plane) tile of this workgroup's 128 rows is an 8 KiB run = 8 DMA
    // wave-instructions of 16 rows.  Lane i of an instruction lands at LDS byte 16 i of its 1 KiB = row i/4, slot i%4, and
    // fetches the k slot that the image keeps there: slot ^ ((row >> 2) & 3)  (row base is a multiple of 16)
    const __amdgpu_buffer_rsrc_t ru = wuni_rsrc(a.U3, (unsigned)((size_t)16 * a.Cout * a.C * 2 * NP));
    const int wvoff = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    const int u3tile = a.Cout * 64;  // bytes of one (f, chunk, plane) tile
    int dma_c = pf * KC;             // flat (f, kc) index of the next weight chunk to fetch
    auto dma_w = [&](const int buf) {
        // 24 instructions per chunk: instruction j = plane j/8, row group j%8
        __bf16 *Wt = lds + buf * BUF + NP * PLX;
        const int base = __builtin_amdgcn_readfirstlane(min(dma_c, NIT - 1) * NP * u3tile + n0 * 64);
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const int j = wave * DPW + i;
            const int p = j >> 3, g = j & 7;
            wdma16(ru, Wt + p * PLW + g * 16 * LKB, wvoff, base + __builtin_amdgcn_readfirstlane(p * u3tile + g * 1024));
        }
        ++dma_c;
    };
    // fp16x2: V arrives pre-split too -- wino_input_h2_kernel wrote it as fp16 planes, chunk-major [16][C/32][2][T][32], scaled per image
    // (wino_v_scale) -- and takes the same road as the filter: a (f, chunk, plane) tile of this workgroup's BM rows is a contiguous run
    // of BM / 16 DMA instructions (16 rows x 64 B each), two instructions per wave and chunk, with the filter's slot swizzle.  No V
    // registers, no split in the loop, no VGPR -> LDS stores.  (Rows past T read the next plane's rows or zeros: never stored.)
    constexpr int DPV = 2;
    static_assert(2 * (BM / 16) == DPV * NW, "two V DMA instructions per wave and chunk");
    auto dma_v = [&](const int buf, const int c) {  // c = flat (f, kc) index of the chunk (clamped like the filter's)
        __bf16 *X = lds + buf * BUF;
        const size_t tile = (size_t)a.Ttot * 64;  // bytes of one (f, chunk, plane) tile (Ttot = T unless the layer's tiles are a slice)
        const __amdgpu_buffer_rsrc_t rv = wuni_rsrc(reinterpret_cast<const char *>(a.V) + (size_t)min(c, NIT - 1) * 2 * tile, (unsigned)(2 * tile));
#pragma unroll
        for (int i = 0; i < DPV; ++i) {
            const int j = wave * DPV + i;
            const int p = j / (BM / 16), g = j % (BM / 16);
            wdma16(rv, X + p * PLX + g * 16 * LKB, wvoff, __builtin_amdgcn_readfirstlane(p * (int)tile + (a.Toff + t0 + g * 16) * 64));
        }
    };
    f32x4 xsA[XR], xsB[XR];
    int ld_f = pf, ld_kc = 0;
    auto load_chunk = [&](f32x4 (&xs)[XR]) {
        const int f = min(ld_f, 15);
        const __amdgpu_buffer_rsrc_t rv = wuni_rsrc(a.V + (size_t)f * vplane, vbytes);
        const int soff = ld_kc * BKT * 4;
#pragma unroll
        for (int i = 0; i < XR; ++i) xs[i] = wbuf_load4(rv, xoff[i], soff);
        if (++ld_kc == KC) {
            ld_kc = 0;
            ++ld_f;
        }
    };
    struct Split {
        wbf16x4 h, m, l;  // (fp16x2: h, m hold the two fp16 planes' bits)
    };
    auto split = [&](const f32x4 v, const int i, Split &o) {
        if constexpr (F16) {
            const f32x4 xs = v * sxr[i];
            const wh16x4 h = __builtin_convertvector(xs, wh16x4);
            const wh16x4 l = __builtin_convertvector(xs - __builtin_convertvector(h, f32x4), wh16x4);
            o.h = __builtin_bit_cast(wbf16x4, h);
            o.m = __builtin_bit_cast(wbf16x4, l);
        } else {
            wsplit3(v, o.h, o.m, o.l);
        }
    };
    auto put = [&](const int buf, const int i, const Split &v) {  // the planes of loader row lr + RPP i
        __bf16 *p = lds + buf * BUF + (lr + RPP * i) * LKB + lcs;
        *reinterpret_cast<wbf16x4 *>(p) = v.h;
        *reinterpret_cast<wbf16x4 *>(p + PLX) = v.m;
        if constexpr (!F16) *reinterpret_cast<wbf16x4 *>(p + 2 * PLX) = v.l;
    };

    f32x16 mf[TN];
    f32x16 yy[4][TN];
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            mf[n][r] = 0.f;
            yy[0][n][r] = yy[1][n][r] = yy[2][n][r] = yy[3][n][r] = 0.f;
        }
    if (tid < BN) {
        const int n = n0 + tid;
        ss[tid] = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
        ss[BN + tid] = (a.shift && n < a.Cout) ? a.shift[n] : 0.f;
    }

    // fragments: row = lane % 32 of the wave's tile / channel block, k = 8 * (lane / 32) .. + 7 of 16-deep step st
    const int frow = lane & 31;
    const int fsw = (frow >> 2) & 3;
    const __bf16 *fX = lds + (wm * 32 + frow) * LKB;
    const __bf16 *fW = lds + NP * PLX + (wn * 64 + frow) * LKB;
    struct Frags {
        wbf16x8 a[NP][TN], b[NP];
    };
    auto rdA = [&](Frags &F, const int buf, const int st, const int p) {
        const int slot = (((2 * st + (lane >> 5)) ^ fsw) & 3) << 3;
#pragma unroll
        for (int n = 0; n < TN; ++n) F.a[p][n] = *reinterpret_cast<const wbf16x8 *>(fW + buf * BUF + p * PLW + n * 32 * LKB + slot);
    };
    auto rdB = [&](Frags &F, const int buf, const int st, const int p) {
        const int slot = (((2 * st + (lane >> 5)) ^ fsw) & 3) << 3;
        F.b[p] = *reinterpret_cast<const wbf16x8 *>(fX + buf * BUF + p * PLX + slot);
    };
    auto fold = [&](f32x16 (&m)[TN], const int f) {
        const int u = f >> 2, v = f & 3;
        const float au0 = (u < 3) ? 1.f : 0.f, au1 = (u == 0) ? 0.f : ((u == 1) ? 1.f : -1.f);
        const float av0 = (v < 3) ? 1.f : 0.f, av1 = (v == 0) ? 0.f : ((v == 1) ? 1.f : -1.f);
        const float c00 = au0 * av0, c01 = au0 * av1, c10 = au1 * av0, c11 = au1 * av1;
#pragma unroll
        for (int n = 0; n < TN; ++n) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = m[n][r];
                yy[0][n][r] = __builtin_fmaf(c00, x, yy[0][n][r]);
                yy[1][n][r] = __builtin_fmaf(c01, x, yy[1][n][r]);
                yy[2][n][r] = __builtin_fmaf(c10, x, yy[2][n][r]);
                yy[3][n][r] = __builtin_fmaf(c11, x, yy[3][n][r]);
                m[n][r] = 0.f;
            }
        }
    };

    // Schedule.  Chunk c lives in stage c % 2; an iteration is the two 16-deep steps of one chunk, each six product terms of two
    // MFMAs, with ONE barrier between the steps:
    //   step 0 (fragments S0(c) in registers): reads the fragments S1(c); splits the staged V chunk c+1 into stage (c+1) % 2
    //   -- wait: weight DMA of chunk c+1 landed (counted vmcnt), own LDS traffic drained; barrier --
    //   step 1 (S1(c)): reads S0(c+1) from the other stage; issues the weight DMA of chunk c+2 into stage c % 2 (nobody reads it any
    //   more: every wave drained its S1(c) reads before the barrier) and the V loads of chunk c+3 (two register sets, consumed at
    //   step 0 two iterations later); folds the plane when it ends.  (The barrier two terms later -- so that the last S1 reads are
    //   not waited for right after their issue -- measured 5 % slower.)
    // Fragment reads run one full step (12 MFMAs) ahead of their use wherever a register set is free: S0 a0 b0 b1 at the first
    // term, a1 after the fourth (the current a1 b1 are dead), a2 b2 after the fifth -- with a single fragment set the reads of a
    // step could only start after its predecessor's last MFMA had issued, and the exposed LDS latency cost 29 % of the kernel
    // (timing-only ablation: 3.80 -> 2.69 ms without the reads).
#define X3W_FENCE __builtin_amdgcn_sched_barrier(0);
#define X3W_MFMA(C, A, Bv)                                                                                                                \
    if constexpr (F16) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wh16x8, A), __builtin_bit_cast(wh16x8, Bv), C, 0, 0, 0); \
    else C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bv, C, 0, 0, 0);
#define X3W_TERM(F, PA, PB)              \
    X3W_MFMA(mf[0], F.a[PA][0], F.b[PB]) \
    X3W_MFMA(mf[1], F.a[PA][1], F.b[PB])
// (one MFMA, then its share of the block's other instructions: the wave issues in order, so what follows an MFMA runs in its shadow)
#define X3W_MIX(NV)                                        \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);    \
    __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);     \
    __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);
    auto step0 = [&](const int cur, Frags &F, Frags &G, const f32x4 (&xs)[XR]) {
        Split s0, s1;
        if constexpr (F16) {  // three terms: h.h, h.l, l.h
            X3W_TERM(F, 0, 0)
            rdA(G, cur, 1, 0);
            rdB(G, cur, 1, 0);
            rdB(G, cur, 1, 1);
            split(xs[0], 0, s0);
            X3W_MIX(10)
            X3W_FENCE
            X3W_TERM(F, 0, 1)
            put(cur ^ 1, 0, s0);
            split(xs[1], 1, s1);
            X3W_MIX(10)
            X3W_FENCE
            X3W_TERM(F, 1, 0)
            put(cur ^ 1, 1, s1);
            rdA(G, cur, 1, 1);
            X3W_MIX(4)
            X3W_FENCE
            return;
        }
        X3W_TERM(F, 0, 0)
        rdA(G, cur, 1, 0);
        rdB(G, cur, 1, 0);
        rdB(G, cur, 1, 1);
        split(xs[0], 0, s0);
        X3W_MIX(12)
        X3W_FENCE
        X3W_TERM(F, 0, 1)
        put(cur ^ 1, 0, s0);
        X3W_MIX(4)
        X3W_FENCE
        X3W_TERM(F, 1, 0)
        split(xs[1], 1, s1);
        X3W_MIX(12)
        X3W_FENCE
        X3W_TERM(F, 1, 1)
        put(cur ^ 1, 1, s1);
        X3W_MIX(4)
        X3W_FENCE
        X3W_TERM(F, 2, 0)
        rdA(G, cur, 1, 1);
        X3W_MIX(4)
        X3W_FENCE
        X3W_TERM(F, 0, 2)
        rdA(G, cur, 1, 2);
        rdB(G, cur, 1, 2);
        X3W_MIX(4)
        X3W_FENCE
    };
    auto step1 = [&](const int cur, Frags &F, Frags &G, f32x4 (&xs)[XR]) {
        if constexpr (F16) {
            X3W_TERM(F, 0, 0)
            dma_w(cur);
            rdA(G, cur ^ 1, 0, 0);
            rdB(G, cur ^ 1, 0, 0);
            rdB(G, cur ^ 1, 0, 1);
            X3W_MIX(6)
            X3W_FENCE
            X3W_TERM(F, 0, 1)
            load_chunk(xs);
            X3W_MIX(6)
            X3W_FENCE
            X3W_TERM(F, 1, 0)
            rdA(G, cur ^ 1, 0, 1);
            X3W_MIX(4)
            X3W_FENCE
            return;
        }
        X3W_TERM(F, 0, 0)
        dma_w(cur);
        rdA(G, cur ^ 1, 0, 0);
        rdB(G, cur ^ 1, 0, 0);
        rdB(G, cur ^ 1, 0, 1);
        X3W_MIX(6)
        X3W_FENCE
        X3W_TERM(F, 0, 1)
        load_chunk(xs);
        X3W_MIX(6)
        X3W_FENCE
        X3W_TERM(F, 1, 0)
        X3W_TERM(F, 1, 1)
        X3W_FENCE
        X3W_TERM(F, 2, 0)
        rdA(G, cur ^ 1, 0, 1);
        X3W_MIX(4)
        X3W_FENCE
        X3W_TERM(F, 0, 2)
        rdA(G, cur ^ 1, 0, 2);
        rdB(G, cur ^ 1, 0, 2);
        X3W_MIX(4)
        X3W_FENCE
    };

    Frags F0, F1;
    if constexpr (F16) {
        // fp16x2 schedule.  Chunk c lives in stage c % NST.  An iteration = the two 16-deep steps of one chunk (three product terms of
        // two MFMAs each) with ONE barrier between them.  Behind that barrier every wave has read all fragments of chunk c (S0(c)
        // during the previous iteration, S1(c) during step 0), so the DMA of chunk c + NST goes straight into the stage of chunk c:
        // it has NST - 1 iterations to land (two stages gave one: 12 MFMAs ~ 0.2 us against an L2 round trip of 0.5 - 1 us -- the
        // waves sat at this wait in every chunk).  The barrier is a BARE s_barrier: __syncthreads() carries a fence, and in front of a
        // fence the compiler completes every LDS-DMA in flight.
        constexpr int OPS = DPW + DPV;  // vector-memory instructions per wave and chunk
        int st = 0;                     // stage of the chunk being multiplied
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            dma_v(i, dma_c);
            dma_w(i);
        }
        __asm__ volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * OPS) : "memory");
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (scale / shift staged above)
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            rdA(F0, 0, 0, p);
            rdB(F0, 0, 0, p);
        }
        int cf = 0, ckc = 0;
        if constexpr (PP > 0) {
            // PING-PONG (round 5).  In the loop below all eight waves run the same instruction stream from the same barrier: the two
            // waves of a SIMD reach their DMA issue (60-185 cycles each, four per wave and chunk), their fragment waits and the barrier
            // together, and the matrix pipe idles through all three (0.375 busy; without the DMAs the launch is 20 % shorter).  Here
            // waves 4-7 (the SIMD partners of waves 0-3: waves go to SIMDs 0, 2, 1, 3, 0, ..) run HALF A CHUNK behind, and every wave
            // bunches its work: a MEMORY phase (the 12 fragment reads of a whole chunk + its 4 DMA pieces) and a COMPUTE phase (the
            // chunk's 12 MFMAs back to back, then the fold when a plane ends).  Between two barriers waves 0-3 run memory(c), compute(c)
            // and waves 4-7 compute(c - 1), memory(c): one wave of each SIMD multiplies while its partner issues memory instructions.
            // Ring: between BAR_c and BAR_c+1 every wave reads stage(c) only, so the DMA of chunk c - 1 + NST goes into stage(c - 1)
            // (all its reads lie before BAR_c); chunk c + 1 has landed before BAR_c+1 (own pieces: counted vmcnt; the others': the
            // barrier).  Same planes, chunks, steps and terms per accumulator as the lockstep loop: the bits do not change.
            constexpr int OPSP = DPW + DPV;
            const bool grpB = wave >= NW / 2;
            int stp = NST - 1;  // stage of the chunk before the one being read
            auto rd_all = [&]() {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    rdA(F0, st, 0, p);
                    rdB(F0, st, 0, p);
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    rdA(F1, st, 1, p);
                    rdB(F1, st, 1, p);
                }
            };
            auto compute = [&]() {
                X3W_FENCE
                X3W_TERM(F0, 0, 0)
                X3W_TERM(F0, 0, 1)
                X3W_TERM(F0, 1, 0)
                X3W_TERM(F1, 0, 0)
                X3W_TERM(F1, 0, 1)
                X3W_TERM(F1, 1, 0)
                X3W_FENCE
            };
            auto plane_end = [&]() {
                if (++ckc == KC) {
                    ckc = 0;
#ifdef A3D_ABLATIONS
                    if (!(a.abl & 16))
#endif
                    fold(mf, cf++);
                }
            };
            auto advance = [&]() {
                stp = st;
                st = st == NST - 1 ? 0 : st + 1;
            };
            // ONE instruction stream for both halves; only the barrier's place differs: waves 0-3 run memory, compute, BARRIER, waves 4-7
            // memory, BARRIER, compute -- so between two barriers the former run memory(c), compute(c) and the latter compute(c - 1),
            // memory(c).
            for (int it = 0; it < nit; ++it) {
                X3W_FENCE
                if (it > 0) plane_end();  // (the fold of the chunk multiplied last, in front of the reads: the fragment registers are free)
#ifdef A3D_ABLATIONS  // timing-only (results wrong): bit 0 no V DMA, 1 no filter DMA, 3 no fragment reads, 5 no MFMAs
                if (!(a.abl & 8)) rd_all();
                if (it > 0) {
                    if (!(a.abl & 1)) dma_v(stp, dma_c);
                    if (!(a.abl & 2)) dma_w(stp);
                    else ++dma_c;
                }
#else
                rd_all();
                if (it > 0) {
                    dma_v(stp, dma_c);  // chunk it - 1 + NST into the stage of chunk it - 1
                    dma_w(stp);
                }
#endif
                __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef A3D_ABLATIONS
                if (grpB && !(a.abl & 4)) {
#else
                if (grpB) {
#endif
                    __asm__ volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * OPSP) : "memory");
                    __builtin_amdgcn_s_barrier();
                }
#ifdef A3D_ABLATIONS
                if (!(a.abl & 32))
#endif
                compute();
#ifdef A3D_ABLATIONS
                if (!grpB && !(a.abl & 4)) {
#else
                if (!grpB) {
#endif
                    __asm__ volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * OPSP) : "memory");
                    __builtin_amdgcn_s_barrier();
                }
                advance();
            }
            plane_end();
            // the DMAs past the last chunk land anywhere in the ring, the epilogue's tiles use the V areas of stages 0 and 1
            __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
        // (Round 4, measured and not taken: the six fragment reads of a step pinned behind the first three MFMAs of the step before it
        // with sched_group_barriers -- hipcc sinks them to the end of each half, in front of the `lgkmcnt(0)` waits -- 2.16 | 2.26 ms on
        // the p2 layer across two boxes: no gain.  Timing-only decomposition of this loop, p2 256 -> 256, 2.30 ms (tools/wino_abl.sh,
        // profiles/r04_wino_ablation.txt): without the V DMA 1.96, without the filter DMA 1.94, without both 1.85, + without the
        // barrier 1.79, without DMA and fragment reads 1.53, without DMA and fold 1.71, MFMAs + epilogue alone 1.375.)
        for (int it = 0; it < nit; ++it) {
#ifdef A3D_ABLATIONS  // timing-only (results wrong): bit 0 no V DMA, 1 no filter DMA, 2 no barrier, 3 no fragment reads, 4 no fold
            const bool rd = !(a.abl & 8);
            X3W_FENCE
            X3W_TERM(F0, 0, 0)
            if (rd) {
                rdA(F1, st, 1, 0);
                rdB(F1, st, 1, 0);
                rdB(F1, st, 1, 1);
            }
            X3W_TERM(F0, 0, 1)
            if (rd) rdA(F1, st, 1, 1);
            X3W_TERM(F0, 1, 0)
            X3W_FENCE
            __asm__ volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * OPS) : "memory");
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (!(a.abl & 4)) __builtin_amdgcn_s_barrier();
            if (!(a.abl & 1)) dma_v(st, dma_c);
            if (!(a.abl & 2)) dma_w(st);
            st = st == NST - 1 ? 0 : st + 1;
            X3W_FENCE
            X3W_TERM(F1, 0, 0)
            if (rd) {
                rdA(F0, st, 0, 0);
                rdB(F0, st, 0, 0);
                rdB(F0, st, 0, 1);
            }
            X3W_TERM(F1, 0, 1)
            if (rd) rdA(F0, st, 0, 1);
            X3W_TERM(F1, 1, 0)
            X3W_FENCE
            if constexpr (!PS) {
                if (++ckc == KC) {
                    ckc = 0;
                    if (!(a.abl & 16)) fold(mf, cf++);
                }
            }
#else
            X3W_FENCE
            X3W_TERM(F0, 0, 0)
            rdA(F1, st, 1, 0);
            rdB(F1, st, 1, 0);
            rdB(F1, st, 1, 1);
            X3W_TERM(F0, 0, 1)
            rdA(F1, st, 1, 1);
            X3W_TERM(F0, 1, 0)
            X3W_FENCE
            __asm__ volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * OPS) : "memory");  // chunk it + 1 has landed (younger: NST - 2 chunks)
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            dma_v(st, dma_c);  // chunk it + NST into the stage of chunk it
            dma_w(st);
            st = st == NST - 1 ? 0 : st + 1;
            X3W_FENCE
            X3W_TERM(F1, 0, 0)
            rdA(F0, st, 0, 0);
            rdB(F0, st, 0, 0);
            rdB(F0, st, 0, 1);
            X3W_TERM(F1, 0, 1)
            rdA(F0, st, 0, 1);
            X3W_TERM(F1, 1, 0)
            X3W_FENCE
            if constexpr (!PS) {
                if (++ckc == KC) {
                    ckc = 0;
                    fold(mf, cf++);
                }
            }
#endif
        }
        // the DMAs past the last chunk land anywhere in the ring, the epilogue's tiles use the V areas of stages 0 and 1
        __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        }
    } else {
    // prologue: W(0), W(1) by DMA; V(0) split into stage 0; V(1), V(2) staged in registers; S0(0) read
    dma_w(0);
    dma_w(1);
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
    for (int p = 0; p < NP; ++p) {
        rdA(F0, 0, 0, p);
        rdB(F0, 0, 0, p);
    }
    X3W_FENCE

    int cf = 0, ckc = 0;
    for (int it = 0; it < nit; it += 2) {
        step0(0, F0, F1, xsB);
        __asm__ volatile("s_waitcnt vmcnt(2)" ::: "memory");  // all but the two youngest (V loads): the weight DMA of chunk it+1 has landed
        __syncthreads();
        step1(0, F1, F0, xsB);
        if constexpr (!PS) {
            if (++ckc == KC) {
                ckc = 0;
                fold(mf, cf++);
            }
        }
        X3W_FENCE
        step0(1, F0, F1, xsA);
        __asm__ volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __syncthreads();
        step1(1, F1, F0, xsA);
        if constexpr (!PS) {
            if (++ckc == KC) {
                ckc = 0;
                fold(mf, cf++);
            }
        }
        X3W_FENCE
    }
    }
#undef X3W_MIX
#undef X3W_TERM
#undef X3W_MFMA
#undef X3W_FENCE

    // Row-major epilogue (as conv_x3_kernel's): a lane holds one TILE and register quads of 4 channels, i.e. a store instruction would
    // touch 32 pixels x 32 bytes.  Every 32-tile x 32-channel block of an output position goes through 4 KiB of LDS and comes back
    // with 8 lanes per tile: 8 pixels x 128 contiguous bytes per instruction.  Same values and operations per element.  No barrier:
    // behind the last barrier every fragment still to be multiplied is in registers; the late weight DMAs land in the FILTER areas
    // and the tiles use the two stages' V areas (WM waves x 4 KiB each).
    static_assert(NP * PLX * 2 >= WM * 4096, "WM 4 KiB tiles per V area");
    float *Tt = reinterpret_cast<float *>(lds + (wave / WM) * BUF) + (wave % WM) * 1024;
    const int tb = t0 + wm * 32;
    const int pr = lane & 31, ph = lane >> 5, qr = lane >> 3, qc = lane & 7;
    if constexpr (PS) {  // the raw plane product, row-major: M[pf][tile][channel]
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 v = {mf[ni][rg * 4 + 0], mf[ni][rg * 4 + 1], mf[ni][rg * 4 + 2], mf[ni][rg * 4 + 3]};
                *reinterpret_cast<f32x4 *>(Tt + pr * 32 + (((rg * 2 + ph) ^ (pr & 7)) << 2)) = v;
            }
            const int n = n0 + wn * 64 + ni * 32 + qc * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = qr + 8 * j;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(Tt + q * 32 + ((qc ^ (q & 7)) << 2));
                const int t = tb + q;
                if (t < a.T && n < a.Cout) *reinterpret_cast<f32x4 *>(a.M + ((size_t)pf * a.T + t) * a.Cout + n) = v;
            }
        }
        return;
    }
#ifdef A3D_ABLATIONS
    if (a.abl & 64) {  // timing-only: no epilogue (one store per wave keeps the accumulators alive)
        float sacc = 0.f;
#pragma unroll
        for (int ij = 0; ij < 4; ++ij)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += yy[ij][ni][r];
        if (sacc == 12345.678f) a.y[tid] = sacc;
        return;
    }
#endif
    const int tyx = a.Ty * a.Tx;
    // (two exact factors, applied one after the other: their product can leave fp32's range for images of extreme magnitude)
    float unx = 1.f;
    const float unw = F16 ? 1.f / a.w_scale : 1.f;
    int jb[4], jy[4], jx[4];  // image / first output row / first output column of the tile this lane stores in pass j (-1: none)
    // ML: the output map of each tile this lane stores (its level's)
    int jH[4], jW[4];
    float *jyp[4], *jya[4];
    if constexpr (ML) {
        {
            int l, tl;
            wino_level_of(a.lv, min(tb + pr, a.T - 1), l, tl);
            const int b = tl / (wino_level_sel(a.lv.Ty, l) * wino_level_sel(a.lv.Tx, l));
            if (tb + pr < a.T) {  // 1 / wino_v_scale of the tile's image (a second source -- channel concat, single-map launches -- shares it)
                float m = wino_level_sel(a.lv.in_amax, l)[b];
                if (a.in_amax2) m = fmaxf(m, a.in_amax2[b]);
                unx = 1.f / (0.25f * a3d_pow2_scale(m));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = tb + qr + 8 * j;
            int l, tl;
            wino_level_of(a.lv, min(t, a.T - 1), l, tl);
            const int Tx = wino_level_sel(a.lv.Tx, l), Ty = wino_level_sel(a.lv.Ty, l);
            const int r = tl / Tx;
            jx[j] = 2 * (tl - r * Tx);
            jy[j] = 2 * (r % Ty);
            jb[j] = t < a.T ? r / Ty : -1;
            jH[j] = wino_level_sel(a.lv.Hl, l);
            jW[j] = wino_level_sel(a.lv.Wl, l);
            jyp[j] = wino_level_sel(a.lv.y, l);
            jya[j] = wino_level_sel(a.lv.y_amax, l);
        }
    } else {
    if (F16 && tb + pr < a.T) unx = 1.f / wino_v_scale(a, (tb + pr) / tyx);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = tb + qr + 8 * j;
        const int r = t / a.Tx;
        jx[j] = 2 * (t - r * a.Tx);
        jy[j] = 2 * (r % a.Ty);
        jb[j] = t < a.T ? r / a.Ty : -1;
        jH[j] = a.Hl;
        jW[j] = a.Wl;
        jyp[j] = a.y;
        jya[j] = a.y_amax;
    }
    }
    float vmax[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ij = 0; ij < 4; ++ij) {
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                f32x4 v = {yy[ij][ni][rg * 4 + 0], yy[ij][ni][rg * 4 + 1], yy[ij][ni][rg * 4 + 2], yy[ij][ni][rg * 4 + 3]};
                if constexpr (F16) v = (v * unx) * unw;  // exact: powers of two
                *reinterpret_cast<f32x4 *>(Tt + pr * 32 + (((rg * 2 + ph) ^ (pr & 7)) << 2)) = v;
            }
            const int nl = wn * 64 + ni * 32 + qc * 4;
            const int n = n0 + nl;
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(ss + nl), sh = *reinterpret_cast<const f32x4 *>(ss + BN + nl);
            f32x4 tv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = qr + 8 * j;
                tv[j] = *reinterpret_cast<const f32x4 *>(Tt + q * 32 + ((qc ^ (q & 7)) << 2));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int oy = jy[j] + (ij >> 1), ox = jx[j] + (ij & 1);
                if (jb[j] < 0 || oy >= jH[j] || ox >= jW[j] || n >= a.Cout) continue;
                const size_t ooff = (((size_t)jb[j] * jH[j] + oy) * jW[j] + ox) * a.Cout + n;
                f32x4 v = tv[j];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = __builtin_fmaf(v[k], sc[k], sh[k]);
                if (a.act == A3D_ACT_RELU) {
                    for (int k = 0; k < 4; ++k) v[k] = v[k] <= 0.f ? 0.f : v[k];
                } else if (a.act == A3D_ACT_LEAKY) {
                    for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.01f * v[k];
                }
                if (a.gate) {  // (single-map launches only: a3d_wino_gemm_levels refuses a gate)
                    const f32x4 g = *reinterpret_cast<const f32x4 *>(a.gate + ooff);
                    for (int k = 0; k < 4; ++k) v[k] = g[k] > 0.f ? v[k] : 0.f;
                }
                vmax[j] = fmaxf(vmax[j], a3d_absmax4(v));
                *reinterpret_cast<f32x4 *>(jyp[j] + ooff) = v;
            }
        }
    }
    if constexpr (ML) {
        // the wave's 32 tiles in one image of one level (the rule at 64 frames: every level's tile count is a multiple of 128): one reduction;
        // otherwise the 8 lanes of a tile reduce and one of them raises the slot of ITS level and image
        int l0, tl0, l1, tl1;
        wino_level_of(a.lv, min(tb, a.T - 1), l0, tl0);
        wino_level_of(a.lv, min(tb + 31, a.T - 1), l1, tl1);
        const int tyx0 = wino_level_sel(a.lv.Ty, l0) * wino_level_sel(a.lv.Tx, l0);
        float *ya0 = wino_level_sel(a.lv.y_amax, l0);
        if (tb < a.T && l0 == l1 && tl0 / tyx0 == tl1 / tyx0) {
            if (ya0) a3d_note_amax(ya0, tl0 / tyx0, fmaxf(fmaxf(vmax[0], vmax[1]), fmaxf(vmax[2], vmax[3])), true);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = vmax[j];
                v = fmaxf(v, __shfl_xor(v, 1, 64));
                v = fmaxf(v, __shfl_xor(v, 2, 64));
                v = fmaxf(v, __shfl_xor(v, 4, 64));
                if (jb[j] >= 0 && qc == 0 && jya[j] && v > jya[j][jb[j]]) atomicMax(reinterpret_cast<int *>(jya[j] + jb[j]), __float_as_int(v));
            }
        }
        return;
    }
    if (a.y_amax) {
        const int tl = min(tb + 31, a.T - 1);
        if (tb < a.T && tb / tyx == tl / tyx) {  // the wave's 32 tiles belong to one image: one reduction
            a3d_note_amax(a.y_amax, tb / tyx, fmaxf(fmaxf(vmax[0], vmax[1]), fmaxf(vmax[2], vmax[3])), true);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // (the 8 lanes of a tile reduce first: one pre-checked atomic per tile, not eight)
                float v = vmax[j];
                v = fmaxf(v, __shfl_xor(v, 1, 64));
                v = fmaxf(v, __shfl_xor(v, 2, 64));
                v = fmaxf(v, __shfl_xor(v, 4, 64));
                a3d_note_amax(a.y_amax, max(jb[j], 0), v, jb[j] >= 0 && qc == 0);
            }
        }
    }
}
// Second launch of the plane-split form: one thread = (tile, channel quad).  The 16 plane products are folded with the SAME
// fused multiply-adds, in the same order, as wino_gemm_x3w_kernel's fold(), then the same epilogue: bit-identical output.
__global__ __launch_bounds__(256) void wino_fold_kernel(const WinoArgs a) {
    const int C4 = a.Cout >> 2;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool ok = idx < (size_t)a.T * C4;
    const int t = ok ? (int)(idx / C4) : 0;
    const int n = ok ? (int)(idx - (size_t)t * C4) * 4 : 0;
    const int tyx = a.Ty * a.Tx;
    const int bimg = t / tyx;
    float vmax = 0.f;
    if (ok) {
        f32x4 yy[4];
#pragma unroll
        for (int ij = 0; ij < 4; ++ij) yy[ij] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            const f32x4 x = *reinterpret_cast<const f32x4 *>(a.M + ((size_t)f * a.T + t) * a.Cout + n);
            const int u = f >> 2, v = f & 3;
            const float au0 = (u < 3) ? 1.f : 0.f, au1 = (u == 0) ? 0.f : ((u == 1) ? 1.f : -1.f);
            const float av0 = (v < 3) ? 1.f : 0.f, av1 = (v == 0) ? 0.f : ((v == 1) ? 1.f : -1.f);
            const float c00 = au0 * av0, c01 = au0 * av1, c10 = au1 * av0, c11 = au1 * av1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                yy[0][k] = __builtin_fmaf(c00, x[k], yy[0][k]);
                yy[1][k] = __builtin_fmaf(c01, x[k], yy[1][k]);
                yy[2][k] = __builtin_fmaf(c10, x[k], yy[2][k]);
                yy[3][k] = __builtin_fmaf(c11, x[k], yy[3][k]);
            }
        }
        const float unx = 1.f / wino_v_scale(a, bimg), unw = 1.f / a.w_scale;
        const int r = t / a.Tx;
        const int ox0 = 2 * (t - r * a.Tx), oy0 = 2 * (r % a.Ty);
        const f32x4 sc = a.scale ? *reinterpret_cast<const f32x4 *>(a.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 sh = a.shift ? *reinterpret_cast<const f32x4 *>(a.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ij = 0; ij < 4; ++ij) {
            const int oy = oy0 + (ij >> 1), ox = ox0 + (ij & 1);
            if (oy >= a.Hl || ox >= a.Wl) continue;
            const size_t ooff = (((size_t)bimg * a.Hl + oy) * a.Wl + ox) * a.Cout + n;
            f32x4 v = (yy[ij] * unx) * unw;  // exact: powers of two
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = __builtin_fmaf(v[k], sc[k], sh[k]);
            if (a.act == A3D_ACT_RELU) {
                for (int k = 0; k < 4; ++k) v[k] = v[k] <= 0.f ? 0.f : v[k];
            } else if (a.act == A3D_ACT_LEAKY) {
                for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.01f * v[k];
            }
            if (a.gate) {
                const f32x4 g = *reinterpret_cast<const f32x4 *>(a.gate + ooff);
                for (int k = 0; k < 4; ++k) v[k] = g[k] > 0.f ? v[k] : 0.f;
            }
            vmax = fmaxf(vmax, a3d_absmax4(v));
            *reinterpret_cast<f32x4 *>(a.y + ooff) = v;
        }
    }
    if (a.y_amax) a3d_note_amax(a.y_amax, bimg, vmax, ok);  // (every lane of the wave gets here)
}
// src [outer][rows][cols] fp32 -> dst [outer][cols/32][3][rows][32] bf16 with src == hi + mid + lo exactly: the chunk-major
// plane layout the split-operand GEMM streams (the `rows` of one plane of one 32-deep chunk are one contiguous run; with a
// plain [rows][cols] plane layout a lane's 16-byte piece of a row is a 64-byte-strided access and the kernel runs 2x slower).
// a3d_conv_desc.w_wino_x3: outer = 16, rows = Cout, cols = Cin + Cin2.
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float *__restrict__ src, __bf16 *__restrict__ dst, int rows, int cols, int chunk, size_t total) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    const size_t per = (size_t)rows * cols;
    const size_t o = i / per, r = i - o * per;
    const int n = (int)(r / cols), c = (int)(r - (size_t)n * cols);
    wbf16x4 h, m, l;
    wsplit3(*reinterpret_cast<const f32x4 *>(src + i), h, m, l);
    __bf16 *d = dst + o * 3 * per + ((size_t)(c / chunk) * 3 * rows + n) * chunk + (c % chunk);
    *reinterpret_cast<wbf16x4 *>(d) = h;
    *reinterpret_cast<wbf16x4 *>(d + (size_t)rows * chunk) = m;
    *reinterpret_cast<wbf16x4 *>(d + (size_t)rows * chunk * 2) = l;
}

extern "C" int a3d_split_bf16x3_chunk(const float *src, void *dst, int outer, int rows, int cols, int chunk, void *stream) {
    if (!src || !dst || outer <= 0 || rows <= 0 || cols <= 0 || (chunk != 16 && chunk != 32) || cols % chunk) return A3D_ERR_ARG;
    a3d_begin();
    const size_t total = (size_t)outer * rows * cols;
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (__bf16 *)dst, rows, cols, chunk,
                       total);
    return a3d_check_launch();
}

// the fp16x2 counterpart (precision 3): src * scale = hi + lo in fp16, dst [outer][cols/chunk][2][rows][chunk]
__global__ __launch_bounds__(256) void split_f16x2_kernel(const float *__restrict__ src, _Float16 *__restrict__ dst, int rows, int cols, int chunk, float scale,
                                                          size_t total) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    const size_t per = (size_t)rows * cols;
    const size_t o = i / per, r = i - o * per;
    const int n = (int)(r / cols), c = (int)(r - (size_t)n * cols);
    const f32x4 xs = *reinterpret_cast<const f32x4 *>(src + i) * scale;
    const wh16x4 h = __builtin_convertvector(xs, wh16x4);
    const wh16x4 l = __builtin_convertvector(xs - __builtin_convertvector(h, f32x4), wh16x4);
    _Float16 *d = dst + o * 2 * per + ((size_t)(c / chunk) * 2 * rows + n) * chunk + (c % chunk);
    *reinterpret_cast<wh16x4 *>(d) = h;
    *reinterpret_cast<wh16x4 *>(d + (size_t)rows * chunk) = l;
}

extern "C" int a3d_split_f16x2_chunk(const float *src, void *dst, int outer, int rows, int cols, int chunk, float scale, void *stream) {
    if (!src || !dst || outer <= 0 || rows <= 0 || cols <= 0 || (chunk != 16 && chunk != 32) || cols % chunk || !(scale > 0.f)) return A3D_ERR_ARG;
    a3d_begin();
    const size_t total = (size_t)outer * rows * cols;
    hipLaunchKernelGGL(split_f16x2_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (_Float16 *)dst, rows, cols, chunk,
                       scale, total);
    return a3d_check_launch();
}

extern "C" int a3d_split_bf16x3(const float *src, void *dst, int outer, int rows, int cols, void *stream) {
    return a3d_split_bf16x3_chunk(src, dst, outer, rows, cols, 32, stream);
}

int a3d_wino_eligible(const a3d_conv_desc *d) {
    if (!d->w_wino) return 0;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1) return 0;
    if (d->res || d->pixshuf || d->stem || d->splitk != 1 || d->m_dev) return 0;
    if (((d->Cin + d->Cin2) & 15) || (d->Cin & 3) || (d->Cin2 & 3) || (d->Cout & 3)) return 0;
    const int Hl = d->ups ? 2 * d->H : d->H, Wl = d->ups ? 2 * d->W : d->W;
    const size_t T = (size_t)d->B * ((Hl + 1) / 2) * ((Wl + 1) / 2);
    const size_t C = (size_t)d->Cin + d->Cin2;
    if (T * C * 4 >= ((size_t)1 << 32) || (size_t)d->Cout * C * 4 >= ((size_t)1 << 32)) return 0;
    return 1;
}

size_t a3d_wino_workspace_bytes(const a3d_conv_desc *d) {
    const int Hl = d->ups ? 2 * d->H : d->H, Wl = d->ups ? 2 * d->W : d->W;
    size_t T = (size_t)d->B * ((Hl + 1) / 2) * ((Wl + 1) / 2);
    if (d->precision == 3 && d->wino_t_total > 0) T = (size_t)d->wino_t_total;  // (the layer's tiles are a slice of a buffer shared by several maps)
    return 16 * T * ((size_t)d->Cin + d->Cin2) * sizeof(float);
}

// Plane-split form (precision 3): worth it while the one-launch form would start at most 48 128-tile workgroups -- they leave most
// of the 256 CUs idle for 16 x C/32 iterations, and 32 x as many 64-tile single-plane workgroups still fit in a few rounds.  Measured
// (frames/s, bound 0 | 16 | 32 | 48 | 96): 1 frame 113 | 145 | 143 | 145 | 143, 4 frames 386 | 426 | 447 | 451 | 453, 64 frames
// 1288 | . | 1278 (= its 0) | . | 1285 (-0.3 %).  A3D_WINO_PS_BLOCKS overrides the bound (0 = never; A/B runs).
// (Round 4, built, bit-identical, measured and removed: the partial LAST round of a multi-round 64-tile launch run plane-split + folded
// while the full rounds keep the one-launch form.  64 frames of a 30x40 level are 600 blocks = one full round of 512 and one that is
// 17 % full -- 0.214 ms where 54 frames, one round, take 0.134.  With the 88 trailing blocks as 1408 plane workgroups + the fold over
// their tiles: 0.218 | 0.216 ms (hybrid | plain), 60x80 256 -> 128 0.475 | 0.416: ring fill, raw-tile epilogue, the M round trip and two
// more launches cost what the short round costs.)
extern "C" size_t a3d_wino_m_bytes(const a3d_conv_desc *d) {
    const long ps_max = a3d_dev_knob("A3D_WINO_PS_BLOCKS", 48);
    if (!d || d->precision != 3 || !a3d_wino_eligible(d)) return 0;
    const int C = d->Cin + d->Cin2;
    if ((C & 63) || ((d->Cout + 63) / 64) % 2 != 0) return 0;  // (an even number of 32-deep chunks per plane; the wide tiles' channel blocks)
    const int Hl = d->ups ? 2 * d->H : d->H, Wl = d->ups ? 2 * d->W : d->W;
    const size_t T = (size_t)d->B * ((Hl + 1) / 2) * ((Wl + 1) / 2);
    const long blocks4 = (long)((T + 127) / 128) * ((d->Cout + X3W_BN - 1) / X3W_BN);
    if (blocks4 > ps_max) return 0;
    return 16 * T * (size_t)d->Cout * sizeof(float);
}

static int wino_launch_input(const a3d_conv_desc *d, hipStream_t s) {
    const int Hl = d->ups ? 2 * d->H : d->H, Wl = d->ups ? 2 * d->W : d->W;
    const int Ty = (Hl + 1) / 2, Tx = (Wl + 1) / 2;
    const int C = d->Cin + d->Cin2;
    const size_t T = (size_t)d->B * Ty * Tx;
    const size_t total = T * (C / 4);
    size_t blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (d->precision == 3) {  // fp16x2: V pre-split into the two fp16 planes, chunk-major (what wino_gemm_x3w_kernel<.., true> DMAs)
        if ((C & 31) || !d->in_amax) return A3D_ERR_ARG;
        if (d->wino_t_total && (d->wino_t_off < 0 || (size_t)d->wino_t_off + T > (size_t)d->wino_t_total)) return A3D_ERR_ARG;
        if ((size_t)(d->wino_t_total ? d->wino_t_total : T) * C * 4 >= ((size_t)1 << 32)) return A3D_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(wino_input_h2_kernel, dim3((int)blocks), dim3(256), 0, s, d->x, d->x2, reinterpret_cast<unsigned char *>(d->workspace), d->in_amax,
                           d->in_amax2, d->B, d->H, d->W, d->Cin, d->Cin2, d->ups, Ty, Tx, d->wino_t_total ? d->wino_t_off : 0, d->wino_t_total ? d->wino_t_total : (int)T);
        return A3D_OK;
    }
    hipLaunchKernelGGL(wino_input_kernel, dim3((int)blocks), dim3(256), 0, s, d->x, d->x2, d->workspace, d->B, d->H, d->W,
                       d->Cin, d->Cin2, d->ups, Ty, Tx);
    return A3D_OK;
}

static int wino_launch_gemm(const a3d_conv_desc *d, hipStream_t s) {
    const int Hl = d->ups ? 2 * d->H : d->H, Wl = d->ups ? 2 * d->W : d->W;
    const int Ty = (Hl + 1) / 2, Tx = (Wl + 1) / 2;
    const size_t T = (size_t)d->B * Ty * Tx;
    WinoArgs a;
    a.V = d->workspace;
    a.U = d->w_wino;
    a.scale = d->scale;
    a.shift = d->shift;
    a.gate = d->gate;
    a.y = d->y;
    a.T = (int)T;
    a.C = d->Cin + d->Cin2;
    a.Cout = d->Cout;
    a.B = d->B;
    a.Hl = Hl;
    a.Wl = Wl;
    a.Ty = Ty;
    a.Tx = Tx;
    a.act = d->act;
    const int mtiles = (int)((T + 63) / 64);
    a.U3 = nullptr;
    a.y_amax = d->y_amax;
    a.in_amax = d->in_amax;
    a.in_amax2 = d->in_amax2;
    a.w_scale = d->w_scale;
    a.M = nullptr;
    a.abl = (int)a3d_dev_knob("A3D_WINO_ABL", 0);
    a.Toff = 0;
    a.Ttot = (int)T;
    a.lv.n = 1;  // (the ping-pong form's epilogue is table-driven: this layer's map is the table)
    for (int k = 0; k < 5; ++k) {
        a.lv.t0[k] = k == 0 ? 0 : (int)T;
        a.lv.Ty[k] = Ty;
        a.lv.Tx[k] = Tx;
        a.lv.Hl[k] = Hl;
        a.lv.Wl[k] = Wl;
        a.lv.y[k] = d->y;
        a.lv.in_amax[k] = d->in_amax;
        a.lv.y_amax[k] = d->y_amax;
    }
    a.lv.t0[5] = (int)T;
    if (d->wino_t_total) {  // a slice of a shared V buffer: the fp16x2 forms only
        if (d->precision != 3 || d->wino_t_off < 0 || (size_t)d->wino_t_off + T > (size_t)d->wino_t_total) return A3D_ERR_ARG;
        a.Toff = d->wino_t_off;
        a.Ttot = d->wino_t_total;
    }
    if (d->precision == 3) {  // fp16x2: the wide kernels only (w_wino_x3 = the filter pre-split by a3d_split_f16x2_chunk(.., 32, w_scale))
        if (!d->w_wino_x3 || (a.C & 31) || !d->in_amax || !(d->w_scale > 0.f) || ((d->Cout + 63) / 64) % 2 != 0 ||
            (size_t)16 * d->Cout * a.C * 4 >= ((size_t)1 << 32))
            return A3D_ERR_ARG;
        a.U3 = reinterpret_cast<const __bf16 *>(d->w_wino_x3);
        const int nt = (d->Cout + X3W_BN - 1) / X3W_BN;
        // 128-tile blocks, one 512-thread workgroup per CU, ping-pong loop -- on every problem size.  Until round 4 the mid-sized problems
        // (256 .. 2400 blocks) took 64-tile blocks with two independent 256-thread workgroups per CU, whose barriers decouple (60x80x256
        // 0.68 | 0.69 ms, 30x40 0.21 | 0.22); with the two halves of the 512-thread workgroup in antiphase the wide form is ahead
        // everywhere (profiles/r05_wino_pp.txt, 64-tile | 128-tile lockstep | 128-tile ping-pong: 30x40x256 0.215 | 0.237 | 0.211,
        // 60x80x256 0.676 | 0.697 | 0.599, 1000 ROIs 0.469 | 0.444 | 0.398, 32 frames of p2 1.200 | 1.197 | 1.091, 15x20x512 0.209 | 0.199
        // | 0.164) and moves two thirds of the narrow form's operand bytes.  Same kernel template, same operation order per output: the
        // three forms are bit-identical (tune 24: 64-tile blocks, tune 23: the lockstep loop; tests/test_gpu_parity.py).
        const int wmx = d->tune == 24 ? 2 : 4;
        static a3d_attr_once attr3;
        if (attr3.needed()) {
            if (hipFuncSetAttribute((const void *)wino_gemm_x3w_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, x3w_lds_bytes(2, 2)) != hipSuccess ||
                hipFuncSetAttribute((const void *)wino_gemm_x3w_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, x3w_lds_bytes(4, 2)) != hipSuccess ||
                hipFuncSetAttribute((const void *)wino_gemm_x3w_kernel<4, true, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, x3w_lds_bytes(4, 2)) != hipSuccess ||
                hipFuncSetAttribute((const void *)wino_gemm_x3w_kernel<2, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, x3w_lds_bytes(2, 2)) != hipSuccess)
                return A3D_ERR_LAUNCH;
            attr3.mark();
        }
        if (d->wino_m && a3d_wino_m_bytes(d)) {  // small problem: one plane per workgroup, then the fold (same bits)
            a.M = d->wino_m;
            a3d_note_variant("wino_gemm_h2w_kernel<2> planes + wino_fold_kernel");
            hipLaunchKernelGGL((wino_gemm_x3w_kernel<2, true, true>), dim3(mtiles * nt, 16), dim3(256), x3w_lds_bytes(2, 2), s, a, nt, mtiles * nt);
            const size_t threads = T * (size_t)(d->Cout / 4);
            hipLaunchKernelGGL(wino_fold_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a);
            return A3D_OK;
        }
        a3d_note_variant("wino_gemm_h2w_kernel<%d>", wmx);
        if (wmx == 4) {
            const int m4 = (int)((T + 127) / 128);
            // ping-pong loop (round 5; tune 23: the lockstep loop it replaces -- A/B runs and the bit-equality test)
            if (d->tune == 23) hipLaunchKernelGGL((wino_gemm_x3w_kernel<4, true>), dim3(m4 * nt), dim3(512), x3w_lds_bytes(4, 2), s, a, nt, m4 * nt);
            else hipLaunchKernelGGL((wino_gemm_x3w_kernel<4, true, false, 1>), dim3(m4 * nt), dim3(512), x3w_lds_bytes(4, 2), s, a, nt, m4 * nt);
        } else {
            hipLaunchKernelGGL((wino_gemm_x3w_kernel<2, true>), dim3(mtiles * nt), dim3(256), x3w_lds_bytes(2, 2), s, a, nt, mtiles * nt);
        }
        return A3D_OK;
    }
    if (d->precision == 2) {  // fp32-grade products on the bf16 pipe (2x); C % 32 == 0 is required by its 32-deep chunks
        if (!d->w_wino_x3 || (a.C & 31)) return A3D_ERR_ARG;
        a.U3 = reinterpret_cast<const __bf16 *>(d->w_wino_x3);
        // (Round 5, built, bit-identical -- fuzz + equality tests --, measured and removed: the fp16x2 form's recipe for this arithmetic, V
        // pre-split into its three bf16 planes by the transform (6 bytes per element), both operands by LDS-DMA through three 48 KiB stages,
        // the workgroup's halves in antiphase.  GEMM per layer, 64 frames, that form | the register-staged 128-tile loop below: p2 256 -> 256
        // 3.497 | 3.526 ms, 60x80 0.954 | 0.916, 30x40 0.328 | 0.342, 276 ROIs 0.188 | 0.191, 15x20x512 0.319 | 0.317 -- a tie, with six
        // MFMAs per product the loop is not short of overlap -- while the transform writes 1.5 x the bytes: the bf16x3 step 59.3 -> 64.5 ms.)
        // 128 tiles x 128 channels with DMA-staged weights (one 512-thread workgroup per CU) or 64 x 64 (three 256-thread workgroups
        // per CU)?  Both are bit-identical, so the choice is free; it goes by the rounds the busiest CU runs.  A CU works through
        // ceil(blocks / 256) blocks, the narrow form three at a time; three narrow blocks are 0.75 of a wide block's work at ~0.87 of
        // its rate, i.e. one narrow round costs ~0.86 of a wide one.  Measured (tools/x3w_check.py, ms wide | narrow): p2 256->256
        // 3.50 | 4.15, 60x80x256 0.97 | 1.08, 276 ROIs of 14x14 0.18 | 0.28 (one partial round instead of two), 1600 ROIs 1.07 | 1.30,
        // 30x40x256 0.34 | 0.34, 15x20x512 0.33 | 0.34.
        // tune 8: the narrow form everywhere; tune 24 | 25 force the 64- | 128-tile wide form (A/B runs and the bit-equality test).
        const int wm_force = d->tune == 24 ? 2 : (d->tune == 25 ? 4 : 0);
        const int ntw = (d->Cout + X3W_BN - 1) / X3W_BN;
        const long blocks_w = (long)((T + 127) / 128) * ntw, blocks_n = (long)mtiles * ((d->Cout + 63) / 64);
        const long rounds_w = (blocks_w + 255) / 256, rounds_n = ((blocks_n + 255) / 256 + 2) / 3;
        const bool wide = d->tune != 8 && ((d->Cout + 63) / 64) % 2 == 0 && (size_t)16 * d->Cout * a.C * 6 < ((size_t)1 << 32) &&
                          (wm_force || 100 * rounds_w <= 86 * rounds_n);
        if (wide) {
            const int nt = ntw;
            const int wmx = wm_force ? wm_force : 4;
            static a3d_attr_once attr_set;
            if (attr_set.needed()) {  // > 64 KiB of dynamic LDS needs the opt-in attribute (once per device)
                if (hipFuncSetAttribute((const void *)wino_gemm_x3w_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, x3w_lds_bytes(2, 3)) != hipSuccess ||
                    hipFuncSetAttribute((const void *)wino_gemm_x3w_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, x3w_lds_bytes(4, 3)) != hipSuccess)
                    return A3D_ERR_LAUNCH;
                attr_set.mark();
            }
            a3d_note_variant("wino_gemm_x3w_kernel<%d>", wmx);
            if (wmx == 4) {
                const int m4 = (int)((T + 127) / 128);
                hipLaunchKernelGGL((wino_gemm_x3w_kernel<4>), dim3(m4 * nt), dim3(512), x3w_lds_bytes(4, 3), s, a, nt, m4 * nt);
            } else {
                hipLaunchKernelGGL((wino_gemm_x3w_kernel<2>), dim3(mtiles * nt), dim3(256), x3w_lds_bytes(2, 3), s, a, nt, mtiles * nt);
            }
            return A3D_OK;
        }
        const int nt = (d->Cout + 63) / 64;
        a3d_note_variant("wino_gemm_x3_kernel");
        hipLaunchKernelGGL(wino_gemm_x3_kernel, dim3(mtiles * nt), dim3(256), 0, s, a, nt, mtiles * nt);
        return A3D_OK;
    }
    // tune >= 200 selects an explicit GEMM variant for A/B measurements: 200 + 10*(TN-1) + (BK==32)
    int tn = 1, bk32 = 1;  // measured best on every layer shape of the detector (64 tiles x 64 channels, 3 waves/SIMD)
    if (d->tune >= 200) {
        tn = (d->tune - 200) / 10 + 1;
        bk32 = (d->tune - 200) % 10;
    }
    const int bn = tn * 64;
    const int ntiles = (d->Cout + bn - 1) / bn;
    const dim3 grid(mtiles * ntiles);
    a3d_note_variant("wino_gemm_kernel<%d,%d>", tn, bk32 ? 32 : 16);
    // ping-pong accumulators (PP) whenever a plane has an even number of 32-deep k-chunks; tune % 10 >= 2 turns it off (A/B)
    if (tn == 1 && !bk32) hipLaunchKernelGGL((wino_gemm_kernel<1, 16>), grid, dim3(256), 0, s, a, ntiles, mtiles * ntiles);
    else if (tn == 1) hipLaunchKernelGGL((wino_gemm_kernel<1, 32>), grid, dim3(256), 0, s, a, ntiles, mtiles * ntiles);
    else if (!bk32) hipLaunchKernelGGL((wino_gemm_kernel<2, 16>), grid, dim3(256), 0, s, a, ntiles, mtiles * ntiles);
    else hipLaunchKernelGGL((wino_gemm_kernel<2, 32>), grid, dim3(256), 0, s, a, ntiles, mtiles * ntiles);
    return A3D_OK;
}

int a3d_conv_launch_wino(const a3d_conv_desc *d, hipStream_t s) {
    if (!a3d_wino_eligible(d) || !d->workspace) return A3D_ERR_UNSUPPORTED;
    int r = wino_launch_input(d, s);
    if (r != A3D_OK) return r;
    r = wino_launch_gemm(d, s);  // (argument / attribute failures launch nothing: report them instead of a clean launch status)
    if (r != A3D_OK) return r;
    return a3d_check_launch();
}

extern "C" int a3d_wino_input_transform(const a3d_conv_desc *d, void *stream) {
    if (!d || !d->x || !a3d_wino_eligible(d) || !d->workspace) return A3D_ERR_ARG;
    a3d_begin();
    const int r = wino_launch_input(d, (hipStream_t)stream);
    if (r != A3D_OK) return r;
    return a3d_check_launch();
}

extern "C" int a3d_wino_gemm_levels(const a3d_conv_desc *lv, int n, void *stream) {
    if (!lv || n < 1 || n > 5) return A3D_ERR_ARG;
    const a3d_conv_desc *d0 = lv;
    if (d0->precision != 3 || !a3d_wino_eligible(d0) || !d0->workspace || !d0->w_wino_x3 || !(d0->w_scale > 0.f) || d0->wino_t_total <= 0) return A3D_ERR_ARG;
    const int C = d0->Cin + d0->Cin2;
    if ((C & 31) || ((d0->Cout + 63) / 64) % 2 != 0 || (size_t)16 * d0->Cout * C * 4 >= ((size_t)1 << 32)) return A3D_ERR_ARG;
    WinoArgs a;
    a.V = d0->workspace;
    a.U = d0->w_wino;
    a.U3 = reinterpret_cast<const __bf16 *>(d0->w_wino_x3);
    a.scale = d0->scale;
    a.shift = d0->shift;
    a.gate = nullptr;
    a.y = nullptr;
    a.y_amax = nullptr;
    a.in_amax = nullptr;
    a.in_amax2 = nullptr;
    a.w_scale = d0->w_scale;
    a.M = nullptr;
    a.abl = 0;
    a.C = C;
    a.Cout = d0->Cout;
    a.act = d0->act;
    a.B = d0->B;
    a.Hl = a.Wl = a.Ty = a.Tx = 1;
    a.Toff = 0;
    a.Ttot = a.T = d0->wino_t_total;
    a.lv.n = n;
    size_t t = 0;
    for (int k = 0; k < 5; ++k) {
        const a3d_conv_desc *d = lv + (k < n ? k : n - 1);
        if (k < n) {
            if (d->precision != 3 || !a3d_wino_eligible(d) || d->workspace != d0->workspace || d->w_wino_x3 != d0->w_wino_x3 || d->Cout != d0->Cout ||
                d->Cin + d->Cin2 != C || d->Cin2 || d->x2 || d->ups || d->gate || d->act != d0->act || d->scale != d0->scale || d->shift != d0->shift ||
                d->w_scale != d0->w_scale || d->wino_t_total != d0->wino_t_total || (size_t)d->wino_t_off != t || !d->in_amax || !d->y)
                return A3D_ERR_ARG;
        }
        const int Ty = (d->H + 1) / 2, Tx = (d->W + 1) / 2;
        a.lv.t0[k] = (int)t;
        a.lv.Ty[k] = Ty;
        a.lv.Tx[k] = Tx;
        a.lv.Hl[k] = d->H;
        a.lv.Wl[k] = d->W;
        a.lv.y[k] = d->y;
        a.lv.in_amax[k] = d->in_amax;
        a.lv.y_amax[k] = d->y_amax;
        if (k < n) t += (size_t)d->B * Ty * Tx;
    }
    a.lv.t0[5] = (int)t;
    if (t != (size_t)d0->wino_t_total || t * C * 4 >= ((size_t)1 << 32)) return A3D_ERR_ARG;
    a3d_begin();
    static a3d_attr_once attr_ml;
    if (attr_ml.needed()) {
        if (hipFuncSetAttribute((const void *)wino_gemm_x3w_kernel<4, true, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, x3w_lds_bytes(4, 2)) != hipSuccess)
            return A3D_ERR_LAUNCH;
        attr_ml.mark();
    }
    const int nt = (a.Cout + X3W_BN - 1) / X3W_BN, m4 = (int)((t + 127) / 128);
    a3d_note_variant("wino_gemm_h2w_kernel<4> levels%d", n);
    hipLaunchKernelGGL((wino_gemm_x3w_kernel<4, true, false, 1>), dim3(m4 * nt), dim3(512), x3w_lds_bytes(4, 2), (hipStream_t)stream, a, nt, m4 * nt);
    return a3d_check_launch();
}

extern "C" int a3d_wino_gemm(const a3d_conv_desc *d, void *stream) {
    if (!d || !d->y || !a3d_wino_eligible(d) || !d->workspace) return A3D_ERR_ARG;
    a3d_begin();
    const int r = wino_launch_gemm(d, (hipStream_t)stream);
    if (r != A3D_OK) return r;
    return a3d_check_launch();
}
