// 3x3 stride-1 pad-1 convolutions as Winograd F(2x2,3x3) in ONE launch: the input transform V = B^T d B is computed
// inside the GEMM's loader, so the 16-plane tensor V (4x the input) never exists in HBM.
//
// Why a second Winograd kernel (conv_wino.hip keeps the two-launch form for the cases this one does not take): the split
// form writes V once (wino_input_kernel: 8.1 ms of pure HBM time per 64-frame step, no FLOPs) and its GEMM streams V back
// in (~2.1 GB moved per average layer against 0.29 GB of algorithmic input + output; profiles/r01_traffic.json).
//
// Loop order.  The split GEMM runs plane-outer (for f: for k-chunk), which would make a fused loader re-transform every
// patch 16 times.  Here the order is k-outer: for each chunk of 8 input channels the workgroup transforms its 64 tiles
// ONCE into all 16 planes (LDS), then issues the 16 plane-GEMMs of depth 8.  Every wave therefore carries the 16 plane
// accumulators M_f to the end (the fold Y_ij = sum_f c_ij,f M_f is linear and runs once, in the epilogue): 16 planes x a 32 x 32
// wave tile of v_mfma_f32_32x32x2_f32 = 256 accumulator registers (the whole AGPR half), one wave per SIMD; the workgroup is
// 4 waves = 64 tiles x 64 channels, one per CU.  k runs in the same order as in conv_wino.hip's GEMM and V is computed by the same
// expressions, so the result is BIT-IDENTICAL to the two-launch form (tests/test_gpu_parity.py asserts equality).
//
//   A operand = U_f (32 output channels x 2 k), B operand = V_f (2 k x 32 tiles)  ->  lane l owns tile (l & 31) and, per register
//   quad, 4 consecutive channels: float4 NHWC stores, as in every other conv kernel of this library.
//
// Measured variants (MI355X, p2 256->256 layer, 32 frames; two-launch form 3.25-3.4 ms):
//   v1 patch through registers, 16x16x4, 2 waves/SIMD: 3.70 ms -> fenced plane steps 3.42 -> LDS-image weights + b128 fragments 3.33
//   (bank conflicts 45 % -> 0) -> 2-D blocks + DMA staging 3.12-3.24 -> this form (32x32x2, 1 wave/SIMD) 3.27, no gain from
//   interleaving two planes' MFMAs.  Ablations of this form: MFMA alone 2.60 ms; + fragment reads + barrier 2.70; + transform 3.0;
//   + DMA 3.27.  The transform's ~120 v_add_f32 per chunk cost their full issue time: the fp32 MFMA runs at exactly the fp32
//   vector rate and evidently shares that datapath, so VALU work does not hide behind it (it does behind bf16 MFMAs).
//
// Data movement (what the first versions of this kernel got wrong, measured with rocprofv3 --pmc):
//   * the 64 tiles of a workgroup are a 2-D BLOCK of BH x BW tiles of one image (8x8, 4x16, 16x4 or 2x32, chosen per layer), so
//     their 4x4 patches overlap to (2BH+2) x (2BW+2) distinct pixels.  Per chunk those pixels' 8 channels (32 B each) go
//     global -> LDS ONCE by LDS-DMA (`buffer_load_dwordx4 ... lds`, two lanes per pixel, no registers); pixels in the zero padding
//     carry offset 0xFFFFFFFF and arrive as zeros.  (Loading every thread's own patch through registers fetched each pixel
//     ~4.7 times in 32-byte pieces of 128-byte lines: the L1 line-fill path, not the matrix pipe, set the pace -- 0.63 MFMA busy.)
//   * the weights are pre-packed in the exact order of the LDS image (a3d_conv_desc.w_wino_cm), so one (chunk, plane, k half) of
//     a 64-channel tile is a contiguous 1 KiB run = one DMA instruction; a lane's fragment for BOTH channel blocks is one
//     ds_read_b128 (two ds_read_b64 get fused into ds_read2_b64, whose 16-lane / 32-bank grouping made 45 % of the LDS cycles
//     bank conflicts).
//   * the transform reads its 12 patch pixels from the staged region, 32 VALU adds, 8 ds_write_b64 into the V image
//     [plane][k half][tile][4] (half 1 stores tile ^ 4: conflict-free 16-lane write groups without padding).
// One iteration = one chunk = 8 steps of {fragment reads of the next two planes, 8 MFMA, a slice of the other work}, fenced so
// that hipcc keeps the order; DMA of chunk c+2 (pixels) and c+1 (weights) is issued at step 0 and waited for at the barrier that
// ends the iteration.  LDS: V and U double-buffered (4 x 32 KiB) + 2 x 13 KiB staging.
#include "conv_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

template <int N, int I = 0, class F>
__device__ __forceinline__ void a3d_static_for(F &&fn) {
    if constexpr (I < N) {
        fn(std::integral_constant<int, I>{});
        a3d_static_for<N, I + 1>(fn);
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// (base pointer and size pass through v_readfirstlane: they ARE wave-uniform, and saying so keeps the descriptor in SGPRs --
// otherwise hipcc wraps every buffer load of the unrolled loop in a waterfall loop)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fmake_rsrc(const void *p, unsigned bytes) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ int funi(int v) { return __builtin_amdgcn_readfirstlane(v); }
// (default cache policy on purpose: `nt` on the pixel stream was measured 1-6 % slower -- a staged line is re-used by the next
// three chunks and by the three other channel tiles of the block)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, float *lds_dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds_dst, 16, voff, soff, 0, 0);
}

struct WinoFusedArgs {
    const float *x;      // [B, H, W, C] NHWC
    const float *Uc;     // Winograd weights in LDS-image order, see a3d_conv_desc.w_wino_cm
    const float *scale, *shift, *gate;
    float *y;            // [B, H, W, Cout]
    int C, Cout, B, H, W, act;
    int lbw;             // log2 of the block width in tiles (block = (64 >> lbw) x (1 << lbw) tiles)
    int NBY, NBX;        // blocks per image
};

constexpr int BN = 64, BKC = 8;                   // channels per workgroup, input channels per chunk
constexpr int HALF = 64 * 4;                      // floats of one k half of one plane: [64 rows][4]
constexpr int PLANE = 2 * HALF;                   // 2 KiB
constexpr int OPBUF = 16 * PLANE;                 // one operand (V or U), all 16 planes, one chunk: 32 KiB
constexpr int STG = 13 * 256;                     // staging buffer: up to 13 DMA instructions of 1 KiB (396 pixels x 32 B)
constexpr int LDS_FLOATS = 4 * OPBUF + 2 * STG;   // 157,696 B

// 4 waves, ONE per SIMD (the 16 plane accumulators of a 32 x 32 wave tile are 256 registers): wave = (tile half wm, channel
// half wn).  v_mfma_f32_32x32x2_f32: lane l supplies A[row l & 31][k = l >> 5] and B[k = l >> 5][col l & 31]; one ds_read_b128 per
// operand (4 consecutive k of the lane's k half) feeds the 4 MFMAs of a plane and chunk.
__global__ __launch_bounds__(256, 1) void wino_fused_kernel(const WinoFusedArgs a, const int ntiles, const int nblk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ __attribute__((aligned(16))) float ss[2 * BN];
    float *const Vl = lds, *const Ul = lds + 2 * OPBUF, *const Sl = lds + 4 * OPBUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: role branches below are scalar
    const int logical = a3d_xcd_remap(blockIdx.x, nblk);
    const int mt = logical / ntiles, nt = logical - mt * ntiles;
    const int n0 = nt * BN;
    const int NCH = a.C / BKC;
    const int lbw = a.lbw, BW = 1 << lbw, BH = 64 >> lbw;
    const int RW = 2 * BW + 2, RH = 2 * BH + 2, P2 = 2 * RW * RH;   // staged region (pixels) and its 16-byte slots
    const int bx = mt % a.NBX, bq = mt / a.NBX;
    const int by = bq % a.NBY, b = bq / a.NBY;
    const int y0 = 2 * by * BH, x0 = 2 * bx * BW;                   // first output pixel of the block

    // ---- DMA role: slot s = 16 bytes = (pixel s >> 1 of the region, channels 4 (s & 1) .. +3); wave w issues instructions w + 4i ----
    const __amdgpu_buffer_rsrc_t rx = fmake_rsrc(a.x, (unsigned)((size_t)a.B * a.H * a.W * a.C * 4));
    int svoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = (wave + 4 * i) * 64 + lane, pidx = s >> 1;
        const int ry = pidx / RW, rxx = pidx - ry * RW;
        const int py = y0 - 1 + ry, px = x0 - 1 + rxx;
        const bool ok = s < P2 && (unsigned)py < (unsigned)a.H && (unsigned)px < (unsigned)a.W;
        svoff[i] = ok ? (((b * a.H + py) * a.W + px) * a.C + (s & 1) * 4) * 4 : -1;
    }
    auto dma_patch = [&](int c, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((wave + 4 * i) * 64 < P2) dma16(rx, Sl + buf * STG + (wave + 4 * i) * 256, svoff[i], funi(c * BKC * 4));
    };
    const int NT = (a.Cout + BN - 1) / BN;
    const int uchunk = 16 * NT * 512 * 4;            // bytes between consecutive chunks of Uc
    const __amdgpu_buffer_rsrc_t ru = fmake_rsrc(a.Uc, (unsigned)((size_t)NCH * uchunk));
    auto dma_u = [&](int c, int buf, int i0, int i1) {  // 32 half planes per chunk, 8 per wave
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < i0 || i >= i1) continue;
            const int hp = wave * 8 + i;
            dma16(ru, Ul + buf * OPBUF + hp * HALF, lane * 16, funi(c * uchunk + ((hp >> 1) * NT + nt) * 2048 + (hp & 1) * 1024));
        }
    };

    // ---- transform role: thread = (tile, channel pair): the whole 4x4 patch -> 16 planes ------------------------------------------
    const int cp = lane & 3;                         // channel pair of the chunk: channels 2cp, 2cp+1
    const int ltile = wave * 16 + (lane >> 2);
    const int sbase = ((2 * (ltile >> lbw)) * RW + 2 * (ltile & (BW - 1))) * 8 + 2 * cp;   // word offset of patch pixel (0, 0)
    const int vbase = (cp >> 1) ? (HALF + (ltile ^ 4) * 4 + (cp & 1) * 2) : (ltile * 4 + (cp & 1) * 2);

    // ---- MFMA role ---------------------------------------------------------------------------------------------------------------
    const int wm = wave >> 1, wn = wave & 1;
    const int kh = lane >> 5, row = lane & 31;
    const int fr_v = kh ? (HALF + ((wm * 32 + row) ^ 4) * 4) : ((wm * 32 + row) * 4);
    const int fr_u = kh * HALF + (wn * 32 + row) * 4;

    f32x16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;

    struct Frag {
        f32x4 v, u;
    };
    auto read_frag = [&](Frag &fr, int buf, int f) {
        fr.v = *reinterpret_cast<const f32x4 *>(Vl + buf * OPBUF + f * PLANE + fr_v);
        fr.u = *reinterpret_cast<const f32x4 *>(Ul + buf * OPBUF + f * PLANE + fr_u);
    };
    // B^T d B in slices.  Column q of the patch (4 pixels) -> column q of m = B^T d; then plane 4u + v from row u of m.
    f32x2 m[4][4];
    auto transform_col = [&](int sbuf, int q) {
        const float *S = Sl + sbuf * STG + sbase + q * 8;
        const f32x2 d0 = *reinterpret_cast<const f32x2 *>(S), d1 = *reinterpret_cast<const f32x2 *>(S + RW * 8);
        const f32x2 d2 = *reinterpret_cast<const f32x2 *>(S + 2 * RW * 8), d3 = *reinterpret_cast<const f32x2 *>(S + 3 * RW * 8);
        m[0][q] = d0 - d2;
        m[1][q] = d1 + d2;
        m[2][q] = d2 - d1;
        m[3][q] = d1 - d3;
    };
    auto transform_out = [&](int buf, int f) {
        const int u = f >> 2, v = f & 3;
        f32x2 o;
        if (v == 0) o = m[u][0] - m[u][2];
        else if (v == 1) o = m[u][1] + m[u][2];
        else if (v == 2) o = m[u][2] - m[u][1];
        else o = m[u][1] - m[u][3];
        *reinterpret_cast<f32x2 *>(Vl + buf * OPBUF + f * PLANE + vbase) = o;
    };

    // one chunk: MFMAs of chunk c (V/U[c & 1]); B^T d B of chunk c+1 (staging[(c+1) & 1] -> V[(c+1) & 1]); DMA of the weights of
    // chunk c+1 (-> U[(c+1) & 1]) and of the pixels of chunk c+2 (-> staging[c & 1], whose previous content was consumed during
    // iteration c-1).  Step f = {fragment reads of plane f+1; 4 MFMA of plane f; one slice of the other work}, fenced.
    auto iteration = [&](const int c, const bool more, const bool more2) {
        const int buf = c & 1, obuf = buf ^ 1;
        Frag fa[2], fb[2];
        read_frag(fa[0], buf, 0);
        read_frag(fa[1], buf, 1);
        // two planes per step, their MFMAs interleaved: consecutive MFMAs never accumulate into the same registers
        a3d_static_for<8>([&](auto fc) {
            constexpr int h = decltype(fc)::value, f = 2 * h;
            Frag(&fr)[2] = (h & 1) ? fb : fa;
            Frag(&fn)[2] = (h & 1) ? fa : fb;
            if (h + 1 < 8) {
                read_frag(fn[0], buf, f + 2);
                read_frag(fn[1], buf, f + 3);
            }
            __builtin_amdgcn_sched_barrier(0);  // the next planes' fragments are in flight BEFORE these planes' MFMAs issue
            if (h < 2) {
                if (more) dma_u(c + 1, obuf, 4 * h, 4 * h + 4);
                if (h == 0 && more2) dma_patch(c + 2, buf);
            }
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[0].u[0], fr[0].v[0], acc[f], 0, 0, 0);
            acc[f + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[1].u[0], fr[1].v[0], acc[f + 1], 0, 0, 0);
            if (more) {
                if (h < 2) {
                    transform_col(obuf, 2 * h);
                    transform_col(obuf, 2 * h + 1);
                } else {
                    transform_out(obuf, 4 * (h - 2));
                }
            }
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[0].u[1], fr[0].v[1], acc[f], 0, 0, 0);
            acc[f + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[1].u[1], fr[1].v[1], acc[f + 1], 0, 0, 0);
            if (more && h >= 2) transform_out(obuf, 4 * (h - 2) + 1);
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[0].u[2], fr[0].v[2], acc[f], 0, 0, 0);
            acc[f + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[1].u[2], fr[1].v[2], acc[f + 1], 0, 0, 0);
            if (more && h >= 2) {
                transform_out(obuf, 4 * (h - 2) + 2);
                transform_out(obuf, 4 * (h - 2) + 3);
            }
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[0].u[3], fr[0].v[3], acc[f], 0, 0, 0);
            acc[f + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[1].u[3], fr[1].v[3], acc[f + 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        // every DMA of this wave has landed in LDS before the barrier publishes the buffers to the other waves
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    if (tid < BN) {  // epilogue vectors of this N tile (visible through the barriers of the main loop)
        const int n = n0 + tid;
        ss[tid] = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
        ss[BN + tid] = (a.shift && n < a.Cout) ? a.shift[n] : 0.f;
    }
    // prologue: pixels of chunk 0 and 1, weights of chunk 0; transform chunk 0
    dma_patch(0, 0);
    dma_u(0, 0, 0, 8);
    if (NCH > 1) dma_patch(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) transform_col(0, q);
#pragma unroll
    for (int f = 0; f < 16; ++f) transform_out(0, f);
    __syncthreads();
    for (int c = 0; c + 2 < NCH; ++c) iteration(c, true, true);
    if (NCH > 1) iteration(NCH - 2, true, false);
    iteration(NCH - 1, false, false);

    // ---- epilogue: fold the 16 planes into the 2x2 outputs (A^T M A, coefficients 0 / +-1), scale / shift / activation -------
    const int tl = wm * 32 + row;
    const int oy0 = y0 + 2 * (tl >> lbw), ox0 = x0 + 2 * (tl & (BW - 1));
    if (oy0 >= a.H || ox0 >= a.W) return;
    f32x16 yv[4];
#pragma unroll
    for (int ij = 0; ij < 4; ++ij)
#pragma unroll
        for (int r = 0; r < 16; ++r) yv[ij][r] = 0.f;
#pragma unroll
    for (int f = 0; f < 16; ++f) {
        const int u = f >> 2, v = f & 3;
        // A^T = [1 1 1 0; 0 1 -1 -1]
        const int au0 = (u < 3) ? 1 : 0, au1 = (u == 0) ? 0 : ((u == 1) ? 1 : -1);
        const int av0 = (v < 3) ? 1 : 0, av1 = (v == 0) ? 0 : ((v == 1) ? 1 : -1);
        const int cf[4] = {au0 * av0, au0 * av1, au1 * av0, au1 * av1};
#pragma unroll
        for (int ij = 0; ij < 4; ++ij) {
            if (cf[ij] == 1) yv[ij] += acc[f];
            else if (cf[ij] == -1) yv[ij] -= acc[f];
        }
    }
#pragma unroll
    for (int ij = 0; ij < 4; ++ij) {
        const int oy = oy0 + (ij >> 1), ox = ox0 + (ij & 1);
        if (oy >= a.H || ox >= a.W) continue;
        const size_t prow = (((size_t)b * a.H + oy) * a.W + ox) * a.Cout;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int nl = wn * 32 + rg * 8 + kh * 4;
            const int n = n0 + nl;
            if (n >= a.Cout) continue;
            f32x4 o = {yv[ij][rg * 4 + 0], yv[ij][rg * 4 + 1], yv[ij][rg * 4 + 2], yv[ij][rg * 4 + 3]};
            const f32x4 sc = *reinterpret_cast<const f32x4 *>(ss + nl), sh = *reinterpret_cast<const f32x4 *>(ss + BN + nl);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = __builtin_fmaf(o[k], sc[k], sh[k]);
            if (a.act == A3D_ACT_RELU) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = o[k] <= 0.f ? 0.f : o[k];
            } else if (a.act == A3D_ACT_LEAKY) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = o[k] > 0.f ? o[k] : 0.01f * o[k];
            }
            if (a.gate) {
                const f32x4 gt = *reinterpret_cast<const f32x4 *>(a.gate + prow + n);
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = gt[k] > 0.f ? o[k] : 0.f;
            }
            *reinterpret_cast<f32x4 *>(a.y + prow + n) = o;
        }
    }
}

// block shape (log2 width in tiles) that wastes the fewest tile slots; *blocks = blocks per image
static int pick_block(int Ty, int Tx, int *nby, int *nbx) {
    int best = -1;
    long best_n = 0;
    const int cand[4] = {3, 4, 2, 5};  // 8x8, 4x16, 16x4, 2x32 (earlier wins ties)
    for (int i = 0; i < 4; ++i) {
        const int bw = 1 << cand[i], bh = 64 >> cand[i];
        const long n = (long)((Ty + bh - 1) / bh) * ((Tx + bw - 1) / bw);
        if (best < 0 || n < best_n) {
            best = cand[i];
            best_n = n;
        }
    }
    const int bw = 1 << best, bh = 64 >> best;
    *nby = (Ty + bh - 1) / bh;
    *nbx = (Tx + bw - 1) / bw;
    return best;
}

}  // namespace

// One-launch Winograd path: plain 3x3 s1 p1 layers (one source, no upsampling) whose descriptor carries the LDS-image
// weights (a3d_conv_desc.w_wino_cm) and whose tile grid fills the 64-tile blocks to at least 3/4.  No workspace.
int a3d_wino_fused_eligible(const a3d_conv_desc *d) {
    if (!d->w_wino_cm || d->precision != 0) return 0;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1) return 0;
    if (d->res || d->pixshuf || d->stem || d->splitk != 1 || d->m_dev || d->ups || d->x2 || d->Cin2 || d->phase) return 0;
    if ((d->Cin & 7) || (d->Cout & 3)) return 0;
    if ((size_t)d->B * d->H * d->W * d->Cin * 4 >= ((size_t)1 << 31)) return 0;
    const int Ty = (d->H + 1) / 2, Tx = (d->W + 1) / 2;
    int nby, nbx;
    pick_block(Ty, Tx, &nby, &nbx);
    if ((long)nby * nbx * 64 * 3 > (long)Ty * Tx * 4) return 0;  // small / odd maps: the linear-tile two-launch form wastes nothing
    // measured (tools/wino_fused_check.py): ahead of the two-launch form from the p3 level (30x40 tiles per image) upwards -- 64- and
    // 128-channel layers by 15-30 %, 256-channel ones by 2-5 % -- behind it on the small maps (p4 and below, the 14x14 ROI heads),
    // where one workgroup per CU leaves the chip half empty in the last round.  A3D_WINO_FUSED_MIN_TILES overrides (A/B runs).
    const long min_tiles = a3d_dev_knob("A3D_WINO_FUSED_MIN_TILES", 1200);
    if ((long)Ty * Tx < min_tiles) return 0;
    if ((size_t)d->B * nby * nbx >= ((size_t)1 << 24)) return 0;
    return 1;
}

int a3d_conv_launch_wino_fused(const a3d_conv_desc *d, hipStream_t s) {
    if (!a3d_wino_fused_eligible(d)) return A3D_ERR_UNSUPPORTED;
    WinoFusedArgs a;
    a.x = d->x;
    a.Uc = d->w_wino_cm;
    a.scale = d->scale;
    a.shift = d->shift;
    a.gate = d->gate;
    a.y = d->y;
    a.C = d->Cin;
    a.Cout = d->Cout;
    a.B = d->B;
    a.H = d->H;
    a.W = d->W;
    a.act = d->act;
    a.lbw = pick_block((d->H + 1) / 2, (d->W + 1) / 2, &a.NBY, &a.NBX);
    const int mtiles = d->B * a.NBY * a.NBX, ntiles = (d->Cout + BN - 1) / BN;
    static a3d_attr_once attr_set;
    if (attr_set.needed()) {  // > 64 KiB of dynamic LDS needs the opt-in attribute (once per device)
        if (hipFuncSetAttribute((const void *)wino_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * 4) != hipSuccess)
            return A3D_ERR_LAUNCH;
        attr_set.mark();
    }
    a3d_note_variant("wino_fused_kernel %dx%d tiles x 64 ch, bk8 (F(2x2,3x3), input transform in the loader)", 64 >> a.lbw, 1 << a.lbw);
    hipLaunchKernelGGL(wino_fused_kernel, dim3(mtiles * ntiles), dim3(256), LDS_FLOATS * 4, s, a, ntiles, mtiles * ntiles);
    return a3d_check_launch();
}
