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
// accumulators M_f to the end (the fold Y_ij = sum_f c_ij,f M_f is linear and runs once, in the epilogue).  To fit 16
// accumulators the wave tile is 16 tiles x 32 channels on v_mfma_f32_16x16x4_f32 (16 planes x 2 channel blocks x 4 = 128
// accumulator registers, two waves per SIMD); the workgroup is 8 waves = 64 tiles x 64 channels, one per CU (128 KiB LDS).
//
//   A operand = U_f (16 output channels x 4 k), B operand = V_f (4 k x 16 tiles)  ->  lane l owns tile (l & 15) and the
//   4 consecutive channels 4*(l >> 4) .. +3 of a block: float4 NHWC stores, as in every other conv kernel of this library.
//
// Per chunk and workgroup: 12 buffer_load_dwordx2 per thread fetch the raw patch (thread = tile x channel pair x row half),
// 32 VALU adds transform it, 8 ds_write_b64 publish 8 of the 16 planes; 4 dwordx4 loads + 4 ds_write_b128 copy the U chunk
// (weights are pre-packed chunk-major [C/8][16][Cout][8], so a chunk of a plane is one contiguous 2 KiB run); then 16 x
// (3 ds_read_b64 + 4 MFMA).  Loads run two chunks ahead of the MFMAs, LDS is double-buffered, one barrier per chunk.
// LDS image per plane and operand: [k half][row][4 floats] -- lanes 0-31 read the 64 consecutive words of half 0, lanes
// 32-63 those of half 1: conflict-free ds_read_b64.
#include "conv_common.h"
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

__device__ __forceinline__ f32x2 fbuf_load2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 fbuf_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// (base pointer and size pass through v_readfirstlane: they ARE wave-uniform, and saying so keeps the descriptor in SGPRs --
// otherwise hipcc wraps every buffer load of the unrolled loop in a waterfall loop)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fmake_rsrc(const void *p, unsigned bytes) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    void *q = reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo);
    return __builtin_amdgcn_make_buffer_rsrc(q, 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ int funi(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct WinoFusedArgs {
    const float *x;      // [B, H, W, C] NHWC
    const float *Uc;     // Winograd weights in LDS-image order, see the loader
    const float *scale, *shift, *gate;
    float *y;            // [B, H, W, Cout]
    int T, C, Cout, B, H, W, Ty, Tx, act;
};

constexpr int BM = 64, BN = 64, BKC = 8;          // tiles x channels per workgroup, input channels per chunk
constexpr int HALF = BM * 4;                      // floats of one k half of one plane: [64 rows][4]
constexpr int SKEW = 48;                          // half 1 starts 16 banks (mod 32) after half 0: conflict-free ds_write_b64 / b128
constexpr int PLANE = 2 * HALF + SKEW;
constexpr int OPBUF = 16 * PLANE;                 // one operand (V or U), all 16 planes, one chunk
constexpr int LDS_FLOATS = 2 * 2 * OPBUF;         // {V, U} x double buffer = 139264 B

__global__ __launch_bounds__(512, 2) void wino_fused_kernel(const WinoFusedArgs a, const int ntiles, const int nblk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ __attribute__((aligned(16))) float ss[2 * BN];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: role branches below are scalar
    const int logical = a3d_xcd_remap(blockIdx.x, nblk);
    const int mt = logical / ntiles, nt = logical - mt * ntiles;
    const int t0 = mt * BM, n0 = nt * BN;
    const int NCH = a.C / BKC;

    // ---- loader role: (tile, channel pair, row half) ------------------------------------------------------------------
    const int cp = lane & 3;                        // channel pair of the chunk: channels 2cp, 2cp+1
    const int ltile = (wave & 3) * 16 + (lane >> 2);
    const int rh = wave >> 2;                       // 0: patch rows 0..2 -> planes u = 0,1;  1: rows 1..3 -> u = 2,3
    // Patch addressing: ONE per-lane byte offset (the tile's pixel (2ty, 2tx), always inside the image) + a validity bit per patch
    // pixel; the 12 pixel displacements (dy in -1..2 relative to the row half, dx in -1..2) are wave-uniform and travel as the
    // scalar offset of the buffer load.  The resource therefore starts one row + one pixel BEFORE x, so every displacement is
    // non-negative; a lane whose pixel lies in the zero padding (or whose tile is past T) loads with offset 0xFFFFFFFF, which the
    // hardware range check turns into zeros.  (12 offset registers -> 2.)
    int pbase = -1;
    unsigned pmask = 0;
    {
        const int t = t0 + ltile;
        if (t < a.T) {
            const int tx = t % a.Tx;
            const int r = t / a.Tx;
            const int ty = r % a.Ty, b = r / a.Ty;
            pbase = ((((b * a.H + 2 * ty) * a.W + 2 * tx) * a.C) + 2 * cp) * 4;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int py = 2 * ty - 1 + rh + (i >> 2), px = 2 * tx - 1 + (i & 3);
                if ((unsigned)py < (unsigned)a.H && (unsigned)px < (unsigned)a.W) pmask |= 1u << i;
            }
        }
    }
    const int pixb = a.C * 4, rowb = a.W * pixb;
    const unsigned xbytes = (unsigned)((size_t)a.B * a.H * a.W * a.C * 4);
    const __amdgpu_buffer_rsrc_t rx = fmake_rsrc(reinterpret_cast<const char *>(a.x) - (rowb + pixb), xbytes + 4u * (unsigned)(rowb + pixb));
    // U chunk copy.  The weights are pre-packed in the exact order of the LDS image (a3d_conv_desc.w_wino_cm):
    //   [C/8][16 planes][Cout/64 tiles][k half 2][channel half 2][row 16][k pair 2][channel block 2][2]
    // so one (chunk, plane, tile, k half) is a contiguous 1 KiB run = ONE `buffer_load_dwordx4 ... lds` of a wave (LDS-DMA: global
    // -> LDS without passing through registers, no ds_write), and a lane's fragment for BOTH channel blocks is one ds_read_b128
    // (two ds_read_b64 get fused into ds_read2_b64 by hipcc, whose 16-lane / 32-bank grouping made 45 % of the LDS cycles of the
    // first version bank conflicts).  32 half planes per chunk, 4 per wave.
    const int NT = (a.Cout + BN - 1) / BN;
    const int uchunk = 16 * NT * 512 * 4;            // bytes between consecutive chunks
    const __amdgpu_buffer_rsrc_t ru = fmake_rsrc(a.Uc, (unsigned)((size_t)(a.C / BKC) * uchunk));
    auto dma_u = [&](int c, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hp = wave * 4 + i, f = hp >> 1, half = hp & 1;
            float *dst = lds + buf * 2 * OPBUF + OPBUF + f * PLANE + half * (HALF + SKEW);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (__attribute__((address_space(3))) void *)dst, 16, lane * 16,
                                                     funi(c * uchunk + ((f * NT + nt) * 512 + half * 256) * 4), 0, 0);
        }
    };

    // ---- software pipeline ------------------------------------------------------------------------------------------------
    // One iteration = one chunk = 16 "plane steps" of 4 MFMAs.  hipcc, left alone, (a) sinks the global loads below the MFMAs so
    // that the next iteration starts by waiting for them and (b) issues every plane's ds_reads right before the MFMAs that need
    // them (measured: 0.55 of the matrix pipe).  The loop is therefore written step by step with a scheduling fence after every
    // step: step f issues the fragment reads of plane f+1, the 4 MFMAs of plane f, and one slice of the other work --
    //   step 0      : the 4 LDS-DMA loads of the U chunk c+1 (waited for just before the barrier that ends the iteration)
    //   steps 0-3   : the 12 patch loads of chunk c+2 (into the register set chunk c's transform freed one iteration ago)
    //   steps 0-7   : B^T d B of chunk c+1 (loaded a FULL iteration ago) and its 8 ds_write_b64
    // so neither pipe waits for the other: every global load has ~4000 cycles to land, an LDS fragment 128.
    f32x2 psA[12], psB[12];                          // raw patch: two register sets (chunk c+1 being transformed, chunk c+2 in flight)
    f32x4 acc[16][2];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[f][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int tb = wave & 3, chh = wave >> 2;        // MFMA role: wave = (tile block, channel half); lane = (row l & 15, k group l >> 4)
    const int g = lane >> 4, row = lane & 15;
    const int fr_v = (g >> 1) * (HALF + SKEW) + (tb * 16 + row) * 4 + (g & 1) * 2;
    const int fr_u = (g >> 1) * (HALF + SKEW) + chh * 128 + row * 8 + (g & 1) * 4;
    const int vbase = (cp >> 1) * (HALF + SKEW) + ltile * 4 + (cp & 1) * 2;

    auto load_patch = [&](f32x2 (&ps)[12], int c, int i0, int i1) {
        const int soff = min(c, NCH - 1) * BKC * 4;  // (the look-ahead past the last chunk re-reads it: in range, never used)
#pragma unroll
        for (int i = 0; i < 12; ++i)
            if (i >= i0 && i < i1) ps[i] = fbuf_load2(rx, ((pmask >> i) & 1u) ? pbase : -1, funi(soff + (rh + (i >> 2)) * rowb + (i & 3) * pixb));
    };

    struct Frag {
        f32x2 v;
        f32x4 u;  // {block 0: k, k+1 | block 1: k, k+1}
    };
    auto read_frag = [&](Frag &fr, int buf, int f) {
        const float *Vb = lds + buf * 2 * OPBUF + fr_v + f * PLANE;
        const float *Ub = lds + buf * 2 * OPBUF + OPBUF + fr_u + f * PLANE;
        fr.v = *reinterpret_cast<const f32x2 *>(Vb);
        fr.u = *reinterpret_cast<const f32x4 *>(Ub);
    };
    // one slice of B^T d B: rows u = 2rh (s = 0) / 2rh+1 (s = 1) of B^T d, then column v of (B^T d) B -> plane 4u + v
    f32x2 mrow[4];
    auto transform_piece = [&](const f32x2 (&ps)[12], int buf, int k) {
        const int s_ = k >> 2, v = k & 3;
        if (v == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 r0 = ps[q], r1 = ps[4 + q], r2 = ps[8 + q];
                if (rh == 0) mrow[q] = s_ ? (r1 + r2) : (r0 - r2);   // m0 = d0 - d2, m1 = d1 + d2
                else mrow[q] = s_ ? (r0 - r2) : (r1 - r0);            // rows are d1, d2, d3:  m2 = d2 - d1, m3 = d1 - d3
            }
        }
        f32x2 o;
        if (v == 0) o = mrow[0] - mrow[2];
        else if (v == 1) o = mrow[1] + mrow[2];
        else if (v == 2) o = mrow[2] - mrow[1];
        else o = mrow[1] - mrow[3];
        *reinterpret_cast<f32x2 *>(lds + buf * 2 * OPBUF + ((2 * rh + s_) * 4 + v) * PLANE + vbase) = o;
    };

    // one chunk: MFMAs of chunk c (LDS[c & 1]) + transform of chunk c+1 out of `cur` into LDS[(c+1) & 1] + loads of chunk c+2 into `nxt`
    auto iteration = [&](const int c, const f32x2 (&cur)[12], f32x2 (&nxt)[12]) {
        const int buf = c & 1, obuf = buf ^ 1;
        const bool more = c + 1 < NCH;
        Frag fa, fb;
        read_frag(fa, buf, 0);
        a3d_static_for<16>([&](auto fc) {
            constexpr int f = decltype(fc)::value;
            Frag &fr = (f & 1) ? fb : fa;
            Frag &fn = (f & 1) ? fa : fb;
            if (f + 1 < 16) read_frag(fn, buf, f + 1);
            __builtin_amdgcn_sched_barrier(0);  // the next plane's fragments are in flight BEFORE this plane's MFMAs issue
            if (f == 0 && more) dma_u(c + 1, obuf);
            if (f < 4) load_patch(nxt, c + 2, 3 * f, 3 * f + 3);
            acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr.u[0], fr.v[0], acc[f][0], 0, 0, 0);
            acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr.u[2], fr.v[0], acc[f][1], 0, 0, 0);
            if (more && f < 8) transform_piece(cur, obuf, f);
            acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr.u[1], fr.v[1], acc[f][0], 0, 0, 0);
            acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr.u[3], fr.v[1], acc[f][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        // this wave's 4 DMA loads were issued before its 12 patch loads: once at most 12 vector-memory operations are outstanding
        // the U chunk has landed in LDS (the barrier then publishes it to the other waves)
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __syncthreads();
    };

    if (tid < BN) {  // epilogue vectors of this N tile (visible through the barriers of the main loop)
        const int n = n0 + tid;
        ss[tid] = (a.scale && n < a.Cout) ? a.scale[n] : 1.f;
        ss[BN + tid] = (a.shift && n < a.Cout) ? a.shift[n] : 0.f;
    }
    // prologue: chunk 0 -> LDS[0]; chunk 1 -> set B (transformed during iteration 0); iteration 0 loads chunk 2 into set A
    dma_u(0, 0);
    load_patch(psA, 0, 0, 12);
#pragma unroll
    for (int k = 0; k < 8; ++k) transform_piece(psA, 0, k);
    load_patch(psB, 1, 0, 12);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __syncthreads();
    for (int c = 0; c < NCH; c += 2) {  // NCH = C / 8 is even (C % 16 == 0)
        iteration(c, psB, psA);
        iteration(c + 1, psA, psB);
    }

    // ---- epilogue: fold the 16 planes into the 2x2 outputs (A^T M A, coefficients 0 / +-1), scale / shift / activation -------
    const int t = t0 + tb * 16 + row;
    if (t >= a.T) return;
    const int tx = t % a.Tx;
    const int r = t / a.Tx;
    const int ty = r % a.Ty, b = r / a.Ty;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int nl = chh * 32 + nb * 16 + g * 4;
        const int n = n0 + nl;
        if (n >= a.Cout) continue;
        f32x4 yv[4];
#pragma unroll
        for (int ij = 0; ij < 4; ++ij) yv[ij] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            const int u = f >> 2, v = f & 3;
            // A^T = [1 1 1 0; 0 1 -1 -1]
            const int au0 = (u < 3) ? 1 : 0, au1 = (u == 0) ? 0 : ((u == 1) ? 1 : -1);
            const int av0 = (v < 3) ? 1 : 0, av1 = (v == 0) ? 0 : ((v == 1) ? 1 : -1);
            const int cf[4] = {au0 * av0, au0 * av1, au1 * av0, au1 * av1};
#pragma unroll
            for (int ij = 0; ij < 4; ++ij) {
                if (cf[ij] == 1) yv[ij] += acc[f][nb];
                else if (cf[ij] == -1) yv[ij] -= acc[f][nb];
            }
        }
        const f32x4 sc = *reinterpret_cast<const f32x4 *>(ss + nl), sh = *reinterpret_cast<const f32x4 *>(ss + BN + nl);
#pragma unroll
        for (int ij = 0; ij < 4; ++ij) {
            const int oy = 2 * ty + (ij >> 1), ox = 2 * tx + (ij & 1);
            if (oy >= a.H || ox >= a.W) continue;
            const size_t ooff = (((size_t)b * a.H + oy) * a.W + ox) * a.Cout + n;
            f32x4 o = yv[ij];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = __builtin_fmaf(o[k], sc[k], sh[k]);
            if (a.act == A3D_ACT_RELU) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = o[k] > 0.f ? o[k] : 0.f;
            } else if (a.act == A3D_ACT_LEAKY) {
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = o[k] > 0.f ? o[k] : 0.01f * o[k];
            }
            if (a.gate) {
                const f32x4 gt = *reinterpret_cast<const f32x4 *>(a.gate + ooff);
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = gt[k] > 0.f ? o[k] : 0.f;
            }
            *reinterpret_cast<f32x4 *>(a.y + ooff) = o;
        }
    }
}

}  // namespace

// One-launch Winograd path: plain 3x3 s1 p1 layers (one source, no upsampling) whose descriptor carries the chunk-major
// weights (a3d_conv_desc.w_wino_cm).  No workspace.
int a3d_wino_fused_eligible(const a3d_conv_desc *d) {
    if (!d->w_wino_cm || d->precision != 0) return 0;
    if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1) return 0;
    if (d->res || d->pixshuf || d->stem || d->splitk != 1 || d->m_dev || d->ups || d->x2 || d->Cin2 || d->phase) return 0;
    if ((d->Cin & 15) || (d->Cout & 3)) return 0;  // chunks of 8 channels, two per unrolled loop trip
    const size_t T = (size_t)d->B * ((d->H + 1) / 2) * ((d->W + 1) / 2);
    if ((size_t)d->B * d->H * d->W * d->Cin * 4 >= ((size_t)1 << 31) || (size_t)16 * d->Cout * d->Cin * 4 >= ((size_t)1 << 31)) return 0;
    if (T >= ((size_t)1 << 30)) return 0;
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
    a.Ty = (d->H + 1) / 2;
    a.Tx = (d->W + 1) / 2;
    a.T = d->B * a.Ty * a.Tx;
    a.C = d->Cin;
    a.Cout = d->Cout;
    a.B = d->B;
    a.H = d->H;
    a.W = d->W;
    a.act = d->act;
    const int mtiles = (a.T + BM - 1) / BM, ntiles = (d->Cout + BN - 1) / BN;
    static bool attr_set = false;
    if (!attr_set) {  // > 64 KiB of dynamic LDS needs the opt-in attribute (once per process)
        if (hipFuncSetAttribute((const void *)wino_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * 4) != hipSuccess)
            return A3D_ERR_LAUNCH;
        attr_set = true;
    }
    a3d_note_variant("wino_fused_kernel 64x64 bk8 (F(2x2,3x3), input transform in the loader)");
    hipLaunchKernelGGL(wino_fused_kernel, dim3(mtiles * ntiles), dim3(512), LDS_FLOATS * 4, s, a, ntiles, mtiles * ntiles);
    return a3d_check_launch();
}
