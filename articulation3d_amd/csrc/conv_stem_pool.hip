// The ResNet stem in ONE launch (round 4): 7x7 stride-2 pad-3 convolution + FrozenBN + ReLU + 3x3 stride-2 pad-1 max-pool, fp16x2 arithmetic.
// Replaces, in the default arithmetic, conv_x3_kernel<1, STEM> followed by maxpool3x3s2_kernel (detectron2 BasicStem.forward, reached
// from pkg/modeling/meta_arch/planercnn.py:150): 0.88 + 0.34 ms per 64 frames, a 1.26 GB intermediate written and read back, and a
// loader that fetches every input pixel ~12 times through L1 (a 7x7 window of a 4-channel pixel per output, three rewrites of that
// kernel tied at 0.8 ms).
//
// Here a workgroup owns a 6 x 9 tile of POOLED pixels.  The 31 x 43 patch of the (normalised, NHWC4) input under it is loaded once, split
// once into the two fp16 planes (per-image scale, as every fp16x2 kernel) and kept in LDS as [row][column][4 channels]; the B fragment of
// a 16-deep chunk -- 4 filter columns x 4 channels of one filter row -- for 32 consecutive conv pixels of a row is then 32 x 16 contiguous
// bytes of that image at stride 16 (conflict-free ds_read_b128).  The FILTER is stationary in registers: 14 chunks x 2 planes x one
// 32-channel block per wave = 112 VGPRs, loaded once per (persistent) workgroup.  8 waves = 2 channel blocks x 4 quarters of the
// 13 x 19 conv pixels (8 blocks of 32) under the pooled tile; conv outputs go through the shared epilogue into an LDS staging tile,
// and the pool reads its 3 x 3 windows from there.  Nothing but the pooled tensor is written.
//
// Bits: the chunk order (filter row, column half), the three product terms per chunk (h.h, h.l, l.h), the exact power-of-two
// un-scaling and the fused multiply-add epilogue are conv_x3_kernel's, and the pool's comparison order is maxpool3x3s2_kernel's:
// the output equals the two-launch form's bit for bit (tests/test_gpu_parity.py).
#include "conv_common.h"
#include <limits.h>
#include <stdlib.h>

namespace {
typedef _Float16 sp_h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 sp_h16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int SP_NKC = 14;  // 16-deep chunks: 7 filter rows x 2 column halves
constexpr int SP_SP = 68;   // staging pitch in floats (64 channels + 4: b128 stores of 16 lanes cover all banks)
// PH x PW: the pooled tile of a workgroup; NW waves = 2 channel blocks x NW / 2 pairs of 32-pixel blocks of the conv pixels under it.
//   6 x 9, 8 waves: 13 x 19 = 247 conv pixels (8 blocks), 92 KiB of LDS -> ONE workgroup per CU: its phases (patch, conv, pool) run one
//     after the other and the matrix pipe idles through two of them (0.72 ms per 64 frames: conv 0.30, pool 0.19, the rest 0.22)
//   3 x 8, 4 waves: 7 x 17 = 119 conv pixels (4 blocks), 47 KiB -> TWO workgroups per CU out of phase with each other
template <int PH, int PW, int NW>
struct SpCfg {
    static constexpr int NT = 64 * NW;
    static constexpr int CH = 2 * PH + 1, CW = 2 * PW + 1, NQ = CH * CW;  // conv pixels under the tile
    static constexpr int NBLK = NW;                                        // 32-pixel blocks (two per wave pair)
    static constexpr int IH = 2 * CH + 5, IW = 2 * CW + 6, NPIX = IH * IW;  // input patch (+ the zero-weight 8th filter column)
    static constexpr int PLANE = NPIX * 8;                                  // bytes of one plane: 4 fp16 per pixel
    static constexpr int LI = (NPIX + NT - 1) / NT;                         // patch pixels per thread
    static constexpr int PI = (PH * PW * 16 + NT - 1) / NT;                 // pool items per thread
    static constexpr int LDS = 2 * PLANE + 32 * NBLK * SP_SP * 4 + 128 * 4;
    static_assert(NQ <= 32 * NBLK, "the blocks cover the conv pixels");
};

template <int PH, int PW, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) void stem_pool_kernel(const a3d_conv_desc d, const int Hp, const int Wp, const int tiles_x, const int tiles_y, const int total) {
    using K = SpCfg<PH, PW, NW>;
    constexpr int SP_PH = PH, SP_PW = PW, SP_CW = K::CW, SP_NQ = K::NQ, SP_IW = K::IW, SP_NPIX = K::NPIX, SP_PLANE = K::PLANE, SP_LI = K::LI,
                  SP_PI = K::PI, NT = K::NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_lds[];
    unsigned char *const Xh = sp_lds, *const Xl = sp_lds + SP_PLANE;
    float *const stg = reinterpret_cast<float *>(sp_lds + 2 * SP_PLANE);
    float *const ss = stg + 32 * K::NBLK * SP_SP;  // scale[64] | shift[64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = wave & 1, part = wave >> 1;
    const int khalf = lane >> 5, frow = lane & 31;

    // ---- the wave's filter fragments: w_x3 [chunk][plane][64][16] fp16, row nb * 32 + lane % 32, k half lane / 32
    sp_h16x8 Wf[SP_NKC][2];
    {
        const sp_h16x8 *w = reinterpret_cast<const sp_h16x8 *>(d.w_x3);
#pragma unroll
        for (int kc = 0; kc < SP_NKC; ++kc)
#pragma unroll
            for (int p = 0; p < 2; ++p) Wf[kc][p] = w[((kc * 2 + p) * 64 + nb * 32 + frow) * 2 + khalf];
    }
    a3d_stage_scale_shift(ss, d, 0, 64, tid);
    // ---- the wave's two blocks of conv pixels: q = 32 (2 part + j) + lane % 32 -> (row q / CW, column q % CW) of the CH x CW conv pixels
    int bbase[2], qq[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = (2 * part + j) * 32 + frow;
        qq[j] = q;
        const int qc = min(q, SP_NQ - 1);
        const int cy = qc / SP_CW, cx = qc - cy * SP_CW;
        bbase[j] = ((2 * cy) * SP_IW + 2 * cx + 2 * khalf) * 8;  // bytes: patch pixel of filter tap (0, 2 khalf) of this conv pixel
    }
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(d.x), 0, (int)((size_t)d.B * d.H * d.W * 16), 0x00020000);
    const float unw = 1.f / d.w_scale;
    const bool relu = d.act == A3D_ACT_RELU;
    const int tpi = tiles_x * tiles_y;
    // pool items of this thread: (pooled pixel, channel quad) = idx / 16, idx % 16 for idx = tid, tid + 512 (the same in every tile)
    int pl_stg[SP_PI], pl_pyx[SP_PI], pl_c4[SP_PI];
#pragma unroll
    for (int it = 0; it < SP_PI; ++it) {
        const int idx = tid + NT * it;
        const int pp = idx >> 4, ppy = pp / SP_PW, ppx = pp - ppy * SP_PW;
        pl_c4[it] = (idx & 15) * 4;
        pl_pyx[it] = (ppy << 8) | ppx;
        pl_stg[it] = idx < SP_PH * SP_PW * 16 ? ((2 * ppy) * SP_CW + 2 * ppx) * SP_SP + pl_c4[it] : -1;
    }

    // patch loads of tile t into registers (issued one tile ahead: in flight across the previous tile's MFMAs)
    f32x4 xs[SP_LI];
    auto load_patch = [&](const int t) {
#ifdef A3D_ABLATIONS
        if (d.tune & 32) return;
#endif
        const int b = t / tpi, tr = t - b * tpi;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int iy0 = 2 * (2 * ty * SP_PH - 1) - 3, ix0 = 2 * (2 * tx * SP_PW - 1) - 3;
#pragma unroll
        for (int i = 0; i < SP_LI; ++i) {
            const int j = tid + NT * i;
            const int r = j / SP_IW, c = j - r * SP_IW;
            const int y = iy0 + r, x = ix0 + c;
            const bool inb = t < total && j < SP_NPIX && (unsigned)y < (unsigned)d.H && (unsigned)x < (unsigned)d.W;
            xs[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, inb ? ((b * d.H + y) * d.W + x) * 16 : -1, 0, 0));
        }
    };
    // a workgroup walks a CONTIGUOUS range of tiles: the image changes a couple of times per workgroup, so the two things that depend on
    // the image -- its input scale (a global read) and its output maximum (a read + an atomic) -- cost one memory round trip per image
    // instead of two per tile (1-2 us each against ~5 us of work)
    const int per = (total + gridDim.x - 1) / gridDim.x;
    const int t_begin = blockIdx.x * per, t_end = min(total, t_begin + per);
    int b_cur = -1, b_sx = -1;
    float sx_p = 1.f, vmax = 0.f;  // sx_p: the scale the patch in LDS (or on its way there) was split under = that of image b_sx
    auto store_patch = [&](const float sx) {  // split, store (first input row / column of a tile's patch: 2 cy0 - 3, 2 cx0 - 3)
#ifdef A3D_ABLATIONS
        if (d.tune & 16) return;
#endif
#pragma unroll
        for (int i = 0; i < SP_LI; ++i) {
            const int j = tid + NT * i;
            if (j >= SP_NPIX) continue;
            const f32x4 v = xs[i] * sx;
            const sp_h16x4 h = __builtin_convertvector(v, sp_h16x4);
            const sp_h16x4 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), sp_h16x4);
            *reinterpret_cast<sp_h16x4 *>(Xh + j * 8) = h;
            *reinterpret_cast<sp_h16x4 *>(Xl + j * 8) = l;
        }
    };
    // Order inside an iteration: [barrier] conv of tile t, [barrier] patch of tile t + 1 into LDS, pool + stores of tile t, loads of
    // tile t + 2.  The loads are consumed a whole conv phase after their issue, and the counted wait in front of the patch stores has
    // nothing younger than its loads in flight -- with the pool's stores issued BEFORE that wait (the first form of this loop) every
    // tile waited for its stores to be acknowledged: 2.4 of 8 us.
    if (t_begin < t_end) {
        load_patch(t_begin);
        b_sx = t_begin / tpi;
        sx_p = a3d_in_scale(d, b_sx);
        store_patch(sx_p);
    }
    load_patch(t_begin + 1 < t_end ? t_begin + 1 : total);
    for (int t = t_begin; t < t_end; ++t) {
        const int b = t / tpi, tr = t - b * tpi;
        const int ty = tr / tiles_x, tx = tr - ty * tiles_x;
        const int py0 = ty * SP_PH, px0 = tx * SP_PW;
        const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;  // first conv row / column under the tile (the pool's pad row / column: -1)
        if (b != b_cur) {
            if (b_cur >= 0 && d.y_amax) a3d_note_amax(d.y_amax, b_cur, vmax, true);
            vmax = 0.f;
            b_cur = b;
        }
        const float unx = 1.f / sx_p;  // (the patch in LDS is tile t's: split under image b's scale)
        // (BARE barriers: __syncthreads() carries a fence, and in front of a fence the compiler waits for every global access in flight)
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // ---- conv: the wave's two blocks of 32 pixels x its 32 channels, side by side (two independent accumulator chains; per
        // accumulator the order is chunk by chunk h.h, h.l, l.h); the fragments of chunk kc + 1 are requested before chunk kc multiplies
        {
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            sp_h16x8 bh[2][2], bl[2][2];  // [chunk parity][block]
            auto rd = [&](const int kc, const int set) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int off = bbase[j] + ((kc >> 1) * SP_IW + 4 * (kc & 1)) * 8;
                    bh[set][j] = *reinterpret_cast<const sp_h16x8 *>(Xh + off);
                    bl[set][j] = *reinterpret_cast<const sp_h16x8 *>(Xl + off);
                }
            };
#ifdef A3D_ABLATIONS
            if (!(d.tune & 1)) {
#endif
            rd(0, 0);
#pragma unroll
            for (int kc = 0; kc < SP_NKC; ++kc) {
                const int s = kc & 1;
                if (kc + 1 < SP_NKC) rd(kc + 1, s ^ 1);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[kc][0], bh[s][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[kc][0], bh[s][1], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[kc][0], bl[s][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[kc][0], bl[s][1], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[kc][1], bh[s][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Wf[kc][1], bh[s][1], acc[1], 0, 0, 0);
            }
#ifdef A3D_ABLATIONS
            }
#endif
            // accumulator register r of lane l: channel nb * 32 + (r / 4) * 8 + (l / 32) * 4 + r % 4, pixel block row l % 32
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int n = nb * 32 + rg * 8 + khalf * 4;
                    f32x4 v = {acc[j][rg * 4 + 0], acc[j][rg * 4 + 1], acc[j][rg * 4 + 2], acc[j][rg * 4 + 3]};
                    v = (v * unx) * unw;  // exact: powers of two
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    v = a3d_epilogue_math(d, v, *reinterpret_cast<const f32x4 *>(ss + n), *reinterpret_cast<const f32x4 *>(ss + 64 + n), false, zero);
                    *reinterpret_cast<f32x4 *>(stg + qq[j] * SP_SP + n) = v;
                }
        }
        __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + 1 < t_end) {  // the next tile's patch (every wave is past its fragment reads of this one)
            const int bn = (t + 1) / tpi;
            if (bn != b_sx) {
                b_sx = bn;
                sx_p = a3d_in_scale(d, bn);
            }
            store_patch(sx_p);
        }
        // ---- pool: thread = (pooled pixel, channel quad); the window in maxpool3x3s2_kernel's order, a NaN wins
#ifdef A3D_ABLATIONS
        if (!(d.tune & 2))
#endif
#pragma unroll
        for (int it = 0; it < SP_PI; ++it) {
            if (pl_stg[it] < 0) continue;
            const int ppy = pl_pyx[it] >> 8, ppx = pl_pyx[it] & 255;
            const int py = py0 + ppy, px = px0 + ppx;
            if (py >= Hp || px >= Wp) continue;
            const float *sp = stg + pl_stg[it];
            f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (relu) {
                // behind a ReLU every value is +0, positive or NaN: as SIGNED INTEGERS the (sign-cleared) bit patterns order exactly like
                // the floats and every NaN sorts above +inf, so one v_max_i32 per element is maxpool3x3s2_kernel's "larger, or NaN, wins"
                // -- four vector instructions per element otherwise, and the pool phase was bound by them.  (A NaN comes out with its
                // sign bit cleared: the only difference to the two-launch form.)
                i32x4 mi = {INT_MIN, INT_MIN, INT_MIN, INT_MIN};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    if ((unsigned)(cy0 + 2 * ppy + dy) >= (unsigned)d.Ho) continue;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        if ((unsigned)(cx0 + 2 * ppx + dx) >= (unsigned)d.Wo) continue;
                        const i32x4 v = *reinterpret_cast<const i32x4 *>(sp + (dy * SP_CW + dx) * SP_SP) & 0x7FFFFFFF;
#pragma unroll
                        for (int k = 0; k < 4; ++k) mi[k] = max(mi[k], v[k]);
                    }
                }
                m = __builtin_bit_cast(f32x4, mi);
            } else {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    if ((unsigned)(cy0 + 2 * ppy + dy) >= (unsigned)d.Ho) continue;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        if ((unsigned)(cx0 + 2 * ppx + dx) >= (unsigned)d.Wo) continue;
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(sp + (dy * SP_CW + dx) * SP_SP);
#pragma unroll
                        for (int k = 0; k < 4; ++k) m[k] = (v[k] > m[k] || v[k] != v[k]) ? v[k] : m[k];
                    }
                }
            }
#ifdef A3D_ABLATIONS
            if (!(d.tune & 4))
#endif
            *reinterpret_cast<f32x4 *>(d.y + (((size_t)b * Hp + py) * Wp + px) * 64 + pl_c4[it]) = m;
            vmax = fmaxf(vmax, a3d_absmax4(m));  // (max of the pooled tile = max of the conv pixels under it)
        }
        load_patch(t + 2 < t_end ? t + 2 : total);
        // (the next iteration's first barrier: the patch is complete, and every thread is past this pool before the staging tile is
        // written again)
    }
    if (b_cur >= 0 && d.y_amax) a3d_note_amax(d.y_amax, b_cur, vmax, true);
}
template <int PH, int PW, int NW>
int sp_launch(const a3d_conv_desc *d, hipStream_t s, const int Hp, const int Wp, const int slots) {
    using K = SpCfg<PH, PW, NW>;
    const int tiles_y = (Hp + PH - 1) / PH, tiles_x = (Wp + PW - 1) / PW;
    const long total = (long)d->B * tiles_x * tiles_y;
    if (total >= (1l << 30)) return A3D_ERR_UNSUPPORTED;
    static a3d_attr_once attr;
    if (attr.needed()) {
        if (hipFuncSetAttribute((const void *)stem_pool_kernel<PH, PW, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS) != hipSuccess) return A3D_ERR_LAUNCH;
        attr.mark();
    }
    const int grid = (int)(total < slots ? total : slots);  // persistent: every workgroup walks a contiguous range of tiles (the filter is loaded once)
    hipLaunchKernelGGL((stem_pool_kernel<PH, PW, NW>), dim3(grid), dim3(K::NT), K::LDS, s, *d, Hp, Wp, tiles_x, tiles_y, (int)total);
    return a3d_check_launch();
}
}  // namespace

// d: the stem's descriptor as for a3d_conv2d_nhwc_f32 (stem = 1, precision 3, w_x3, in_amax, w_scale; Ho x Wo = the CONV output size),
// except that d->y (and d->y_amax) is the POOLED tensor [B, (Ho - 1) / 2 + 1, (Wo - 1) / 2 + 1, 64].
extern "C" int a3d_stem_conv_pool(const a3d_conv_desc *d, void *stream) {
    if (!d || !d->x || !d->w_x3 || !d->y || !d->in_amax || !(d->w_scale > 0.f)) return A3D_ERR_ARG;
    if (!d->stem || d->precision != 3 || d->Cout != 64 || d->KH != 7 || d->KW != 7 || d->stride != 2 || d->pad != 3 || d->Kpad != 224) return A3D_ERR_UNSUPPORTED;
    if (d->Ho != (d->H + 6 - 7) / 2 + 1 || d->Wo != (d->W + 6 - 7) / 2 + 1 || d->x2 || d->res || d->gate || d->ups || d->phase || d->pixshuf || d->splitk != 1 || d->m_dev)
        return A3D_ERR_UNSUPPORTED;
    if ((size_t)d->B * d->H * d->W * 16 >= ((size_t)1 << 31)) return A3D_ERR_UNSUPPORTED;
    const int Hp = (d->Ho - 1) / 2 + 1, Wp = (d->Wo - 1) / 2 + 1;
    a3d_begin();
    // (one attribute query per call -- a few hundred ns, no 1 KiB hipDeviceProp_t fill on the host-bound 1-2 frame path; no static: the
    // library keeps no process-global state and a second device of the process may have another CU count)
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const int form = (int)a3d_dev_knob("A3D_STEM_TILE", 0);  // (developer builds, A/B runs: 2 = the 3 x 8 tile, two workgroups per CU)
    a3d_note_variant("stem_pool_kernel");
    // (measured at 64 frames: 6 x 9 tiles 0.705 ms, 3 x 8 tiles with two workgroups per CU 0.728 -- what a tile costs beside its MFMAs is
    // vector work of the pool and the epilogue, which a second workgroup on the same SIMDs does not hide)
    if (form == 2) return sp_launch<3, 8, 4>(d, (hipStream_t)stream, Hp, Wp, 2 * cus);
    return sp_launch<6, 9, 8>(d, (hipStream_t)stream, Hp, Wp, cus);
}
