// HBM-bound NHWC helpers of the detection path: frame normalisation, the stem max-pool, the FPN
// p6 subsample, bilinear resizes of the depth head and its 64->1 3x3 prediction conv.
// One thread owns one float4 of channels (16 B/lane, coalesced); grid-stride loops capped at 2048 WGs.
#include "a3d_common.h"
#include "../../include/a3d.h"

static inline int grid_for(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b ? b : 1));
}

// ---- (x - mean)/std, uint8 HWC BGR -> fp32 NHWC4 ------------------------------------------------
__global__ __launch_bounds__(256) void preprocess_u8_kernel(const uint8_t *__restrict__ in, float *__restrict__ out,
                                                            size_t npix, float m0, float m1, float m2, float s0,
                                                            float s1, float s2) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const uint8_t *p = in + i * 3;
        f32x4 v = {((float)p[0] - m0) / s0, ((float)p[1] - m1) / s1, ((float)p[2] - m2) / s2, 0.f};
        *reinterpret_cast<f32x4 *>(out + i * 4) = v;
    }
}

__global__ __launch_bounds__(256) void preprocess_chw_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                             int B, size_t hw, float m0, float m1, float m2, float s0,
                                                             float s1, float s2) {
    const size_t npix = (size_t)B * hw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / hw, r = i - b * hw;
        const float *p = in + b * 3 * hw + r;
        f32x4 v = {(p[0] - m0) / s0, (p[hw] - m1) / s1, (p[2 * hw] - m2) / s2, 0.f};
        *reinterpret_cast<f32x4 *>(out + i * 4) = v;
    }
}

extern "C" int a3d_preprocess_u8hwc(const uint8_t *frames, float *out, int B, int H, int W, const float mean[3],
                                    const float std[3], void *stream) {
    if (!frames || !out || B <= 0 || H <= 0 || W <= 0) return A3D_ERR_ARG;
    const size_t npix = (size_t)B * H * W;
    a3d_begin();
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, frames, out, npix,
                       mean[0], mean[1], mean[2], std[0], std[1], std[2]);
    return a3d_check_launch();
}

extern "C" int a3d_preprocess_f32chw(const float *images, float *out, int B, int H, int W, const float mean[3],
                                     const float std[3], void *stream) {
    if (!images || !out || B <= 0 || H <= 0 || W <= 0) return A3D_ERR_ARG;
    const size_t hw = (size_t)H * W;
    a3d_begin();
    hipLaunchKernelGGL(preprocess_chw_kernel, dim3(grid_for(hw * B)), dim3(256), 0, (hipStream_t)stream, images, out, B,
                       hw, mean[0], mean[1], mean[2], std[0], std[1], std[2]);
    return a3d_check_launch();
}

// ---- input front end: cv2.resize(frame, (Wd, Hd)) + channel flip + (x - mean)/std in one pass -----------
// Replaces the host-side `cv2.resize(im, (640, 480))` + `im[:, :, ::-1]` of the reference's frame loop
// (tools/inference.py:216-218) together with the uint8 -> float cast and normalisation (arti_vis.py:58, planercnn.py:188-196).
// cv2.resize on uint8 with the default INTER_LINEAR is OpenCV's fixed-point bilinear (imgproc/resize.cpp, as published):
//   fx = (float)((dx + 0.5) * (Ws / Wd) - 0.5);  sx = floor(fx);  fx -= sx;  clamped to the image with fx = 0 at the borders;
//   11-bit coefficients a = cvRound(w * 2048) (round half to even);  horizontal pass in int32, then
//   dst = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
//   an exact 2x2 decimation is switched to INTER_AREA: (s00 + s01 + s10 + s11 + 2) >> 2;  equal sizes copy.
// HBM-bound byte work: one thread per output pixel reads 4 source pixels x 3 bytes and writes one float4 (+ 3 bytes).
__global__ __launch_bounds__(256) void preprocess_resize_u8_kernel(const uint8_t *__restrict__ in, float *__restrict__ out,
                                                                   uint8_t *__restrict__ out_u8, int B, int Hs, int Ws, int Hd, int Wd,
                                                                   int swap_rb, float m0, float m1, float m2, float s0, float s1, float s2) {
    const size_t npix = (size_t)B * Hd * Wd;
    const double scale_x = (double)Ws / Wd, scale_y = (double)Hs / Hd;
    const bool same = Hs == Hd && Ws == Wd, area2 = Hs == 2 * Hd && Ws == 2 * Wd;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const int dx = (int)(i % Wd);
        const size_t r = i / Wd;
        const int dy = (int)(r % Hd), b = (int)(r / Hd);
        const uint8_t *src = in + (size_t)b * Hs * Ws * 3;
        int v[3];
        if (same) {
            const uint8_t *p = src + ((size_t)dy * Ws + dx) * 3;
            v[0] = p[0], v[1] = p[1], v[2] = p[2];
        } else if (area2) {
            const uint8_t *p = src + ((size_t)(2 * dy) * Ws + 2 * dx) * 3, *q = p + (size_t)Ws * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = (p[c] + p[3 + c] + q[c] + q[3 + c] + 2) >> 2;
        } else {
            float fx = (float)((dx + 0.5) * scale_x - 0.5), fy = (float)((dy + 0.5) * scale_y - 0.5);
            int sx = (int)floorf(fx), sy = (int)floorf(fy);
            fx -= (float)sx;
            fy -= (float)sy;
            if (sx < 0) fx = 0.f, sx = 0;
            if (sx >= Ws - 1) fx = 0.f, sx = Ws - 1;
            if (sy < 0) fy = 0.f, sy = 0;
            if (sy >= Hs - 1) fy = 0.f, sy = Hs - 1;
            const int a0 = __float2int_rn((1.f - fx) * 2048.f), a1 = __float2int_rn(fx * 2048.f);
            const int b0 = __float2int_rn((1.f - fy) * 2048.f), b1 = __float2int_rn(fy * 2048.f);
            const int sx1 = min(sx + 1, Ws - 1), sy1 = min(sy + 1, Hs - 1);
            const uint8_t *p00 = src + ((size_t)sy * Ws + sx) * 3, *p01 = src + ((size_t)sy * Ws + sx1) * 3;
            const uint8_t *p10 = src + ((size_t)sy1 * Ws + sx) * 3, *p11 = src + ((size_t)sy1 * Ws + sx1) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int r0 = p00[c] * a0 + p01[c] * a1, r1 = p10[c] * a0 + p11[c] * a1;
                v[c] = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
            }
        }
        if (out_u8) {  // the resized frame in the SOURCE channel order (what the reference keeps in `frames`, inference.py:217)
            uint8_t *o = out_u8 + i * 3;
            o[0] = (uint8_t)v[0], o[1] = (uint8_t)v[1], o[2] = (uint8_t)v[2];
        }
        const int c0 = swap_rb ? v[2] : v[0], c2 = swap_rb ? v[0] : v[2];
        f32x4 o4 = {((float)c0 - m0) / s0, ((float)v[1] - m1) / s1, ((float)c2 - m2) / s2, 0.f};
        *reinterpret_cast<f32x4 *>(out + i * 4) = o4;
    }
}

extern "C" int a3d_preprocess_resize_u8(const uint8_t *frames, float *out, uint8_t *out_u8, int B, int Hs, int Ws, int Hd, int Wd,
                                        int swap_rb, const float mean[3], const float std[3], void *stream) {
    if (!frames || !out || B <= 0 || Hs <= 0 || Ws <= 0 || Hd <= 0 || Wd <= 0) return A3D_ERR_ARG;
    const size_t npix = (size_t)B * Hd * Wd;
    a3d_begin();
    hipLaunchKernelGGL(preprocess_resize_u8_kernel, dim3(grid_for(npix)), dim3(256), 0, (hipStream_t)stream, frames, out, out_u8, B, Hs,
                       Ws, Hd, Wd, swap_rb, mean[0], mean[1], mean[2], std[0], std[1], std[2]);
    return a3d_check_launch();
}

// ---- max-pool 3x3 s2 p1 -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float *__restrict__ x, float *__restrict__ y, int B,
                                                           int H, int W, int C4, int Ho, int Wo) {
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t p = i / C4;
        const int ow = (int)(p % Wo);
        p /= Wo;
        const int oh = (int)(p % Ho);
        const int b = (int)(p / Ho);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int dy = 0; dy < 3; ++dy) {
            const int ih = oh * 2 - 1 + dy;
            if ((unsigned)ih >= (unsigned)H) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int iw = ow * 2 - 1 + dx;
                if ((unsigned)iw >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(x + (((size_t)b * H + ih) * W + iw) * (C4 * 4) + c * 4);
                for (int k = 0; k < 4; ++k) m[k] = (v[k] > m[k] || v[k] != v[k]) ? v[k] : m[k];  // (a NaN in the window wins, like torch's max_pool2d)
            }
        }
        *reinterpret_cast<f32x4 *>(y + i * 4) = m;
    }
}

extern "C" int a3d_maxpool3x3s2_nhwc(const float *x, float *y, int B, int H, int W, int C, void *stream) {
    if (!x || !y || (C & 3) || B <= 0) return A3D_ERR_ARG;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)B * Ho * Wo * (C / 4);
    a3d_begin();
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W,
                       C / 4, Ho, Wo);
    return a3d_check_launch();
}

// ---- kernel-1 stride-2 pool: y[b,oh,ow] = x[b,2oh,2ow] -------------------------------------------
__global__ __launch_bounds__(256) void subsample2_kernel(const float *__restrict__ x, float *__restrict__ y, int B,
                                                         int H, int W, int C4, int Ho, int Wo) {
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t p = i / C4;
        const int ow = (int)(p % Wo);
        p /= Wo;
        const int oh = (int)(p % Ho);
        const int b = (int)(p / Ho);
        *reinterpret_cast<f32x4 *>(y + i * 4) =
            *reinterpret_cast<const f32x4 *>(x + (((size_t)b * H + 2 * oh) * W + 2 * ow) * (C4 * 4) + c * 4);
    }
}

extern "C" int a3d_subsample2_nhwc(const float *x, float *y, int B, int H, int W, int C, void *stream) {
    if (!x || !y || (C & 3) || B <= 0) return A3D_ERR_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t total = (size_t)B * Ho * Wo * (C / 4);
    a3d_begin();
    hipLaunchKernelGGL(subsample2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C / 4,
                       Ho, Wo);
    return a3d_check_launch();
}

// ---- bilinear resize, align_corners=False (aten upsample_bilinear2d semantics) -------------------
__device__ __forceinline__ void src_index(int o, float scale, int in_size, int &i0, int &i1, float &l1) {
    float s = scale * ((float)o + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

template <int VEC>
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float *__restrict__ x, float *__restrict__ y, int B,
                                                              int H, int W, int C, int Ho, int Wo, float sh, float sw) {
    const int CV = C / VEC;
    const size_t total = (size_t)B * Ho * Wo * CV;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * VEC;
        size_t p = i / CV;
        const int ow = (int)(p % Wo);
        p /= Wo;
        const int oh = (int)(p % Ho);
        const int b = (int)(p / Ho);
        int y0, y1, x0, x1;
        float ly, lx;
        src_index(oh, sh, H, y0, y1, ly);
        src_index(ow, sw, W, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float *base = x + (size_t)b * H * W * C + c;
        for (int k = 0; k < VEC; ++k) {
            const float v00 = base[((size_t)y0 * W + x0) * C + k], v01 = base[((size_t)y0 * W + x1) * C + k];
            const float v10 = base[((size_t)y1 * W + x0) * C + k], v11 = base[((size_t)y1 * W + x1) * C + k];
            y[i * VEC + k] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
        }
    }
}

// One channel (the depth map, depth_head.py:82,88), Wo % 4 == 0: a thread owns four consecutive outputs of a row and stores them as one
// 16-byte vector (round 6: the one-output form wrote 78 MB in 4-byte stores, 0.15 ms at 64 frames).  Per output the arithmetic is the
// generic kernel's: the same bits.
__global__ __launch_bounds__(256) void resize_bilinear_c1x4_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int H, int W,
                                                                   int Ho, int Wo, float sh, float sw) {
    const int W4 = Wo >> 2;
    const size_t total = (size_t)B * Ho * W4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ow0 = (int)(i % W4) * 4;
        size_t p = i / W4;
        const int oh = (int)(p % Ho);
        const int b = (int)(p / Ho);
        int y0, y1;
        float ly;
        src_index(oh, sh, H, y0, y1, ly);
        const float hy = 1.f - ly;
        const float *r0 = x + ((size_t)b * H + y0) * W, *r1 = x + ((size_t)b * H + y1) * W;
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int x0, x1;
            float lx;
            src_index(ow0 + k, sw, W, x0, x1, lx);
            const float hx = 1.f - lx;
            o[k] = hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
        }
        *reinterpret_cast<f32x4 *>(y + ((size_t)b * Ho + oh) * Wo + ow0) = o;
    }
}

extern "C" int a3d_resize_bilinear_nhwc(const float *x, float *y, int B, int H, int W, int C, int Ho, int Wo,
                                        void *stream) {
    if (!x || !y || B <= 0 || C <= 0) return A3D_ERR_ARG;
    const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
    a3d_begin();
    if ((C & 3) == 0) {
        const size_t total = (size_t)B * Ho * Wo * (C / 4);
        hipLaunchKernelGGL(resize_bilinear_kernel<4>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, B,
                           H, W, C, Ho, Wo, sh, sw);
    } else if (C == 1 && (Wo & 3) == 0 && ((size_t)y & 15) == 0) {
        hipLaunchKernelGGL(resize_bilinear_c1x4_kernel, dim3(grid_for((size_t)B * Ho * (Wo >> 2))), dim3(256), 0, (hipStream_t)stream, x, y, B,
                           H, W, Ho, Wo, sh, sw);
    } else {
        const size_t total = (size_t)B * Ho * Wo * C;
        hipLaunchKernelGGL(resize_bilinear_kernel<1>, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, B,
                           H, W, C, Ho, Wo, sh, sw);
    }
    return a3d_check_launch();
}

// ---- 3x3 pad-1 conv to one channel -------------------------------------------------------------------
// HBM-bound (reads C floats per output).  Each lane owns a float4 of channels (C = 64 -> 16 lanes per pixel group); a
// group produces a run of 4 horizontally adjacent outputs from a 3 x 6 input window (18 float4 loads for 4 outputs
// instead of 36), the filter lives in registers, and the per-pixel sums are reduced across the group's lanes with
// xor-shuffles.
__global__ __launch_bounds__(256) void conv3x3_to1_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                          float bias, float *__restrict__ y, int B, int H, int W,
                                                          int C) {
    const int lanes_per_pix = C >> 2;           // 16 for C=64
    const int grp_per_blk = 256 / lanes_per_pix;
    const int sub = threadIdx.x % lanes_per_pix;
    const int Wr = (W + 3) >> 2;  // 4-pixel runs per row
    f32x4 k[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) k[t] = *reinterpret_cast<const f32x4 *>(w + t * C + sub * 4);
    // one block = grp_per_blk consecutive runs; the blocks of an XCD are a contiguous band of rows (a3d_xcd_remap), so the two
    // neighbour rows a block reads come from the same L2 that the neighbouring blocks filled (round-robin placement put the three
    // readers of a row on three XCDs: 3.07x the input fetched, 3.9 GB per launch)
    // round 3: a group owns a 4 x 4 block of outputs (RB rows x 4 columns) and slides down the RB + 2 input rows it needs, so a row of
    // the map is fetched 1.5 x instead of 3 x (the launch was moving 3.8 GB from L2 for 1.26 GB of input: 0.54 ms, 2.3 TB/s of HBM).
    // Every output still adds its taps in the order dy, input column, dx: bit-identical to the one-row form.
    constexpr int RB = 4;
    {
        const int Hb = (H + RB - 1) / RB;
        const size_t nruns = (size_t)B * Hb * Wr;
        const size_t r = (size_t)a3d_xcd_remap(blockIdx.x, gridDim.x) * grp_per_blk + threadIdx.x / lanes_per_pix;
        if (r >= nruns) return;
        const int ow0 = (int)(r % Wr) * 4;
        const size_t t = r / Wr;
        const int oh0 = (int)(t % Hb) * RB;
        const int b = (int)(t / Hb);
        float acc[RB][4];
#pragma unroll
        for (int q = 0; q < RB; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[q][j] = 0.f;
#pragma unroll
        for (int ri = 0; ri < RB + 2; ++ri) {  // input row oh0 - 1 + ri feeds output rows q = ri - dy, dy = 0..2
            const int ih = oh0 - 1 + ri;
            if ((unsigned)ih >= (unsigned)H) continue;
            const float *row = x + ((size_t)b * H + ih) * W * C + sub * 4;
            f32x4 v[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const int iw = ow0 - 1 + c;
                v[c] = (unsigned)iw < (unsigned)W ? *reinterpret_cast<const f32x4 *>(row + (size_t)iw * C) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int dy = 2; dy >= 0; --dy) {  // (output rows in increasing order; the order WITHIN an output is set by ri alone)
                const int q = ri - dy;
                if (q < 0 || q >= RB) continue;
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    if ((unsigned)(ow0 - 1 + c) >= (unsigned)W) continue;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int j = c - dx;
                        if (j < 0 || j > 3) continue;
                        const f32x4 kk = k[dy * 3 + dx];
                        acc[q][j] += v[c][0] * kk[0] + v[c][1] * kk[1] + v[c][2] * kk[2] + v[c][3] * kk[3];
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
            if (oh0 + q >= H) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = acc[q][j];
                for (int off = lanes_per_pix >> 1; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
                if (sub == 0 && ow0 + j < W) y[((size_t)b * H + oh0 + q) * W + ow0 + j] = a + bias;
            }
        }
    }
}

extern "C" int a3d_conv3x3_to1_nhwc(const float *x, const float *w, float bias, float *y, int B, int H, int W, int C,
                                    void *stream) {
    if (!x || !w || !y || B <= 0) return A3D_ERR_ARG;
    const int lpp = C >> 2;
    if ((C & 3) || lpp > 64 || (lpp & (lpp - 1))) return A3D_ERR_UNSUPPORTED;
    const size_t nruns = (size_t)B * ((H + 3) / 4) * ((W + 3) / 4);  // 4 x 4 output blocks
    const int gpb = 256 / lpp;
    const size_t blocks = (nruns + gpb - 1) / gpb;
    if (blocks >= ((size_t)1 << 31)) return A3D_ERR_UNSUPPORTED;
    a3d_begin();
    hipLaunchKernelGGL(conv3x3_to1_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, B, H, W,
                       C);
    return a3d_check_launch();
}


// ---- second half of a 3x3 pad-1 convolution to one channel whose per-pixel tap products a phase-5 conv launch stored (a3d_conv_desc.dot_y) ----
// y[b][oh][ow] = bias + sum_t g[b][t][oh + t / 3 - 1][ow + t % 3 - 1]: nine shifted planes, taps in order, zero outside the map.
__global__ __launch_bounds__(256) void tapsum9_kernel(const float *__restrict__ g, float bias, float *__restrict__ y, int B, int H, int W) {
    const size_t total = (size_t)B * H * W, plane = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ow = (int)(i % W);
        const size_t r = i / W;
        const int oh = (int)(r % H);
        const size_t b = r / H;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ih = oh + t / 3 - 1, iw = ow + t % 3 - 1;
            if ((unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W) s += g[(b * 9 + t) * plane + (size_t)ih * W + iw];
        }
        y[i] = s + bias;
    }
}
// W % 4 == 0 (the detector's 240 x 320 depth map): a thread owns FOUR consecutive outputs of a row -- per tap plane one 16-byte load of the
// columns above them plus the one neighbour its shift needs, one 16-byte store -- instead of nine 4-byte loads and a 4-byte store per output
// (round 6: 142 us for 197 MB at 64 frames was 1.4 TB/s).  Every output adds its taps in the order 0..8 as above: the same bits.
__global__ __launch_bounds__(256) void tapsum9x4_kernel(const float *__restrict__ g, float bias, float *__restrict__ y, int B, int H, int W) {
    const int W4 = W >> 2;
    const size_t total = (size_t)B * H * W4, plane = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ow = (int)(i % W4) * 4;
        const size_t r = i / W4;
        const int oh = (int)(r % H);
        const size_t b = r / H;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ih = oh + t / 3 - 1, dx = t % 3 - 1;
            if ((unsigned)ih >= (unsigned)H) continue;
            const float *row = g + (b * 9 + t) * plane + (size_t)ih * W;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(row + ow);
            if (dx == 0) {
                s += v;
            } else if (dx < 0) {
                if (ow > 0) s[0] += row[ow - 1];
                s[1] += v[0];
                s[2] += v[1];
                s[3] += v[2];
            } else {
                s[0] += v[1];
                s[1] += v[2];
                s[2] += v[3];
                if (ow + 4 < W) s[3] += row[ow + 4];
            }
        }
        *reinterpret_cast<f32x4 *>(y + (b * H + oh) * (size_t)W + ow) = s + bias;
    }
}
extern "C" int a3d_tapsum9(const float *g, float bias, float *y, int B, int H, int W, void *stream) {
    if (!g || !y || B <= 0 || H <= 0 || W <= 0) return A3D_ERR_ARG;
    a3d_begin();
    if ((W & 3) == 0 && (((size_t)g | (size_t)y) & 15) == 0)
        hipLaunchKernelGGL(tapsum9x4_kernel, dim3(grid_for((size_t)B * H * (W >> 2))), dim3(256), 0, (hipStream_t)stream, g, bias, y, B, H, W);
    else
        hipLaunchKernelGGL(tapsum9_kernel, dim3(grid_for((size_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, g, bias, y, B, H, W);
    return a3d_check_launch();
}

// ---- per-image maxima of a tensor no kernel of this library produced (a3d_conv_desc.in_amax) --------------------------------
// Any row count (the grid strides over rows), any row length and alignment (scalar head / tail around the 16-byte body).
// Non-finite values do not count (conv_common.h a3d_finite_mag): the scale of an image comes from its finite values.
__global__ __launch_bounds__(256) void absmax_rows_kernel(const float *__restrict__ x, float *__restrict__ out, int B, size_t n) {
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        const float *row = x + (size_t)b * n;
        const size_t head = min(n, (size_t)((4 - ((reinterpret_cast<size_t>(row) >> 2) & 3)) & 3));  // floats before the first 16-byte boundary
        const size_t n4 = (n - head) >> 2;
        float m = 0.f;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(row + head + i * 4);
            m = fmaxf(m, a3d_absmax4(v));
        }
        if (blockIdx.x == 0) {
            for (size_t i = threadIdx.x; i < head; i += blockDim.x) m = fmaxf(m, a3d_finite_mag(row[i]));
            for (size_t i = head + (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, a3d_finite_mag(row[i]));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
        if ((threadIdx.x & 63) == 0 && m > out[b]) atomicMax(reinterpret_cast<int *>(out + b), __float_as_int(m));
    }
}

extern "C" int a3d_absmax_rows(const float *x, float *out, int B, size_t n, void *stream) {
    if (!x || !out || B <= 0 || n == 0 || ((size_t)x & 3)) return A3D_ERR_ARG;
    a3d_begin();
    size_t bx = (n / 4 + 255) / 256;
    if (bx > 64) bx = 64;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(absmax_rows_kernel, dim3((unsigned)bx, (unsigned)(B < 65535 ? B : 65535)), dim3(256), 0, (hipStream_t)stream, x, out, B, n);
    return a3d_check_launch();
}

// ---- activation pre-split for a3d_conv_desc.x_h2 (include/a3d.h) ---------------------------------------------------------------
// x [B][n] fp32 -> dst [B][n/16][h | l][16] fp16 of x * s(b): the split the fp16x2 loaders perform on the fly (conv_bf16x3_wide.hip
// wx_split2h), done once per tensor so that the consuming kernel takes both operands by LDS-DMA.  A thread owns 8 consecutive values
// = one 16-byte half of a chunk's h row and of its l row (two 16-byte loads, two 16-byte stores); same bytes in and out.
#include "conv_common.h"
typedef _Float16 ps_h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ps_h16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void presplit_f16x2_kernel(const float *__restrict__ x, unsigned char *__restrict__ dst, const float *__restrict__ amax,
                                                             const float *__restrict__ amax2, int B, size_t n8) {
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
        float am = amax[b];
        if (amax2) am = fmaxf(am, amax2[b]);
        const float sc = a3d_pow2_scale(am);
        const float *row = x + (size_t)b * n8 * 8;
        unsigned char *drow = dst + (size_t)b * n8 * 32;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
            const f32x4 x0 = *reinterpret_cast<const f32x4 *>(row + i * 8) * sc, x1 = *reinterpret_cast<const f32x4 *>(row + i * 8 + 4) * sc;
            const ps_h16x4 h0 = __builtin_convertvector(x0, ps_h16x4), h1 = __builtin_convertvector(x1, ps_h16x4);
            const ps_h16x4 l0 = __builtin_convertvector(x0 - __builtin_convertvector(h0, f32x4), ps_h16x4);
            const ps_h16x4 l1 = __builtin_convertvector(x1 - __builtin_convertvector(h1, f32x4), ps_h16x4);
            unsigned char *d = drow + (i >> 1) * 64 + (i & 1) * 16;
            *reinterpret_cast<ps_h16x8 *>(d) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            *reinterpret_cast<ps_h16x8 *>(d + 32) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }
}

extern "C" int a3d_presplit_f16x2(const float *x, void *dst, const float *amax, const float *amax2, int B, size_t n, void *stream) {
    if (!x || !dst || !amax || B <= 0 || n == 0 || (n & 15) || ((size_t)x & 15) || ((size_t)dst & 15)) return A3D_ERR_ARG;
    a3d_begin();
    size_t bx = (n / 8 + 255) / 256;
    const size_t cap = B >= 64 ? 64 : 2048 / (size_t)B;
    if (bx > cap) bx = cap;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(presplit_f16x2_kernel, dim3((unsigned)bx, (unsigned)(B < 65535 ? B : 65535)), dim3(256), 0, (hipStream_t)stream, x,
                       (unsigned char *)dst, amax, amax2, B, n / 8);
    return a3d_check_launch();
}
