// Tails of the per-ROI heads and the post-processing of the detections:
//   linear_small   skinny output layers (1024 -> 3 / 2+1 / 2, 256 -> 1) with fused L2-normalise or
//                  sigmoid: plane_head.py:80-82, axis_head.py:106-107,120, mask predictor + sigmoid
//   paste_lsq      detector_postprocess score/empty filter (postprocessing.py:47-60), mask paste
//                  (mask_ops.py:41-60,128-129) and the per-ROI plane-offset least squares
//                  (arti_vis.py:90-99,125-149) fused: the pasted mask is consumed in registers.
// Built with -ffp-contract=off so the paste coordinates round like the reference's separate ops.
#include "a3d_common.h"
#include "../../include/a3d.h"

#define SMALL_N_MAX 8

// one wave per row; lanes stride K with float4; N <= 8 accumulators reduced by xor-shuffles.  Round 6: a wave walks FOUR of its rows at a
// time (RB) -- four independent row loads in flight, the filter quads loaded once for the four -- because a row of the mask predictor is
// 1 KiB (one load per wave) and the one-row loop ran at the latency of that load (0.9 TB/s on 221 MB).  Per row the operations and their
// order are unchanged: the same bits.
template <int RB>
__device__ __forceinline__ void linear_small_rows(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                  float *__restrict__ y, const int (&rows)[RB], const int nrows, const int lane, int K, int N,
                                                  int norm_n, int act) {
    float acc[RB][SMALL_N_MAX];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int n = 0; n < SMALL_N_MAX; ++n) acc[r][n] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        f32x4 xv[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r)
            if (r < nrows) xv[r] = *reinterpret_cast<const f32x4 *>(x + (size_t)rows[r] * K + k);
#pragma unroll
        for (int n = 0; n < SMALL_N_MAX; ++n) {
            if (n < N) {
                const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + (size_t)n * K + k);
#pragma unroll
                for (int r = 0; r < RB; ++r)
                    if (r < nrows) acc[r][n] += xv[r][0] * wv[0] + xv[r][1] * wv[1] + xv[r][2] * wv[2] + xv[r][3] * wv[3];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        if (r >= nrows) break;
#pragma unroll
        for (int n = 0; n < SMALL_N_MAX; ++n)
            for (int off = 32; off > 0; off >>= 1) acc[r][n] += __shfl_xor(acc[r][n], off, 64);
        if (lane == 0) {
            float v[SMALL_N_MAX];
            for (int n = 0; n < N; ++n) v[n] = acc[r][n] + (bias ? bias[n] : 0.f);
            if (norm_n > 0) {  // F.normalize(p=2, eps=1e-12)
                float ss = 0.f;
                for (int n = 0; n < norm_n; ++n) ss += v[n] * v[n];
                const float den = fmaxf(sqrtf(ss), 1e-12f);
                for (int n = 0; n < norm_n; ++n) v[n] = v[n] / den;
            }
            if (act == 3) {
                for (int n = 0; n < N; ++n) v[n] = 1.0f / (1.0f + expf(-v[n]));
            }
            for (int n = 0; n < N; ++n) y[(size_t)rows[r] * N + n] = v[n];
        }
    }
}

__global__ __launch_bounds__(256) void linear_small_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                           const float *__restrict__ bias, float *__restrict__ y, int M,
                                                           const int *__restrict__ m_dev, int K, int N, int norm_n,
                                                           int act) {
    if (m_dev) M = min(M, *m_dev);
    const int lane = threadIdx.x & 63;
    constexpr int RB = 4;
    const int stride = gridDim.x * 4;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += stride * RB) {
        int rows[RB];
        int nrows = 0;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            rows[r] = row + r * stride;
            nrows += rows[r] < M ? 1 : 0;  // (rows ascend: the live ones are a prefix)
        }
        linear_small_rows<RB>(x, w, bias, y, rows, nrows, lane, K, N, norm_n, act);
    }
}

extern "C" int a3d_linear_small(const float *x, const float *w, const float *bias, float *y, int M, const int *m_dev,
                                int K, int N, int norm_n, int sigmoid, void *stream) {
    if (!x || !w || !y || M < 0 || (K & 3) || N < 1 || N > SMALL_N_MAX || norm_n > N) return A3D_ERR_ARG;
    if (M == 0) return A3D_OK;
    int blocks = (M + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    a3d_begin();
    hipLaunchKernelGGL(linear_small_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, M, m_dev, K,
                       N, norm_n, sigmoid ? 3 : 0);
    return a3d_check_launch();
}

// ------------------------------------------------------------------------------------------------
// paste + threshold + plane-offset LSQ.  One workgroup per detection slot (b, r).
// ------------------------------------------------------------------------------------------------
struct PasteArgs {
    const float *boxes;     // [B, R, 4]
    const float *scores;    // [B, R]
    const int *count;       // [B]
    const int *row_offset;  // [B] compact row of (b, 0) in the per-ROI head outputs
    const float *mask_prob; // [rows, MS, MS]
    const float *normals;   // [rows, 3] unit normals (pred_plane), or NULL
    const float *depth;     // [B, H, W] or NULL
    int B, R, MS, H, W;
    float post_score_thresh, mask_thresh;
    float fx, cx, cy;       // rays: ((x - cx)/fx, (y - cy)/fx, 1), arti_vis.py:101-123
    int clip;
    unsigned char *masks;   // [B, R, H, W] or NULL
    float *planes;          // [B, R, 3]  normal * offset in the reference's output axes
    int *area;              // [B, R] mask pixel count
    int *keep;              // [B, R] 1 if the detection survives detector_postprocess
    float *out_boxes;       // [B, R, 4] clipped boxes
};

// 1024 threads per detection (round 6; 256 before): a frame's handful of detections is a handful of workgroups, each walking the whole image
// (64 frames: 149 -> 86 us; the one-frame pass 3.37 -> 3.23 ms).  The same workgroup shape for every batch, so a detection's sums -- per-thread
// partial sums in float64, lanes by xor-shuffle, then the sixteen waves in order -- do not depend on the batch it arrives in.
constexpr int PASTE_THREADS = 1024, PASTE_WAVES = PASTE_THREADS / 64;
__global__ __launch_bounds__(PASTE_THREADS) void paste_lsq_kernel(const PasteArgs a) {
    __shared__ int redc[PASTE_WAVES];
    const int slot = blockIdx.x;
    const int b = slot / a.R, r = slot - b * a.R;
    const size_t npix = (size_t)a.H * a.W;
    unsigned char *mout = a.masks ? a.masks + (size_t)slot * npix : nullptr;
    const int cnt = min(a.count[b], a.R);
    bool live = r < cnt;
    float x0 = 0.f, y0 = 0.f, x1 = 0.f, y1 = 0.f;
    if (live) {
        const float *bx = a.boxes + (size_t)slot * 4;
        // output size == input size on this path (planercnn.py:213-214): scale 1, then clip
        x0 = bx[0], y0 = bx[1], x1 = bx[2], y1 = bx[3];
        if (a.clip) {
            x0 = fminf(fmaxf(x0, 0.f), (float)a.W);
            y0 = fminf(fmaxf(y0, 0.f), (float)a.H);
            x1 = fminf(fmaxf(x1, 0.f), (float)a.W);
            y1 = fminf(fmaxf(y1, 0.f), (float)a.H);
            live = a.scores[slot] >= a.post_score_thresh && (x1 - x0) > 0.f && (y1 - y0) > 0.f;
        }
    }
    if (threadIdx.x == 0) {
        a.keep[slot] = live ? 1 : 0;
        a.out_boxes[slot * 4 + 0] = x0;
        a.out_boxes[slot * 4 + 1] = y0;
        a.out_boxes[slot * 4 + 2] = x1;
        a.out_boxes[slot * 4 + 3] = y1;
    }
    if (!live) {
        if (mout && (npix & 15) == 0 && (((size_t)slot * npix) & 15) == 0)
            for (size_t i = threadIdx.x; i < npix / 16; i += blockDim.x) reinterpret_cast<uint4 *>(mout)[i] = uint4{0, 0, 0, 0};
        else if (mout)
            for (size_t i = threadIdx.x; i < npix; i += blockDim.x) mout[i] = 0;
        if (threadIdx.x == 0) {
            a.area[slot] = 0;
            a.planes[slot * 3 + 0] = a.planes[slot * 3 + 1] = a.planes[slot * 3 + 2] = 0.f;
        }
        return;
    }
    const int row = a.row_offset[b] + r;
    const float *mp = a.mask_prob + (size_t)row * a.MS * a.MS;
    // scannet -> suncg axis swap of the predicted normal (arti_vis.py:130-131), renormalise (:140-141)
    float n0 = 0.f, n1 = 0.f, n2 = 0.f, p0 = 0.f, p1 = 0.f, p2 = 0.f;
    const bool lsq = a.normals && a.depth;
    if (lsq) {
        const float *pl = a.normals + (size_t)row * 3;
        p0 = pl[0];
        p1 = -pl[2];
        p2 = pl[1];
        const float nrm = sqrtf(p0 * p0 + p1 * p1 + p2 * p2);
        const float den = fmaxf(nrm, 1e-8f);
        n0 = p0 / den;
        n1 = p1 / den;
        n2 = p2 / den;
    }
    const float *dep = a.depth ? a.depth + (size_t)b * npix : nullptr;
    const float MSf = (float)a.MS;
    const float dxb = x1 - x0, dyb = y1 - y0;
    double sum = 0.0;
    int cntpix = 0;
    // 16 pixels per thread-iteration along x so the uint8 mask is stored as one 16-byte vector
    const int groups_per_row = (a.W + 15) / 16;
    const bool vec_store = (a.W & 15) == 0;  // 16-byte mask stores need 16-pixel-aligned rows; other widths store bytes
    for (int gidx = threadIdx.x; gidx < a.H * groups_per_row; gidx += blockDim.x) {
        const int py = gidx / groups_per_row, gx = (gidx - py * groups_per_row) * 16;
        float ty = ((float)py + 0.5f) - y0;
        ty = ty / dyb;
        ty = ty * 2.f;
        ty = ty - 1.f;
        // F.grid_sample (ATen CPU kernel, the oracle's reference) un-normalises with ONE fused multiply-add and sums the
        // four taps as an FMA chain nw -> ne -> sw -> se; reproducing exactly that rounding keeps the thresholded mask
        // bit-identical (checked against 2e5 random samples: 0 mismatches; the unfused form gives 19 % 1-ulp diffs).
        const float iy = __fmaf_rn(ty + 1.f, MSf, -1.f) / 2.f;
        const float fy = floorf(iy);
        const int iyn = (int)fy;
        const float n = iy - fy, s = 1.f - n;
        const bool rowhit = iy > -1.f && iy < MSf;
        unsigned char bytes[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            unsigned char bit = 0;
            if (rowhit && gx + j < a.W) {
                const int px = gx + j;
                float tx = ((float)px + 0.5f) - x0;
                tx = tx / dxb;
                tx = tx * 2.f;
                tx = tx - 1.f;
                const float ix = __fmaf_rn(tx + 1.f, MSf, -1.f) / 2.f;
                if (ix > -1.f && ix < MSf) {
                    const float fx = floorf(ix);
                    const int ixw = (int)fx;
                    const float w = ix - fx, e = 1.f - w;
                    const float nw = e * s, ne = w * s, sw = e * n, se = w * n;
                    const bool yn_ok = (unsigned)iyn < (unsigned)a.MS, ys_ok = (unsigned)(iyn + 1) < (unsigned)a.MS;
                    const bool xw_ok = (unsigned)ixw < (unsigned)a.MS, xe_ok = (unsigned)(ixw + 1) < (unsigned)a.MS;
                    const float vnw = (yn_ok && xw_ok) ? mp[iyn * a.MS + ixw] : 0.f;
                    const float vne = (yn_ok && xe_ok) ? mp[iyn * a.MS + ixw + 1] : 0.f;
                    const float vsw = (ys_ok && xw_ok) ? mp[(iyn + 1) * a.MS + ixw] : 0.f;
                    const float vse = (ys_ok && xe_ok) ? mp[(iyn + 1) * a.MS + ixw + 1] : 0.f;
                    const float val = __fmaf_rn(vse, se, __fmaf_rn(vsw, sw, __fmaf_rn(vne, ne, vnw * nw)));
                    if (val >= a.mask_thresh) {
                        bit = 1;
                        ++cntpix;
                        if (lsq) {
                            const float dv = dep[(size_t)py * a.W + px];
                            // rays are built in float64 and cast to fp32 (arti_vis.py:50,101-123)
                            const float X = (float)(((double)px - (double)a.cx) / (double)a.fx) * dv;
                            const float Y = (float)(((double)py - (double)a.cy) / (double)a.fx) * dv;
                            sum += (double)(n0 * X + n1 * Y + n2 * dv);
                        }
                    }
                }
            }
            bytes[j] = bit;
        }
        if (mout && !vec_store) {
            for (int j = 0; j < 16 && gx + j < a.W; ++j) mout[(size_t)py * a.W + gx + j] = bytes[j];
        } else if (mout) {
            uint4 pk;
            pk.x = bytes[0] | (bytes[1] << 8) | (bytes[2] << 16) | (bytes[3] << 24);
            pk.y = bytes[4] | (bytes[5] << 8) | (bytes[6] << 16) | (bytes[7] << 24);
            pk.z = bytes[8] | (bytes[9] << 8) | (bytes[10] << 16) | (bytes[11] << 24);
            pk.w = bytes[12] | (bytes[13] << 8) | (bytes[14] << 16) | (bytes[15] << 24);
            *reinterpret_cast<uint4 *>(mout + (size_t)py * a.W + gx) = pk;
        }
    }
    // workgroup reduction (fixed order: lanes by xor-shuffle, then the waves in order)
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off, 64);
        cntpix += __shfl_xor(cntpix, off, 64);
    }
    const int wave = threadIdx.x >> 6;
    __shared__ double sred[PASTE_WAVES];
    if ((threadIdx.x & 63) == 0) {
        sred[wave] = sum;
        redc[wave] = cntpix;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        int c = 0;
        for (int w = 0; w < PASTE_WAVES; ++w) {
            tot += sred[w];
            c += redc[w];
        }
        a.area[slot] = c;
        float o0, o1, o2;
        if (!lsq) {
            o0 = o1 = o2 = 0.f;
        } else if (c == 0) {  // empty mask keeps the (swapped) plane, arti_vis.py:136-138
            o0 = p0;
            o1 = p1;
            o2 = p2;
        } else {
            const float offn = (float)(tot / (double)c);
            o0 = n0 * offn;
            o1 = n1 * offn;
            o2 = n2 * offn;
        }
        // swap back (x, y, z) -> (x, z, -y), arti_vis.py:146-147
        a.planes[slot * 3 + 0] = o0;
        a.planes[slot * 3 + 1] = o2;
        a.planes[slot * 3 + 2] = -o1;
    }
}

extern "C" int a3d_paste_lsq(const a3d_paste_desc *d, void *stream) {
    if (!d || !d->boxes || !d->scores || !d->count || !d->row_offset || !d->mask_prob || !d->planes || !d->area ||
        !d->keep || !d->out_boxes)
        return A3D_ERR_ARG;
    if (d->B <= 0 || d->R <= 0 || d->W <= 0 || d->H <= 0 || d->MS <= 0) return A3D_ERR_ARG;
    PasteArgs a;
    a.boxes = d->boxes;
    a.scores = d->scores;
    a.count = d->count;
    a.row_offset = d->row_offset;
    a.mask_prob = d->mask_prob;
    a.normals = d->normals;
    a.depth = d->depth;
    a.B = d->B;
    a.R = d->R;
    a.MS = d->MS;
    a.H = d->H;
    a.W = d->W;
    a.post_score_thresh = d->post_score_thresh;
    a.mask_thresh = d->mask_thresh;
    a.fx = d->focal;
    a.cx = d->cx;
    a.cy = d->cy;
    a.clip = d->clip_boxes;
    a.masks = d->masks;
    a.planes = d->planes;
    a.area = d->area;
    a.keep = d->keep;
    a.out_boxes = d->out_boxes;
    a3d_begin();
    hipLaunchKernelGGL(paste_lsq_kernel, dim3(d->B * d->R), dim3(PASTE_THREADS), 0, (hipStream_t)stream, a);
    return a3d_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Stand-alone plane-offset least squares from already-pasted dense masks (one image):
// PlaneRCNN_Branch.override_depth, arti_vis.py:125-149.  One workgroup per detection.
// ------------------------------------------------------------------------------------------------
// MT = float (0 / 1 masks as create_instances holds them) or unsigned char (the pasted bool masks as they are: no float copy of
// 1.2 MB per mask).  Same pixel -> thread assignment, same accumulation order and same ray values as before for both (the rays
// ((double)px - cx) / fx, cast to float, are tabulated per column / row instead of being divided out per pixel).
template <typename MT>
__global__ __launch_bounds__(256) void plane_offset_dense_kernel(const float *__restrict__ depth,
                                                                 const MT *__restrict__ masks,
                                                                 const float *__restrict__ normals,
                                                                 float *__restrict__ out, int H, int W, float fx,
                                                                 float cx, float cy) {
    __shared__ double sred[4];
    __shared__ int redc[4];
    __shared__ float rayx[2048], rayy[2048];
    const int d = blockIdx.x;
    const size_t npix = (size_t)H * W;
    const MT *m = masks + (size_t)d * npix;
    const float *pl = normals + (size_t)d * 3;
    const float p0 = pl[0], p1 = -pl[2], p2 = pl[1];
    const float den = fmaxf(sqrtf(p0 * p0 + p1 * p1 + p2 * p2), 1e-8f);
    const float n0 = p0 / den, n1 = p1 / den, n2 = p2 / den;
    for (int i = threadIdx.x; i < W; i += blockDim.x) rayx[i] = (float)(((double)i - (double)cx) / (double)fx);
    for (int i = threadIdx.x; i < H; i += blockDim.x) rayy[i] = (float)(((double)i - (double)cy) / (double)fx);
    __syncthreads();
    double sum = 0.0;
    int c = 0;
    int py = threadIdx.x / W, px = threadIdx.x - py * W;  // pixel i = threadIdx.x + 256 k, advanced without divisions
    const int dy = (int)blockDim.x / W, dx = (int)blockDim.x - dy * W;
    for (size_t i = threadIdx.x; i < npix; i += blockDim.x) {
        if (m[i] != (MT)0) {
            const float dv = depth[i];
            const float X = rayx[px] * dv;
            const float Y = rayy[py] * dv;
            sum += (double)(n0 * X + n1 * Y + n2 * dv);
            ++c;
        }
        px += dx;
        py += dy;
        if (px >= W) {
            px -= W;
            ++py;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off, 64);
        c += __shfl_xor(c, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        sred[threadIdx.x >> 6] = sum;
        redc[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double tot = sred[0] + sred[1] + sred[2] + sred[3];
        const int cc = redc[0] + redc[1] + redc[2] + redc[3];
        float o0 = p0, o1 = p1, o2 = p2;
        if (cc > 0) {
            const float offn = (float)(tot / (double)cc);
            o0 = n0 * offn;
            o1 = n1 * offn;
            o2 = n2 * offn;
        }
        out[d * 3 + 0] = o0;
        out[d * 3 + 1] = o2;
        out[d * 3 + 2] = -o1;
    }
}

extern "C" int a3d_plane_offset_dense(const float *depth, const float *masks, const float *normals, float *out, int D,
                                      int H, int W, float focal, float cx, float cy, void *stream) {
    if (!depth || !masks || !normals || !out || D < 0 || H > 2048 || W > 2048) return A3D_ERR_ARG;
    if (D == 0) return A3D_OK;
    a3d_begin();
    hipLaunchKernelGGL(plane_offset_dense_kernel<float>, dim3(D), dim3(256), 0, (hipStream_t)stream, depth, masks, normals, out,
                       H, W, focal, cx, cy);
    return a3d_check_launch();
}

// The pasted bool masks as they are (one byte per pixel), 1024 threads per detection, four pixels per thread and step (one 32-bit
// mask load; the depth quad only where a byte is set).  Per-thread double sums, then a wave / workgroup tree.
__global__ __launch_bounds__(1024) void plane_offset_dense_u8_kernel(const float *__restrict__ depth, const unsigned char *__restrict__ masks,
                                                                     const float *__restrict__ normals, float *__restrict__ out, int H, int W,
                                                                     float fx, float cx, float cy) {
    __shared__ double sred[16];
    __shared__ int redc[16];
    __shared__ float rayx[2048], rayy[2048];
    const int d = blockIdx.x;
    const int npix = H * W;
    const unsigned char *m = masks + (size_t)d * npix;
    const float *pl = normals + (size_t)d * 3;
    const float p0 = pl[0], p1 = -pl[2], p2 = pl[1];
    const float den = fmaxf(sqrtf(p0 * p0 + p1 * p1 + p2 * p2), 1e-8f);
    const float n0 = p0 / den, n1 = p1 / den, n2 = p2 / den;
    for (int i = threadIdx.x; i < W; i += 1024) rayx[i] = (float)(((double)i - (double)cx) / (double)fx);
    for (int i = threadIdx.x; i < H; i += 1024) rayy[i] = (float)(((double)i - (double)cy) / (double)fx);
    __syncthreads();
    double sum = 0.0;
    int c = 0;
    const bool quads = ((npix | W) & 3) == 0 && (reinterpret_cast<size_t>(m) & 3) == 0;  // rows hold whole quads (uniform)
    if (quads) {
        for (int q = threadIdx.x; q < (npix >> 2); q += 1024) {
            const unsigned w = *reinterpret_cast<const unsigned *>(m + 4 * (size_t)q);
            if (!w) continue;
            const int i = 4 * q, py = i / W, px = i - py * W;
            const float ry = rayy[py];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if ((w >> (8 * k)) & 0xFFu) {
                    const float dv = depth[i + k];
                    sum += (double)(n0 * (rayx[px + k] * dv) + n1 * (ry * dv) + n2 * dv);
                    ++c;
                }
            }
        }
    } else {
        for (int i = threadIdx.x; i < npix; i += 1024) {
            if (m[i]) {
                const int py = i / W, px = i - py * W;
                const float dv = depth[i];
                sum += (double)(n0 * (rayx[px] * dv) + n1 * (rayy[py] * dv) + n2 * dv);
                ++c;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_xor(sum, off, 64);
        c += __shfl_xor(c, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        sred[threadIdx.x >> 6] = sum;
        redc[threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        int cc = 0;
        for (int w = 0; w < 16; ++w) {
            tot += sred[w];
            cc += redc[w];
        }
        float o0 = p0, o1 = p1, o2 = p2;
        if (cc > 0) {
            const float offn = (float)(tot / (double)cc);
            o0 = n0 * offn;
            o1 = n1 * offn;
            o2 = n2 * offn;
        }
        out[d * 3 + 0] = o0;
        out[d * 3 + 1] = o2;  // (back to the reference's output axes, arti_vis.py:145-147: as plane_offset_dense_kernel)
        out[d * 3 + 2] = -o1;
    }
}

extern "C" int a3d_plane_offset_dense_u8(const float *depth, const unsigned char *masks, const float *normals, float *out, int D,
                                         int H, int W, float focal, float cx, float cy, void *stream) {
    if (!depth || !masks || !normals || !out || D < 0 || H > 2048 || W > 2048) return A3D_ERR_ARG;
    if (D == 0) return A3D_OK;
    a3d_begin();
    hipLaunchKernelGGL(plane_offset_dense_u8_kernel, dim3(D), dim3(1024), 0, (hipStream_t)stream, depth, masks, normals, out, H, W, focal, cx, cy);
    return a3d_check_launch();
}
