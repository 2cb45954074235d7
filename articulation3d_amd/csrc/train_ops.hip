// Small kernels of the training step: everything between the MFMA contractions.
// Reference call sites: the training branch PlaneRCNN.forward (pkg/modeling/meta_arch/planercnn.py:83-123), the box
// branch of PlaneRCNNROIHeads (pkg/modeling/roi_heads/roi_heads.py:93-117,190-204) and the detectron2 trainer behind
// tools/train_net.py:84-117.  The arithmetic they replace is detectron2's (Matcher, Box2BoxTransform.get_deltas,
// RPN.losses, FastRCNNOutputLayers.losses, torch.optim.SGD) -- restated in oracle/train_oracle.py.
//
// All of these are HBM-bound element / row kernels; built with -ffp-contract=off so IoU and delta arithmetic round like
// the reference's separate mul/add/div (the matcher's labels are compared bit-exactly).
#include "a3d_common.h"
#include "../../include/a3d.h"

// ------------------------------------------------------------------------------------------------
// weights: [Cout][KH][KW][Cin] -> data-gradient filter [Cin][KH][KW][Cout], taps flipped, rows scaled by the folded-BN
// scale of the forward conv.  dgrad of a stride-1 conv is then the forward kernel on dy with this filter.
// ------------------------------------------------------------------------------------------------
__global__ void weight_transpose_kernel(const float *__restrict__ w, const float *__restrict__ scale, float *__restrict__ wt,
                                        int Cout, int T, int Cin, int KH, int KW) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        float v = 0.f;
        if (co < Cout && ci < Cin) {
            v = w[((size_t)co * T + tap) * Cin + ci];
            if (scale) v *= scale[co];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    const int kh = tap / KW, kw = tap - kh * KW;
    const int ftap = (KH - 1 - kh) * KW + (KW - 1 - kw);
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < Cin && co < Cout) wt[((size_t)ci * T + ftap) * Cout + co] = tile[tx][r];
    }
}

extern "C" int a3d_weight_transpose(const float *w, const float *scale, float *wt, int Cout, int KH, int KW, int Cin, void *stream) {
    if (!w || !wt || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0) return A3D_ERR_ARG;
    a3d_begin();
    hipLaunchKernelGGL(weight_transpose_kernel, dim3((Cin + 31) / 32, (Cout + 31) / 32, KH * KW), dim3(256), 0, (hipStream_t)stream,
                       w, scale, wt, Cout, KH * KW, Cin, KH, KW);
    return a3d_check_launch();
}

// n filters in one launch (a3d_weight_transpose_batch): a block finds its filter by binary search over the items' first-block prefix.
__global__ void weight_transpose_batch_kernel(const a3d_transpose_item *__restrict__ table, int n) {
    __shared__ float tile[32][33];
    int lo = 0, hi = n - 1;
    const int bid = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].block0 <= bid) lo = mid;
        else hi = mid - 1;
    }
    const a3d_transpose_item it = table[lo];
    const int T = it.KH * it.KW, nx = (it.Cin + 31) / 32, ny = (it.Cout + 31) / 32;
    int r_ = bid - it.block0;
    const int tap = r_ / (nx * ny);
    r_ -= tap * nx * ny;
    const int by = r_ / nx, bx = r_ - by * nx;
    const int ci0 = bx * 32, co0 = by * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int co = co0 + r, ci = ci0 + tx;
        float v = 0.f;
        if (co < it.Cout && ci < it.Cin) {
            v = it.w[((size_t)co * T + tap) * it.Cin + ci];
            if (it.scale) v *= it.scale[co];
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    const int kh = tap / it.KW, kw = tap - kh * it.KW;
    const int ftap = (it.KH - 1 - kh) * it.KW + (it.KW - 1 - kw);
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci0 + r, co = co0 + tx;
        if (ci < it.Cin && co < it.Cout) it.wt[((size_t)ci * T + ftap) * it.Cout + co] = tile[tx][r];
    }
}

extern "C" int a3d_weight_transpose_batch(const a3d_transpose_item *table, int n, int total_blocks, void *stream) {
    if (!table || n <= 0 || total_blocks <= 0) return A3D_ERR_ARG;
    a3d_begin();
    hipLaunchKernelGGL(weight_transpose_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, table, n);
    return a3d_check_launch();
}

// Winograd-domain weights U = G g G^T of a packed 3x3 filter [Cout][3][3][Cin] -> [16][Cout][Cin] (see conv_wino.hip).
// Needed every step in training because the filter changes; at inference this is done once at load time.
__global__ void wino_weight_kernel(const float *__restrict__ w, float *__restrict__ U, int Cout, int Cin) {
    const size_t n = (size_t)Cout * Cin;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i / Cin), ci = (int)(i - (size_t)co * Cin);
        float g[3][3];
        for (int p = 0; p < 3; ++p)
            for (int q = 0; q < 3; ++q) g[p][q] = w[((size_t)co * 9 + p * 3 + q) * Cin + ci];
        float t[4][3];  // G g
        for (int q = 0; q < 3; ++q) {
            t[0][q] = g[0][q];
            t[1][q] = 0.5f * (g[0][q] + g[1][q] + g[2][q]);
            t[2][q] = 0.5f * (g[0][q] - g[1][q] + g[2][q]);
            t[3][q] = g[2][q];
        }
        for (int u = 0; u < 4; ++u) {
            const float v0 = t[u][0], v1 = 0.5f * (t[u][0] + t[u][1] + t[u][2]), v2 = 0.5f * (t[u][0] - t[u][1] + t[u][2]), v3 = t[u][2];
            U[(size_t)(u * 4 + 0) * n + i] = v0;
            U[(size_t)(u * 4 + 1) * n + i] = v1;
            U[(size_t)(u * 4 + 2) * n + i] = v2;
            U[(size_t)(u * 4 + 3) * n + i] = v3;
        }
    }
}

extern "C" int a3d_wino_weight_transform(const float *w, float *U, int Cout, int Cin, void *stream) {
    if (!w || !U || Cout <= 0 || Cin <= 0) return A3D_ERR_ARG;
    const size_t n = (size_t)Cout * Cin;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    a3d_begin();
    hipLaunchKernelGGL(wino_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, U, Cout, Cin);
    return a3d_check_launch();
}

// ------------------------------------------------------------------------------------------------
// spatial gradient plumbing
// ------------------------------------------------------------------------------------------------
// y [B,Ho,Wo,C] = 0 except y[b,2i,2j,:] = x[b,i,j,:]   (x [B,H,W,C], H = ceil(Ho/2)): backward of a stride-2 1x1 conv's
// input sub-sampling and of the kernel-1 stride-2 pool that makes p6.  accumulate=1 adds into y instead.
__global__ void zero_insert2_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int H, int W, int C, int Ho, int Wo,
                                    int accumulate) {
    const int C4 = C >> 2;
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t r = i / C4;
        const int ow = (int)(r % Wo);
        r /= Wo;
        const int oh = (int)(r % Ho), b = (int)(r / Ho);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!((oh | ow) & 1)) v = *reinterpret_cast<const f32x4 *>(x + (((size_t)b * H + (oh >> 1)) * W + (ow >> 1)) * C + c4 * 4);
        f32x4 *o = reinterpret_cast<f32x4 *>(y + i * 4);
        *o = accumulate ? *o + v : v;
    }
}

extern "C" int a3d_zero_insert2_nhwc(const float *x, float *y, int B, int H, int W, int C, int Ho, int Wo, int accumulate, void *stream) {
    if (!x || !y || B <= 0 || (C & 3) || (Ho + 1) / 2 != H || (Wo + 1) / 2 != W) return A3D_ERR_ARG;
    const size_t total = (size_t)B * Ho * Wo * (C >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    a3d_begin();
    hipLaunchKernelGGL(zero_insert2_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C, Ho, Wo, accumulate);
    return a3d_check_launch();
}

// y [B,H,W,C] += sum of the 2x2 block of x [B,2H,2W,C]: backward of the nearest-x2 upsampling in the FPN top-down path.
__global__ void sumpool2_add_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int H, int W, int C) {
    const int C4 = C >> 2;
    const size_t total = (size_t)B * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        size_t r = i / C4;
        const int w = (int)(r % W);
        r /= W;
        const int h = (int)(r % H), b = (int)(r / H);
        const float *p = x + (((size_t)b * 2 * H + 2 * h) * (2 * W) + 2 * w) * C + c4 * 4;
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(p), a1 = *reinterpret_cast<const f32x4 *>(p + C);
        const f32x4 a2 = *reinterpret_cast<const f32x4 *>(p + (size_t)2 * W * C), a3 = *reinterpret_cast<const f32x4 *>(p + (size_t)2 * W * C + C);
        f32x4 *o = reinterpret_cast<f32x4 *>(y + i * 4);
        *o = *o + ((a0 + a1) + (a2 + a3));
    }
}

extern "C" int a3d_sumpool2_add_nhwc(const float *x, float *y, int B, int H, int W, int C, void *stream) {
    if (!x || !y || B <= 0 || (C & 3)) return A3D_ERR_ARG;
    const size_t total = (size_t)B * H * W * (C >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    a3d_begin();
    hipLaunchKernelGGL(sumpool2_add_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C);
    return a3d_check_launch();
}

// Bias gradient: out[c] (+)= sum over the M rows of dy [M, C].  Two deterministic stages: row slices -> workspace, then
// the slices are added in order.
__global__ void colsum_partial_kernel(const float *__restrict__ dy, float *__restrict__ ws, int M, int C, int rows_per) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;  // 4 row phases
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    float s = 0.f;
    if (c < C)
        for (int r = r0 + sub; r < r1; r += 4) s += dy[(size_t)r * C + c];
    __shared__ float red[4][64];
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && c < C) ws[(size_t)blockIdx.y * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// (64 channels per workgroup, the slices dealt over 4 row phases and combined in a fixed order: up to 256 slices one after the
// other per thread took longer than the first stage)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float *__restrict__ ws, float *__restrict__ out, int C, int slices, int accumulate) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int sub = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int k = sub; k < slices; k += 4) s += ws[(size_t)k * C + c];
    __shared__ float red[4][64];
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && c < C) {
        const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        out[c] = accumulate ? out[c] + t : t;
    }
}

// (C % 4 == 0: a lane owns 4 channels and keeps four independent 16-byte loads in flight -- the scalar form above ran the wide layers'
// 315 MB gradients at 0.7 TB/s, 2.5 ms of a 31 ms step at 16 images; fixed summation order: rows r, r+16, .. per accumulator, then
// ((a0 + a1) + (a2 + a3)), the 4 row phases as above)
template <bool BF>  // BF: dy is stored as bf16 (widening is exact: the same sums as on the widened values)
__global__ __launch_bounds__(256) void colsum_partial4_kernel(const void *__restrict__ dyv, float *__restrict__ ws, int M, int C, int rows_per) {
    const int c = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int sub = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    auto ld = [&](const int r) -> f32x4 {
        if constexpr (BF) {
            const uint2 q = *reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned short *>(dyv) + (size_t)r * C + c);
            f32x4 v;
            v[0] = __uint_as_float(q.x << 16);
            v[1] = __uint_as_float(q.x & 0xFFFF0000u);
            v[2] = __uint_as_float(q.y << 16);
            v[3] = __uint_as_float(q.y & 0xFFFF0000u);
            return v;
        } else {
            return *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(dyv) + (size_t)r * C + c);
        }
    };
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    if (c < C) {
        int r = r0 + sub;
        for (; r + 12 < r1; r += 16) {
            const f32x4 v0 = ld(r), v1 = ld(r + 4), v2 = ld(r + 8), v3 = ld(r + 12);
            a0 += v0;
            a1 += v1;
            a2 += v2;
            a3 += v3;
        }
        for (; r < r1; r += 4) a0 += ld(r);
    }
    const f32x4 s = (a0 + a1) + (a2 + a3);
    __shared__ f32x4 red[4][64];
    red[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && c < C)
        *reinterpret_cast<f32x4 *>(ws + (size_t)blockIdx.y * C + c) = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

#define A3D_COLSUM_SLICES 256
extern "C" size_t a3d_colsum_workspace_bytes(int C) { return (size_t)A3D_COLSUM_SLICES * C * sizeof(float); }
static int colsum_launch(const void *dy, bool bf, float *out, float *workspace, int M, int C, int accumulate, void *stream) {
    if (!dy || !out || !workspace || M <= 0 || C <= 0) return A3D_ERR_ARG;
    // slices: ~512 rows each, between 32 and the workspace's 256 (the second stage adds them one after the other)
    int want = M / 512;
    want = want < 32 ? 32 : (want > A3D_COLSUM_SLICES ? A3D_COLSUM_SLICES : want);
    const int rows_per = (M + want - 1) / want;
    const int slices = (M + rows_per - 1) / rows_per;
    a3d_begin();
    if (bf) {
        if ((C & 3) || (reinterpret_cast<size_t>(dy) & 7)) return A3D_ERR_ARG;
        hipLaunchKernelGGL(colsum_partial4_kernel<true>, dim3((C + 255) / 256, slices), dim3(256), 0, (hipStream_t)stream, dy, workspace, M, C, rows_per);
    } else if ((C & 3) == 0 && (reinterpret_cast<size_t>(dy) & 15) == 0)
        hipLaunchKernelGGL(colsum_partial4_kernel<false>, dim3((C + 255) / 256, slices), dim3(256), 0, (hipStream_t)stream, dy, workspace, M, C, rows_per);
    else
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((C + 63) / 64, slices), dim3(256), 0, (hipStream_t)stream, (const float *)dy, workspace, M, C, rows_per);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, workspace, out, C, slices, accumulate);
    return a3d_check_launch();
}
extern "C" int a3d_colsum(const float *dy, float *out, float *workspace, int M, int C, int accumulate, void *stream) {
    return colsum_launch(dy, false, out, workspace, M, C, accumulate, stream);
}
extern "C" int a3d_colsum_bf16(const void *dy, float *out, float *workspace, int M, int C, int accumulate, void *stream) {
    return colsum_launch(dy, true, out, workspace, M, C, accumulate, stream);
}

// ------------------------------------------------------------------------------------------------
// Matcher (detectron2.modeling.matcher.Matcher on pairwise_iou(gt, boxes)): per box the best ground-truth index and a
// label from the IoU thresholds; optionally "low quality" matches (every box that attains a gt's best IoU gets label 1).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pair_iou(const float *g, float ga, float x1, float y1, float x2, float y2, float ba) {
    float w = fminf(g[2], x2) - fmaxf(g[0], x1);
    float h = fminf(g[3], y2) - fmaxf(g[1], y1);
    w = w > 0.f ? w : 0.f;
    h = h > 0.f ? h : 0.f;
    const float inter = w * h;
    return inter > 0.f ? inter / (ga + ba - inter) : 0.f;
}

#define A3D_MATCH_MAX_GT 64
__global__ void match_pass1_kernel(const a3d_match_desc d) {
    __shared__ float gt[A3D_MATCH_MAX_GT][4], ga[A3D_MATCH_MAX_GT];
    __shared__ unsigned int best[A3D_MATCH_MAX_GT];
    const int b = blockIdx.y;
    const int G = min(d.gt_count[b], d.Gmax);
    if (threadIdx.x < A3D_MATCH_MAX_GT) {
        best[threadIdx.x] = 0u;
        if (threadIdx.x < G) {
            const float *g = d.gt_boxes + ((size_t)b * d.Gmax + threadIdx.x) * 4;
            for (int k = 0; k < 4; ++k) gt[threadIdx.x][k] = g[k];
            ga[threadIdx.x] = (g[2] - g[0]) * (g[3] - g[1]);
        }
    }
    __syncthreads();
    const int nb = d.box_count ? min(d.box_count[b], d.N) : d.N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        const float *bx = d.boxes + ((size_t)b * d.box_batch_stride + i) * 4;
        const float x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3];
        const float ba = (x2 - x1) * (y2 - y1);
        float bv = -1.f;
        int bi = 0;
        for (int g = 0; g < G; ++g) {
            const float v = pair_iou(gt[g], ga[g], x1, y1, x2, y2, ba);
            if (v > bv) {  // first maximum wins, like torch.max(dim=0)
                bv = v;
                bi = g;
            }
            if (d.allow_low_quality) atomicMax(&best[g], __float_as_uint(v));  // IoU >= 0: uint order == float order
        }
        if (G == 0) bv = 0.f;
        int label;
        if (d.n_thresholds == 2) label = bv < d.thresholds[0] ? d.labels[0] : (bv < d.thresholds[1] ? d.labels[1] : d.labels[2]);
        else label = bv < d.thresholds[0] ? d.labels[0] : d.labels[1];
        if (G == 0) label = d.labels[0];
        d.matched_idx[(size_t)b * d.N + i] = bi;
        d.label[(size_t)b * d.N + i] = (signed char)label;
        if (d.matched_iou) d.matched_iou[(size_t)b * d.N + i] = bv;
    }
    if (d.allow_low_quality) {
        __syncthreads();
        if (threadIdx.x < G) atomicMax(&d.gt_best[(size_t)b * d.Gmax + threadIdx.x], best[threadIdx.x]);
    }
}
__global__ void match_pass2_kernel(const a3d_match_desc d) {
    __shared__ float gt[A3D_MATCH_MAX_GT][4], ga[A3D_MATCH_MAX_GT], best[A3D_MATCH_MAX_GT];
    const int b = blockIdx.y;
    const int G = min(d.gt_count[b], d.Gmax);
    if (threadIdx.x < G) {
        const float *g = d.gt_boxes + ((size_t)b * d.Gmax + threadIdx.x) * 4;
        for (int k = 0; k < 4; ++k) gt[threadIdx.x][k] = g[k];
        ga[threadIdx.x] = (g[2] - g[0]) * (g[3] - g[1]);
        best[threadIdx.x] = __uint_as_float(d.gt_best[(size_t)b * d.Gmax + threadIdx.x]);
    }
    __syncthreads();
    const int nb = d.box_count ? min(d.box_count[b], d.N) : d.N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        const float *bx = d.boxes + ((size_t)b * d.box_batch_stride + i) * 4;
        const float x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3];
        const float ba = (x2 - x1) * (y2 - y1);
        bool hit = false;
        for (int g = 0; g < G; ++g) hit |= pair_iou(gt[g], ga[g], x1, y1, x2, y2, ba) == best[g];
        if (hit) d.label[(size_t)b * d.N + i] = 1;
    }
}

extern "C" int a3d_match_boxes(const a3d_match_desc *d, void *stream) {
    if (!d || !d->boxes || !d->gt_boxes || !d->gt_count || !d->matched_idx || !d->label || d->B <= 0 || d->N <= 0) return A3D_ERR_ARG;
    if (d->Gmax < 1 || d->Gmax > A3D_MATCH_MAX_GT || d->n_thresholds < 1 || d->n_thresholds > 2) return A3D_ERR_ARG;
    if (d->allow_low_quality && !d->gt_best) return A3D_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int blocks = (d->N + 255) / 256;
    if (blocks > 128) blocks = 128;
    a3d_begin();
    if (d->allow_low_quality) (void)hipMemsetAsync(d->gt_best, 0, (size_t)d->B * d->Gmax * sizeof(unsigned int), s);
    hipLaunchKernelGGL(match_pass1_kernel, dim3(blocks, d->B), dim3(256), 0, s, *d);
    if (d->allow_low_quality) hipLaunchKernelGGL(match_pass2_kernel, dim3(blocks, d->B), dim3(256), 0, s, *d);
    return a3d_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Box2BoxTransform.get_deltas
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void get_deltas(const float *src, const float *tgt, const float *w, float out[4]) {
    const float sw = src[2] - src[0], sh = src[3] - src[1];
    const float sx = src[0] + 0.5f * sw, sy = src[1] + 0.5f * sh;
    const float tw = tgt[2] - tgt[0], th = tgt[3] - tgt[1];
    const float tx = tgt[0] + 0.5f * tw, ty = tgt[1] + 0.5f * th;
    out[0] = w[0] * (tx - sx) / sw;
    out[1] = w[1] * (ty - sy) / sh;
    out[2] = w[2] * logf(tw / sw);
    out[3] = w[3] * logf(th / sh);
}

// ------------------------------------------------------------------------------------------------
// RPN.losses forward + backward in one pass over the head outputs: binary cross entropy with logits on the sampled
// anchors (label >= 0), L1 (smooth-L1 with beta 0) on the deltas of positive anchors, both divided by `normalizer`;
// gradients are written in the head's own layout so the 1x1 head conv's dgrad / wgrad consume them directly.
// Per-workgroup loss partials go to `partials` [grid][2]; a3d_sum_partials adds them in order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rpn_loss_kernel(const a3d_rpn_loss_desc d, const int level, const int level_off, float *partials,
                                                       const int part_off) {
    const int Hf = d.Hf[level], Wf = d.Wf[level], A = d.A, CH = d.CH;
    const int cells = Hf * Wf;
    const float *head = d.head[level];
    float *dhead = d.dhead[level];
    float lc = 0.f, ll = 0.f;
    const int total = d.B * cells;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / cells, cell = i - b * cells;
        const int y = cell / Wf, x = cell - y * Wf;
        const float *h = head + (size_t)i * CH;
        float *g = dhead + (size_t)i * CH;
        for (int c = 5 * A; c < CH; ++c) g[c] = 0.f;
        for (int a = 0; a < A; ++a) {
            const size_t ai = (size_t)b * d.Atotal + level_off + (size_t)cell * A + a;
            const int lab = d.labels[ai];
            float gl = 0.f, gd[4] = {0.f, 0.f, 0.f, 0.f};
            if (lab >= 0) {
                const float z = h[a], t = (float)lab;
                lc += fmaxf(z, 0.f) - z * t + log1pf(expf(-fabsf(z)));
                const float sg = 1.f / (1.f + expf(-z));
                gl = (sg - t) / d.normalizer;
            }
            if (lab == 1) {
                const float *ca = d.cell_anchors[level][a];
                const float sx = (float)(x * d.stride[level]), sy = (float)(y * d.stride[level]);
                const float anc[4] = {ca[0] + sx, ca[1] + sy, ca[2] + sx, ca[3] + sy};
                const int gi = d.matched_idx[ai];
                float tg[4];
                get_deltas(anc, d.gt_boxes + ((size_t)b * d.Gmax + gi) * 4, d.weights, tg);
                for (int k = 0; k < 4; ++k) {
                    const float df = h[A + a * 4 + k] - tg[k];
                    ll += fabsf(df);
                    gd[k] = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) / d.normalizer;
                }
            }
            g[a] = gl;
            for (int k = 0; k < 4; ++k) g[A + a * 4 + k] = gd[k];
        }
    }
    __shared__ float r0[256], r1[256];
    r0[threadIdx.x] = lc;
    r1[threadIdx.x] = ll;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            r0[threadIdx.x] += r0[threadIdx.x + s];
            r1[threadIdx.x] += r1[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partials[(size_t)(part_off + blockIdx.x) * 2 + 0] = r0[0];
        partials[(size_t)(part_off + blockIdx.x) * 2 + 1] = r1[0];
    }
}

__global__ void sum_partials_kernel(const float *partials, int n, int width, float scale, float *out) {
    if (threadIdx.x < width) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += (double)partials[(size_t)i * width + threadIdx.x];
        out[threadIdx.x] = (float)(s * (double)scale);
    }
}

#define A3D_LOSS_BLOCKS 64
extern "C" size_t a3d_loss_workspace_bytes(void) { return (size_t)5 * A3D_LOSS_BLOCKS * 2 * sizeof(float); }

extern "C" int a3d_rpn_loss(const a3d_rpn_loss_desc *d, void *stream) {
    if (!d || !d->labels || !d->matched_idx || !d->gt_boxes || !d->loss || !d->workspace) return A3D_ERR_ARG;
    if (d->B <= 0 || d->L < 1 || d->L > 5 || d->A < 1 || d->A > 3 || d->CH < 5 * d->A || !(d->normalizer > 0.f)) return A3D_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    a3d_begin();
    int off = 0, nparts = 0;
    for (int l = 0; l < d->L; ++l) {
        if (!d->head[l] || !d->dhead[l]) return A3D_ERR_ARG;
        const int total = d->B * d->Hf[l] * d->Wf[l];
        int blocks = (total + 255) / 256;
        if (blocks > A3D_LOSS_BLOCKS) blocks = A3D_LOSS_BLOCKS;
        hipLaunchKernelGGL(rpn_loss_kernel, dim3(blocks), dim3(256), 0, s, *d, l, off, d->workspace, nparts);
        off += d->Hf[l] * d->Wf[l] * d->A;
        nparts += blocks;
    }
    if (off != d->Atotal) return A3D_ERR_ARG;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, s, d->workspace, nparts, 2, 1.0f / d->normalizer, d->loss);
    return a3d_check_launch();
}

// ------------------------------------------------------------------------------------------------
// FastRCNNOutputLayers.losses forward + backward: mean softmax cross entropy over all sampled rows + L1 on the
// ground-truth class's 4 deltas of the foreground rows, divided by the row count.  pred row layout = the fused
// predictor's: [0, K] class scores (K = background), [K+1, K+1+4K) deltas (class-major); dpred has the same pitch.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void box_loss_kernel(const a3d_box_loss_desc d, float *partials) {
    const int K = d.num_classes;
    float lc = 0.f, lb = 0.f;
    int live = d.M;
    if (d.count) {  // ragged: only the first count[b] rows of every R-row block carry loss
        live = 0;
        for (int b = 0; b < d.M / d.R; ++b) live += min(d.count[b], d.R);
    }
    const float inv = 1.f / (float)max(live, 1);
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < d.M; r += gridDim.x * blockDim.x) {
        const float *p = d.pred + (size_t)r * d.pitch;
        float *g = d.dpred + (size_t)r * d.pitch;
        for (int c = 0; c < d.pitch; ++c) g[c] = 0.f;
        if (d.count && (r % d.R) >= d.count[r / d.R]) continue;
        const int cls = d.gt_classes[r];
        float mx = p[0];
        for (int c = 1; c <= K; ++c) mx = fmaxf(mx, p[c]);
        float se = 0.f;
        for (int c = 0; c <= K; ++c) se += expf(p[c] - mx);
        const float lse = mx + logf(se);
        lc += lse - p[cls];
        for (int c = 0; c <= K; ++c) g[c] = (expf(p[c] - lse) - (c == cls ? 1.f : 0.f)) * inv;
        if (cls >= 0 && cls < K) {
            float tg[4];
            get_deltas(d.boxes + (size_t)r * 4, d.gt_boxes + (size_t)r * 4, d.weights, tg);
            for (int k = 0; k < 4; ++k) {
                const int c = K + 1 + cls * 4 + k;
                const float df = p[c] - tg[k];
                lb += fabsf(df);
                g[c] = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * inv;
            }
        }
    }
    __shared__ float r0[256], r1[256];
    r0[threadIdx.x] = lc;
    r1[threadIdx.x] = lb;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            r0[threadIdx.x] += r0[threadIdx.x + s];
            r1[threadIdx.x] += r1[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partials[(size_t)blockIdx.x * 2 + 0] = r0[0] * inv;  // (normalised here: the live row count lives on the device)
        partials[(size_t)blockIdx.x * 2 + 1] = r1[0] * inv;
    }
}

extern "C" int a3d_box_loss(const a3d_box_loss_desc *d, void *stream) {
    if (!d || !d->pred || !d->dpred || !d->gt_classes || !d->boxes || !d->gt_boxes || !d->loss || !d->workspace) return A3D_ERR_ARG;
    if (d->M <= 0 || d->num_classes < 1 || d->pitch < 1 + 5 * d->num_classes) return A3D_ERR_ARG;
    if (d->count && (d->R <= 0 || d->M % d->R)) return A3D_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    int blocks = (d->M + 255) / 256;
    if (blocks > A3D_LOSS_BLOCKS) blocks = A3D_LOSS_BLOCKS;
    a3d_begin();
    hipLaunchKernelGGL(box_loss_kernel, dim3(blocks), dim3(256), 0, s, *d, d->workspace);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, s, d->workspace, blocks, 2, 1.0f, d->loss);
    return a3d_check_launch();
}

// ------------------------------------------------------------------------------------------------
// torch.optim.SGD with momentum and weight decay over one flat parameter buffer:
//   d = grad_scale * g + wd * p;  buf = first ? d : momentum * buf + d;  p -= lr * buf
// grad_scale = 1 / world_size after the gradient all-reduce (sum) of data-parallel training.
// ------------------------------------------------------------------------------------------------
template <bool GB16>  // GB16: the gradient arrives as bf16 (the all-reduced payload itself: no widening pass over the flat buffer)
__global__ void sgd_kernel(float *__restrict__ p, const void *__restrict__ g, float *__restrict__ buf, size_t n4, float lr, float momentum,
                           float wd, float grad_scale, int first) {
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 pv = reinterpret_cast<f32x4 *>(p)[i];
        f32x4 gv;
        if constexpr (GB16) gv = __builtin_convertvector(reinterpret_cast<const b4 *>(g)[i], f32x4);
        else gv = reinterpret_cast<const f32x4 *>(g)[i];
        const f32x4 dv = grad_scale * gv + wd * pv;
        f32x4 bv = first ? dv : momentum * reinterpret_cast<f32x4 *>(buf)[i] + dv;
        reinterpret_cast<f32x4 *>(buf)[i] = bv;
        reinterpret_cast<f32x4 *>(p)[i] = pv - lr * bv;
    }
}

static int sgd_launch(float *p, const void *g, bool g_bf16, float *buf, size_t n, float lr, float momentum, float wd, float grad_scale, int first,
                      void *stream) {
    if (!p || !g || !buf || (n & 3)) return A3D_ERR_ARG;
    const size_t n4 = n >> 2;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    a3d_begin();
    if (g_bf16) hipLaunchKernelGGL(sgd_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, buf, n4, lr, momentum, wd, grad_scale, first);
    else hipLaunchKernelGGL(sgd_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, buf, n4, lr, momentum, wd, grad_scale, first);
    return a3d_check_launch();
}

extern "C" int a3d_sgd_momentum(float *p, const float *g, float *buf, size_t n, float lr, float momentum, float wd, float grad_scale,
                                int first, void *stream) {
    return sgd_launch(p, g, false, buf, n, lr, momentum, wd, grad_scale, first, stream);
}

// The same update reading the gradient as bf16 -- the payload of the bf16 gradient all-reduce where the collective left it.  Widening is
// exact, so the update equals a3d_bf16_to_f32 followed by a3d_sgd_momentum bit for bit, without the pass over the flat buffer.
extern "C" int a3d_sgd_momentum_bf16g(float *p, const void *g_bf16, float *buf, size_t n, float lr, float momentum, float wd, float grad_scale,
                                      int first, void *stream) {
    return sgd_launch(p, g_bf16, true, buf, n, lr, momentum, wd, grad_scale, first, stream);
}


// ---- bf16 payload of the data-parallel gradient all-reduce (BASELINE configs[4]: "bf16 ... grad all-reduce over xGMI") ------------
// torch's DDP bf16_compress_hook: the gradient is divided by the world size, rounded to bf16 (nearest even), summed by the collective
// in bf16 and widened back.  Two HBM-bound passes over the flat gradient buffer; the collective moves half the bytes (82 MB, not 164).
__global__ __launch_bounds__(256) void f32_to_bf16_scaled_kernel(const float *__restrict__ src, __bf16 *__restrict__ dst, size_t n4, float scale) {
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 v = reinterpret_cast<const f32x4 *>(src)[i] * scale;
        reinterpret_cast<b4 *>(dst)[i] = __builtin_convertvector(v, b4);
    }
}
__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const __bf16 *__restrict__ src, float *__restrict__ dst, size_t n4) {
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<f32x4 *>(dst)[i] = __builtin_convertvector(reinterpret_cast<const b4 *>(src)[i], f32x4);
}

extern "C" int a3d_f32_to_bf16_scaled(const float *src, void *dst, size_t n, float scale, void *stream) {
    if (!src || !dst || (n & 3)) return A3D_ERR_ARG;
    if (n == 0) return A3D_OK;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    a3d_begin();
    hipLaunchKernelGGL(f32_to_bf16_scaled_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (__bf16 *)dst, n / 4, scale);
    return a3d_check_launch();
}

extern "C" int a3d_bf16_to_f32(const void *src, float *dst, size_t n, void *stream) {
    if (!src || !dst || (n & 3)) return A3D_ERR_ARG;
    if (n == 0) return A3D_OK;
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    a3d_begin();
    hipLaunchKernelGGL(bf16_to_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)src, dst, n / 4);
    return a3d_check_launch();
}
