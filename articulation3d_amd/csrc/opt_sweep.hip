// Hypothesis sweeps of the temporal optimiser on the GPU.
// Replaces the inner blocks of optimize_planes_3dc / optimize_planes_3d_trans
// (pkg/utils/opt_utils.py:400-456 + 462-476, 540-596 + 600-611, 700-748 + 753-768, 838-905):
//   a detection's mask is lifted onto its plane (get_pcd, pkg/utils/vis.py:86-102), moved by every hypothesis of the
//   sweep (45 rotations about the 3-D axis, or 20 translations along it), re-projected (project2D, vis.py:62-83), written
//   into one binary mask per hypothesis, and compared by IoU with the masks of the tracked detections.
// The reference does this with a Python loop over hypotheses and over tracked frames (one scatter + two full-image
// reductions each); here it is one launch for all hypotheses and one for the whole IoU matrix, on bit-packed masks
// (9 600 words per 480x640 mask): HBM / L2-bound byte work, no MFMA.
// Built with -ffp-contract=off: the projection rounds like the reference's separate multiplies / adds, the pixel index is
// a truncation, so a fused multiply-add could move a point across a pixel boundary.
#include "a3d_common.h"
#include "../../include/a3d.h"

// masks [n,H,W] uint8 (non-zero = set) -> bits [n, words], words = ceil(H*W/32); bit i of word w = pixel 32w+i
// (gridDim.y is capped at 65535 masks: the kernels stride over n, so a long clip's detections pack in one call)
__global__ void pack_bits_kernel(const unsigned char *__restrict__ m, unsigned int *__restrict__ bits, size_t npix, int words, int count) {
    for (int n = blockIdx.y; n < count; n += gridDim.y) {
        for (int w = blockIdx.x * blockDim.x + threadIdx.x; w < words; w += gridDim.x * blockDim.x) {
            unsigned int v = 0;
            const size_t base = (size_t)n * npix + (size_t)w * 32;
            for (int i = 0; i < 32; ++i)
                if ((size_t)w * 32 + i < npix && m[base + i]) v |= 1u << i;
            bits[(size_t)n * words + w] = v;
        }
    }
}

extern "C" int a3d_masks_pack_bits(const unsigned char *masks, unsigned int *bits, int n, int H, int W, void *stream) {
    if (!masks || !bits || n <= 0 || H <= 0 || W <= 0) return A3D_ERR_ARG;
    const size_t npix = (size_t)H * W;
    const int words = (int)((npix + 31) / 32);
    a3d_begin();
    hipLaunchKernelGGL(pack_bits_kernel, dim3((words + 255) / 256, n < 65535 ? n : 65535), dim3(256), 0, (hipStream_t)stream, masks, bits, npix,
                       words, n);
    return a3d_check_launch();
}

__global__ void unpack_bits_kernel(const unsigned int *__restrict__ bits, unsigned char *__restrict__ m, size_t npix, int words, int count) {
    for (int n = blockIdx.y; n < count; n += gridDim.y)
        for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x)
            m[(size_t)n * npix + p] = (bits[(size_t)n * words + (p >> 5)] >> (p & 31)) & 1u;
}

extern "C" int a3d_masks_unpack_bits(const unsigned int *bits, unsigned char *masks, int n, int H, int W, void *stream) {
    if (!masks || !bits || n <= 0 || H <= 0 || W <= 0) return A3D_ERR_ARG;
    const size_t npix = (size_t)H * W;
    a3d_begin();
    hipLaunchKernelGGL(unpack_bits_kernel, dim3((int)((npix + 255) / 256), n < 65535 ? n : 65535), dim3(256), 0, (hipStream_t)stream, bits, masks,
                       npix, (int)((npix + 31) / 32), n);
    return a3d_check_launch();
}

// One thread per source pixel; the A hypotheses (R | t about `pivot`) sit in LDS.
__global__ __launch_bounds__(256) void project_hypotheses_kernel(const a3d_sweep_desc d, const int words) {
    __shared__ float xf[A3D_SWEEP_MAX_HYP][12];
    for (int i = threadIdx.x; i < d.A * 12; i += blockDim.x) xf[i / 12][i % 12] = d.xforms[i];
    __syncthreads();
    const int npix = d.H * d.W;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix || !d.mask[p]) return;
    const int y = p / d.W, x = p - y * d.W;
    // get_pcd in float64: ray = K^-1 [x, y, 1], depth = offset / (normal . ray), point = depth * ray; then .float()
    const double rx = ((double)x - d.cx) / d.focal, ry = ((double)y - d.cy) / d.focal;
    const double den = (double)d.normal[0] * rx + (double)d.normal[1] * ry + (double)d.normal[2];
    const double depth = (double)d.offset / den;
    const float px = (float)(depth * rx), py = (float)(depth * ry), pz = (float)depth;
    const float f = (float)d.focal, cx = (float)d.cx, cy = (float)d.cy;
    for (int a = 0; a < d.A; ++a) {
        const float *m = xf[a];
        const float qx = px - d.pivot[0], qy = py - d.pivot[1], qz = pz - d.pivot[2];
        const float tx = (m[0] * qx + m[1] * qy + m[2] * qz) + d.pivot[0] + m[9];
        const float ty = (m[3] * qx + m[4] * qy + m[5] * qz) + d.pivot[1] + m[10];
        const float tz = (m[6] * qx + m[7] * qy + m[8] * qz) + d.pivot[2] + m[11];
        // project2D: K @ point, divide by z, truncate (.long()), clamp to the image
        const float u = (f * tx + cx * tz) / tz, v = (f * ty + cy * tz) / tz;
        int col = (u != u) ? 0 : (u >= 2147483520.f ? 2147483647 : (u <= -2147483648.f ? -2147483647 - 1 : (int)u));
        int row = (v != v) ? 0 : (v >= 2147483520.f ? 2147483647 : (v <= -2147483648.f ? -2147483647 - 1 : (int)v));
        col = min(max(col, 0), d.W - 1);
        row = min(max(row, 0), d.H - 1);
        const int q = row * d.W + col;
        atomicOr(&d.out_bits[(size_t)a * words + (q >> 5)], 1u << (q & 31));
    }
}

extern "C" int a3d_project_hypotheses(const a3d_sweep_desc *d, void *stream) {
    if (!d || !d->mask || !d->xforms || !d->out_bits || d->H <= 0 || d->W <= 0 || d->A < 1 || d->A > A3D_SWEEP_MAX_HYP) return A3D_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int npix = d->H * d->W, words = (npix + 31) / 32;
    a3d_begin();
    (void)hipMemsetAsync(d->out_bits, 0, (size_t)d->A * words * sizeof(unsigned int), s);
    hipLaunchKernelGGL(project_hypotheses_kernel, dim3((npix + 255) / 256), dim3(256), 0, s, *d, words);
    return a3d_check_launch();
}

// iou[f][a] = popcount(target[f] & proj[a]) / popcount(target[f] | proj[a])  (0/0 -> NaN, like the reference's division)
__global__ __launch_bounds__(256) void mask_iou_kernel(const unsigned int *__restrict__ target, const unsigned int *__restrict__ proj, float *__restrict__ iou,
                                                       int A, int words) {
    const int a = blockIdx.x, f = blockIdx.y;
    const unsigned int *t = target + (size_t)f * words, *p = proj + (size_t)a * words;
    int in = 0, un = 0;
    for (int w = threadIdx.x; w < words; w += blockDim.x) {
        const unsigned int x = t[w], y = p[w];
        in += __popc(x & y);
        un += __popc(x | y);
    }
    __shared__ int si[256], su[256];
    si[threadIdx.x] = in;
    su[threadIdx.x] = un;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            si[threadIdx.x] += si[threadIdx.x + s];
            su[threadIdx.x] += su[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) iou[(size_t)f * A + a] = (float)si[0] / (float)su[0];
}

extern "C" int a3d_mask_iou_matrix(const unsigned int *target_bits, const unsigned int *proj_bits, float *iou, int F, int A, int H, int W,
                                   void *stream) {
    if (!target_bits || !proj_bits || !iou || F <= 0 || A <= 0 || H <= 0 || W <= 0) return A3D_ERR_ARG;
    a3d_begin();
    hipLaunchKernelGGL(mask_iou_kernel, dim3(A, F), dim3(256), 0, (hipStream_t)stream, target_bits, proj_bits, iou, A, (H * W + 31) / 32);
    return a3d_check_launch();
}
