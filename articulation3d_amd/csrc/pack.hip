// Fixed-size detection records: the payload of the per-frame all-gather that feeds the host-side
// temporal optimiser.  One record = the fields create_instances() materialises for one detection
// (pkg/utils/arti_vis.py:162-186): box xyxy (4), score, class, plane normal*offset (3), rot axis (3),
// tran axis (2) and the 28x28 soft mask (re-pasted deterministically by the receiver) = 798 floats.
#include "a3d_common.h"
#include "../../include/a3d.h"

#define REC_HEAD 14

struct PackArgs {
    const float *boxes, *scores;
    const int *classes, *count, *row_offset, *keep;
    const float *planes, *rot_axis, *tran_axis, *mask_prob;
    int B, R, MS;
    float *records;
    int *rec_count;
};

__global__ __launch_bounds__(256) void pack_kernel(const PackArgs a) {
    __shared__ int dst[1024];
    __shared__ int s_total;
    const int b = blockIdx.x;
    const int cnt = min(a.count[b], a.R);
    if (threadIdx.x == 0) {
        int t = 0;
        for (int r = 0; r < a.R; ++r) {
            const bool k = r < cnt && a.keep[b * a.R + r];
            dst[r] = k ? t : -1;
            t += k ? 1 : 0;
        }
        s_total = t;
        a.rec_count[b] = t;
    }
    __syncthreads();
    const int mm = a.MS * a.MS, rec = REC_HEAD + mm;
    const int total = s_total;
    float *out = a.records + (size_t)b * a.R * rec;
    for (int r = 0; r < a.R; ++r) {
        const int d = dst[r];
        if (d < 0) continue;
        const int slot = b * a.R + r, row = a.row_offset[b] + r;
        float *o = out + (size_t)d * rec;
        if (threadIdx.x < REC_HEAD) {
            const int t = threadIdx.x;
            float v;
            if (t < 4) v = a.boxes[slot * 4 + t];
            else if (t == 4) v = a.scores[slot];
            else if (t == 5) v = (float)a.classes[slot];
            else if (t < 9) v = a.planes ? a.planes[slot * 3 + (t - 6)] : 0.f;
            else if (t < 12) v = a.rot_axis ? a.rot_axis[(size_t)row * 3 + (t - 9)] : 0.f;
            else v = a.tran_axis ? a.tran_axis[(size_t)row * 2 + (t - 12)] : 0.f;
            o[t] = v;
        }
        if (a.mask_prob)
            for (int i = threadIdx.x; i < mm; i += blockDim.x) o[REC_HEAD + i] = a.mask_prob[(size_t)row * mm + i];
    }
    // zero the unused tail so the gathered buffer is deterministic
    for (size_t i = (size_t)total * rec + threadIdx.x; i < (size_t)a.R * rec; i += blockDim.x) out[i] = 0.f;
}

extern "C" int a3d_record_floats(int MS) { return REC_HEAD + MS * MS; }

extern "C" int a3d_detections_pack(const a3d_pack_desc *d, void *stream) {
    if (!d || !d->boxes || !d->scores || !d->classes || !d->count || !d->row_offset || !d->keep || !d->records ||
        !d->rec_count)
        return A3D_ERR_ARG;
    if (d->B <= 0 || d->R <= 0 || d->R > 1024 || d->MS <= 0) return A3D_ERR_ARG;
    PackArgs a;
    a.boxes = d->boxes;
    a.scores = d->scores;
    a.classes = d->classes;
    a.count = d->count;
    a.row_offset = d->row_offset;
    a.keep = d->keep;
    a.planes = d->planes;
    a.rot_axis = d->rot_axis;
    a.tran_axis = d->tran_axis;
    a.mask_prob = d->mask_prob;
    a.B = d->B;
    a.R = d->R;
    a.MS = d->MS;
    a.records = d->records;
    a.rec_count = d->rec_count;
    a3d_begin();
    hipLaunchKernelGGL(pack_kernel, dim3(d->B), dim3(256), 0, (hipStream_t)stream, a);
    return a3d_check_launch();
}

// sizeof() of every descriptor struct, so a binding can verify its mirror of the layouts (ids in include/a3d.h).
// ------------------------------------------------------------------------------------------------
// COCO run-length encoding of pasted masks (the `segmentation` field of the reference's per-frame record:
// pkg/utils/arti_vis.py:66-67 -> detectron2 instances_to_coco_json -> pycocotools mask.encode).  cocoapi walks a mask
// COLUMN-major (Fortran order) and emits the lengths of its alternating 0 / 1 runs.  The device side finds the run boundaries:
// pos[k] = column-major index i >= 1 with m[i] != m[i-1], in increasing order; the host turns them into run lengths and the
// compressed ASCII string (a few hundred integers instead of a 307 KB mask over PCIe, and no 307 200-element host scan).
// Byte / index work, L2-bound, deterministic.
// ------------------------------------------------------------------------------------------------
// Implementation: one 1024-thread workgroup per mask.  (1) The mask is re-read in COLUMN-major order and packed 32 pixels per word
// into LDS (thread t builds words t, t + 1024, ...: 32 independent byte loads each, neighbouring threads walk neighbouring columns
// segments).  (2) A word's boundaries are the set bits of w ^ ((w << 1) | last bit of the previous word); popcounts are scanned
// over the words (wave shuffles + one LDS hop).  (3) Every word writes the indices of its set bits at its offset.  A first version
// walked each column with one thread (480 dependent compares behind 480 strided loads, twice): 0.55 ms per frame; this form ~0.05.
#define RLE_MAX_WORDS 12288  // 48 KiB of LDS: masks of up to 393 216 pixels (480 x 640 = 9 600 words)
__global__ __launch_bounds__(1024) void mask_rle_kernel(const unsigned char *__restrict__ masks, int H, int W, int cap, int *__restrict__ pos,
                                                        int *__restrict__ count, int *__restrict__ first) {
    __shared__ unsigned words[RLE_MAX_WORDS];
    __shared__ int wave_sum[16];
    __shared__ int s_carry;
    const unsigned char *m = masks + (size_t)blockIdx.x * H * W;
    int *out = pos + (size_t)blockIdx.x * cap;
    const int N = H * W, NW = (N + 31) >> 5;
    for (int j = threadIdx.x; j < NW; j += 1024) {
        unsigned w = 0;
        int i = j << 5;
        int x = i / H, y = i - x * H;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) {
            if (i + k < N) w |= (m[(size_t)y * W + x] != 0 ? 1u : 0u) << k;
            if (++y == H) {
                y = 0;
                ++x;
            }
        }
        words[j] = w;
    }
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    // boundaries per word; exclusive scan in passes of 1024 words
    for (int j0 = 0; j0 < NW; j0 += 1024) {
        const int j = j0 + threadIdx.x;
        unsigned t = 0;
        if (j < NW) {
            const unsigned w = words[j];
            const unsigned prev = j > 0 ? (words[j - 1] >> 31) : (w & 1u);  // (no boundary at i = 0)
            t = w ^ ((w << 1) | prev);
            const int valid = N - (j << 5);
            if (valid < 32) t &= (1u << valid) - 1u;
        }
        const int c = __popc(t);
        int incl = c;
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(incl, off, 64);
            if ((threadIdx.x & 63) >= off) incl += u;
        }
        if ((threadIdx.x & 63) == 63) wave_sum[threadIdx.x >> 6] = incl;
        __syncthreads();
        int base = s_carry + incl - c;
        for (int q = 0; q < (int)(threadIdx.x >> 6); ++q) base += wave_sum[q];
        int k = base;
        while (t) {  // the indices of the set bits, in increasing order
            const int b = __ffs(t) - 1;
            t &= t - 1;
            if (k < cap) out[k] = (j << 5) + b;
            ++k;
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = base + c;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        count[blockIdx.x] = s_carry;
        first[blockIdx.x] = (int)(words[0] & 1u);
    }
}

extern "C" int a3d_mask_rle(const unsigned char *masks, int D, int H, int W, int cap, int *pos, int *count, int *first, void *stream) {
    if (!masks || !pos || !count || !first || D < 0 || H <= 0 || W <= 0 || (size_t)H * W > (size_t)RLE_MAX_WORDS * 32 || cap <= 0) return A3D_ERR_ARG;
    if (D == 0) return A3D_OK;
    a3d_begin();
    hipLaunchKernelGGL(mask_rle_kernel, dim3(D), dim3(1024), 0, (hipStream_t)stream, masks, H, W, cap, pos, count, first);
    return a3d_check_launch();
}

extern "C" size_t a3d_struct_size(int id) {
    switch (id) {
        case 0: return sizeof(a3d_conv_desc);
        case 1: return sizeof(a3d_rpn_desc);
        case 2: return sizeof(a3d_boxdet_desc);
        case 3: return sizeof(a3d_roialign_desc);
        case 4: return sizeof(a3d_paste_desc);
        case 5: return sizeof(a3d_pack_desc);
        case 6: return sizeof(a3d_wgrad_desc);
        case 7: return sizeof(a3d_roialign_bwd_desc);
        case 8: return sizeof(a3d_match_desc);
        case 9: return sizeof(a3d_rpn_loss_desc);
        case 10: return sizeof(a3d_box_loss_desc);
        case 11: return sizeof(a3d_roi_sample_desc);
        case 12: return sizeof(a3d_sweep_desc);
        case 13: return sizeof(a3d_transpose_item);
        default: return 0;
    }
}
