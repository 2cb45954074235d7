// Device-side label sub-sampling of the training step: detectron2's subsample_labels
// (RPN._subsample_labels: 256 anchors per image, at most half positive; ROIHeads._sample_proposals: 512 proposals per
// image, at most a quarter foreground) without leaving the GPU, so the training step has no host synchronisation.
//
// torch.randperm(n)[:k] draws a uniformly random k-subset.  The same distribution is produced here by giving every
// candidate an independent pseudo-random key (a 31-bit hash of (seed, image, index), ties broken by the index) and keeping
// the k smallest keys of each class: a counter-based RNG, reproducible for a given seed and independent of the launch
// geometry.  Anchors (76 740 per image): 48-bit radix select of the k-th smallest key, 8 bits per pass with an LDS
// histogram -- the selection kernel of proposals.hip run on keys instead of scores.  Proposals (<= ~1 000 per image):
// every candidate counts the smaller keys of its class (rank), which also gives the output order.
#include "a3d_common.h"
#include "../../include/a3d.h"

typedef unsigned long long u64;

__device__ __forceinline__ unsigned int mix32(unsigned int x) {  // murmur3 finaliser
    x ^= x >> 16;
    x *= 0x85ebca6bu;
    x ^= x >> 13;
    x *= 0xc2b2ae35u;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ u64 sample_key(unsigned long long seed, int b, int i) {
    const unsigned int h = mix32(mix32((unsigned int)seed ^ (unsigned int)(seed >> 32) ^ 0x9e3779b9u * (unsigned int)(b + 1)) ^
                                 0x85ebca6bu * (unsigned int)(i + 1));
    return ((u64)(h >> 1) << 17) | (u64)i;  // 31 + 17 bits; i < 2^17
}

// k-th smallest key (1-based k) among the elements of class `cls` (label == cls) of one image; 0 if k == 0
__device__ u64 kth_smallest(const signed char *lab, int n, int cls, int k, unsigned long long seed, int b, unsigned int *hist, u64 *s_prefix,
                            int *s_krem) {
    if (k <= 0) return 0;
    if (threadIdx.x == 0) {
        *s_prefix = 0;
        *s_krem = k;
    }
    for (int shift = 40; shift >= 0; shift -= 8) {
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        const u64 prefix = *s_prefix;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            if (lab[i] != cls) continue;
            const u64 c = sample_key(seed, b, i);
            if ((c >> (shift + 8)) == prefix) atomicAdd(&hist[(unsigned)(c >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int krem = *s_krem, bin = 0;
            unsigned cum = 0;
            for (; bin < 255; ++bin) {
                if (cum + hist[bin] >= (unsigned)krem) break;
                cum += hist[bin];
            }
            *s_krem = krem - (int)cum;
            *s_prefix = (prefix << 8) | (u64)bin;
        }
        __syncthreads();
    }
    return *s_prefix;
}

// Round 4: three scans of the labels instead of fourteen.  The radix select above re-reads all N labels and re-hashes their keys in each
// of its 6 passes per class (0.33 ms per step for ONE workgroup per image: 4 % of the step at the reference's 2 images per GPU).  Here
// one scan builds the top-byte histograms of both classes, a second one gathers the (~N / 256) candidates of each class's deciding bin
// into LDS, the k-th smallest key is found among those by rank counting, and the third scan writes the labels: the same thresholds, the
// same output.  A class whose deciding bin overflows the candidate list (it cannot with hashed keys; 2048 slots for ~300) falls back to
// the radix select.
#define A3D_SL_CAP 2048
__global__ __launch_bounds__(1024) void sample_labels_kernel(const signed char *__restrict__ labels, signed char *__restrict__ out, int N, int num,
                                                             int max_pos, unsigned long long seed) {
    __shared__ unsigned int hist[2][256];
    __shared__ u64 cand[2][A3D_SL_CAP];
    __shared__ u64 s_prefix, s_thr[2];
    __shared__ int s_krem, s_k[2], s_bin[2], s_rem[2], s_n[2];
    const int b = blockIdx.x;
    const signed char *lab = labels + (size_t)b * N;
    signed char *o = out + (size_t)b * N;
    for (int i = threadIdx.x; i < 512; i += blockDim.x) (&hist[0][0])[i] = 0;
    if (threadIdx.x < 2) s_n[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const int l = lab[i];
        if (l == 0 || l == 1) atomicAdd(&hist[l][(unsigned)(sample_key(seed, b, i) >> 40) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // class 1 first: the negatives fill what the positives leave
        unsigned tot[2] = {0, 0};
        for (int c = 0; c < 2; ++c)
            for (int j = 0; j < 256; ++j) tot[c] += hist[c][j];
        const int k_pos = min((int)tot[1], max_pos);
        const int k_neg = min((int)tot[0], num - k_pos);
        s_k[1] = k_pos;
        s_k[0] = k_neg;
        for (int c = 0; c < 2; ++c) {
            int bin = 0;
            unsigned cum = 0;
            for (; bin < 255; ++bin) {
                if (cum + hist[c][bin] >= (unsigned)s_k[c]) break;
                cum += hist[c][bin];
            }
            s_bin[c] = bin;
            s_rem[c] = s_k[c] - (int)cum;  // rank (1-based) of the threshold key inside its bin
        }
    }
    __syncthreads();
    const int k_pos = s_k[1], k_neg = s_k[0];
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const int l = lab[i];
        if (l != 0 && l != 1) continue;
        if (s_k[l] <= 0) continue;
        const u64 key = sample_key(seed, b, i);
        if ((int)((unsigned)(key >> 40) & 255u) != s_bin[l]) continue;
        const int slot = atomicAdd(&s_n[l], 1);
        if (slot < A3D_SL_CAP) cand[l][slot] = key;
    }
    __syncthreads();
    for (int c = 0; c < 2; ++c) {
        const int n = s_n[c];
        if (s_k[c] <= 0 || n > A3D_SL_CAP) continue;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {  // keys are distinct (their low 17 bits are the index)
            const u64 ki = cand[c][i];
            int rank = 0;
            for (int j = 0; j < n; ++j) rank += cand[c][j] < ki;
            if (rank == s_rem[c] - 1) s_thr[c] = ki;
        }
    }
    __syncthreads();
    u64 t_pos = s_thr[1], t_neg = s_thr[0];
    if (k_pos > 0 && s_n[1] > A3D_SL_CAP) {  // (never with hashed keys)
        t_pos = kth_smallest(lab, N, 1, k_pos, seed, b, hist[0], &s_prefix, &s_krem);
        __syncthreads();
    }
    if (k_neg > 0 && s_n[0] > A3D_SL_CAP) {
        t_neg = kth_smallest(lab, N, 0, k_neg, seed, b, hist[0], &s_prefix, &s_krem);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const int l = lab[i];
        signed char r = -1;
        if (l == 1 && k_pos > 0 && sample_key(seed, b, i) <= t_pos) r = 1;
        if (l == 0 && k_neg > 0 && sample_key(seed, b, i) <= t_neg) r = 0;
        o[i] = r;
    }
}

extern "C" int a3d_sample_labels(const signed char *labels, signed char *out, int B, int N, int num, int max_pos, unsigned long long seed,
                                 void *stream) {
    if (!labels || !out || B <= 0 || N <= 0 || N >= (1 << 17) || num <= 0 || max_pos < 0 || max_pos > num) return A3D_ERR_ARG;
    a3d_begin();
    hipLaunchKernelGGL(sample_labels_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, labels, out, N, num, max_pos, seed);
    return a3d_check_launch();
}

// ------------------------------------------------------------------------------------------------
// ROIHeads.label_and_sample_proposals after the matcher, in one launch per image:
//   boxes = [proposals | ground truth] (add_ground_truth_to_proposals), class = gt class of the matched box or K
//   (background), sample <= max_fg foreground + background up to `num`, and gather the sampled boxes, classes and matched
//   ground-truth boxes into fixed [B, num] outputs (foreground first, random order inside each class).
// ------------------------------------------------------------------------------------------------
#define A3D_SAMPLE_MAXN 2048
__global__ __launch_bounds__(1024) void sample_rois_kernel(const a3d_roi_sample_desc d) {
    __shared__ u64 key[A3D_SAMPLE_MAXN];
    __shared__ short cls[A3D_SAMPLE_MAXN];
    __shared__ int s_fg, s_bg;
    const int b = blockIdx.x;
    const int n = min(d.box_count[b], d.N);
    const int G = d.gt_count[b];
    if (threadIdx.x == 0) s_fg = s_bg = 0;
    __syncthreads();
    int f = 0, g = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        int c = d.num_classes;
        if (G > 0 && d.match_label[(size_t)b * d.N + i] == 1) c = d.gt_classes[(size_t)b * d.Gmax + d.matched_idx[(size_t)b * d.N + i]];
        cls[i] = (short)c;
        key[i] = sample_key(d.seed, b, i);
        f += c < d.num_classes;
        g += c == d.num_classes;
    }
    atomicAdd(&s_fg, f);
    atomicAdd(&s_bg, g);
    __syncthreads();
    const int k_fg = min(s_fg, d.max_fg);
    const int k_bg = min(s_bg, d.num - k_fg);
    if (threadIdx.x == 0) d.out_count[b] = k_fg + k_bg;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const bool fg = cls[i] < d.num_classes;
        const u64 ki = key[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += ((cls[j] < d.num_classes) == fg) && key[j] < ki;
        const int slot = fg ? (rank < k_fg ? rank : -1) : (rank < k_bg ? k_fg + rank : -1);
        if (slot < 0) continue;
        const size_t o = (size_t)b * d.num + slot;
        const float *bx = d.boxes + ((size_t)b * d.N + i) * 4;
        const int gi = G > 0 ? d.matched_idx[(size_t)b * d.N + i] : 0;
        const float *gb = G > 0 ? d.gt_boxes + ((size_t)b * d.Gmax + gi) * 4 : bx;
        for (int c = 0; c < 4; ++c) {
            d.out_boxes[o * 4 + c] = bx[c];
            d.out_gt_boxes[o * 4 + c] = gb[c];
        }
        d.out_classes[o] = cls[i];
        d.out_index[o] = i;
    }
    // slots past the count: zero boxes, background class (their rows carry no loss and no gradient)
    for (int s = k_fg + k_bg + threadIdx.x; s < d.num; s += blockDim.x) {
        const size_t o = (size_t)b * d.num + s;
        for (int c = 0; c < 4; ++c) d.out_boxes[o * 4 + c] = d.out_gt_boxes[o * 4 + c] = 0.f;
        d.out_classes[o] = d.num_classes;
        d.out_index[o] = -1;
    }
}

extern "C" int a3d_sample_rois(const a3d_roi_sample_desc *d, void *stream) {
    if (!d || !d->boxes || !d->box_count || !d->gt_boxes || !d->gt_classes || !d->gt_count || !d->matched_idx || !d->match_label) return A3D_ERR_ARG;
    if (!d->out_boxes || !d->out_gt_boxes || !d->out_classes || !d->out_index || !d->out_count) return A3D_ERR_ARG;
    if (d->B <= 0 || d->N <= 0 || d->N > A3D_SAMPLE_MAXN || d->num <= 0 || d->max_fg < 0 || d->max_fg > d->num || d->Gmax < 1) return A3D_ERR_ARG;
    a3d_begin();
    hipLaunchKernelGGL(sample_rois_kernel, dim3(d->B), dim3(1024), 0, (hipStream_t)stream, *d);
    return a3d_check_launch();
}

// add_ground_truth_to_proposals: out[b] = [proposals[b, :count[b]] | gt[b, :gt_count[b]] | zeros], out_count = count + gt_count
__global__ void append_gt_kernel(const float *__restrict__ props, const int *__restrict__ count, const float *__restrict__ gt,
                                 const int *__restrict__ gt_count, float *__restrict__ out, int *__restrict__ out_count, int R, int Gmax) {
    const int b = blockIdx.x;
    const int n = min(count[b], R), g = min(gt_count[b], Gmax);
    const int N = R + Gmax;
    for (int i = threadIdx.x; i < N * 4; i += blockDim.x) {
        const int r = i >> 2, c = i & 3;
        float v = 0.f;
        if (r < n) v = props[((size_t)b * R + r) * 4 + c];
        else if (r < n + g) v = gt[((size_t)b * Gmax + (r - n)) * 4 + c];
        out[(size_t)b * N * 4 + i] = v;
    }
    if (threadIdx.x == 0) out_count[b] = n + g;
}

extern "C" int a3d_append_gt_boxes(const float *props, const int *count, const float *gt, const int *gt_count, float *out, int *out_count,
                                   int B, int R, int Gmax, void *stream) {
    if (!props || !count || !gt || !gt_count || !out || !out_count || B <= 0 || R <= 0 || Gmax <= 0) return A3D_ERR_ARG;
    a3d_begin();
    hipLaunchKernelGGL(append_gt_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, props, count, gt, gt_count, out, out_count, R, Gmax);
    return a3d_check_launch();
}
