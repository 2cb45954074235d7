// Proposal / detection selection kernels (integer + fp32 index work, latency-bound):
//   rpn_select      per (image, level): radix-select the top-k objectness logits, sort them, decode
//                   the anchor deltas, clip, flag invalid boxes           
//   box_candidates  per (image, class): softmax, per-class delta decode, clip, score threshold, sort
//                                                                            
//   group_nms       per group (<=1024 score-sorted boxes): IoU bitmask in LDS + wavefront scan
//                                                                            
//   merge_topk      per image: merge the kept boxes of its groups by (score desc, position asc),
//                   keep the first K
// Built with -ffp-contract=off: box arithmetic must round exactly like the separate mul/add/div
// sequence of the reference operators so that keep-masks are bit-exact on identical inputs.
#include "a3d_common.h"
#include "../../include/a3d.h"

#define GROUP_CAP 1024

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t ordered_key(float f) {
    f = f + 0.0f;  // -0 -> +0 so that equal floats give equal keys
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// Descending bitonic sort of n (power of two) 64-bit keys in LDS by the whole workgroup.
__device__ void bitonic_sort_desc(u64 *k, int n) {
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (n >> 1); t += blockDim.x) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const u64 a = k[lo], b = k[hi];
                if ((a < b) == desc) {
                    k[lo] = b;
                    k[hi] = a;
                }
            }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ bool finite4(float a, float b, float c, float d) {
    return isfinite(a) && isfinite(b) && isfinite(c) && isfinite(d);
}

// Box2BoxTransform.apply_deltas + Boxes.clip, op order of SURVEY.md A.5.
__device__ __forceinline__ void decode_clip(const float ax1, const float ay1, const float ax2, const float ay2, float dx,
                                            float dy, float dw, float dh, float wx, float wy, float ww, float wh,
                                            float clampv, float img_w, float img_h, float out[4], bool &finite) {
    const float w = ax2 - ax1, h = ay2 - ay1;
    const float cx = ax1 + 0.5f * w, cy = ay1 + 0.5f * h;
    dx = dx / wx;
    dy = dy / wy;
    dw = dw / ww;
    dh = dh / wh;
    dw = fminf(dw, clampv);
    dh = fminf(dh, clampv);
    const float pcx = dx * w + cx, pcy = dy * h + cy;
    const float pw = expf(dw) * w, ph = expf(dh) * h;
    float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
    finite = finite4(x1, y1, x2, y2);
    out[0] = fminf(fmaxf(x1, 0.f), img_w);
    out[1] = fminf(fmaxf(y1, 0.f), img_h);
    out[2] = fminf(fmaxf(x2, 0.f), img_w);
    out[3] = fminf(fmaxf(y2, 0.f), img_h);
}

// ------------------------------------------------------------------------------------------------
// rpn_select: grid (L, B), block 1024.
// head: [B, Hf, Wf, CH] with channels [0,A) = objectness, [A, 5A) = deltas (a*4 + coord).
// ------------------------------------------------------------------------------------------------
struct RpnLevel {
    const float *head;
    int Hf, Wf, stride;
    float base[3][4];  // cell anchors (A = 3)
};
struct RpnSelectArgs {
    RpnLevel lv[5];
    int L, A, CH, pre_topk;
    float wx, wy, ww, wh, clampv, img_w, img_h, min_size;
    float *g_boxes;    // [B*L][GROUP_CAP][4]
    float *g_scores;   // [B*L][GROUP_CAP]
    int *g_pos;        // [B*L][GROUP_CAP]
    int *g_valid;      // [B*L][GROUP_CAP]
    int *g_n;          // [B*L]
};

template <int CAP>
__global__ __launch_bounds__(1024) void rpn_select_kernel(const RpnSelectArgs a) {
    constexpr int CB = (CAP == 1024) ? 10 : 11;  // bits of a slot index inside a group
    __shared__ u64 sel[CAP];
    __shared__ unsigned int hist[256];
    __shared__ u64 s_prefix;
    __shared__ int s_krem, s_cnt, s_done;
    const int l = blockIdx.x, b = blockIdx.y, g = b * a.L + l;
    const RpnLevel &lv = a.lv[l];
    const int n = lv.Hf * lv.Wf * a.A;
    const int k = min(a.pre_topk, min(n, CAP));
    const float *head = lv.head + (size_t)b * lv.Hf * lv.Wf * a.CH;
    int nb = 8;
    while ((1 << nb) < n) nb += 8;
    const u64 idx_mask = ((u64)1 << nb) - 1;
    const int total_bits = 32 + nb;
    auto composite = [&](int i) -> u64 {
        const int pix = i / a.A, an = i - pix * a.A;
        const float s = head[(size_t)pix * a.CH + an];
        return ((u64)ordered_key(s) << nb) | (idx_mask - (u64)i);
    };
    if (threadIdx.x == 0) {
        s_prefix = 0;
        s_krem = k;
        s_cnt = 0;
        s_done = 0;
    }
    // Round 6: the walks over the level's logits issue their loads EIGHT at a time.  Each of the 6-7 radix passes and the final pick reads one
    // 64-byte-strided logit per anchor from L2; with the load, its test and an LDS atomic in one loop body the compiler issued load, wait,
    // atomic per anchor -- 56 dependent L2 round trips per thread and pass on the finest level (57 600 anchors), ~400 per launch: the
    // latency of this kernel (0.17-0.30 ms), of every single-frame pass and of the training step's proposal stage.  Same keys, same
    // passes, same selection: bit for bit.
    constexpr int UB = 8;
    auto walk = [&](auto &&visit) {
        for (int i0 = threadIdx.x; i0 < n; i0 += blockDim.x * UB) {
            u64 c[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = i0 + u * (int)blockDim.x;
                c[u] = composite(min(i, n - 1));
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
                if (i0 + u * (int)blockDim.x < n) visit(c[u]);
        }
    };
    // radix select (8 bits per pass, most significant first) of the k-th largest composite key
    for (int shift = total_bits - 8; shift >= 0; shift -= 8) {
        if (threadIdx.x < 256) hist[threadIdx.x] = 0;
        __syncthreads();
        const u64 prefix = s_prefix;
        walk([&](const u64 c) {
            if ((c >> (shift + 8)) == prefix) atomicAdd(&hist[(unsigned)(c >> shift) & 255u], 1u);
        });
        __syncthreads();
        // The bin that holds the k-th key: the first bin, from 255 downwards, at which the running count reaches what is still wanted (bin 0
        // if none does).  Round 6: by wave 0 -- lane l owns bins 4 l .. 4 l + 3, a shuffle scan gives it the count above its bins, the
        // highest lane whose bins cross the mark wins -- instead of one thread walking up to 255 LDS reads one after the other in each of the
        // 6-7 passes (~10 us a pass).  (Measured and dropped in the same round: ballot-aggregated histogram atomics, 171 -> 262 us.)
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            const unsigned krem = (unsigned)s_krem;
            unsigned h[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) h[q] = hist[4 * lane + q];
            const unsigned tot = h[0] + h[1] + h[2] + h[3];
            unsigned above = tot;  // inclusive suffix sum over lanes l' >= l ...
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned v = __shfl_down(above, off, 64);
                if (lane + off < 64) above += v;
            }
            above -= tot;  // ... made exclusive: keys in bins above this lane's
            int fbin = -1;
            unsigned fcum = 0, cum = above;
#pragma unroll
            for (int q = 3; q >= 0; --q) {
                const int bin = 4 * lane + q;
                if (fbin < 0 && (bin == 0 || cum + h[q] >= krem)) {
                    fbin = bin;
                    fcum = cum;
                }
                cum += h[q];
            }
            const u64 found = __ballot(fbin >= 0);  // (lane 0 always finds: bin 0 is the fall-through)
            const int win = 63 - __builtin_clzll(found);
            const int bin = __builtin_amdgcn_readlane(fbin, win);
            const unsigned cumw = (unsigned)__builtin_amdgcn_readlane((int)fcum, win);
            if (lane == 0) {
                // (when the chosen bin and the bins above it hold EXACTLY what is still wanted, every key of that bin is selected: the bin's
                // lower bound is the threshold and the remaining passes -- typically the 16 index bits behind a unique score -- add nothing)
                const bool exact = cumw + hist[bin] == krem;
                s_krem = (int)(krem - cumw);
                s_prefix = exact ? (((prefix << 8) | (u64)bin) << shift) : ((prefix << 8) | (u64)bin);
                s_done = exact ? 1 : 0;
            }
        }
        __syncthreads();
        if (s_done) break;
    }
    const u64 T = s_prefix;
    for (int i = threadIdx.x; i < CAP; i += blockDim.x) sel[i] = 0;
    __syncthreads();
    walk([&](const u64 c) {
        if (c >= T) {
            const int slot = atomicAdd(&s_cnt, 1);
            if (slot < CAP) sel[slot] = c;
        }
    });
    bitonic_sort_desc(sel, CAP);
    for (int r = threadIdx.x; r < CAP; r += blockDim.x) {
        float box[4] = {0.f, 0.f, 0.f, 0.f};
        float score = 0.f;
        int valid = 0;
        if (r < k) {
            const u64 c = sel[r];
            const int i = (int)(idx_mask - (c & idx_mask));
            score = key_to_float((uint32_t)(c >> nb));
            const int pix = i / a.A, an = i - pix * a.A;
            const int y = pix / lv.Wf, x = pix - y * lv.Wf;
            const float sx = (float)(x * lv.stride), sy = (float)(y * lv.stride);
            const float *dl = head + (size_t)pix * a.CH + a.A + an * 4;
            bool fin;
            decode_clip(sx + lv.base[an][0], sy + lv.base[an][1], sx + lv.base[an][2], sy + lv.base[an][3], dl[0], dl[1],
                        dl[2], dl[3], a.wx, a.wy, a.ww, a.wh, a.clampv, a.img_w, a.img_h, box, fin);
            const bool nonempty = (box[2] - box[0]) > a.min_size && (box[3] - box[1]) > a.min_size;
            valid = (fin && isfinite(score) && nonempty) ? 1 : 0;
        }
        const size_t o = (size_t)g * CAP + r;
        *reinterpret_cast<f32x4 *>(a.g_boxes + o * 4) = f32x4{box[0], box[1], box[2], box[3]};
        a.g_scores[o] = score;
        a.g_pos[o] = (l << CB) | r;
        a.g_valid[o] = valid;
    }
    if (threadIdx.x == 0) a.g_n[g] = k;
}

// ------------------------------------------------------------------------------------------------
// box_candidates: grid (C, B), block 1024.  pred: [B*R, CH] with channels [0, C] = class logits
// (last = background), [C+1, C+1+4C) = per-class deltas.  proposals [B][R][4], counts [B].
// ------------------------------------------------------------------------------------------------
struct BoxCandArgs {
    const float *pred;
    const float *prop_boxes;
    const int *prop_count;
    int R, C, CH;
    float wx, wy, ww, wh, clampv, img_w, img_h, score_thresh;
    float *g_boxes, *g_scores;
    int *g_pos, *g_valid, *g_n;
};

__global__ __launch_bounds__(1024) void box_candidates_kernel(const BoxCandArgs a) {
    __shared__ u64 sel[GROUP_CAP];
    __shared__ int s_cnt;
    const int c = blockIdx.x, b = blockIdx.y, g = b * a.C + c;
    const int nprop = min(a.prop_count[b], min(a.R, GROUP_CAP));
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    for (int r = threadIdx.x; r < GROUP_CAP; r += blockDim.x) {
        u64 key = 0;
        if (r < nprop) {
            const float *p = a.pred + ((size_t)b * a.R + r) * a.CH;
            // softmax over C+1 logits (max-subtracted, as aten's softmax)
            float mx = p[0];
            for (int j = 1; j <= a.C; ++j) mx = fmaxf(mx, p[j]);
            float den = 0.f;
            for (int j = 0; j <= a.C; ++j) den += expf(p[j] - mx);
            const float prob = expf(p[c] - mx) / den;
            bool row_ok = isfinite(den);  // valid_mask: every decoded box and score of the ROW finite
            for (int j = 0; j <= a.C; ++j) row_ok = row_ok && isfinite(expf(p[j] - mx) / den);
            const float *pbx = a.prop_boxes + ((size_t)b * a.R + r) * 4;
            for (int j = 0; j < a.C; ++j) {
                const float *dl = p + a.C + 1 + j * 4;
                float tmp[4];
                bool fin;
                decode_clip(pbx[0], pbx[1], pbx[2], pbx[3], dl[0], dl[1], dl[2], dl[3], a.wx, a.wy, a.ww, a.wh, a.clampv,
                            a.img_w, a.img_h, tmp, fin);
                row_ok = row_ok && fin;
            }
            if (row_ok && prob > a.score_thresh) {
                key = ((u64)ordered_key(prob) << 32) | (u64)(0xffffffffu - (uint32_t)r);
                atomicAdd(&s_cnt, 1);
            }
        }
        sel[r] = key;
    }
    bitonic_sort_desc(sel, GROUP_CAP);
    const int n = s_cnt;
    for (int r = threadIdx.x; r < GROUP_CAP; r += blockDim.x) {
        float box[4] = {0.f, 0.f, 0.f, 0.f};
        float score = 0.f;
        int valid = 0, pos = 0;
        if (r < n) {
            const u64 key = sel[r];
            const int row = (int)(0xffffffffu - (uint32_t)(key & 0xffffffffu));
            score = key_to_float((uint32_t)(key >> 32));
            const float *p = a.pred + ((size_t)b * a.R + row) * a.CH + a.C + 1 + c * 4;
            const float *pb = a.prop_boxes + ((size_t)b * a.R + row) * 4;
            bool fin;
            decode_clip(pb[0], pb[1], pb[2], pb[3], p[0], p[1], p[2], p[3], a.wx, a.wy, a.ww, a.wh, a.clampv, a.img_w,
                        a.img_h, box, fin);
            valid = fin ? 1 : 0;
            pos = row * a.C + c;
        }
        const size_t o = (size_t)g * GROUP_CAP + r;
        *reinterpret_cast<f32x4 *>(a.g_boxes + o * 4) = f32x4{box[0], box[1], box[2], box[3]};
        a.g_scores[o] = score;
        a.g_pos[o] = pos;
        a.g_valid[o] = valid;
    }
    if (threadIdx.x == 0) a.g_n[g] = n;
}

// ------------------------------------------------------------------------------------------------
// group_nms: grid (G), block 1024.  Boxes of a group are already score-descending.
// LDS: 1024 x 16 suppression words (128 KiB) + boxes (16 KiB).
// ------------------------------------------------------------------------------------------------
// CAP = 1024: the suppression words live in LDS (128 KiB).  CAP = 2048 (training's PRE_NMS_TOPK 2000): 2048 x 32 words =
// 512 KiB per group do not fit, they go to a global scratch area (`gmask`, L2-resident) -- same algorithm, same results.
// GM: the suppression words live in global memory (`gmask`, filled by nms_mask_kernel<CAP>): always at CAP 2048; at CAP 1024 when the launch
// has fewer groups than the chip has CUs (round 6) -- one workgroup per group computing its 512 K IoUs alone left a single frame's five RPN
// groups at 0.19 ms on five CUs; spread over the chip the words take ~10 us and the group's own workgroup only scans.  Same words, same scan.
template <int CAP, bool GM = (CAP > 1024)>
__global__ __launch_bounds__(1024) void group_nms_kernel(const float *__restrict__ g_boxes, const int *__restrict__ g_valid,
                                                         const int *__restrict__ g_n, int *__restrict__ g_keep,
                                                         float thr, u64 *__restrict__ gmask, const int have_mask, const int inner) {
    constexpr int W = CAP / 64;  // suppression words per row
    __shared__ u64 smask[GM ? 1 : CAP * (CAP / 64)];
    __shared__ f32x4 sb[CAP];
    __shared__ unsigned char sv[CAP];
    // One workgroup per CU (144 KiB of LDS), so 64 frames x 5 levels = 320 groups take two rounds of the chip.  Groups are g = image *
    // inner + level; walking them level-major puts the 1000-box levels in the first round and leaves the coarsest level's short
    // groups (240 anchors) for the second one instead of a random fifth of everything.
    const int nimg = gridDim.x / inner;
    const int g = (blockIdx.x % nimg) * inner + blockIdx.x / nimg;
    u64 *mask = GM ? gmask + (size_t)g * CAP * W : smask;
    const int n = min(g_n[g], CAP);
    for (int i = threadIdx.x; i < CAP; i += blockDim.x) {
        sb[i] = *reinterpret_cast<const f32x4 *>(g_boxes + ((size_t)g * CAP + i) * 4);
        sv[i] = (i < n) ? (unsigned char)g_valid[(size_t)g * CAP + i] : 0;
    }
    __syncthreads();
    const int nwords = (n + 63) >> 6;
    for (int wd = threadIdx.x; wd < (have_mask ? 0 : n * W); wd += blockDim.x) {  // (have_mask: nms_mask_kernel already filled it)
        const int i = wd / W, w = wd % W;
        u64 bits = 0;
        if (w < nwords && (w << 6) + 63 > i && sv[i]) {
            const f32x4 bi = sb[i];
            const float ai = (bi[2] - bi[0]) * (bi[3] - bi[1]);
            const int j0 = w << 6;
            for (int t = 0; t < 64; ++t) {
                const int j = j0 + t;
                if (j <= i || j >= n || !sv[j]) continue;
                const f32x4 bj = sb[j];
                const float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]);
                const float xx2 = fminf(bi[2], bj[2]), yy2 = fminf(bi[3], bj[3]);
                const float iw = fmaxf(0.f, xx2 - xx1), ih = fmaxf(0.f, yy2 - yy1);
                const float inter = iw * ih;
                const float aj = (bj[2] - bj[0]) * (bj[3] - bj[1]);
                const float ovr = inter / (ai + aj - inter);
                if (ovr > thr) bits |= (u64)1 << t;
            }
        }
        mask[wd] = bits;
    }
    __syncthreads();
    // Wavefront scan by wave 0: lane l < W carries removed-word l.  Per 64-box block: resolve the
    // block with its diagonal words in registers (v_readlane chain), then OR the kept rows' words
    // into the carried state (4 row subsets x 16 words over the 64 lanes, xor-shuffle reduce).
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        u64 R = 0;
        for (int wi = 0; wi < nwords; ++wi) {
            const int row = (wi << 6) + lane;
            const u64 diag = (row < n) ? mask[row * W + wi] : 0;
            const u64 validbits = __ballot(row < n && sv[row]);
            // The 64-step chain of a block runs on the SCALAR unit (round 6): removed / kept words are wave-uniform, a row's diagonal word
            // comes out of its lane by v_readlane with a constant lane.  (With __shfl the compiler kept the chain in vector registers:
            // two ds_bpermute + a wait per step, ~100 cycles x 64 steps x 16 blocks = 45 us per 1000-box group on ONE wave -- the
            // latency of this launch, of every single-frame pass and of the training step's proposal stage.)  Same bit logic: same keep masks.
            const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
            const unsigned rlo = (unsigned)R, rhi = (unsigned)(R >> 32);
            u64 cur = ((u64)(unsigned)__builtin_amdgcn_readlane((int)rhi, wi) << 32) | (unsigned)__builtin_amdgcn_readlane((int)rlo, wi);  // removed bits of this block so far
            u64 kept = 0;
            // (one step per KEPT box of the block, not per box: the lowest box still alive is kept, its row's diagonal word -- bits of the
            // LATER boxes it suppresses -- joins the removed set, and everything removed drops out of the candidates)
            u64 avail = validbits & ~cur;
            while (avail) {
                const int bq = __builtin_ctzll(avail);
                const u64 d = ((u64)(unsigned)__builtin_amdgcn_readlane((int)dhi, bq) << 32) | (unsigned)__builtin_amdgcn_readlane((int)dlo, bq);
                kept |= (u64)1 << bq;
                cur |= d;
                avail &= ~cur & ~((u64)1 << bq);
            }
            if (row < CAP) g_keep[(size_t)g * CAP + row] = (int)((kept >> lane) & 1);
            const int w = lane % W, sub = lane / W;  // 64 / W row subsets of W rows each
            u64 acc = 0;
            if constexpr (!GM) {
#pragma unroll
                for (int t = 0; t < W; ++t) {  // (branch-free: sixteen independent LDS reads in flight instead of a read behind every test)
                    const int rr = sub * W + t;
                    const u64 m = mask[((wi << 6) + rr) * W + w];
                    acc |= ((kept >> rr) & 1) ? m : 0;
                }
            } else {
                // the words live in global memory here: the kept rows of this lane's subset in batches of eight requests in flight (one
                // at a time behind a branch, each waited out a memory round trip: ~12 us per 64-box block, 0.38 ms per launch)
                u64 todo = (kept >> (sub * W)) & (W == 64 ? ~(u64)0 : (((u64)1 << W) - 1));
                while (__any(todo != 0)) {
                    u64 m8[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const bool on = todo != 0;
                        const int t = on ? __ffsll((long long)todo) - 1 : 0;
                        todo &= todo - 1;
                        m8[q] = on ? mask[((wi << 6) + sub * W + t) * W + w] : 0;
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc |= m8[q];
                }
            }
#pragma unroll
            for (int off = W; off < 64; off <<= 1) acc |= __shfl_xor(acc, off, 64);
            R |= acc;  // lanes >= W carry copies; only lanes < W are read via __shfl(R, wi)
        }
        for (int row = (nwords << 6) + lane; row < CAP; row += 64) g_keep[(size_t)g * CAP + row] = 0;
    }
}

// Suppression words of a 2048-slot group computed by 16 workgroups instead of one (the single-workgroup form spends
// 2 ms per launch on the 2M IoUs of a full group while 246 CUs idle): grid (G, 16), block y owns rows [128y, 128y+128).
template <int CAP>
__global__ __launch_bounds__(256) void nms_mask_kernel(const float *__restrict__ g_boxes, const int *__restrict__ g_valid,
                                                       const int *__restrict__ g_n, float thr, u64 *__restrict__ gmask) {
    constexpr int W = CAP / 64;
    const int ROWS = CAP / (int)gridDim.y;  // (rows per workgroup: 128 at 16 workgroups per 2048-slot group, 32 at 64 -- and at 32 per 1024-slot group)
    __shared__ f32x4 sb[CAP];
    __shared__ unsigned char sv[CAP];
    const int g = blockIdx.x;
    const int n = min(g_n[g], CAP);
    const int r0 = blockIdx.y * ROWS;
    if (r0 >= n) return;
    for (int i = threadIdx.x; i < CAP; i += blockDim.x) {
        sb[i] = *reinterpret_cast<const f32x4 *>(g_boxes + ((size_t)g * CAP + i) * 4);
        sv[i] = (i < n) ? (unsigned char)g_valid[(size_t)g * CAP + i] : 0;
    }
    __syncthreads();
    const int nwords = (n + 63) >> 6;
    const int r1 = min(r0 + ROWS, n);
    u64 *mask = gmask + (size_t)g * CAP * W;
    // Round 6: a WAVE forms one 64-box word per step -- lane t owns box j = 64 w + t of the word (in registers across the rows), the row's box
    // is an LDS broadcast, the word is the ballot of the 64 comparisons.  (Before, a THREAD looped over the 64 boxes of its word: half
    // the lanes of a wave sat in the lower triangle, the rest diverged on per-box tests, and every pair re-read box j from LDS -- 0.33 ms
    // chip-wide for the 320 groups of a 64-frame step.)  Same pairs, same expression per pair: the same words.
    // words that hold no pair (at or below the diagonal, past the last box, rows of dead boxes): zero
    for (int wd = threadIdx.x; wd < ROWS * W; wd += blockDim.x) {
        const int i = r0 + wd / W, w = wd % W;
        if (i < n && !(w < nwords && (w << 6) + 63 > i && sv[i])) mask[(size_t)i * W + w] = 0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int w = r0 >> 6; w < nwords; ++w) {  // (a word below r0 >> 6 lies under the diagonal for every row of this block)
        const int j = (w << 6) + lane;
        const f32x4 bj = sb[min(j, CAP - 1)];
        const bool vj = j < n && sv[min(j, CAP - 1)];
        const float aj = (bj[2] - bj[0]) * (bj[3] - bj[1]);
        for (int i = r0 + wave; i < r1; i += nw) {
            if (!((w << 6) + 63 > i) || !sv[i]) continue;  // (wave-uniform)
            const f32x4 bi = sb[i];
            const float ai = (bi[2] - bi[0]) * (bi[3] - bi[1]);
            const float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]);
            const float xx2 = fminf(bi[2], bj[2]), yy2 = fminf(bi[3], bj[3]);
            const float iw = fmaxf(0.f, xx2 - xx1), ih = fmaxf(0.f, yy2 - yy1);
            const float inter = iw * ih;
            const float ovr = inter / (ai + aj - inter);
            const u64 bits = __ballot(vj && j > i && ovr > thr);
            if (lane == 0) mask[(size_t)i * W + w] = bits;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// merge_topk: grid (B), block 1024.  groups of image b: [b*NG, (b+1)*NG).
// ------------------------------------------------------------------------------------------------
#define MERGE_CAP 8192
template <int CAP, int MCAP>
__global__ __launch_bounds__(1024) void merge_topk_kernel(const float *__restrict__ g_boxes,
                                                          const float *__restrict__ g_scores,
                                                          const int *__restrict__ g_pos, const int *__restrict__ g_keep,
                                                          const int *__restrict__ g_n, int NG, int K,
                                                          float *__restrict__ out_boxes, float *__restrict__ out_scores,
                                                          int *__restrict__ out_cat, int *__restrict__ out_pos,
                                                          int *__restrict__ out_count) {
    constexpr int CB = (CAP == 1024) ? 10 : 11;
    __shared__ u64 keys[MCAP];
    __shared__ unsigned short ref[MCAP == 8192 ? 8192 : 5 * CAP];  // indexed by position (group << CB | slot)
    __shared__ int s_cnt;
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < MCAP; i += blockDim.x) keys[i] = 0;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    for (int gl = 0; gl < NG; ++gl) {
        const int g = b * NG + gl;
        const int n = min(g_n[g], CAP);
        for (int r = threadIdx.x; r < n; r += blockDim.x) {
            const size_t o = (size_t)g * CAP + r;
            if (g_keep[o]) {
                const int pos = g_pos[o];
                const int slot = atomicAdd(&s_cnt, 1);
                keys[slot] = ((u64)ordered_key(g_scores[o]) << 32) | (u64)(0xffffffffu - (uint32_t)pos);
                ref[pos] = (unsigned short)((gl << CB) | r);
            }
        }
    }
    __syncthreads();
    const int total = s_cnt;
    int np2 = 2;
    while (np2 < total) np2 <<= 1;
    bitonic_sort_desc(keys, np2);
    const int cnt = min(total, K);
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        f32x4 box = {0.f, 0.f, 0.f, 0.f};
        float sc = 0.f;
        int cat = -1, pos = -1;
        if (i < cnt) {
            pos = (int)(0xffffffffu - (uint32_t)(keys[i] & 0xffffffffu));
            const int rf = ref[pos];
            cat = rf >> CB;
            const size_t o = (size_t)(b * NG + cat) * CAP + (rf & (CAP - 1));
            box = *reinterpret_cast<const f32x4 *>(g_boxes + o * 4);
            sc = g_scores[o];
        }
        *reinterpret_cast<f32x4 *>(out_boxes + ((size_t)b * K + i) * 4) = box;
        out_scores[(size_t)b * K + i] = sc;
        out_cat[(size_t)b * K + i] = cat;
        out_pos[(size_t)b * K + i] = pos;
    }
    if (threadIdx.x == 0) out_count[b] = cnt;
}

// ================================= C ABI ==========================================================
// Launches of up to this many groups compute their suppression words chip-wide into global scratch (nms_mask_kernel<1024>) and scan from there
// (group_nms_kernel<1024, true>); larger ones keep them in the LDS of a group's own workgroup.  The words of a full group are 512 K IoUs --
// ~0.2 ms of ONE CU: a single frame's five RPN groups ran on five CUs (0.19 ms a launch, now 0.05), and the 320 groups of a 64-frame step
// took two rounds of the 256 CUs (0.47 ms, the second round a quarter full).  1024 groups x 128 KiB = 128 MiB of scratch at most.
#define A3D_NMS_SPLIT_GROUPS 1024
extern "C" size_t a3d_group_buffers_bytes(int n_groups) {
    // boxes(16) + scores(4) + pos(4) + valid(4) + keep(4) per slot, + n per group (+ the global suppression words of a small launch)
    size_t b = (size_t)n_groups * GROUP_CAP * 32 + (size_t)n_groups * 4 + 256;
    if (n_groups <= A3D_NMS_SPLIT_GROUPS) b += 8 + (size_t)n_groups * GROUP_CAP * (GROUP_CAP / 64) * sizeof(u64);
    return b;
}


// Workspace of a3d_rpn_proposals: group buffers with 1024 slots (pre_topk <= 1024) or 2048 slots plus the global
// suppression words of the 2048-candidate NMS (training's PRE_NMS_TOPK_TRAIN 2000).
extern "C" size_t a3d_rpn_workspace_bytes(int B, int L, int pre_topk) {
    const size_t G = (size_t)B * L;
    if (pre_topk <= GROUP_CAP) return a3d_group_buffers_bytes((int)G);
    const size_t cap = 2048;
    return G * cap * 32 + G * 4 + 256 + G * cap * (cap / 64) * sizeof(u64);
}

struct GroupBufs {
    float *boxes, *scores;
    int *pos, *valid, *keep, *n;
    u64 *mask;
};
static GroupBufs carve(void *ws, int G, int cap = GROUP_CAP) {
    GroupBufs gb;
    char *p = (char *)ws;
    gb.boxes = (float *)p;
    p += (size_t)G * cap * 16;
    gb.scores = (float *)p;
    p += (size_t)G * cap * 4;
    gb.pos = (int *)p;
    p += (size_t)G * cap * 4;
    gb.valid = (int *)p;
    p += (size_t)G * cap * 4;
    gb.keep = (int *)p;
    p += (size_t)G * cap * 4;
    gb.n = (int *)p;
    p += (size_t)G * 4 + 256;
    gb.mask = (u64 *)(((uintptr_t)p + 7) & ~(uintptr_t)7);  // (the 2048-slot form, and 1024-slot launches of <= A3D_NMS_SPLIT_GROUPS groups, own bytes here)
    return gb;
}
// the 1024-slot NMS of G groups whose buffers `gb` were carved from a3d_group_buffers_bytes(G)
static void launch_group_nms_1024(const GroupBufs &gb, int G, float thr, int inner, hipStream_t s) {
    if (G <= A3D_NMS_SPLIT_GROUPS) {  // few groups: the words by the whole chip (32-row blocks), then one scanning workgroup per group
        hipLaunchKernelGGL(nms_mask_kernel<1024>, dim3(G, 32), dim3(256), 0, s, gb.boxes, gb.valid, gb.n, thr, gb.mask);
        hipLaunchKernelGGL((group_nms_kernel<1024, true>), dim3(G), dim3(1024), 0, s, gb.boxes, gb.valid, gb.n, gb.keep, thr, gb.mask, 1, inner);
    } else {
        hipLaunchKernelGGL((group_nms_kernel<1024, false>), dim3(G), dim3(1024), 0, s, gb.boxes, gb.valid, gb.n, gb.keep, thr, nullptr, 0, inner);
    }
}

extern "C" int a3d_rpn_proposals(const a3d_rpn_desc *d, void *stream) {
    if (!d || d->L < 1 || d->L > 5 || d->A != 3 || !d->workspace || !d->out_boxes || !d->out_scores || !d->out_count)
        return A3D_ERR_ARG;
    if (d->pre_topk > 2048 || d->post_topk <= 0 || d->L * GROUP_CAP > MERGE_CAP) return A3D_ERR_UNSUPPORTED;
    if (d->CH < 5 * d->A) return A3D_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int G = d->B * d->L;
    const int cap = d->pre_topk <= GROUP_CAP ? GROUP_CAP : 2048;  // workspace: a3d_rpn_workspace_bytes()
    GroupBufs gb = carve(d->workspace, G, cap);
    RpnSelectArgs a;
    for (int l = 0; l < d->L; ++l) {
        a.lv[l].head = d->head[l];
        a.lv[l].Hf = d->Hf[l];
        a.lv[l].Wf = d->Wf[l];
        a.lv[l].stride = d->stride[l];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) a.lv[l].base[i][j] = d->cell_anchors[l][i][j];
        if ((size_t)d->Hf[l] * d->Wf[l] * d->A >= ((size_t)1 << 24)) return A3D_ERR_UNSUPPORTED;
    }
    a.L = d->L;
    a.A = d->A;
    a.CH = d->CH;
    a.pre_topk = d->pre_topk;
    a.wx = d->weights[0];
    a.wy = d->weights[1];
    a.ww = d->weights[2];
    a.wh = d->weights[3];
    a.clampv = d->scale_clamp;
    a.img_w = (float)d->img_w;
    a.img_h = (float)d->img_h;
    a.min_size = d->min_size;
    a.g_boxes = gb.boxes;
    a.g_scores = gb.scores;
    a.g_pos = gb.pos;
    a.g_valid = gb.valid;
    a.g_n = gb.n;
    a3d_begin();
    if (cap == GROUP_CAP) {
        hipLaunchKernelGGL(rpn_select_kernel<1024>, dim3(d->L, d->B), dim3(1024), 0, s, a);
        launch_group_nms_1024(gb, G, d->nms_thresh, d->L, s);
        hipLaunchKernelGGL((merge_topk_kernel<1024, MERGE_CAP>), dim3(d->B), dim3(1024), 0, s, gb.boxes, gb.scores, gb.pos, gb.keep, gb.n,
                           d->L, d->post_topk, d->out_boxes, d->out_scores, d->out_level, d->out_pos, d->out_count);
    } else {
        hipLaunchKernelGGL(rpn_select_kernel<2048>, dim3(d->L, d->B), dim3(1024), 0, s, a);
        // (the suppression words of a group by 16 workgroups -- or 64 where the batch is a few images: the first row block of a group carries
        // most of its upper-triangular work, and 10 groups x 16 blocks left the launch at the length of that one block: 0.23 ms)
        hipLaunchKernelGGL(nms_mask_kernel<2048>, dim3(G, G <= 40 ? 64 : 16), dim3(256), 0, s, gb.boxes, gb.valid, gb.n, d->nms_thresh, gb.mask);
        hipLaunchKernelGGL(group_nms_kernel<2048>, dim3(G), dim3(1024), 0, s, gb.boxes, gb.valid, gb.n, gb.keep, d->nms_thresh, gb.mask, 1, d->L);
        hipLaunchKernelGGL((merge_topk_kernel<2048, 16384>), dim3(d->B), dim3(1024), 0, s, gb.boxes, gb.scores, gb.pos, gb.keep, gb.n,
                           d->L, d->post_topk, d->out_boxes, d->out_scores, d->out_level, d->out_pos, d->out_count);
    }
    return a3d_check_launch();
}

extern "C" int a3d_box_detections(const a3d_boxdet_desc *d, void *stream) {
    if (!d || !d->pred || !d->prop_boxes || !d->prop_count || !d->workspace || !d->out_boxes || !d->out_scores ||
        !d->out_classes || !d->out_count)
        return A3D_ERR_ARG;
    if (d->R > GROUP_CAP || d->C < 1 || d->C * GROUP_CAP > MERGE_CAP || d->R * d->C > MERGE_CAP) return A3D_ERR_UNSUPPORTED;
    if (d->CH < 5 * d->C + 1) return A3D_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int G = d->B * d->C;
    GroupBufs gb = carve(d->workspace, G);
    BoxCandArgs a;
    a.pred = d->pred;
    a.prop_boxes = d->prop_boxes;
    a.prop_count = d->prop_count;
    a.R = d->R;
    a.C = d->C;
    a.CH = d->CH;
    a.wx = d->weights[0];
    a.wy = d->weights[1];
    a.ww = d->weights[2];
    a.wh = d->weights[3];
    a.clampv = d->scale_clamp;
    a.img_w = (float)d->img_w;
    a.img_h = (float)d->img_h;
    a.score_thresh = d->score_thresh;
    a.g_boxes = gb.boxes;
    a.g_scores = gb.scores;
    a.g_pos = gb.pos;
    a.g_valid = gb.valid;
    a.g_n = gb.n;
    a3d_begin();
    hipLaunchKernelGGL(box_candidates_kernel, dim3(d->C, d->B), dim3(1024), 0, s, a);
    launch_group_nms_1024(gb, G, d->nms_thresh, d->C, s);
    hipLaunchKernelGGL((merge_topk_kernel<1024, MERGE_CAP>), dim3(d->B), dim3(1024), 0, s, gb.boxes, gb.scores, gb.pos, gb.keep, gb.n, d->C,
                       d->topk, d->out_boxes, d->out_scores, d->out_classes, d->out_pos, d->out_count);
    return a3d_check_launch();
}

// Stand-alone batched NMS over caller-sorted groups (unit-parity entry for the keep masks).
extern "C" int a3d_group_nms(const float *g_boxes, const int *g_valid, const int *g_n, int *g_keep, int n_groups,
                             float thresh, void *stream) {
    if (!g_boxes || !g_valid || !g_n || !g_keep || n_groups <= 0) return A3D_ERR_ARG;
    a3d_begin();
    hipLaunchKernelGGL((group_nms_kernel<1024, false>), dim3(n_groups), dim3(1024), 0, (hipStream_t)stream, g_boxes, g_valid, g_n,
                       g_keep, thresh, nullptr, 0, 1);
    return a3d_check_launch();
}
