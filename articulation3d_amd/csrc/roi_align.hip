// ROIPooler = FPN level assignment + ROIAlign over NHWC pyramids.
// Replaces detectron2 ROIPooler -> torchvision.ops.roi_align reached from
// pkg/modeling/roi_heads/roi_heads.py:185 (box, 7x7 aligned), :236 (mask, 14x14 ratio 2),
// :250 (plane) and :268 (axis) (14x14 adaptive).
//
// HBM/L2-bound gather.  NHWC makes every bilinear corner a contiguous C-vector: one wave owns one
// output bin, lane l owns channels 4l..4l+3 (C = 256 -> exactly one float4 per lane), so every corner
// read and the bin store are 1 KiB coalesced accesses.  One workgroup (4 waves) owns one ROI and
// walks its P*P bins; the ROI geometry is computed once per workgroup.
//
// Measured alternative (round 1, rejected): staging each ROI's cell window in LDS per 32-channel slice
// (one workgroup per ROI x slice, then bins from LDS) ran the 7x7 box pooler in 3.07 ms per 32 frames
// against 1.93 ms for this direct form: proposal windows are only ~100 cells, so the extra workgroups,
// barrier and 12-wave occupancy cost more than the duplicate corner reads that L2 already absorbs.
//
// Round-2 counters of the 7x7 box pooler at 64 x 1000 boxes (tools/pmc_roi.sh, gpurun_out -> DESIGN.md 5a): 27.8 M wave loads =
// 31.6 GB of L1 accesses, 12.2 GB requested from L2 (L1 hit rate 0.57), 4.1 GB missing L2 (hit rate 0.73), mean L2 read latency
// 345 cycles, TA busy 0.53, the L1 in "pending miss" stall 0.65 of the time; 21 waves per CU.  No unit is saturated: the
// one-load-at-a-time walk and the batched walk now run in the same time (2.59 ms before the spatial order, 2.39 with it), and
// giving each wave its own contiguous run of bins instead of every fourth bin was slower (2.52 ms: the four waves of a
// workgroup no longer share their neighbouring cells in L1).
//
// Round 6.  Measured and not taken: bins of more than NC = 9 cells (3 x 3 and 4 x 4 sampling lattices: most proposals of p3..p5) walked in
// groups of 9 loads issued together instead of one load at a time -- bit-identical, 2.44 ms against 2.39-2.44: the launch is not
// latency-serialised.  A bin reads ~20 cells of 1 KiB for 1 KiB of output: 64 000 x 49 bins x ~20 KiB = 31 GB through the L1 / texture
// path, which delivers ~50-80 GB/s per CU on gathers like this one = 1.6-2.4 ms whatever the schedule; what can be cut is the duplicate
// cell reads of neighbouring bins (980 cell loads per ROI against 441 distinct cells).  TAKEN: the rolling-window walk of the 7 x 7 box
// pooler (roi_align_fpn_kernel<4, false, true>, below): 675 loads per typical ROI, 2.04 ms against 2.43 (tools/roi_bench.py).
#include "conv_common.h"  // a3d_pow2_scale: the block exponent of the fp16x2 split (out_h2)
#ifndef A3D_ROI_NC
#define A3D_ROI_NC 9
#endif
#include <stdlib.h>
#include "../../include/a3d.h"

struct RoiArgs {
    const float *feat[4];
    int Hf[4], Wf[4];
    float scale[4];
    int L, C;
    const float *boxes;
    const int *count;
    const int *row_offset;
    int R, P, ratio, aligned;
    float *out;
    int *out_level;
    int serial;  // A/B + test hook (A3D_ROI_SERIAL=1): the one-load-at-a-time bin walk the batched form replaced
    int rolling; // round 6 (a3d_roialign_desc.serial == 2): the rolling-window walk below, where the ROI's geometry allows it
    const int *order;  // optional [B*R]: slot walked by workgroup (b, rank); see a3d_roialign_desc.order_ws
    int nblk;
    float *out_amax;            // optional [rows]: max |pooled[row]| over the finite pooled values (a3d_roialign_desc.out_amax)
    const float *level_amax[4]; // optional per level [B]: maxima of the pyramid level, for the window monitor below
    int *window_count;          // optional: number of live ROIs fainter than 2^-A3D_ROI_WINDOW_LOG2 of their level's maximum
    unsigned char *out_h2;      // H2 form: [rows][P*P*C/16][h | l][16] fp16 (a3d_roialign_desc.out_h2) instead of `out`
};
#define A3D_ROI_WINDOW_LOG2 16

// ---- spatial order of an image's boxes ---------------------------------------------------------------------------------------
// Proposals arrive score-sorted, i.e. in random spatial order, and all ~1000 boxes of an image are in flight at once across the
// 8 XCDs: every XCD's 4 MiB L2 then sees boxes all over a 26 MB pyramid and re-fetches each feature cell from the Infinity
// Cache / HBM ~10 times (measured: 17 GB fetched past L2 per 64-frame step against 1.7 GB of distinct cells -- the kernel ran
// at the memory-side rate, not at the L1 rate).  Sorting the boxes by (level, y, x) and giving each XCD whole images makes the
// ~128 workgroups an XCD runs side by side neighbours in the pyramid.  One workgroup per image: bitonic sort of <= 1024 keys in LDS.
__global__ __launch_bounds__(256) void roi_order_kernel(const float *boxes, const int *count, int *order, int R, int L) {
    __shared__ unsigned keys[1024];
    __shared__ int idx[1024];
    const int b = blockIdx.x;
    const int cnt = count ? min(count[b], R) : R;
    for (int r = threadIdx.x; r < 1024; r += 256) {
        unsigned k = 0xFFFFFFFFu;
        if (r < cnt) {
            const float *bx = boxes + ((size_t)b * R + r) * 4;
            const float size = sqrtf((bx[2] - bx[0]) * (bx[3] - bx[1]));
            float lvf = floorf(4.0f + log2f(size / 224.0f + 1e-8f));
            lvf = fminf(fmaxf(lvf, 2.0f), (float)(2 + L - 1));
            const int lv = (int)lvf - 2;
            // band of 64 px (2 cells at the coarsest level, 16 at the finest) in y, then x
            const int yb = min(max((int)((bx[1] + bx[3]) * (0.5f / 64.0f)), 0), 1023);
            const int xb = min(max((int)((bx[0] + bx[2]) * 0.5f), 0), 16383);
            k = ((unsigned)lv << 28) | ((unsigned)yb << 14) | (unsigned)xb;
        }
        keys[r] = k;
        idx[r] = r;
    }
    __syncthreads();
    for (int k = 2; k <= 1024; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < 1024; i += 256) {
                const int p = i ^ j;
                if (p > i) {
                    const bool up = (i & k) == 0;
                    const unsigned a0 = keys[i], a1 = keys[p];
                    const int i0 = idx[i], i1 = idx[p];
                    const bool gt = a0 > a1 || (a0 == a1 && i0 > i1);  // ties by slot: a total order, deterministic
                    if (gt == up) {
                        keys[i] = a1;
                        keys[p] = a0;
                        idx[i] = i1;
                        idx[p] = i0;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int r = threadIdx.x; r < R; r += 256) order[(size_t)b * R + r] = idx[r];  // ranks >= count hold dead slots (skipped by the walker)
}

// H2 (a3d_roialign_desc.out_h2; the 7x7 box pooler in the default arithmetic): the pooled row leaves the kernel ALREADY split into the
// two scaled fp16 planes the fp16x2 GEMM multiplies, in a3d_conv_desc.x_h2's layout, so that the box head's fc1 moves both operands
// global -> LDS by LDS-DMA.  The scale is the power of two of the ROI's OWN maximum, which is only known once every bin is pooled:
// the workgroup keeps its P*P*C pooled floats in LDS (49 KiB at 7 x 7 x 256: three workgroups per CU), reduces the maximum,
// and then splits and stores the row -- the same bytes as the fp32 row (2 + 2 per element), the bits fc1's loader computed from it.
// NW waves per workgroup: 7 for the 7x7 pooler (wave w pools column w of every bin row: the waves of a workgroup stay neighbours in
// the pyramid, and 49 bins are 7 rounds with no idle wave), 4 otherwise.
typedef _Float16 ra_h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 ra_h16x8 __attribute__((ext_vector_type(8)));
template <int NW, bool H2, bool ROLL = false>
__global__ __launch_bounds__(64 * NW, H2 ? 3 : 1) void roi_align_fpn_kernel(const RoiArgs a) {
    extern __shared__ __attribute__((aligned(16))) float pooled_lds[];  // H2: [P*P][C]
    int slot = blockIdx.x;
    if (a.order) {  // (image, rank) in XCD-contiguous order -> the slot at that rank of the image's spatial order
        const int logical = a3d_xcd_remap(blockIdx.x, a.nblk);
        slot = (logical / a.R) * a.R + a.order[logical];
    }
    const int b = slot / a.R, r = slot - b * a.R;
    const int cnt = a.count ? a.count[b] : a.R;
    if (r >= cnt) return;
    const int row = (a.row_offset ? a.row_offset[b] : b * a.R) + r;
    const float *bx = a.boxes + (size_t)slot * 4;
    const float bx1 = bx[0], by1 = bx[1], bx2 = bx[2], by2 = bx[3];
    // level = floor(4 + log2(sqrt(area)/224 + 1e-8)) clamped to [2, 2+L-1]
    const float size = sqrtf((bx2 - bx1) * (by2 - by1));
    float lvf = floorf(4.0f + log2f(size / 224.0f + 1e-8f));
    lvf = fminf(fmaxf(lvf, 2.0f), (float)(2 + a.L - 1));
    const int lv = (int)lvf - 2;
    if (a.out_level && threadIdx.x == 0) a.out_level[row] = lv;
    const float *feat = a.feat[lv] + (size_t)b * a.Hf[lv] * a.Wf[lv] * a.C;
    const int H = a.Hf[lv], W = a.Wf[lv];
    const float s = a.scale[lv];
    const float off = a.aligned ? 0.5f : 0.0f;
    const float x1 = bx1 * s - off, y1 = by1 * s - off, x2 = bx2 * s - off, y2 = by2 * s - off;
    float rw = x2 - x1, rh = y2 - y1;
    if (!a.aligned) {
        rw = fmaxf(rw, 1.0f);
        rh = fmaxf(rh, 1.0f);
    }
    const float bh = rh / (float)a.P, bw = rw / (float)a.P;
    const int gh = a.ratio > 0 ? a.ratio : (int)ceilf(rh / (float)a.P);
    const int gw = a.ratio > 0 ? a.ratio : (int)ceilf(rw / (float)a.P);
    const float count = (float)max(gh * gw, 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C4 = a.C >> 2;
    float *orow = a.out + (size_t)row * a.P * a.P * a.C;
    float lmax = 0.f;  // max |pooled| over this lane's stores (finite values only), reduced per ROI at the end
    __shared__ float wave_max[NW];
    auto emit = [&](const int bin, const int c, const f32x4 o) {  // channels c .. c + 3 of bin `bin`
        lmax = fmaxf(lmax, a3d_absmax4(o));
        if constexpr (H2) *reinterpret_cast<f32x4 *>(pooled_lds + (size_t)bin * a.C + c) = o;
        else *reinterpret_cast<f32x4 *>(orow + (size_t)bin * a.C + c) = o;
    };

    // Separable form.  The samples of a bin form a gh x gw lattice and a bilinear weight factorises into a y part and
    // an x part, so  sum_samples sum_corners w*f  ==  sum_rows sum_cols WY[row] * WX[col] * f[row][col]  with
    // WY[row] = sum over the bin's valid y-samples of (hy if yl == row) + (ly if yh == row), WX likewise.  A bin then reads
    // (gh+1)(gw+1) cells instead of 4*gh*gw (gh, gw are 2-4 for typical proposals: ~2.2x fewer gathered bytes, which is
    // what bounds this kernel).  Tables are built once per ROI: one thread per bin row / bin column.
    constexpr int KMAX = 16, PMAX = 16;
    __shared__ float WY[PMAX][KMAX], WX[PMAX][KMAX];
    __shared__ int Y0[PMAX], NY[PMAX], X0[PMAX], NX[PMAX];
    const bool separable = gh < KMAX && gw < KMAX && a.P <= PMAX;  // uniform per workgroup
    if (separable) {
        if (threadIdx.x < 2 * a.P) {
            const bool isx = threadIdx.x >= a.P;
            const int p = isx ? threadIdx.x - a.P : threadIdx.x;
            const int g = isx ? gw : gh, L = isx ? W : H;
            const float start = isx ? x1 : y1, bsz = isx ? bw : bh;
            float *wt = isx ? WX[p] : WY[p];
            for (int k = 0; k < KMAX; ++k) wt[k] = 0.f;
            int base = -1, last = -1;
            for (int i = 0; i < g; ++i) {
                float v = start + (float)p * bsz + ((float)i + 0.5f) * bsz / (float)g;
                if (v < -1.0f || v > (float)L) continue;
                if (v <= 0.f) v = 0.f;
                int lo = (int)v, hi;
                if (lo >= L - 1) {
                    hi = lo = L - 1;
                    v = (float)lo;
                } else
                    hi = lo + 1;
                const float l = v - (float)lo, h = 1.0f - l;
                if (base < 0) base = lo;
                wt[lo - base] += h;
                wt[hi - base] += l;
                last = hi;
            }
            (isx ? X0 : Y0)[p] = base < 0 ? 0 : base;
            (isx ? NX : NY)[p] = base < 0 ? 0 : last - base + 1;
        }
        __syncthreads();
        // A bin's (ny x nx) cells are independent 1 KiB loads.  Walking them with run-time loop bounds made hipcc issue one load,
        // wait, accumulate, issue the next: ~9 serialized L2 round trips per bin, and the kernel ran at the SUM of their latencies
        // (2.5 ms per 64-frame step for a 3.2 GB output).  The common case (<= 16 cells per bin: sampling grids up to 3 x 3) now
        // issues every load of the bin first -- 16 independent requests in flight per wave -- and accumulates afterwards in the
        // SAME order (ky outer, kx inner), so the results are bit-identical to the serialized form.
        // ---- rolling-window walk (round 6, opt-in: a3d_roialign_desc.serial == 2; 7 x 7 bins, C = 256, four waves) -------------------------
        // The bin-by-bin walk reads (ny x nx) cells per bin: ~980 cell loads of 1 KiB per ROI for ~441 distinct cells, and the launch is
        // bound by the L1 / texture path.  Here a wave owns TWO adjacent bin rows (wave 3: the last one alone) and walks the cell COLUMNS of
        // the ROI once: per column it loads the union of its bin rows' cell rows, forms the two row-weighted column sums, and adds them into
        // the accumulators of the (at most three) bins whose x window holds the column -- three ROLLING accumulators per bin row, emitted and
        // shifted when a bin's window ends, so every register index is static.  675 cell loads per typical ROI.  The sums run (column, row)
        // instead of (row, column): equal to the bin-by-bin form to fp32 rounding, not bit for bit.  ROIs whose geometry does not fit
        // (a bin without samples, more than three bins on a column = bins narrower than a cell, more than RMAX = 8 cell rows per pair) take the
        // bin-by-bin walk: the choice is a function of the ROI alone.
        constexpr int RMAX = 8;
        bool roll = ROLL && a.rolling && C4 == 64 && a.P == 7 && NW == 4;
        if (roll) {
#pragma unroll 1
            for (int p = 0; p < 7; ++p) {
                const int ex = X0[p] + NX[p] - 1, ey = Y0[p] + NY[p] - 1;
                roll = roll && NX[p] > 0 && NY[p] > 0;
                if (p < 6) roll = roll && X0[p] <= X0[p + 1] && ex <= X0[p + 1] + NX[p + 1] - 1 && Y0[p] <= Y0[p + 1] && ey <= Y0[p + 1] + NY[p + 1] - 1;
                if (p < 4) roll = roll && X0[p + 3] > ex;
                if ((p & 1) == 0) roll = roll && (p == 6 ? NY[p] : Y0[p + 1] + NY[p + 1] - Y0[p]) <= RMAX;
            }
        }
        if (roll) {
            const int phA = 2 * wave, phB = min(2 * wave + 1, 6);
            const bool two = wave < 3;
            const int r_lo = Y0[phA];
            const int nrows = (two ? Y0[phB] + NY[phB] : Y0[phA] + NY[phA]) - r_lo;  // <= RMAX
            // row weights of the two bin rows over the union of their cell rows: wave-uniform, kept in scalar registers
            float wA[RMAX], wB[RMAX];
#pragma unroll
            for (int u = 0; u < RMAX; ++u) {
                const int kA = u, kB = r_lo + u - Y0[phB];
                const float a_ = (u < nrows && kA < NY[phA]) ? WY[phA][kA] : 0.f;
                const float b_ = (two && u < nrows && kB >= 0 && kB < NY[phB]) ? WY[phB][kB] : 0.f;
                wA[u] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a_)));
                wB[u] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b_)));
            }
            const int x_lo = X0[0], x_hi = X0[6] + NX[6] - 1;
            const float *base = feat + ((size_t)r_lo * W) * a.C + lane * 4;
            f32x4 accA[3], accB[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) accA[j] = accB[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            int p = 0;  // the lowest bin column whose window has not ended
            auto load_col = [&](f32x4 (&v)[RMAX], const int x) {  // (a column past x_hi is clamped: loaded, never used)
                const int xc = min(x, x_hi);
#pragma unroll
                for (int u = 0; u < RMAX; ++u)
                    if (u < nrows) v[u] = *reinterpret_cast<const f32x4 *>(base + ((size_t)u * W + xc) * a.C);
            };
            auto use_col = [&](const f32x4 (&v)[RMAX], const int x) {
                f32x4 cA = {0.f, 0.f, 0.f, 0.f}, cB = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < RMAX; ++u)
                    if (u < nrows) {
                        cA += wA[u] * v[u];
                        cB += wB[u] * v[u];
                    }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int q = min(p + j, 6), k = x - X0[q];
                    const float w = (p + j < 7 && k >= 0 && k < NX[q]) ? WX[q][k] : 0.f;
                    accA[j] += w * cA;
                    accB[j] += w * cB;
                }
                while (p < 7 && x == X0[p] + NX[p] - 1) {  // bin column p is complete (two bins may end on one column)
                    emit(phA * 7 + p, lane * 4, accA[0] / count);
                    if (two) emit(phB * 7 + p, lane * 4, accB[0] / count);
                    accA[0] = accA[1];
                    accA[1] = accA[2];
                    accA[2] = f32x4{0.f, 0.f, 0.f, 0.f};
                    accB[0] = accB[1];
                    accB[1] = accB[2];
                    accB[2] = f32x4{0.f, 0.f, 0.f, 0.f};
                    ++p;
                }
            };
            // two columns in flight: column x + 1 is requested before column x is consumed
            f32x4 v0[RMAX], v1[RMAX];
            load_col(v0, x_lo);
            for (int x = x_lo; x <= x_hi; x += 2) {
                load_col(v1, x + 1);
                use_col(v0, x);
                load_col(v0, x + 2);
                if (x + 1 <= x_hi) use_col(v1, x + 1);
            }
        } else {
        constexpr int NC = A3D_ROI_NC;  // cells held in registers per bin (sampling grid + 1 in each direction)
        const int nbins = a.P * a.P;
        auto issue = [&](f32x4 (&v)[NC], int bin) -> int {  // all loads of one bin, nothing waited for
            const int ph = bin / a.P, pw = bin - ph * a.P;
            const int ny = NY[ph], nx = NX[pw];
            const int ncell = ny * nx;
            if (ncell > NC) return ncell;
            const float *base = feat + ((size_t)Y0[ph] * W + X0[pw]) * a.C + lane * 4;
            int ky = 0, kx = 0;
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                if (i < ncell) v[i] = *reinterpret_cast<const f32x4 *>(base + ((size_t)ky * W + kx) * a.C);
                if (++kx == nx) {
                    kx = 0;
                    ++ky;
                }
            }
            return ncell;
        };
        auto finish = [&](const f32x4 (&v)[NC], int bin, int ncell) {
            const int ph = bin / a.P, pw = bin - ph * a.P;
            const int ny = NY[ph], nx = NX[pw];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (ncell <= NC) {
                int ky = 0, kx = 0;
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    if (i < ncell) acc += (WY[ph][ky] * WX[pw][kx]) * v[i];
                    if (++kx == nx) {
                        kx = 0;
                        ++ky;
                    }
                }
            } else {  // large sampling lattices: the one-load-at-a-time walk
                for (int ky = 0; ky < ny; ++ky) {
                    const float wy = WY[ph][ky];
                    const float *frow = feat + ((size_t)(Y0[ph] + ky) * W + X0[pw]) * a.C + lane * 4;
                    for (int kx = 0; kx < nx; ++kx) acc += (wy * WX[pw][kx]) * *reinterpret_cast<const f32x4 *>(frow + (size_t)kx * a.C);
                }
            }
            emit(bin, lane * 4, acc / count);
        };
        if (C4 == 64 && !a.serial) {
            // A bin's (ny x nx) cells are independent 1 KiB loads.  Walking them with run-time loop bounds made hipcc issue one load,
            // wait, accumulate, issue the next: ~9 serialized L2 round trips per bin.  Every load of a bin is now issued before the
            // first is consumed; accumulation order is unchanged (ky outer, kx inner): bit-identical results.  Measured and rejected:
            // a second register set that prefetches the next bin (144 VGPRs, 3 workgroups per CU instead of 4-5: 3.1 ms vs 2.6 ms).
            f32x4 v[NC];
            for (int bin = wave; bin < nbins; bin += NW) finish(v, bin, issue(v, bin));
        } else
        for (int bin = wave; bin < nbins; bin += NW) {
            const int ph = bin / a.P, pw = bin - ph * a.P;
            const int ry0 = Y0[ph], ny = NY[ph], rx0 = X0[pw], nx = NX[pw];
            for (int c4 = lane; c4 < C4; c4 += 64) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                for (int ky = 0; ky < ny; ++ky) {
                    const float wy = WY[ph][ky];
                    const float *frow = feat + ((size_t)(ry0 + ky) * W + rx0) * a.C + c4 * 4;
                    for (int kx = 0; kx < nx; ++kx) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(frow + (size_t)kx * a.C);
                        acc += (wy * WX[pw][kx]) * v;
                    }
                }
                emit(bin, c4 * 4, acc / count);
            }
        }
        }  // (bin-by-bin walk)
    } else
    // general path (very large sampling grids): per-sample evaluation, as torchvision writes it
    for (int bin = wave; bin < a.P * a.P; bin += NW) {
        const int ph = bin / a.P, pw = bin - ph * a.P;
        for (int c4 = lane; c4 < C4; c4 += 64) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int iy = 0; iy < gh; ++iy) {
                float y = y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
                if (y < -1.0f || y > (float)H) continue;
                if (y <= 0.f) y = 0.f;
                int yl = (int)y, yh;
                if (yl >= H - 1) {
                    yh = yl = H - 1;
                    y = (float)yl;
                } else
                    yh = yl + 1;
                const float ly = y - (float)yl, hy = 1.0f - ly;
                for (int ix = 0; ix < gw; ++ix) {
                    float x = x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                    if (x < -1.0f || x > (float)W) continue;
                    if (x <= 0.f) x = 0.f;
                    int xl = (int)x, xh;
                    if (xl >= W - 1) {
                        xh = xl = W - 1;
                        x = (float)xl;
                    } else
                        xh = xl + 1;
                    const float lx = x - (float)xl, hx = 1.0f - lx;
                    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                    const f32x4 v1 = *reinterpret_cast<const f32x4 *>(feat + ((size_t)yl * W + xl) * a.C + c4 * 4);
                    const f32x4 v2 = *reinterpret_cast<const f32x4 *>(feat + ((size_t)yl * W + xh) * a.C + c4 * 4);
                    const f32x4 v3 = *reinterpret_cast<const f32x4 *>(feat + ((size_t)yh * W + xl) * a.C + c4 * 4);
                    const f32x4 v4 = *reinterpret_cast<const f32x4 *>(feat + ((size_t)yh * W + xh) * a.C + c4 * 4);
                    acc += w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
                }
            }
            emit(bin, c4 * 4, acc / count);
        }
    }
    // The ROI's own maximum (fp16x2: the power-of-two scale of every layer that consumes this row).  One workgroup owns the ROI, so
    // the reduction needs no atomics and its result does not depend on anything but the ROI.  A level-wide bound instead (round 2)
    // scaled a faint ROI by the hottest cell of its image.  Window monitor: the ROI's features inherit an ABSOLUTE error of
    // ~2^-40 of their LEVEL's maximum from the backbone's per-image block exponents; relative to the ROI that is 2^-40 x
    // (level max / ROI max), i.e. fp32-grade while the ratio stays below ~2^16.  ROIs past that are counted, never silently passed.
    if (a.out_amax || H2) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off, 64));
        if (lane == 0) wave_max[wave] = lmax;
        __syncthreads();  // (H2: also orders every wave's pooled bins in LDS before the split pass below)
        float m = wave_max[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = fmaxf(m, wave_max[w]);
        if (threadIdx.x == 0 && a.out_amax) {
            a.out_amax[row] = m;
            if (a.window_count && a.level_amax[lv]) {
                const float lm = a.level_amax[lv][b];
                if (m > 0.f && lm < 1.7e38f && m * (float)(1 << A3D_ROI_WINDOW_LOG2) < lm) atomicAdd(a.window_count, 1);
            }
        }
        if constexpr (H2) {
            // x * s = h + l, s = the scale fc1 derives from out_amax[row] (a3d_in_scale): conv_bf16x3_wide.hip's wx_split2h on 8 channels
            // per thread = one 16-byte half of a chunk's h row and of its l row
            const float sc = a3d_pow2_scale(m);
            unsigned char *hrow = a.out_h2 + (size_t)row * a.P * a.P * a.C * 4;
            const int n8 = (a.P * a.P * a.C) >> 3;
            for (int i = threadIdx.x; i < n8; i += 64 * NW) {
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(pooled_lds + (size_t)i * 8);
                const f32x4 v1 = *reinterpret_cast<const f32x4 *>(pooled_lds + (size_t)i * 8 + 4);
                const f32x4 x0 = v0 * sc, x1 = v1 * sc;
                const ra_h16x4 h0 = __builtin_convertvector(x0, ra_h16x4), h1 = __builtin_convertvector(x1, ra_h16x4);
                const ra_h16x4 l0 = __builtin_convertvector(x0 - __builtin_convertvector(h0, f32x4), ra_h16x4);
                const ra_h16x4 l1 = __builtin_convertvector(x1 - __builtin_convertvector(h1, f32x4), ra_h16x4);
                unsigned char *dst = hrow + (size_t)(i >> 1) * 64 + (i & 1) * 16;
                *reinterpret_cast<ra_h16x8 *>(dst) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
                *reinterpret_cast<ra_h16x8 *>(dst + 32) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    }
}

extern "C" int a3d_roi_align_fpn(const a3d_roialign_desc *d, void *stream) {
    if (!d || !d->boxes || (!d->out && !d->out_h2) || d->L < 1 || d->L > 4 || (d->C & 3) || d->B <= 0 || d->R <= 0 || d->P <= 0)
        return A3D_ERR_ARG;
    RoiArgs a;
    for (int l = 0; l < 4; ++l) {
        a.feat[l] = l < d->L ? d->feat[l] : nullptr;
        a.Hf[l] = d->Hf[l];
        a.Wf[l] = d->Wf[l];
        a.scale[l] = d->scale[l];
        if (l < d->L && !d->feat[l]) return A3D_ERR_ARG;
    }
    a.L = d->L;
    a.C = d->C;
    a.boxes = d->boxes;
    a.count = d->count;
    a.row_offset = d->row_offset;
    a.R = d->R;
    a.P = d->P;
    a.ratio = d->sampling_ratio;
    a.aligned = d->aligned;
    a.out = d->out;
    a.out_level = d->out_level;
    a.serial = d->serial == 1;
    a.rolling = d->serial == 2;
    a.order = nullptr;
    a.nblk = d->B * d->R;
    a.out_amax = d->out_amax;
    a.out_h2 = (unsigned char *)d->out_h2;
    a.window_count = d->window_count;
    for (int l = 0; l < 4; ++l) a.level_amax[l] = l < d->L ? d->level_amax[l] : nullptr;
    a3d_begin();
    if (d->order_ws && d->R <= 1024) {
        hipLaunchKernelGGL(roi_order_kernel, dim3(d->B), dim3(256), 0, (hipStream_t)stream, d->boxes, d->count, d->order_ws, d->R, d->L);
        a.order = d->order_ws;
    }
    if (d->out_h2) {  // pre-split output: the pooled row is staged in LDS (P*P*C floats), C = 256 (one float4 per lane and bin)
        const size_t lds = (size_t)d->P * d->P * d->C * 4;
        if (d->C != 256 || (d->C & 15) || lds > 120 * 1024) return A3D_ERR_UNSUPPORTED;
        if (d->P == 7) {
            static a3d_attr_once attr7;
            if (attr7.needed()) {
                if (hipFuncSetAttribute((const void *)roi_align_fpn_kernel<7, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return A3D_ERR_LAUNCH;
                attr7.mark();
            }
            hipLaunchKernelGGL((roi_align_fpn_kernel<7, true>), dim3(d->B * d->R), dim3(448), lds, (hipStream_t)stream, a);
        } else {
            static a3d_attr_once attr4;
            if (attr4.needed()) {
                if (hipFuncSetAttribute((const void *)roi_align_fpn_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024) != hipSuccess) return A3D_ERR_LAUNCH;
                attr4.mark();
            }
            hipLaunchKernelGGL((roi_align_fpn_kernel<4, true>), dim3(d->B * d->R), dim3(256), lds, (hipStream_t)stream, a);
        }
        return a3d_check_launch();
    }
    if (a.rolling && d->P == 7 && d->C == 256)  // (its own instantiation: the rolling walk's registers must not cost the other poolers their occupancy)
        hipLaunchKernelGGL((roi_align_fpn_kernel<4, false, true>), dim3(d->B * d->R), dim3(256), 0, (hipStream_t)stream, a);
    else if (d->P >= 12)
        // the 14 x 14 poolers (mask, plane / axis): 196 bins per ROI and a handful of ROIs per frame -- sixteen waves per ROI instead of four
        // (round 6: a single frame's four detections were four workgroups walking 49 bins per wave one memory round trip at a time, 95 us a
        // launch).  Bins are computed exactly as before, only dealt to more waves: the same bits, the same per-ROI maximum.
        hipLaunchKernelGGL((roi_align_fpn_kernel<16, false>), dim3(d->B * d->R), dim3(1024), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((roi_align_fpn_kernel<4, false>), dim3(d->B * d->R), dim3(256), 0, (hipStream_t)stream, a);
    return a3d_check_launch();
}

// exclusive prefix sum of per-image counts (B <= 1024): offsets[b], total -> offsets[B]
__global__ void count_offsets_kernel(const int *count, int *offsets, int B, int cap) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int s = 0;
        for (int b = 0; b < B; ++b) {
            offsets[b] = s;
            s += min(count[b], cap);
        }
        offsets[B] = s;
    }
}

extern "C" int a3d_count_offsets(const int *count, int *offsets, int B, int cap, void *stream) {
    if (!count || !offsets || B <= 0) return A3D_ERR_ARG;
    a3d_begin();
    hipLaunchKernelGGL(count_offsets_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, count, offsets, B, cap);
    return a3d_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Backward of the pooler (training step): dfeat[level] += scatter(dout).  Same geometry and the same
// separable weight tables as the forward kernel; one wave owns one bin, lane l owns channels 4l..4l+3 and issues
// float atomics (hardware red.add in L2) on the (gh+1)(gw+1) cells of the bin.  The summation order across ROIs is not
// fixed, so results are reproducible to fp32 rounding only (documented tolerance in the parity test).
// ------------------------------------------------------------------------------------------------
struct RoiBwdArgs {
    float *dfeat[4];
    int Hf[4], Wf[4];
    float scale[4];
    int L, C;
    const float *boxes;
    const int *count;
    const int *row_offset;
    int R, P, ratio, aligned;
    const float *dout;
};

__global__ __launch_bounds__(256) void roi_align_fpn_backward_kernel(const RoiBwdArgs a) {
    const int slot = blockIdx.x;
    const int b = slot / a.R, r = slot - b * a.R;
    const int cnt = a.count ? a.count[b] : a.R;
    if (r >= cnt) return;
    const int row = (a.row_offset ? a.row_offset[b] : b * a.R) + r;
    const float *bx = a.boxes + (size_t)slot * 4;
    const float bx1 = bx[0], by1 = bx[1], bx2 = bx[2], by2 = bx[3];
    const float size = sqrtf((bx2 - bx1) * (by2 - by1));
    float lvf = floorf(4.0f + log2f(size / 224.0f + 1e-8f));
    lvf = fminf(fmaxf(lvf, 2.0f), (float)(2 + a.L - 1));
    const int lv = (int)lvf - 2;
    float *feat = a.dfeat[lv] + (size_t)b * a.Hf[lv] * a.Wf[lv] * a.C;
    const int H = a.Hf[lv], W = a.Wf[lv];
    const float s = a.scale[lv];
    const float off = a.aligned ? 0.5f : 0.0f;
    const float x1 = bx1 * s - off, y1 = by1 * s - off, x2 = bx2 * s - off, y2 = by2 * s - off;
    float rw = x2 - x1, rh = y2 - y1;
    if (!a.aligned) {
        rw = fmaxf(rw, 1.0f);
        rh = fmaxf(rh, 1.0f);
    }
    const float bh = rh / (float)a.P, bw = rw / (float)a.P;
    const int gh = a.ratio > 0 ? a.ratio : (int)ceilf(rh / (float)a.P);
    const int gw = a.ratio > 0 ? a.ratio : (int)ceilf(rw / (float)a.P);
    const float count = (float)max(gh * gw, 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int C4 = a.C >> 2;
    const float *orow = a.dout + (size_t)row * a.P * a.P * a.C;

    // weight tables per bin row / bin column: up to KMAX cells each; longer sampling lattices are walked in windows
    constexpr int KMAX = 16, PMAX = 16;
    __shared__ float WY[PMAX][KMAX], WX[PMAX][KMAX];
    __shared__ int Y0[PMAX], NY[PMAX], X0[PMAX], NX[PMAX];
    if (a.P > PMAX) return;  // refused on the host side
    // sample lattices longer than KMAX-1 are processed in passes of (KMAX-1) samples per axis
    const int passes_y = (gh + KMAX - 2) / (KMAX - 1), passes_x = (gw + KMAX - 2) / (KMAX - 1);
    for (int py = 0; py < passes_y; ++py)
        for (int px = 0; px < passes_x; ++px) {
            __syncthreads();
            if (threadIdx.x < 2 * a.P) {
                const bool isx = threadIdx.x >= a.P;
                const int p = isx ? threadIdx.x - a.P : threadIdx.x;
                const int g = isx ? gw : gh, L = isx ? W : H;
                const int i0 = (isx ? px : py) * (KMAX - 1), i1 = min(g, i0 + KMAX - 1);
                const float start = isx ? x1 : y1, bsz = isx ? bw : bh;
                float *wt = isx ? WX[p] : WY[p];
                for (int k = 0; k < KMAX; ++k) wt[k] = 0.f;
                int base = -1, last = -1;
                for (int i = i0; i < i1; ++i) {
                    float v = start + (float)p * bsz + ((float)i + 0.5f) * bsz / (float)g;
                    if (v < -1.0f || v > (float)L) continue;
                    if (v <= 0.f) v = 0.f;
                    int lo = (int)v, hi;
                    if (lo >= L - 1) {
                        hi = lo = L - 1;
                        v = (float)lo;
                    } else
                        hi = lo + 1;
                    const float l = v - (float)lo, h = 1.0f - l;
                    if (base < 0) base = lo;
                    wt[lo - base] += h;
                    wt[hi - base] += l;
                    last = hi;
                }
                (isx ? X0 : Y0)[p] = base < 0 ? 0 : base;
                (isx ? NX : NY)[p] = base < 0 ? 0 : last - base + 1;
            }
            __syncthreads();
            for (int bin = wave; bin < a.P * a.P; bin += 4) {
                const int ph = bin / a.P, pw = bin - ph * a.P;
                const int ry0 = Y0[ph], ny = NY[ph], rx0 = X0[pw], nx = NX[pw];
                // lane l owns channels l, l+64, ...: one atomic instruction covers 64 CONSECUTIVE floats (2 cache lines); with a
                // float4 of channels per lane the four instructions of a cell each touched all 8 lines of the 1 KB vector
                for (int c = lane; c < a.C; c += 64) {
                    const float g = orow[(size_t)bin * a.C + c] / count;
                    for (int ky = 0; ky < ny; ++ky) {
                        const float wy = WY[ph][ky];
                        float *frow = feat + ((size_t)(ry0 + ky) * W + rx0) * a.C + c;
                        for (int kx = 0; kx < nx; ++kx) {
                            const float w = wy * WX[pw][kx];
                            if (w != 0.f) unsafeAtomicAdd(frow + (size_t)kx * a.C, w * g);
                        }
                    }
                }
            }
        }
}

extern "C" int a3d_roi_align_fpn_backward(const a3d_roialign_bwd_desc *d, void *stream) {
    if (!d || !d->boxes || !d->dout || d->L < 1 || d->L > 4 || (d->C & 3) || d->B <= 0 || d->R <= 0 || d->P <= 0 || d->P > 16)
        return A3D_ERR_ARG;
    if (d->sampling_ratio > 0) return A3D_ERR_UNSUPPORTED;  // adaptive lattices only (samples <= 1 cell apart): the box pooler
    RoiBwdArgs a;
    for (int l = 0; l < 4; ++l) {
        a.dfeat[l] = l < d->L ? d->dfeat[l] : nullptr;
        a.Hf[l] = d->Hf[l];
        a.Wf[l] = d->Wf[l];
        a.scale[l] = d->scale[l];
        if (l < d->L && !d->dfeat[l]) return A3D_ERR_ARG;
    }
    a.L = d->L;
    a.C = d->C;
    a.boxes = d->boxes;
    a.count = d->count;
    a.row_offset = d->row_offset;
    a.R = d->R;
    a.P = d->P;
    a.ratio = d->sampling_ratio;
    a.aligned = d->aligned;
    a.dout = d->dout;
    a3d_begin();
    hipLaunchKernelGGL(roi_align_fpn_backward_kernel, dim3(d->B * d->R), dim3(256), 0, (hipStream_t)stream, a);
    return a3d_check_launch();
}


// ------------------------------------------------------------------------------------------------
// Backward of the pooler without atomics (round 4).  The scatter form above issues one float atomic per (bin, cell, channel): 3.5 GB of
// memory-side atomics per step at 16 images (8192 ROIs x ~430 cell updates x 1 KiB), 2.75 ms at the 1.3 TB/s that path sustains, and
// a summation order that changes from run to run.  Here the pyramid is cut into 8 x 8-cell tiles; a first launch marks, per (image,
// tile), the ROIs whose sampling region touches the tile (one bit per ROI slot); a second launch gives every tile ONE workgroup that
// walks its marked ROIs in slot order, accumulates their contributions to its 64 cells in registers (thread = channel: no conflicts,
// no atomics) and adds the tile to dfeat once.  Every cell is written by exactly one workgroup and summed in a fixed order: the result
// is bit-reproducible.  Per (bin, cell) the weight is the same sum, in the same order, of the same bilinear terms as the scatter
// form's separable tables.
// ------------------------------------------------------------------------------------------------
constexpr int RBT = 8;  // tile edge in cells
struct RoiBwdGArgs {
    RoiBwdArgs a;
    unsigned *mask;     // [B][total_tiles][words]
    int tiles_off[5];   // first tile of each level
    int tx[4];          // tiles per row of each level
    int words, total_tiles, B;
};

struct RoiGeom {
    int lv, gh, gw;
    float x1, y1, bh, bw, rw, rh;
};
__device__ __forceinline__ RoiGeom roi_bwd_geom(const RoiBwdArgs &a, const float *bx) {
    RoiGeom q;
    const float bx1 = bx[0], by1 = bx[1], bx2 = bx[2], by2 = bx[3];
    const float size = sqrtf((bx2 - bx1) * (by2 - by1));
    float lvf = floorf(4.0f + log2f(size / 224.0f + 1e-8f));
    lvf = fminf(fmaxf(lvf, 2.0f), (float)(2 + a.L - 1));
    q.lv = (int)lvf - 2;
    const float s = a.scale[q.lv];
    const float off = a.aligned ? 0.5f : 0.0f;
    q.x1 = bx1 * s - off;
    q.y1 = by1 * s - off;
    float rw = bx2 * s - off - q.x1, rh = by2 * s - off - q.y1;
    if (!a.aligned) {
        rw = fmaxf(rw, 1.0f);
        rh = fmaxf(rh, 1.0f);
    }
    q.rw = rw;
    q.rh = rh;
    q.bh = rh / (float)a.P;
    q.bw = rw / (float)a.P;
    q.gh = a.ratio > 0 ? a.ratio : (int)ceilf(rh / (float)a.P);
    q.gw = a.ratio > 0 ? a.ratio : (int)ceilf(rw / (float)a.P);
    return q;
}

__global__ __launch_bounds__(256) void roi_bwd_mark_kernel(const RoiBwdGArgs g) {
    const RoiBwdArgs &a = g.a;
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= g.B * a.R) return;
    const int b = slot / a.R, r = slot - b * a.R;
    if (r >= (a.count ? a.count[b] : a.R)) return;
    const RoiGeom q = roi_bwd_geom(a, a.boxes + (size_t)slot * 4);
    const int H = a.Hf[q.lv], W = a.Wf[q.lv];
    // every sample lies inside (y1, y1 + rh); a sample at v touches the cells floor(v) and floor(v) + 1 (clamped into the map)
    const int ymin = max(0, (int)floorf(q.y1)), ymax = min(H - 1, (int)floorf(q.y1 + q.rh) + 1);
    const int xmin = max(0, (int)floorf(q.x1)), xmax = min(W - 1, (int)floorf(q.x1 + q.rw) + 1);
    if (ymax < ymin || xmax < xmin) return;
    unsigned *m = g.mask + ((size_t)b * g.total_tiles + g.tiles_off[q.lv]) * g.words + (r >> 5);
    for (int ty = ymin / RBT; ty <= ymax / RBT; ++ty)
        for (int tx = xmin / RBT; tx <= xmax / RBT; ++tx) atomicOr(m + (size_t)(ty * g.tx[q.lv] + tx) * g.words, 1u << (r & 31));
}

// thread = channel (C <= 256): the tile's 64 cell sums live in REGISTERS (static indices: an LDS read-modify-write per (bin, cell) chained
// its latencies, 17 us per ROI and tile); per ROI the live bins' values (<= 7 x 7) are requested together, then per bin row
// t[k] = sum over bin columns WX[pw][k] g[ph][pw] and acc[j][k] += WY[ph][j] t[k] -- the separable form of the bilinear weights.
__global__ __launch_bounds__(256) void roi_bwd_gather_kernel(const RoiBwdGArgs g) {
    const RoiBwdArgs &a = g.a;
    extern __shared__ __attribute__((aligned(16))) float rb_box[];  // the image's boxes [R][4]
    constexpr int PM = 7;
    __shared__ float WYt[PM][RBT], WXt[PM][RBT];
    __shared__ int J0[PM], J1[PM], K0[PM], K1[PM];  // cell range of each bin row / column inside the tile (J1 < J0: none)
    const int b = blockIdx.x / g.total_tiles, tile = blockIdx.x - b * g.total_tiles;
    int lv = 0;
    while (lv + 1 < a.L && tile >= g.tiles_off[lv + 1]) ++lv;
    const int tl = tile - g.tiles_off[lv];
    const int ty = tl / g.tx[lv], tx = tl - ty * g.tx[lv];
    const int H = a.Hf[lv], W = a.Wf[lv], y0 = ty * RBT, x0 = tx * RBT;
    const int tid = threadIdx.x;
    const unsigned *mw = g.mask + ((size_t)b * g.total_tiles + tile) * g.words;
    unsigned anyb = 0;
    for (int w = 0; w < g.words; ++w) anyb |= mw[w];
    if (!anyb) return;  // (dfeat += 0; uniform: every thread read the same words)
    for (int i = tid; i < a.R * 4; i += 256) rb_box[i] = a.boxes[(size_t)b * a.R * 4 + i];  // one round trip for all of the image's boxes
    const int row0 = a.row_offset ? a.row_offset[b] : b * a.R;
    const int c = min(tid, a.C - 1);  // (threads past C shadow the last channel and store nothing)
    float acc[RBT * RBT];
#pragma unroll
    for (int i = 0; i < RBT * RBT; ++i) acc[i] = 0.f;
    for (int w = 0; w < g.words; ++w) {
        unsigned bits = mw[w];
        while (bits) {
            const int r = w * 32 + __ffs(bits) - 1;
            bits &= bits - 1;
            __syncthreads();  // boxes staged / the tables of the previous ROI are no longer read
            const RoiGeom q = roi_bwd_geom(a, rb_box + r * 4);  // (q.lv == lv: the mark kernel put the bit on this level's tile)
            const float rcount = (float)max(q.gh * q.gw, 1);
            if (tid < 2 * a.P * RBT) {  // thread = (axis, bin row / column p, tile cell j): its table entry, samples in order, low cell first
                const bool isx = tid >= a.P * RBT;
                const int e = isx ? tid - a.P * RBT : tid;
                const int p = e / RBT, j = e - p * RBT;
                const int gg = isx ? q.gw : q.gh, L = isx ? W : H, cell = (isx ? x0 : y0) + j;
                const float start = isx ? q.x1 : q.y1, bsz = isx ? q.bw : q.bh;
                float wt = 0.f;
                int lo_min = 1 << 30, hi_max = -1;
                for (int i = 0; i < gg; ++i) {
                    float v = start + (float)p * bsz + ((float)i + 0.5f) * bsz / (float)gg;
                    if (v < -1.0f || v > (float)L) continue;
                    if (v <= 0.f) v = 0.f;
                    int lo = (int)v, hi;
                    if (lo >= L - 1) {
                        hi = lo = L - 1;
                        v = (float)lo;
                    } else
                        hi = lo + 1;
                    const float l = v - (float)lo, h = 1.0f - l;
                    if (lo == cell) wt += h;
                    if (hi == cell) wt += l;
                    lo_min = min(lo_min, lo);
                    hi_max = max(hi_max, hi);
                }
                (isx ? WXt : WYt)[p][j] = wt;
                if (j == 0) {  // the cells this bin row / column touches, clipped to the tile
                    const int c0 = isx ? x0 : y0;
                    (isx ? K0 : J0)[p] = max(lo_min - c0, 0);
                    (isx ? K1 : J1)[p] = min(hi_max - c0, RBT - 1);
                }
            }
            __syncthreads();
            // live bins: rows pa .. pb x columns qa .. qb (the ranges are monotone in p: one contiguous block)
            int pa = a.P, pb = -1, qa = a.P, qb = -1;
            for (int p = 0; p < a.P; ++p) {
                if (J1[p] >= J0[p]) {
                    pa = min(pa, p);
                    pb = p;
                }
                if (K1[p] >= K0[p]) {
                    qa = min(qa, p);
                    qb = p;
                }
            }
            if (pb < pa || qb < qa) continue;
            const int nph = pb - pa + 1, nq = qb - qa + 1;
            const float *orow = a.dout + ((size_t)(row0 + r) * a.P * a.P + (size_t)pa * a.P + qa) * a.C + c;
            float gv[PM][PM];
#pragma unroll
            for (int ip = 0; ip < PM; ++ip)
#pragma unroll
                for (int iq = 0; iq < PM; ++iq) gv[ip][iq] = (ip < nph && iq < nq) ? orow[(size_t)(ip * a.P + iq) * a.C] : 0.f;
#pragma unroll
            for (int ip = 0; ip < PM; ++ip) {
                if (ip >= nph) break;
                float t[RBT];
#pragma unroll
                for (int k = 0; k < RBT; ++k) t[k] = 0.f;
#pragma unroll
                for (int iq = 0; iq < PM; ++iq) {
                    if (iq >= nq) break;
                    const float gq = gv[ip][iq] / rcount;
#pragma unroll
                    for (int k = 0; k < RBT; ++k) t[k] = __builtin_fmaf(WXt[qa + iq][k], gq, t[k]);
                }
#pragma unroll
                for (int j = 0; j < RBT; ++j) {
                    const float wy = WYt[pa + ip][j];
#pragma unroll
                    for (int k = 0; k < RBT; ++k) acc[j * RBT + k] = __builtin_fmaf(wy, t[k], acc[j * RBT + k]);
                }
            }
        }
    }
    if (tid >= a.C) return;
    float *feat = a.dfeat[lv] + (size_t)b * H * W * a.C + c;
#pragma unroll
    for (int j = 0; j < RBT; ++j)
#pragma unroll
        for (int k = 0; k < RBT; ++k)
            if (y0 + j < H && x0 + k < W) feat[((size_t)(y0 + j) * W + x0 + k) * a.C] += acc[j * RBT + k];
}

static int roi_bwd_fill(const a3d_roialign_bwd_desc *d, RoiBwdGArgs &g) {
    if (!d || !d->boxes || !d->dout || d->L < 1 || d->L > 4 || (d->C & 3) || d->B <= 0 || d->R <= 0 || d->P <= 0 || d->P > 16) return A3D_ERR_ARG;
    if (d->sampling_ratio > 0) return A3D_ERR_UNSUPPORTED;
    RoiBwdArgs &a = g.a;
    int off = 0;
    for (int l = 0; l < 4; ++l) {
        a.dfeat[l] = l < d->L ? d->dfeat[l] : nullptr;
        a.Hf[l] = d->Hf[l];
        a.Wf[l] = d->Wf[l];
        a.scale[l] = d->scale[l];
        if (l < d->L && !d->dfeat[l]) return A3D_ERR_ARG;
        g.tiles_off[l] = off;
        g.tx[l] = l < d->L ? (d->Wf[l] + RBT - 1) / RBT : 0;
        if (l < d->L) off += g.tx[l] * ((d->Hf[l] + RBT - 1) / RBT);
    }
    g.tiles_off[4] = off;
    for (int l = d->L; l < 4; ++l) g.tiles_off[l] = off;
    g.total_tiles = off;
    g.words = (d->R + 31) / 32;
    g.B = d->B;
    a.L = d->L;
    a.C = d->C;
    a.boxes = d->boxes;
    a.count = d->count;
    a.row_offset = d->row_offset;
    a.R = d->R;
    a.P = d->P;
    a.ratio = d->sampling_ratio;
    a.aligned = d->aligned;
    a.dout = d->dout;
    return A3D_OK;
}

extern "C" size_t a3d_roi_align_bwd_workspace_bytes(const a3d_roialign_bwd_desc *d) {
    RoiBwdGArgs g;
    if (roi_bwd_fill(d, g) != A3D_OK) return 0;
    return (size_t)d->B * g.total_tiles * g.words * sizeof(unsigned);
}

extern "C" int a3d_roi_align_fpn_backward_gather(const a3d_roialign_bwd_desc *d, void *workspace, void *stream) {
    RoiBwdGArgs g;
    const int rc = roi_bwd_fill(d, g);
    if (rc != A3D_OK) return rc;
    if (!workspace) return A3D_ERR_ARG;
    const size_t lds = (size_t)d->R * 4 * sizeof(float);
    if (lds > 60 * 1024 || d->P > 7 || d->C > 256) return A3D_ERR_UNSUPPORTED;  // (the caller keeps the scatter form)
    g.mask = (unsigned *)workspace;
    hipStream_t s = (hipStream_t)stream;
    a3d_begin();
    if (hipMemsetAsync(workspace, 0, (size_t)d->B * g.total_tiles * g.words * sizeof(unsigned), s) != hipSuccess) return A3D_ERR_LAUNCH;
    hipLaunchKernelGGL(roi_bwd_mark_kernel, dim3((d->B * d->R + 255) / 256), dim3(256), 0, s, g);
    hipLaunchKernelGGL(roi_bwd_gather_kernel, dim3(d->B * g.total_tiles), dim3(256), lds, s, g);
    return a3d_check_launch();
}

// ---- upper bound of max |pooled[row]| for the fp16x2 split of the layers that consume pooled ROI features ----------------------
// A pooled value is a convex combination of cells of ONE pyramid level of the ROI's image, so max over that image's levels bounds it.
__global__ __launch_bounds__(256) void roi_amax_kernel(const float *a0, const float *a1, const float *a2, const float *a3, const int *count,
                                                       const int *row_offset, int B, int R, float *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * R) return;
    const int b = i / R, r = i - b * R;
    if (r >= (count ? min(count[b], R) : R)) return;
    float m = a0[b];
    if (a1) m = fmaxf(m, a1[b]);
    if (a2) m = fmaxf(m, a2[b]);
    if (a3) m = fmaxf(m, a3[b]);
    out[(row_offset ? row_offset[b] : b * R) + r] = m;
}

extern "C" int a3d_roi_amax(const float *const level_amax[4], int L, const int *count, const int *row_offset, int B, int R, float *out, void *stream) {
    if (!level_amax || L < 1 || L > 4 || !level_amax[0] || !out || B <= 0 || R <= 0) return A3D_ERR_ARG;
    a3d_begin();
    hipLaunchKernelGGL(roi_amax_kernel, dim3((B * R + 255) / 256), dim3(256), 0, (hipStream_t)stream, level_amax[0], L > 1 ? level_amax[1] : nullptr,
                       L > 2 ? level_amax[2] : nullptr, L > 3 ? level_amax[3] : nullptr, count, row_offset, B, R, out);
    return a3d_check_launch();
}
