/*
 * CPU oracle kernels for the PlaneRCNN detection hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the two third-party operators the reference reaches through
 * torchvision (roi_align, nms).  torchvision/detectron2 are NOT vendored in the reference
 * tree (reference setup.py:10 lists them unpinned), so these follow the published
 * algorithm as written down in SURVEY.md Appendix A.6 / A.7; call sites in the reference:
 *   articulation3d/articulation3d/modeling/roi_heads/roi_heads.py:50-55,74-79,185,236,250,268
 *   (ROIPooler -> roi_align) and the d2 RPN / FastRCNNOutputLayers -> batched_nms.
 * PARITY UNPINNED for these two operators: the reference holds no test or golden vector
 * for them; the definition below IS the parity target.
 *
 * Built by oracle/Makefile into oracle/_build/liba3d_oracle.so; loaded with ctypes by
 * oracle/planercnn_oracle.py.  Never linked into the product library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Bilinear sample of one channel plane, Appendix A.7. */
static inline float bilinear(const float *p, int H, int W, float y, float x) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.0f;
    if (y <= 0.0f) y = 0.0f;
    if (x <= 0.0f) x = 0.0f;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
    float ly = y - (float)yl, lx = x - (float)xl, hy = 1.0f - ly, hx = 1.0f - lx;
    float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    return w1 * p[yl * W + xl] + w2 * p[yl * W + xh] + w3 * p[yh * W + xl] + w4 * p[yh * W + xh];
}

/*
 * feat: N x C x H x W (NCHW).  rois: K x 5 (batch_idx, x1, y1, x2, y2).
 * out: K x C x P x P.  sampling_ratio <= 0 -> adaptive ceil(roi/P).
 */
void orc_roi_align_nchw(const float *feat, int N, int C, int H, int W, const float *rois, int K,
                        int P, float scale, int sampling_ratio, int aligned, float *out) {
    (void)N;
#pragma omp parallel for schedule(dynamic, 4)
    for (int k = 0; k < K; ++k) {
        const float *r = rois + (size_t)k * 5;
        int b = (int)r[0];
        float off = aligned ? 0.5f : 0.0f;
        float x1 = r[1] * scale - off, y1 = r[2] * scale - off;
        float x2 = r[3] * scale - off, y2 = r[4] * scale - off;
        float rw = x2 - x1, rh = y2 - y1;
        if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
        float bh = rh / (float)P, bw = rw / (float)P;
        int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)P);
        int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)P);
        float count = (float)(gh * gw > 1 ? gh * gw : 1);
        for (int c = 0; c < C; ++c) {
            const float *plane = feat + ((size_t)b * C + c) * H * W;
            float *o = out + ((size_t)k * C + c) * P * P;
            for (int ph = 0; ph < P; ++ph)
                for (int pw = 0; pw < P; ++pw) {
                    float acc = 0.0f;
                    for (int iy = 0; iy < gh; ++iy) {
                        float y = y1 + ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
                        for (int ix = 0; ix < gw; ++ix) {
                            float x = x1 + pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                            acc += bilinear(plane, H, W, y, x);
                        }
                    }
                    o[ph * P + pw] = acc / count;
                }
        }
    }
}

/*
 * The same operator in double precision: the float64 YARDSTICK of oracle/exact.py (how far the fp32 CPU path and the
 * HIP path each are from the exactly evaluated graph).  Identical control flow, every float replaced by double.
 */
static inline double bilinear_f64(const double *p, int H, int W, double y, double x) {
    if (y < -1.0 || y > (double)H || x < -1.0 || x > (double)W) return 0.0;
    if (y <= 0.0) y = 0.0;
    if (x <= 0.0) x = 0.0;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (double)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (double)xl; } else xh = xl + 1;
    double ly = y - (double)yl, lx = x - (double)xl, hy = 1.0 - ly, hx = 1.0 - lx;
    return hy * hx * p[yl * W + xl] + hy * lx * p[yl * W + xh] + ly * hx * p[yh * W + xl] + ly * lx * p[yh * W + xh];
}

void orc_roi_align_nchw_f64(const double *feat, int N, int C, int H, int W, const double *rois, int K,
                            int P, double scale, int sampling_ratio, int aligned, double *out) {
    (void)N;
#pragma omp parallel for schedule(dynamic, 4)
    for (int k = 0; k < K; ++k) {
        const double *r = rois + (size_t)k * 5;
        int b = (int)r[0];
        double off = aligned ? 0.5 : 0.0;
        double x1 = r[1] * scale - off, y1 = r[2] * scale - off;
        double x2 = r[3] * scale - off, y2 = r[4] * scale - off;
        double rw = x2 - x1, rh = y2 - y1;
        if (!aligned) { rw = fmax(rw, 1.0); rh = fmax(rh, 1.0); }
        double bh = rh / (double)P, bw = rw / (double)P;
        int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceil(rh / (double)P);
        int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceil(rw / (double)P);
        double count = (double)(gh * gw > 1 ? gh * gw : 1);
        for (int c = 0; c < C; ++c) {
            const double *plane = feat + ((size_t)b * C + c) * H * W;
            double *o = out + ((size_t)k * C + c) * P * P;
            for (int ph = 0; ph < P; ++ph)
                for (int pw = 0; pw < P; ++pw) {
                    double acc = 0.0;
                    for (int iy = 0; iy < gh; ++iy) {
                        double y = y1 + ph * bh + ((double)iy + 0.5) * bh / (double)gh;
                        for (int ix = 0; ix < gw; ++ix) {
                            double x = x1 + pw * bw + ((double)ix + 0.5) * bw / (double)gw;
                            acc += bilinear_f64(plane, H, W, y, x);
                        }
                    }
                    o[ph * P + pw] = acc / count;
                }
        }
    }
}

/*
 * Greedy NMS, Appendix A.6.  boxes: n x 4 xyxy ALREADY in score-descending order;
 * cat: n category ids (suppression only within equal ids).  keep[i] = 1 if kept.
 */
void orc_nms_sorted(const float *boxes, const int32_t *cat, int n, float thr, uint8_t *keep) {
    uint8_t *sup = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    for (int i = 0; i < n; ++i) {
        keep[i] = 0;
        if (sup[i]) continue;
        keep[i] = 1;
        const float *a = boxes + (size_t)i * 4;
        float areaA = (a[2] - a[0]) * (a[3] - a[1]);
        for (int j = i + 1; j < n; ++j) {
            if (sup[j] || cat[j] != cat[i]) continue;
            const float *q = boxes + (size_t)j * 4;
            float xx1 = fmaxf(a[0], q[0]), yy1 = fmaxf(a[1], q[1]);
            float xx2 = fminf(a[2], q[2]), yy2 = fminf(a[3], q[3]);
            float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
            float inter = w * h;
            float areaB = (q[2] - q[0]) * (q[3] - q[1]);
            float ovr = inter / (areaA + areaB - inter);
            if (ovr > thr) sup[j] = 1;
        }
    }
    free(sup);
}

/*
 * Gradient of orc_roi_align_nchw with respect to feat (torchvision roi_align_backward as published:
 * every sample scatters grad_bin / count times its four bilinear weights).  dfeat must be zeroed by the
 * caller; accumulation is double precision per element so the oracle's value does not depend on the
 * scatter order.  Used by oracle/train_oracle.py (the box branch of the training step,
 * reference call site pkg/modeling/roi_heads/roi_heads.py:185 under autograd).
 */
void orc_roi_align_backward_nchw(const float *dout, int N, int C, int H, int W, const float *rois, int K,
                                 int P, float scale, int sampling_ratio, int aligned, double *dfeat) {
    (void)N;
    for (int k = 0; k < K; ++k) {
        const float *r = rois + (size_t)k * 5;
        int b = (int)r[0];
        float off = aligned ? 0.5f : 0.0f;
        float x1 = r[1] * scale - off, y1 = r[2] * scale - off;
        float x2 = r[3] * scale - off, y2 = r[4] * scale - off;
        float rw = x2 - x1, rh = y2 - y1;
        if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
        float bh = rh / (float)P, bw = rw / (float)P;
        int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)P);
        int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)P);
        float count = (float)(gh * gw > 1 ? gh * gw : 1);
        for (int ph = 0; ph < P; ++ph)
            for (int pw = 0; pw < P; ++pw)
                for (int iy = 0; iy < gh; ++iy) {
                    float y = y1 + ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
                    if (y < -1.0f || y > (float)H) continue;
                    if (y <= 0.0f) y = 0.0f;
                    int yl = (int)y, yh;
                    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
                    float ly = y - (float)yl, hy = 1.0f - ly;
                    for (int ix = 0; ix < gw; ++ix) {
                        float x = x1 + pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                        if (x < -1.0f || x > (float)W) continue;
                        if (x <= 0.0f) x = 0.0f;
                        int xl = (int)x, xh;
                        if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
                        float lx = x - (float)xl, hx = 1.0f - lx;
                        float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                        for (int c = 0; c < C; ++c) {
                            double g = (double)dout[(((size_t)k * C + c) * P + ph) * P + pw] / (double)count;
                            double *plane = dfeat + ((size_t)b * C + c) * H * W;
                            plane[yl * W + xl] += g * w1;
                            plane[yl * W + xh] += g * w2;
                            plane[yh * W + xl] += g * w3;
                            plane[yh * W + xh] += g * w4;
                        }
                    }
                }
    }
}
