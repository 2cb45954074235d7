/*
 * a3d.h -- C ABI of liba3d_hip.so: the MI355X (gfx950) kernels of the PlaneRCNN per-frame
 * detection path of Articulation3D.
 *
 * The reference is pure Python on top of detectron2 / torchvision / torch operators; it has no FFI
 * of its own.  Each entry point below therefore names the reference call site (file:line under
 * /root/reference/articulation3d/articulation3d/, "pkg/") whose operator it replaces.  A maintainer
 * binds these with ctypes exactly as articulation3d_amd/_lib.py does.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name says host; no allocation inside; caller owns
 *     all buffers, workspaces are sized by the *_workspace_bytes helpers;
 *   - all activations are fp32 NHWC (channels innermost);
 *   - stream-ordered on `stream` (a hipStream_t passed as void*), re-entrant, no global state;
 *   - returns 0 (A3D_OK) or a negative error code; nothing is thrown.
 */
#ifndef A3D_H
#define A3D_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define A3D_OK 0
#define A3D_ERR_ARG (-1)
#define A3D_ERR_LAUNCH (-2)
#define A3D_ERR_UNSUPPORTED (-3)

#define A3D_ACT_NONE 0
#define A3D_ACT_RELU 1
#define A3D_ACT_LEAKY 2 /* slope 0.01, pkg/modeling/depth_net/depth_head.py:36 */

int a3d_version(void);
/* sizeof() of a descriptor struct, for bindings to verify their mirror of the layout.  id: 0 a3d_conv_desc, 1 a3d_rpn_desc,
 * 2 a3d_boxdet_desc, 3 a3d_roialign_desc, 4 a3d_paste_desc, 5 a3d_pack_desc, 6 a3d_wgrad_desc, 7 a3d_roialign_bwd_desc,
 * 8 a3d_match_desc, 9 a3d_rpn_loss_desc, 10 a3d_box_loss_desc, 11 a3d_roi_sample_desc, 12 a3d_sweep_desc, 13 a3d_transpose_item; 0 for an
 * unknown id. */
size_t a3d_struct_size(int id);

/* ------------------------------------------------------------------------------------------------
 * Pre-processing.  Replaces PlaneRCNN.preprocess_image (pkg/modeling/meta_arch/planercnn.py:188-196)
 * + the HWC->CHW float cast of PlaneRCNN_Branch.inference (pkg/utils/arti_vis.py:58).
 * Output: [B,H,W,4] fp32, channel 3 = 0, value = (x - mean[c]) / std[c].
 * ---------------------------------------------------------------------------------------------- */
int a3d_preprocess_u8hwc(const uint8_t *frames /*[B,H,W,3]*/, float *out, int B, int H, int W,
                         const float mean[3], const float std[3], void *stream);
int a3d_preprocess_f32chw(const float *images /*[B,3,H,W]*/, float *out, int B, int H, int W,
                          const float mean[3], const float std[3], void *stream);

/* Input front end: frames of any size, uint8 HWC in the reader's channel order, to the detector's input in ONE
 * pass: `cv2.resize(im, (Wd, Hd))` (tools/inference.py:216; OpenCV's fixed-point INTER_LINEAR on uint8, the 2x2 INTER_AREA
 * switch at an exact 2x decimation, plain copy at equal sizes) + the RGB -> BGR flip `im[:, :, ::-1]` (:218, swap_rb = 1) +
 * float cast + (x - mean[c]) / std[c] (arti_vis.py:58, planercnn.py:188-196).  out [B,Hd,Wd,4] fp32 (channel 3 = 0), mean / std in
 * the OUTPUT channel order; out_u8 (optional) [B,Hd,Wd,3] = the resized frame in the source order (the reference's `frames` list). */
int a3d_preprocess_resize_u8(const uint8_t *frames /*[B,Hs,Ws,3]*/, float *out, uint8_t *out_u8, int B, int Hs, int Ws, int Hd, int Wd,
                             int swap_rb, const float mean[3], const float std[3], void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fused convolution / linear as an fp32-MFMA implicit GEMM.
 * Replaces every Conv2d / Linear / ConvTranspose2d reached on the path:
 *   backbone + FPN + RPN head (planercnn.py:150,168 -> detectron2), box head (roi_heads.py:186-187),
 *   mask head (roi_heads.py:237), plane head (plane_head.py:71-80), axis head (axis_head.py:95-120),
 *   depth head (depth_head.py:32-46,80-87).
 *   y[m, n] = act( scale[n] * sum_k X[m, k] * w[n, k] + shift[n] + res[m, n] )
 * m = (b, oh, ow); k = (kh, kw, c) with c running over source 0 then source 1 (channel concat).
 * ---------------------------------------------------------------------------------------------- */
typedef struct a3d_conv_desc {
    const float *x;     /* source 0, NHWC [B,H,W,Cin]                                              */
    const float *x2;    /* optional source 1 [B,H,W,Cin2] concatenated after source 0, or NULL     */
    const float *w;     /* packed weights [Cout][Kpad], k = (kh*KW + kw)*(Cin+Cin2) + c            */
    const float *scale; /* [Cout] or NULL (=1)   -- folded BatchNorm scale                         */
    const float *shift; /* [Cout] or NULL (=0)   -- bias / folded BatchNorm shift                  */
    const float *res;   /* residual [B,Ho,Wo,Cout] (or [B,Ho/2,Wo/2,Cout] when res_ups) or NULL    */
    float *y;           /* [B,Ho,Wo,Cout]  (pixshuf: [B,2Ho,2Wo,Cout/4])                           */
    float *workspace;   /* split-K partials or the Winograd-domain input, a3d_conv_workspace_bytes(); may be NULL
                           if splitk == 1 and w_wino == NULL                                       */
    int B, H, W, Cin, Cin2;
    int Ho, Wo, Cout; /* Cout % 4 == 0 */
    int KH, KW, stride, pad;
    int Kpad;    /* row length of w in floats, multiple of 32, >= KH*KW*(Cin+Cin2)                  */
    int ups;     /* 1: the logical input is the nearest x2 upsampling of the sources (depth deconv) */
    int act;     /* A3D_ACT_*                                                                       */
    int res_ups; /* 1: residual is nearest x2 upsampled (FPN top-down add)                          */
    int pixshuf; /* 1: ConvTranspose2d k2 s2 as GEMM with columns (dy,dx,co) scattered to 2x2 blocks */
    int stem;    /* 1: x is [B,H,W,4] (a3d_preprocess_*), 7x7 s2 p3, w packed [Cout][7][8][4]        */
    int splitk;  /* >= 1; >1 writes partials to workspace and reduces in a second launch (precision 1: plain output rows
                    only -- no res_ups / pixshuf / phase --, splitk <= Kpad / 32)                        */
    const int *m_dev; /* optional DEVICE int: live row count (<= B*Ho*Wo); tiles past it exit at once,
                         so ragged per-ROI batches need no host synchronisation                      */
    int tune;         /* 0 = library picks the kernel variant; 1 = (the round-1 general kernel: removed in round 4, A3D_ERR_UNSUPPORTED);
                         2 = direct implicit GEMM even when w_wino is given; 5 / 6 = never / always use the persistent
                         pointwise kernel on eligible 1x1 layers; 7 = two-launch Winograd where the one-launch kernel would
                         run; precision 2: 8 = the narrow split-operand kernels (64-wide Winograd GEMM, 128 x 128 direct),
                         9 = the wide direct kernel whatever the problem size, 10 / 11 = its narrow kernel with 128 x 64 /
                         128 x 128 tiles, 12 = per-lane accumulator stores instead of the row-major epilogue through LDS (the
                         same bits; A/B runs); precision 3: 13 / 14 = always / never the activation-stationary pointwise kernel on an
                         eligible 1x1 layer (Cin 64 / 128 / 256, Cout % 128 == 0; the same bits); 17 = the small-grid pointwise kernel (one wave per
                         32 x 32 output tile, csrc/conv_sg_h2.hip; Cin % 64 == 0, Cout % 32 == 0) whatever the grid size -- tune 0 takes it for
                         launches of up to 1280 such tiles (single frames), 10 / 11 never (the same bits); 15 / 16 = the tap-outer / the patch-resident
                         form of a 3x3 s1 p1 layer (phase 5: every launch is patch-resident by default, 15 is the bit-equality link to
                         the four-launch form; phase 0: 16 forces the patch-resident kernel on a layer whose map its tiles fit badly;
                         the two forms reduce in different orders: equal to fp32 rounding, not bit for bit); Winograd layers: 23 = the
                         lockstep loop of the 128-tile fp16x2 GEMM instead of its ping-pong loop, 24 = 64-tile blocks (two workgroups per
                         CU), 25 (precision 2) = 128-tile blocks whatever the problem size -- all the same bits (A/B runs, equality tests);
                         precision 1 with w_bf16: 30 / 31 = the LDS-DMA kernel with 128- / 256-channel tiles whatever the size, 32 = never it;
                         33 / 34 = always / never the activation-stationary pointwise kernel (csrc/conv_bf16xs.hip) on a 1x1 layer it can
                         run (the same bits);
                         >= 100: explicit tile variant.
                         `tune` is the ONLY way to choose a variant: the library reads no environment variable and has no process-global
                         switch (developer builds with -DA3D_ABLATIONS excepted, see csrc/a3d_common.h). */
    int phase;        /* 0, or 1..4 = output phase (dy,dx) = ((phase-1)>>1, (phase-1)&1) of a 3x3 pad-1 convolution over a
                         nearest-x2 upsampled input, evaluated on the SOURCE grid as a 2x2 convolution with pre-summed
                         taps (KH = KW = 2, stride 1, pad ignored; taps read source rows oh-1+dy .. oh+dy): the four
                         phases write the interleaved pixels (2oh+dy, 2ow+dx) of y [B,2Ho,2Wo,Cout] -- 4/9 of the FLOPs
                         of convolving the upsampled tensor (depth decoder, depth_head.py:40-46).
                         5 = ALL FOUR phases in one launch (fp16x2 only, precision 3 with w_x3): KH = KW = 3, stride 1, pad 1 on
                         the source grid, Cout = 4 x the real channel count C (32 | C), y [B,2Ho,2Wo,C]; GEMM column
                         128 g + 32 p + c holds phase p = 2 dy + dx of output channel 32 g + c, its filter = that phase's
                         pre-summed 2x2 taps at rows kh - dy, columns kw - dx of the 3x3 neighbourhood (zero elsewhere;
                         scale / shift in the same column order).  The kernel skips the zero blocks: the 16 tap-phase
                         products of the four-launch form in the same order per output -- bit-identical -- with every
                         activation chunk loaded and split once for the phases that share it (round 3)                */
    const float *w_wino; /* optional Winograd-domain weights U = G g G^T, [16][Cout][Cin+Cin2] (3x3 s1 p1 only):
                            when given (and workspace holds a3d_conv_workspace_bytes) the layer runs as
                            F(2x2,3x3): 2.25x fewer MFMA cycles, same result within fp32 rounding           */
    const float *gate;   /* optional [B,Ho,Wo,Cout]: outputs are zeroed where gate <= 0 -- the ReLU backward of the
                            training step (section "Training step" below): y = dgrad(...) * (forward activation > 0).
                            Plain output layout only (no pixshuf / phase)                                     */
    int precision;       /* 0: fp32 MFMA (the inference / parity path).  1: bf16 MFMA with fp32 accumulation -- both operands
                            are rounded to bf16 (nearest-even) while they are staged in LDS, tensors stay fp32 in memory: the
                            arithmetic of torch.autocast(bfloat16), which the reference's training config asks for
                            (plain convolutions / linears only: no stem, ups, phase, pixshuf, concat, split-K).
                            2: fp32-grade on the bf16 pipe -- each fp32 operand is split EXACTLY into three bf16 terms in LDS and
                            six bf16 MFMAs per k step reproduce the fp32 product to 2^-24 relative (csrc/conv_bf16x3.hip);
                            same layer kinds as 1 plus the phase convs and their equal-width 2-source concat, Cin % 16 == 0.
                            Opt-in: the default everywhere is 0                                                  */
    const void *w_wino_x3; /* precision 2 on a Winograd layer: w_wino split into three bf16 planes, chunk-major
                            [16][(Cin+Cin2)/32][3][Cout][32] (a3d_split_bf16x3 with outer = 16, rows = Cout, cols = Cin+Cin2);
                            the layer then runs F(2x2,3x3) with the split-operand GEMM                                */
    const float *w_wino_cm; /* optional: the Winograd-domain weights chunk-major in the one-launch kernel's LDS-image order,
                            [Cin/8][16][ceil(Cout/64)][2 h][64 r][4 j] = w_wino[f][64 t + r][8 c + 4 h + j], zero for channels
                            past Cout (ops.winograd_weights_chunk_major).
                            With it a plain 3x3 s1 p1 layer (one source, no upsampling, precision 0, tune 0) runs
                            the ONE-launch Winograd kernel that transforms the input inside the GEMM loader
                            (csrc/conv_wino_fused.hip): no workspace, no 16-plane tensor in HBM                          */
    const void *w_x3;    /* optional, precision 2 on a direct (non-Winograd) layer: w split into three bf16 planes in 16-deep
                            chunk-major order [Kpad/16][3][Cout][16] (a3d_split_bf16x3_chunk with outer = 1, rows = Cout,
                            cols = Kpad, chunk = 16).  With it, wide and large layers run the 256 x 256-tile kernel that
                            moves the weight planes global -> LDS by LDS-DMA (csrc/conv_bf16x3_wide.hip); results are
                            bit-identical to the kernel that splits w on the fly                                       */
    /* ---- precision 3 ("fp16x2"): fp32-grade products from an exact-to-2^-22 two-way fp16 split of each operand, THREE fp16 MFMAs
     * per k step (csrc/conv_bf16x3.hip).  fp16 has 5 exponent bits, so every operand row is scaled by a power of two taken from the
     * largest magnitude of ITS image / ROI -- a frame's result never depends on what else is in the batch:                        */
    const float *in_amax;  /* DEVICE [B]: max |x[b]| (an upper bound is fine) per input image b; required at precision 3            */
    const float *in_amax2; /* DEVICE [B] for source 1 (x2), or NULL                                                                 */
    float *y_amax;         /* optional DEVICE [B], zero-initialised by the caller: the launch raises y_amax[b] to max |y[b]| (atomic
                              max; any precision of the split-operand kernels) so that the next layer has its in_amax for free      */
    float w_scale;         /* precision 3: power of two that puts max |w| (Winograd layers: max |w_wino|) in [2^14, 2^15)           */
    float *wino_m;         /* optional, precision 3 Winograd layers: a3d_wino_m_bytes() of scratch.  With it SMALL problems (a few
                              workgroups in the one-launch form: single frames, the coarsest pyramid levels) run plane-split: 16 x as
                              many workgroups each compute ONE Winograd plane's product into wino_m [16][tiles][Cout], and a second
                              launch folds the planes in the same order and applies the epilogue -- the same bits, a 16th of the
                              per-workgroup latency                                                                                 */
    int io_bf16;           /* precision 1 only (the training step's autocast arithmetic): which tensors are STORED as bf16 in HBM
                              instead of fp32 -- bit 0: x, bit 1: y, bit 2: res, bit 3: gate (x / y / res / gate then point at bf16
                              data).  The kernel rounds its operands to bf16 anyway, so a bf16 x gives the same products; y is
                              rounded to nearest even on the way out.  BASELINE configs[4]: bf16 activations and gradients.       */
    /* ---- precision 3 with the ACTIVATIONS pre-split by their producer (round 4): x_h2 replaces x (x may then be NULL).  Layout
     * [B,H,W][Cin/16][h | l][16] fp16 -- per pixel (or per row of a linear layer) and 16-channel chunk the two fp16 planes of
     * x * s(b), s(b) = the power of two that puts in_amax[b] in [2^14, 2^15): exactly the bits the loaders of the split-operand
     * kernels compute from the fp32 tensor, and the same bytes (2 + 2 per element).  Producers: a3d_roi_align_fpn with out_h2 (the
     * workgroup that pools a ROI knows the ROI's maximum before it stores) and a3d_presplit_f16x2 (one pass over a finished tensor).
     * The wide direct kernel then moves BOTH operands global -> LDS by LDS-DMA through a 4-stage ring: no activation registers, no
     * split arithmetic and no VGPR -> LDS stores in its loop; results are bit-identical to the launch on the fp32 tensor. */
    const void *x_h2;      /* source 0 pre-split, or NULL                                                                            */
    const void *x2_h2;     /* source 1 pre-split (required with x_h2 when Cin2 > 0), or NULL                                        */
    /* ---- phase 5 only (round 4): the layer's output feeds NOTHING but a 3x3 pad-1 convolution to ONE channel (the depth head's
     * depth_pred behind deconv5, pkg/modeling/depth_net/depth_head.py:51,88).  With dot_w / dot_y set the launch does not store its
     * [B,2H,2W,C] output (y may be NULL; C = Cout / 4 must be 64); it stores, per output pixel, the nine dot products of the pixel's
     * channel vector with the nine taps' weights, dot_y [B][9][2H][2W], and a3d_tapsum9 adds the shifted planes: a 177 MB round trip
     * instead of the 1.26 GB one (64 frames), one launch less.  Equal to the two launches to fp32 rounding of the 576-term sum. */
    const float *dot_w;    /* [9][C] = the one-channel filter as [kh][kw][c], or NULL                                                */
    float *dot_y;          /* [B][9][2H][2W], or NULL                                                                                */
    /* ---- Winograd layers, precision 3 (round 5): the layer's transformed tiles are a SLICE of a larger V buffer -- its tiles occupy
     * [wino_t_off, wino_t_off + tiles) of `wino_t_total` tiles per (plane, chunk, h | l) run of `workspace`.  0 / 0 = the layer's own
     * buffer.  Several layers that apply the SAME filter to different maps (the RPN head's 3x3 conv over the pyramid levels,
     * SURVEY K5) then transform into one buffer and a3d_wino_gemm_levels multiplies all of it in ONE launch. */
    int wino_t_off, wino_t_total;
    /* ---- precision 1 (round 5): the filter ALSO as bf16, [Cout][Kpad] in w's layout, rounded to nearest even from w (the training step keeps
     * fp32 master weights and rounds its flat parameter buffer once per step: a3d_f32_to_bf16_scaled).  With it large launches run the
     * kernel that moves both operands global -> LDS by LDS-DMA on 256-pixel tiles (csrc/conv_bf16w.hip); the products are those of the
     * kernel that rounds w on the fly: bit-identical results.  NULL: that kernel.  HBM-bound 1x1 layers (Cin 32 / 128 / 256 / 512, residual and
     * gate stored bf16 or absent) run csrc/conv_bf16xs.hip with it: a wave's 32 pixels stay in registers as MFMA fragments while the
     * workgroup walks every output channel -- again the same bits. */
    const void *w_bf16;
} a3d_conv_desc;

size_t a3d_conv_workspace_bytes(const a3d_conv_desc *d);
/* bytes of a3d_conv_desc.wino_m if this launch would use it (0: the problem is large enough for the one-launch form, or not a
 * precision-3 Winograd layer) */
size_t a3d_wino_m_bytes(const a3d_conv_desc *d);
int a3d_conv2d_nhwc_f32(const a3d_conv_desc *d, void *stream);
/* The ResNet stem in ONE launch (round 4, fp16x2 arithmetic): 7x7 s2 p3 convolution + folded FrozenBN + ReLU + 3x3 s2 p1 max-pool
 * (detectron2 BasicStem.forward behind pkg/modeling/meta_arch/planercnn.py:150).  d as for a3d_conv2d_nhwc_f32 on the stem (stem = 1,
 * precision 3, w_x3, in_amax, w_scale, Ho x Wo = the CONV output size) except that d->y / d->y_amax are the POOLED tensor
 * [B, (Ho - 1) / 2 + 1, (Wo - 1) / 2 + 1, 64] and its maxima.  Bit-identical to the conv launch followed by a3d_maxpool3x3s2_nhwc.
 * A3D_ERR_UNSUPPORTED: not that layer / arithmetic (the caller runs the two launches). */
int a3d_stem_conv_pool(const a3d_conv_desc *d, void *stream);
/* y[b][oh][ow] = bias + sum over t = 3 kh + kw of g[b][t][oh + kh - 1][ow + kw - 1] (zero outside): the second half of a 3x3 pad-1
 * convolution to one channel whose per-pixel tap products a phase-5 launch stored (a3d_conv_desc.dot_y). */
int a3d_tapsum9(const float *g, float bias, float *y, int B, int H, int W, void *stream);
/* src [outer][rows][cols] fp32 (cols % 32 == 0) -> dst [outer][cols/32][3][rows][32] bf16 with src == hi + mid + lo exactly
 * (round-to-nearest-even at each level). */
int a3d_split_bf16x3(const float *src, void *dst, int outer, int rows, int cols, void *stream);
/* The same with `chunk`-deep column chunks (16 or 32; cols % chunk == 0): dst [outer][cols/chunk][3][rows][chunk]. */
int a3d_split_bf16x3_chunk(const float *src, void *dst, int outer, int rows, int cols, int chunk, void *stream);
/* The fp16x2 counterpart (precision 3): src * scale == hi + lo (fp16, to 2^-22 relative); dst [outer][cols/chunk][2][rows][chunk]. */
int a3d_split_f16x2_chunk(const float *src, void *dst, int outer, int rows, int cols, int chunk, float scale, void *stream);
/* Activation pre-split for a3d_conv_desc.x_h2: x [B][n] fp32 (n % 16 == 0: pixels x channels of image b, channels % 16 == 0) ->
 * dst [B][n/16][h | l][16] fp16 of x * s(b), s(b) from amax[b] as the kernels derive it (a3d_conv_desc.in_amax); amax2 optional
 * (second source of a channel concat: both share max(amax[b], amax2[b])).  One HBM-bound pass, same byte count in and out. */
int a3d_presplit_f16x2(const float *x, void *dst, const float *amax, const float *amax2, int B, size_t n, void *stream);
/* out[b] = max(out[b], max |x[b, 0..n)|) for b < B (out zero-initialised by the caller): the in_amax of a tensor that no kernel of
 * this library produced. */
int a3d_absmax_rows(const float *x, float *out, int B, size_t n, void *stream);

/* The two launches of the Winograd form, individually (a3d_conv2d_nhwc_f32 issues both when d->w_wino is set):
 * x (+x2) -> d->workspace = V[16][tiles][Cin+Cin2]   (HBM-bound), then V, d->w_wino -> y   (MFMA-bound). */
int a3d_wino_input_transform(const a3d_conv_desc *d, void *stream);
int a3d_wino_gemm(const a3d_conv_desc *d, void *stream);
/* ONE GEMM launch for n <= 5 Winograd layers (precision 3) that share filter (w_wino_x3, w_scale, scale, shift, act), channel counts and
 * `workspace`, whose slices [wino_t_off, wino_t_off + tiles) are consecutive and fill wino_t_total: the shared-filter RPN conv over the
 * pyramid levels (StandardRPNHead, planercnn.py:168) as one launch over the concatenated tiles, with a per-level table for the output
 * maps, the per-image scales and the recorded maxima.  The partial rounds of the small levels vanish (p3-p6 at 64 frames: 5 + 2 + 1 + 1
 * rounds of the chip -> 6.25) and every output element is computed exactly as by its own launch: bit-identical. */
int a3d_wino_gemm_levels(const a3d_conv_desc *levels, int n, void *stream);
/* Back-to-back pointwise pair across a ResNet bottleneck boundary (round 6; detectron2 BottleneckBlock reached from
 * pkg/modeling/meta_arch/planercnn.py:29,150): d1 = conv3 + FrozenBN + residual + ReLU of block i, d2 = conv1 + FrozenBN + ReLU of block i + 1,
 * ONE launch.  d1 as for a3d_conv2d_nhwc_f32 on the activation-stationary fp16x2 kernel (1x1 stride 1, precision 3, w_x3, in_amax, w_scale,
 * res set, Cin 64 or 128); d2: 1x1 stride 1 over d1's output (Cin = d1->Cout, Cout 64 / 128 for d1->Cin 64 / 128), precision 2 with
 * w_x3 = its bf16x3 planes [Cin/16][3][Cout][16] (a3d_split_bf16x3_chunk), scale / shift / act, y, optional y_amax; no residual.
 * d1->y (and d1->y_amax) are stored exactly as the single launch stores them; d2->y equals a3d_conv2d_nhwc_f32(d2) on that tensor bit for
 * bit -- without reading it back from HBM.  A3D_ERR_UNSUPPORTED: not such a pair (issue the two launches). */
int a3d_conv_b2b(const a3d_conv_desc *d1, const a3d_conv_desc *d2, void *stream);
/* The kernel instantiation the LAST conv launch of the calling thread dispatched, as it appears in a rocprofv3 kernel trace
 * ("conv_pw_kernel<2,2,16> 128x128 persistent", "wino_gemm_kernel<1,32>", ...); "" before the first launch.  For measurement
 * code: launches are labelled with what the dispatcher did, not with a host-side copy of its selection rules.
 * This label is the library's one piece of state beside the per-device "dynamic LDS opted in" bits: a thread_local buffer, written and
 * read by the calling thread only (re-entrancy is not affected; it carries no configuration). */
const char *a3d_last_conv_variant(void);

/* Max-pool 3x3 stride 2 pad 1 (ResNet stem) and kernel-1 stride-2 pool (FPN LastLevelMaxPool = p6). */
int a3d_maxpool3x3s2_nhwc(const float *x, float *y, int B, int H, int W, int C, void *stream);
int a3d_subsample2_nhwc(const float *x, float *y, int B, int H, int W, int C, void *stream);

/* Bilinear resize, align_corners=False (depth_head.py:82,88). */
int a3d_resize_bilinear_nhwc(const float *x, float *y, int B, int H, int W, int C, int Ho, int Wo, void *stream);

/* 3x3 pad-1 convolution to ONE output channel (depth_head.py:68 depth_pred). w: [3][3][C], y: [B,H,W]. */
int a3d_conv3x3_to1_nhwc(const float *x, const float *w, float bias, float *y, int B, int H, int W, int C,
                         void *stream);

/* ------------------------------------------------------------------------------------------------
 * RPN proposal selection.  Replaces detectron2 RPN.predict_proposals / find_top_rpn_proposals reached
 * from planercnn.py:168 (anchor grid, Box2BoxTransform.apply_deltas, per-level top-k, clip, non-empty
 * filter, batched_nms with level as category, first post_topk): SURVEY.md A.4-A.6.
 * head[l]: [B, Hf, Wf, CH] fp32, channels [0,A) objectness logits, [A,5A) deltas (a*4+coord).
 * Outputs are fixed-size: slots >= out_count[b] are zero boxes / level -1.
 * ---------------------------------------------------------------------------------------------- */
typedef struct a3d_rpn_desc {
    const float *head[5];
    int Hf[5], Wf[5], stride[5];
    float cell_anchors[5][3][4]; /* per level, per anchor: x1,y1,x2,y2 around (0,0) */
    int B, L, A, CH;
    int img_h, img_w;
    int pre_topk, post_topk;
    float nms_thresh, min_size;
    float weights[4];
    float scale_clamp;
    void *workspace;    /* a3d_group_buffers_bytes(B*L) */
    float *out_boxes;   /* [B, post_topk, 4] xyxy */
    float *out_scores;  /* [B, post_topk] objectness logits */
    int *out_level;     /* [B, post_topk] */
    int *out_pos;       /* [B, post_topk] (level << 10) | rank-in-level : tie-break position */
    int *out_count;     /* [B] */
} a3d_rpn_desc;

size_t a3d_group_buffers_bytes(int n_groups);
/* bytes of a3d_rpn_desc.workspace: pre_topk <= 1024 (inference) equals a3d_group_buffers_bytes(B*L); up to 2048 (the
 * training configuration's PRE_NMS_TOPK_TRAIN 2000) adds the global suppression words of the 2048-candidate NMS. */
size_t a3d_rpn_workspace_bytes(int B, int L, int pre_topk);
int a3d_rpn_proposals(const a3d_rpn_desc *d, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fast R-CNN box inference.  Replaces FastRCNNOutputLayers.inference reached from
 * pkg/modeling/roi_heads/roi_heads.py:206 (softmax, per-class apply_deltas, clip, score > thresh,
 * batched_nms with class as category, first topk): SURVEY.md A.8.
 * pred: [B*R, CH]: channels [0,C] class logits (C = background), [C+1, C+1+4C) per-class deltas.
 * ---------------------------------------------------------------------------------------------- */
typedef struct a3d_boxdet_desc {
    const float *pred;
    const float *prop_boxes; /* [B, R, 4] */
    const int *prop_count;   /* [B] */
    int B, R, C, CH;
    int img_h, img_w;
    float score_thresh, nms_thresh;
    int topk;
    float weights[4];
    float scale_clamp;
    void *workspace;   /* a3d_group_buffers_bytes(B*C) */
    float *out_boxes;  /* [B, topk, 4] */
    float *out_scores; /* [B, topk] */
    int *out_classes;  /* [B, topk] (-1 in unused slots) */
    int *out_pos;      /* [B, topk] proposal_row*C + class */
    int *out_count;    /* [B] */
} a3d_boxdet_desc;

int a3d_box_detections(const a3d_boxdet_desc *d, void *stream);

/* Greedy NMS over caller-sorted groups of <= 1024 boxes each (torchvision.ops.nms semantics, strict >).
 * g_boxes [G,1024,4] score-descending, g_valid [G,1024], g_n [G] -> g_keep [G,1024]. */
int a3d_group_nms(const float *g_boxes, const int *g_valid, const int *g_n, int *g_keep, int n_groups, float thresh,
                  void *stream);

/* ------------------------------------------------------------------------------------------------
 * ROIPooler: FPN level assignment + ROIAlign.  Replaces detectron2 ROIPooler / torchvision roi_align
 * built at pkg/modeling/roi_heads/roi_heads.py:50-55,74-79 and called at :185,:236,:250,:268
 * (SURVEY.md A.7).  feat[l]: NHWC [B,Hf,Wf,C]; boxes [B,R,4]; count[b] live boxes of image b (NULL = R).
 * Output row of box (b, r) = (row_offset ? row_offset[b] : b*R) + r, layout [row, P, P, C].
 * ---------------------------------------------------------------------------------------------- */
typedef struct a3d_roialign_desc {
    const float *feat[4];
    int Hf[4], Wf[4];
    float scale[4];
    int L, C;
    const float *boxes;
    const int *count;
    const int *row_offset;
    int B, R;
    int P, sampling_ratio, aligned;
    float *out;
    int *out_level; /* optional [rows] */
    int *order_ws;  /* optional [B*R] int32 workspace (R <= 1024): the library first sorts every image's live boxes by (level, y, x)
                       into it and walks them in that order, with each XCD taking whole images, so that workgroups running
                       side by side touch neighbouring feature cells (L2 hits instead of re-fetches).  The OUTPUT rows do not
                       move and the values are bit-identical to the unsorted walk: only the schedule changes. */
    float *out_amax; /* optional [rows]: max |pooled[row]| over the row's finite values, written by the workgroup that pools the
                        row (no atomics; rows of dead slots are not touched).  It is a3d_conv_desc.in_amax of the layers that consume
                        the pooled rows: every ROI is scaled by ITS OWN maximum (round 2 used a3d_roi_amax's level-wide bound, which
                        scaled a faint ROI by the hottest cell of its image). */
    const float *level_amax[4]; /* optional, with window_count: per level [B] = max |feat[l][b]| */
    int *window_count; /* optional: += number of live ROIs whose own maximum is below 2^-16 of their level's.  The backbone's per-image
                          block exponents leave every feature with an absolute error of ~2^-40 of its level's maximum; relative to
                          such a ROI that exceeds fp32's own rounding.  The default arithmetic counts these ROIs instead of passing
                          them silently (0 on every frame of the test and bench clips; A3D_PRECISION=2 is the remedy). */
    void *out_h2;      /* optional, INSTEAD of out (round 4; C == 256, P*P*C*4 <= 120 KiB: the 7x7 box pooler): the pooled rows pre-split
                          for the fp16x2 GEMM that consumes them, [rows][P*P*C/16][h | l][16] fp16 of pooled * s(row), s(row) = the
                          power of two that puts out_amax[row] in [2^14, 2^15) -- a3d_conv_desc.x_h2 of the box head's fc1
                          (roi_heads.py:185-187).  The workgroup keeps the row in LDS until its maximum is known.  Same pooled
                          values as `out` (the split is exact to 2^-22 of the row's maximum). */
    int serial;        /* walk of a ROI's cells.  0 = bin by bin, a bin's loads issued together.  1 = bin by bin, one load at a time (the
                          same bits as 0: schedule only; kept for the bit-equality test and tools/roi_bench.py).  2 (round 6; P = 7,
                          C = 256, `out`) = the rolling-window walk: a wave owns two adjacent bin rows and walks the ROI's cell
                          columns once -- 675 instead of 980 cell loads per typical ROI -- summing in (column, row) order: equal to
                          0 / 1 to fp32 rounding, not bit for bit; ROIs whose geometry does not fit (a bin without samples, bins
                          narrower than a cell, more than 8 cell rows per bin-row pair) take walk 0 -- a function of the ROI alone.
                          The Python layer passes 2 for the 7x7 box pooler (ops.ROI_ROLLING). */
} a3d_roialign_desc;

int a3d_roi_align_fpn(const a3d_roialign_desc *d, void *stream);

/* out[row of (b, r)] = max over the L levels of level_amax[l][b], for the live boxes r < count[b]: an upper bound of max |pooled[row]|
 * (rows as in a3d_roi_align_fpn).  Superseded on the detection path by a3d_roialign_desc.out_amax (the ROI's own maximum); kept for
 * callers that pool with their own kernel. */
int a3d_roi_amax(const float *const level_amax[4], int L, const int *count, const int *row_offset, int B, int R, float *out, void *stream);

/* offsets[b] = sum_{i<b} min(count[i], cap); offsets[B] = total.  (compacts ragged per-image ROI lists) */
int a3d_count_offsets(const int *count, int *offsets, int B, int cap, void *stream);

/* Skinny output layer: y[m, 0:N] = x[m, :] . w[n, :] + bias[n], N <= 8, with the first norm_n outputs
 * L2-normalised (F.normalize eps 1e-12) and/or sigmoid.  plane_head.py:80-82, axis_head.py:106-107,120,
 * Mask R-CNN predictor + sigmoid (roi_heads.py:237). */
int a3d_linear_small(const float *x, const float *w, const float *bias, float *y, int M, const int *m_dev, int K,
                     int N, int norm_n, int sigmoid, void *stream);

/* ------------------------------------------------------------------------------------------------
 * detector_postprocess + paste_masks_in_image + override_depth fused.  Replaces
 * pkg/modeling/postprocessing.py:47-69, pkg/layers/mask_ops.py:41-60,128-129 and
 * pkg/utils/arti_vis.py:90-99,125-149.  One slot per (image b, detection r < R).
 * ---------------------------------------------------------------------------------------------- */
typedef struct a3d_paste_desc {
    const float *boxes;      /* [B,R,4] pred_boxes */
    const float *scores;     /* [B,R] */
    const int *count;        /* [B] */
    const int *row_offset;   /* [B] compact row of (b,0) in mask_prob / normals */
    const float *mask_prob;  /* [rows, MS, MS] sigmoid mask probabilities */
    const float *normals;    /* [rows, 3] pred_plane unit normals, or NULL */
    const float *depth;      /* [B,H,W] or NULL */
    int B, R, MS, H, W;
    float post_score_thresh; /* 0.1, planercnn.py:217 */
    float mask_thresh;       /* 0.5, MODEL.ROI_MASK_HEAD.MASK_THRESHOLD */
    float focal, cx, cy;     /* 571.623718, 319.5, 239.5: arti_vis.py:101-104 */
    int clip_boxes;          /* 1: Boxes.clip to the image first (detector_postprocess, postprocessing.py:58);
                                0: paste with the boxes as given (bare paste_masks_in_image, mask_ops.py:68) */
    unsigned char *masks;    /* [B,R,H,W] 0/1 or NULL (mask never materialised) */
    float *planes;           /* [B,R,3] normal * offset (process()'s pred_plane) */
    int *area;               /* [B,R] */
    int *keep;               /* [B,R] survives score>=thresh and non-empty clip */
    float *out_boxes;        /* [B,R,4] clipped */
} a3d_paste_desc;

int a3d_paste_lsq(const a3d_paste_desc *d, void *stream);

/* Plane-offset least squares from dense masks of ONE image: PlaneRCNN_Branch.override_depth
 * (pkg/utils/arti_vis.py:125-149).  depth [H,W], masks [D,H,W] (non-zero = inside), normals [D,3] -> out [D,3]. */
int a3d_plane_offset_dense(const float *depth, const float *masks, const float *normals, float *out, int D, int H,
                           int W, float focal, float cx, float cy, void *stream);
/* The same with the pasted masks as they are (uint8 / bool bytes, non-zero = set): no float copy of the masks.  Identical results. */
int a3d_plane_offset_dense_u8(const float *depth, const unsigned char *masks, const float *normals, float *out, int D, int H, int W,
                              float focal, float cx, float cy, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fixed-size detection records = payload of the frame-sharded all-gather.  One record
 * carries the fields create_instances() builds (pkg/utils/arti_vis.py:162-186):
 *   [0:4] box xyxy  [4] score  [5] class  [6:9] plane normal*offset  [9:12] rot axis  [12:14] tran axis
 *   [14:14+MS*MS] soft mask.   Kept detections of image b are compacted to records[b, 0:rec_count[b]].
 * ---------------------------------------------------------------------------------------------- */
typedef struct a3d_pack_desc {
    const float *boxes;     /* [B,R,4] clipped boxes (a3d_paste_lsq out_boxes) */
    const float *scores;    /* [B,R] */
    const int *classes;     /* [B,R] */
    const int *count;       /* [B] */
    const int *row_offset;  /* [B] */
    const int *keep;        /* [B,R] */
    const float *planes;    /* [B,R,3] or NULL */
    const float *rot_axis;  /* [rows,3] or NULL */
    const float *tran_axis; /* [rows,2] or NULL */
    const float *mask_prob; /* [rows,MS,MS] or NULL */
    int B, R, MS;
    float *records;         /* [B,R,a3d_record_floats(MS)] */
    int *rec_count;         /* [B] */
} a3d_pack_desc;

int a3d_record_floats(int MS);
int a3d_detections_pack(const a3d_pack_desc *d, void *stream);

/* COCO run-length encoding of pasted masks: the `segmentation` field of the reference's per-frame record
 * (pkg/utils/arti_vis.py:66-67 -> instances_to_coco_json -> pycocotools mask.encode; decoded again at :179-186).  masks [D,H,W] uint8
 * (non-zero = set), H * W <= 393 216.  pos[d, k] = the k-th COLUMN-major index i >= 1 with m[i] != m[i-1] (increasing), count[d] = how many
 * there are (only the first `cap` are stored: re-run with cap >= max count if it was exceeded), first[d] = m[0] != 0.  The run
 * lengths are the differences of {0, pos..., H*W}, with a leading 0 when first is set (cocoapi's convention). */
int a3d_mask_rle(const unsigned char *masks, int D, int H, int W, int cap, int *pos, int *count, int *first, void *stream);

/* ================================================================================================
 * Training step (BASELINE configs[4]: config/step1_bbox.yaml).
 * Replaces, under autograd, the training branch of PlaneRCNN.forward (pkg/modeling/meta_arch/planercnn.py:83-123),
 * the box branch of PlaneRCNNROIHeads.forward (pkg/modeling/roi_heads/roi_heads.py:93-117,190-204) and the SGD step
 * of the detectron2 trainer behind tools/train_net.py:84-117.  Data gradients of convolutions reuse
 * a3d_conv2d_nhwc_f32 with a3d_weight_transpose'd filters and the `gate` epilogue (ReLU backward).
 * ============================================================================================== */

/* dw[co][kh][kw][ci] (=/+=) scale[co] * sum_pixels dy[b,oh,ow,co] * x[b, oh*stride+kh-pad, ow*stride+kw-pad, ci]
 * (autograd's weight gradient of every Conv2d / Linear on the trainable path; dw has the packed forward layout). */
typedef struct a3d_wgrad_desc {
    const float *x;     /* forward input NHWC [B,H,W,Cin]                    */
    const float *dy;    /* gradient of the layer output [B,Ho,Wo,Cout]       */
    const float *scale; /* optional [Cout]: folded FrozenBN scale that followed the conv in the forward pass */
    float *dw;          /* [Cout][KH][KW][Cin]                               */
    float *workspace;   /* a3d_wgrad_workspace_bytes()                       */
    int B, H, W, Cin, Ho, Wo, Cout; /* Cin % 4 == 0, Cout % 4 == 0           */
    int KH, KW, stride, pad;
    int splitk;         /* >= 1 pixel slices (summed in slice order: deterministic) */
    int accumulate;     /* 1: dw += (weights shared by several call sites, e.g. the RPN head over 5 levels) */
    int precision;      /* 0: fp32 MFMA; 1: bf16 MFMA, fp32 accumulation; 2: fp32-grade 3-way bf16 split (see a3d_conv_desc.precision) */
    int io_bf16;        /* precision 1 only: bit 0: x is stored as bf16, bit 1: dy is stored as bf16 (dw stays fp32)              */
    int defer_reduce;   /* 1: launch the partial-sum kernel only; the caller keeps `workspace` alive and folds the slices of MANY layers
                           in one launch later (a3d_wgrad_reduce_batch).  Same sums in the same order: bit-identical dw.  Not with
                           accumulate (a chain of launches into one dw is ordered by its reduces).                                  */
} a3d_wgrad_desc;
size_t a3d_wgrad_workspace_bytes(const a3d_wgrad_desc *d);
int a3d_conv_wgrad_nhwc_f32(const a3d_wgrad_desc *d, void *stream);
/* Workgroup tiles of ONE pixel slice and the length of the pixel reduction of the kernel form the library runs this descriptor on (round 4: the
 * bf16 arithmetic has two forms -- csrc/conv_wgrad_tr.hip takes the stride-1 3x3 pad-1 and 1x1 layers with three taps / 256 input channels per
 * workgroup): what a caller sizes `splitk` by.  Returns the form (0: first form, 1 / 3: taps per workgroup of the second) or a negative error. */
int a3d_wgrad_tiles(const a3d_wgrad_desc *d, int *tiles, int *reduction);
/* The slice reduction of n deferred weight-gradient launches in ONE launch (round 4: at the reference's 2 images per GPU the 63 per-layer
 * reduce launches of a training step were 17 us of latency each, 11 % of the step).  table: DEVICE array of n a3d_wgrad_desc whose
 * workspace / scale / dw / Cout / KH / KW / Cin / splitk fields are read (x, dy are not).  Every layer: dw = scale * sum over slices
 * in slice order -- the arithmetic of the per-launch reduce.  tools/train_net.py:84-104 (the backward of DDP's step). */
int a3d_wgrad_reduce_batch(const a3d_wgrad_desc *table, int n, void *stream);

/* wt[ci][KH-1-kh][KW-1-kw][co] = scale[co] * w[co][kh][kw][ci]: the filter of the data gradient. */
int a3d_weight_transpose(const float *w, const float *scale, float *wt, int Cout, int KH, int KW, int Cin, void *stream);
/* The same for n filters in ONE launch: table = DEVICE array of n items.  The data-gradient filters of every trainable layer are
 * re-derived each step: 50 launches of ~6 us each at 2 images per GPU. */
typedef struct a3d_transpose_item {
    const float *w, *scale;
    float *wt;
    int Cout, KH, KW, Cin;
    int block0; /* first 32 x 32 x tap block of this filter in the launch's flat block index (exclusive prefix sum, filled by the caller) */
    int pad_;
} a3d_transpose_item;
int a3d_weight_transpose_batch(const a3d_transpose_item *table, int n, int total_blocks, void *stream);
/* U = G g G^T of a packed 3x3 filter [Cout][3][3][Cin] -> [16][Cout][Cin] (a3d_conv_desc.w_wino), on the device. */
int a3d_wino_weight_transform(const float *w, float *U, int Cout, int Cin, void *stream);

/* y [B,Ho,Wo,C] (=/+=) x [B,ceil(Ho/2),ceil(Wo/2),C] at even pixels, 0 elsewhere (backward of a3d_subsample2_nhwc and
 * of the input sub-sampling of a stride-2 1x1 convolution). */
int a3d_zero_insert2_nhwc(const float *x, float *y, int B, int H, int W, int C, int Ho, int Wo, int accumulate, void *stream);
/* y [B,H,W,C] += 2x2 block sums of x [B,2H,2W,C] (backward of the nearest-x2 upsampling of the FPN top-down path). */
int a3d_sumpool2_add_nhwc(const float *x, float *y, int B, int H, int W, int C, void *stream);
/* out[c] (=/+=) sum_m dy[m,c]  (bias gradients). */
size_t a3d_colsum_workspace_bytes(int C);
int a3d_colsum(const float *dy, float *out, float *workspace, int M, int C, int accumulate, void *stream);
/* dy stored as bf16 (the bf16 training step keeps its activation gradients that way): the same sums, in the same order, as a3d_colsum on
 * the widened values.  C % 4 == 0. */
int a3d_colsum_bf16(const void *dy, float *out, float *workspace, int M, int C, int accumulate, void *stream);

/* Gradient of a3d_roi_align_fpn with respect to the pyramid: dfeat[level] += scatter(dout).  dfeat must hold the
 * gradient accumulated so far (or zeros).  Adaptive sampling (sampling_ratio 0) only: the box pooler. */
typedef struct a3d_roialign_bwd_desc {
    float *dfeat[4];
    int Hf[4], Wf[4];
    float scale[4];
    int L, C;
    const float *boxes;    /* [B,R,4] */
    const int *count;      /* [B] or NULL */
    const int *row_offset; /* [B] or NULL */
    int B, R, P, sampling_ratio, aligned;
    const float *dout;     /* [rows,P,P,C] */
} a3d_roialign_bwd_desc;
int a3d_roi_align_fpn_backward(const a3d_roialign_bwd_desc *d, void *stream);
/* The same gradient WITHOUT atomics (round 4): the pyramid in 8 x 8-cell tiles, a bit per (image, tile, ROI slot) marked by a first launch,
 * one workgroup per tile that sums its ROIs in slot order in LDS and adds the tile to dfeat once -- bit-reproducible, ~8 x fewer bytes
 * to memory.  workspace: a3d_roi_align_bwd_workspace_bytes(d) bytes (zeroed by the call). */
size_t a3d_roi_align_bwd_workspace_bytes(const a3d_roialign_bwd_desc *d);
int a3d_roi_align_fpn_backward_gather(const a3d_roialign_bwd_desc *d, void *workspace, void *stream);

/* detectron2 Matcher over pairwise_iou(gt, boxes) (RPN.label_and_sample_anchors, ROIHeads.label_and_sample_proposals):
 * matched_idx[b,i] = argmax_g IoU (first maximum), label[b,i] = labels[k] for IoU in [thresholds[k-1], thresholds[k]),
 * and with allow_low_quality every box attaining some gt's best IoU gets label 1.  Images without gt: labels[0]. */
typedef struct a3d_match_desc {
    const float *boxes;     /* [*, N, 4]; image b reads boxes + b*box_batch_stride*4 (stride 0: shared anchors) */
    const int *box_count;   /* [B] live boxes per image or NULL (= N) */
    const float *gt_boxes;  /* [B, Gmax, 4] */
    const int *gt_count;    /* [B] */
    int B, N, Gmax, box_batch_stride;
    float thresholds[2];
    int labels[3];
    int n_thresholds;       /* 1 or 2 */
    int allow_low_quality;
    unsigned int *gt_best;  /* scratch [B, Gmax] (allow_low_quality only) */
    int *matched_idx;       /* [B, N] */
    signed char *label;     /* [B, N] */
    float *matched_iou;     /* [B, N] or NULL */
} a3d_match_desc;
int a3d_match_boxes(const a3d_match_desc *d, void *stream);

/* detectron2 subsample_labels on the device (RPN._subsample_labels): out[b,i] = 1 / 0 for a uniformly random subset of
 * at most max_pos positives (label 1) and negatives (label 0) up to `num` in total, -1 elsewhere.  Counter-based RNG:
 * the subset is a function of (seed, b, i) only.  N < 2^17. */
int a3d_sample_labels(const signed char *labels, signed char *out, int B, int N, int num, int max_pos, unsigned long long seed,
                      void *stream);

/* add_ground_truth_to_proposals: out [B, R+Gmax, 4] = [live proposals | ground truth | zeros], out_count = both counts. */
int a3d_append_gt_boxes(const float *props, const int *count, const float *gt, const int *gt_count, float *out, int *out_count,
                        int B, int R, int Gmax, void *stream);

/* ROIHeads._sample_proposals + the gathers that follow it (roi_heads.py:93-95 -> detectron2 label_and_sample_proposals):
 * class of box i = gt_classes[matched_idx[i]] if match_label[i] == 1 else num_classes (background; also when the image
 * has no ground truth); keeps a random subset of <= max_fg foreground boxes and background boxes up to `num`;
 * outputs are fixed-size [B, num] (foreground first), slots >= out_count[b] are zero boxes of the background class. */
typedef struct a3d_roi_sample_desc {
    const float *boxes;        /* [B, N, 4] (a3d_append_gt_boxes output) */
    const int *box_count;      /* [B] */
    const float *gt_boxes;     /* [B, Gmax, 4] */
    const int *gt_classes;     /* [B, Gmax] */
    const int *gt_count;       /* [B] */
    const int *matched_idx;    /* [B, N] (a3d_match_boxes) */
    const signed char *match_label; /* [B, N] */
    int B, N, Gmax, num_classes, num, max_fg;
    unsigned long long seed;
    float *out_boxes;          /* [B, num, 4] */
    float *out_gt_boxes;       /* [B, num, 4] */
    int *out_classes;          /* [B, num] */
    int *out_index;            /* [B, num] index into boxes, -1 for unused slots */
    int *out_count;            /* [B] */
} a3d_roi_sample_desc;
int a3d_sample_rois(const a3d_roi_sample_desc *d, void *stream);

/* RPN.losses + its gradient with respect to the head outputs (layout of a3d_rpn_desc.head).  loss[0] = loss_rpn_cls,
 * loss[1] = loss_rpn_loc.  labels: -1 ignore, 0 negative, 1 positive AFTER sub-sampling; anchor order: level-major,
 * then (y, x, a). */
typedef struct a3d_rpn_loss_desc {
    const float *head[5];
    float *dhead[5];
    int Hf[5], Wf[5], stride[5];
    float cell_anchors[5][3][4];
    int B, L, A, CH, Atotal, Gmax;
    const signed char *labels; /* [B, Atotal] */
    const int *matched_idx;    /* [B, Atotal] */
    const float *gt_boxes;     /* [B, Gmax, 4] */
    float weights[4];          /* Box2BoxTransform weights (1,1,1,1) */
    float normalizer;          /* RPN.BATCH_SIZE_PER_IMAGE * images */
    float *workspace;          /* a3d_loss_workspace_bytes() */
    float *loss;               /* [2] */
} a3d_rpn_loss_desc;
size_t a3d_loss_workspace_bytes(void);
int a3d_rpn_loss(const a3d_rpn_loss_desc *d, void *stream);

/* FastRCNNOutputLayers.losses + gradient with respect to the fused predictor output rows:
 * pred[r] = [K+1 class scores (K = background) | 4K deltas, class-major | padding up to pitch].
 * loss[0] = loss_cls (mean cross entropy), loss[1] = loss_box_reg (L1 on foreground rows / M). */
typedef struct a3d_box_loss_desc {
    const float *pred;     /* [M, pitch] */
    float *dpred;          /* [M, pitch] */
    const int *gt_classes; /* [M] in [0, K] */
    const float *boxes;    /* [M,4] sampled proposal boxes */
    const float *gt_boxes; /* [M,4] matched ground truth (unused for background rows) */
    int M, num_classes, pitch;
    float weights[4];      /* (10,10,5,5) */
    float *workspace;      /* a3d_loss_workspace_bytes() */
    float *loss;           /* [2] */
    const int *count;      /* optional [M / R]: rows are R per image and only the first count[b] of image b are live; the
                              others get zero gradient and the losses are normalised by the live row count.  NULL: all live */
    int R;
} a3d_box_loss_desc;
int a3d_box_loss(const a3d_box_loss_desc *d, void *stream);

/* torch.optim.SGD (momentum, weight decay) over a flat buffer; n % 4 == 0.
 * d = grad_scale*g + wd*p;  buf = first ? d : momentum*buf + d;  p -= lr*buf */
int a3d_sgd_momentum(float *p, const float *g, float *buf, size_t n, float lr, float momentum, float wd, float grad_scale,
                     int first, void *stream);
/* The same update with the gradient read as bf16 (the all-reduced bf16 payload where the collective left it: tools/train_net.py:110's
 * DDP with bf16_compress_hook widens into .grad first; widening is exact, so the result equals a3d_bf16_to_f32 + a3d_sgd_momentum bit
 * for bit, minus one pass over the flat buffer). */
int a3d_sgd_momentum_bf16g(float *p, const void *g_bf16, float *buf, size_t n, float lr, float momentum, float wd, float grad_scale,
                           int first, void *stream);

/* bf16 payload of the data-parallel gradient all-reduce (BASELINE configs[4]; torch DDP's bf16_compress_hook behind
 * tools/train_net.py:110): dst_bf16[i] = bf16(src[i] * scale) (round to nearest even; scale = 1 / world size), and the widening
 * back after the collective.  n % 4 == 0. */
int a3d_f32_to_bf16_scaled(const float *src, void *dst_bf16, size_t n, float scale, void *stream);
int a3d_bf16_to_f32(const void *src_bf16, float *dst, size_t n, void *stream);

/* ================================================================================================
 * Hypothesis sweeps of the temporal optimiser.  Replace the per-hypothesis / per-frame Python loops
 * inside optimize_planes_3dc and optimize_planes_3d_trans (pkg/utils/opt_utils.py:400-476, 540-611, 700-768, 838-905):
 * lift a mask onto its plane (get_pcd, pkg/utils/vis.py:86-102), apply every rigid hypothesis, re-project
 * (project2D, vis.py:62-83) into one binary mask per hypothesis, IoU against the tracked detections' masks.
 * Masks are bit-packed: words = ceil(H*W/32), bit i of word w = pixel 32w+i (row-major).
 * ============================================================================================== */
#define A3D_SWEEP_MAX_HYP 64
int a3d_masks_pack_bits(const unsigned char *masks /*[n,H,W], non-zero = set*/, unsigned int *bits /*[n,words]*/, int n, int H, int W,
                        void *stream);
int a3d_masks_unpack_bits(const unsigned int *bits, unsigned char *masks /*[n,H,W] 0/1*/, int n, int H, int W, void *stream);

typedef struct a3d_sweep_desc {
    const unsigned char *mask; /* [H,W] source mask (pred_mask.nonzero())                                     */
    int H, W;
    float normal[3];           /* unit plane normal in the optimiser's camera frame ((a,b,c) -> (a,-c,b) swap)  */
    float offset;              /* plane offset = |plane|                                                     */
    double focal, cx, cy;      /* vis.py intrinsics: 517.97, W/2, H/2                                          */
    float pivot[3];            /* point the rotations turn about (a 3-D point of the axis); 0 for translations */
    const float *xforms;       /* [A][12]: R (row-major 3x3) then t; point' = R (point - pivot) + pivot + t      */
    int A;                     /* hypotheses, <= A3D_SWEEP_MAX_HYP                                             */
    unsigned int *out_bits;    /* [A, words], cleared by the call                                              */
} a3d_sweep_desc;
int a3d_project_hypotheses(const a3d_sweep_desc *d, void *stream);

/* iou[f*A + a] = |target_f & proj_a| / |target_f | proj_a|  (opt_utils.py:470-475). */
int a3d_mask_iou_matrix(const unsigned int *target_bits /*[F,words]*/, const unsigned int *proj_bits /*[A,words]*/, float *iou, int F,
                        int A, int H, int W, void *stream);

#ifdef __cplusplus
}
#endif
#endif
