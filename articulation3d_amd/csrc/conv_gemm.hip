// Dispatcher of the conv / linear entry point (a3d_conv2d_nhwc_f32, include/a3d.h): argument checks, the workspace size, the record of
// the kernel variant each launch took, and the routing of a descriptor to its kernel family by arithmetic (a3d_conv_desc.precision) and
// layer kind.  The kernels live in conv_gemm_v2.hip (fp32-input MFMA, direct), conv_pw.hip (persistent pointwise), conv_wino*.hip
// (Winograd), conv_bf16.hip (autocast arithmetic), conv_bf16x3*.hip / conv_xs_h2.hip (split-operand arithmetics).
//
// Round 4: the round-1 general kernel (`conv_gemm_kernel`, pointer-addressed gather, tune 1) that used to live here is gone.  The GPU
// suite's dispatcher log showed no descriptor of the detection or training path reaching it -- conv_gemm_v2 refuses only tensors of
// 4 GiB and more (ops.conv2d runs those as blocks of images), a channel concat of unequal widths and non-stem filters with more than
// 32 taps, none of which the reference's architecture contains -- and a kernel nothing runs is a parity surface nothing checks.
// Such a descriptor now gets A3D_ERR_UNSUPPORTED.
#include "conv_common.h"
#include <stdarg.h>
#include <stdio.h>

static int conv_check(const a3d_conv_desc *d) {
    if (!d || (!d->x && !d->x_h2) || !d->w || (!d->y && !d->dot_y)) return A3D_ERR_ARG;
    if ((d->dot_w || d->dot_y) && d->phase != 5) return A3D_ERR_ARG;  // the tap-product epilogue belongs to the fused four-phase form
    if (d->x_h2 && d->precision != 3) return A3D_ERR_ARG;  // pre-split activations exist in the fp16x2 arithmetic only
    if (d->B <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cout <= 0) return A3D_ERR_ARG;
    if ((d->Cout & 3) || (d->Kpad & 31) || d->splitk < 1) return A3D_ERR_ARG;
    if (d->stem) {
        if (d->KH != 7 || d->KW != 7 || d->stride != 2 || d->pad != 3 || d->Kpad != 224 || d->Cin2 || d->ups) return A3D_ERR_ARG;
    } else {
        if ((d->Cin & 31) || (d->Cin2 & 31) || (d->Cin2 && !d->x2 && !d->x2_h2)) return A3D_ERR_ARG;
        if (d->Kpad < d->KH * d->KW * (d->Cin + d->Cin2)) return A3D_ERR_ARG;
    }
    if (d->pixshuf && (d->Cout & 15)) return A3D_ERR_ARG;
    if (d->phase < 0 || d->phase > 5 || (d->phase == 5 && d->precision != 3 && d->precision != 2)) return A3D_ERR_ARG;  // (5: the fused four-phase form: the split-operand arithmetics)
    if (d->gate && (d->pixshuf || d->phase)) return A3D_ERR_ARG;
    if (d->io_bf16 && d->precision != 1) return A3D_ERR_ARG;  // bf16 storage belongs to the bf16 (autocast) arithmetic
    if (d->splitk > 1 && !d->workspace) return A3D_ERR_ARG;
    if ((size_t)d->B * d->H * d->W >= ((size_t)1 << 31)) return A3D_ERR_UNSUPPORTED;
    return A3D_OK;
}

static thread_local char g_last_variant[160] = "";
void a3d_note_variant(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_variant, sizeof(g_last_variant), fmt, ap);
    va_end(ap);
}
extern "C" const char *a3d_last_conv_variant(void) { return g_last_variant; }

extern "C" size_t a3d_conv_workspace_bytes(const a3d_conv_desc *d) {
    if (!d) return 0;
    if (d->tune == 0 && a3d_wino_fused_eligible(d)) return 0;
    if ((d->tune == 0 || d->tune == 7 || d->tune == 8 || (d->tune >= 23 && d->tune <= 25) || d->tune >= 200) && a3d_wino_eligible(d)) return a3d_wino_workspace_bytes(d);
    if (d->splitk <= 1) return 0;
    return (size_t)d->splitk * d->B * d->Ho * d->Wo * d->Cout * sizeof(float);
}

extern "C" int a3d_conv2d_nhwc_f32(const a3d_conv_desc *d, void *stream) {
    const int rc = conv_check(d);
    if (rc != A3D_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    a3d_begin();
    if (d->precision == 1) return a3d_conv_launch_bf16(d, s);  // (A3D_ERR_UNSUPPORTED for layer kinds it does not cover)
    if (d->precision == 2) {  // Winograd layers keep the Winograd form (split-operand GEMM, conv_wino.hip 2x), the rest go direct
        if ((d->tune == 0 || d->tune == 8 || d->tune == 24 || d->tune == 25) && d->workspace && d->w_wino && d->w_wino_x3 && ((d->Cin + d->Cin2) & 31) == 0 && a3d_wino_eligible(d)) return a3d_conv_launch_wino(d, s);
        return a3d_conv_launch_bf16x3(d, s);
    }
    if (d->precision == 3 && d->x_h2) return a3d_conv_launch_bf16x3_wide(d, s);  // pre-split activations: the dual-DMA forms only
    if (d->precision == 3) {  // fp16x2 split: Winograd layers (wide kernels) when their pre-split filter is given, the rest direct
        if ((d->tune == 0 || d->tune == 23 || d->tune == 24) && d->workspace && d->w_wino && d->w_wino_x3 && ((d->Cin + d->Cin2) & 31) == 0 && a3d_wino_eligible(d))
            return a3d_conv_launch_wino(d, s);
        return a3d_conv_launch_bf16x3(d, s);
    }
    if (d->precision != 0) return A3D_ERR_ARG;
    if (d->tune == 0 && a3d_wino_fused_eligible(d)) return a3d_conv_launch_wino_fused(d, s);  // one launch, no V tensor (tune 7: the two-launch form)
    if ((d->tune == 0 || d->tune == 7 || d->tune >= 200) && d->workspace && a3d_wino_eligible(d)) return a3d_conv_launch_wino(d, s);
    if (d->tune == 0 || d->tune == 6) {  // persistent pointwise kernel for the 1x1 layers (tune 5: never, 6: whenever eligible)
        const int r1 = a3d_conv_launch_pw(d, s, d->tune == 6);
        if (r1 != A3D_ERR_UNSUPPORTED) return r1;
    }
    if (d->tune == 1) return A3D_ERR_UNSUPPORTED;  // (tune 1 selected the round-1 general kernel: removed, see the file header)
    return a3d_conv_launch_v2(d, s);  // A3D_ERR_UNSUPPORTED for the descriptors it does not take
}

extern "C" int a3d_version(void) { return 1; }
