// Shared pieces of the conv-GEMM kernels: tile constants and the fused epilogue.
#pragma once
#include "a3d_common.h"
#include "../../include/a3d.h"

#define BK 32
#define LDK 36

template <int ACT>
__device__ __forceinline__ float a3d_act(float v) {
    if (ACT == A3D_ACT_RELU) return v <= 0.f ? 0.f : v;  // (NaN stays NaN, like torch.relu)
    if (ACT == A3D_ACT_LEAKY) return v > 0.f ? v : 0.01f * v;
    return v;
}

__device__ __forceinline__ f32x4 apply_epilogue(const a3d_conv_desc &d, f32x4 v, int n, size_t res_row) {
    f32x4 s = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (d.scale) s = *reinterpret_cast<const f32x4 *>(d.scale + n);
    if (d.shift) sh = *reinterpret_cast<const f32x4 *>(d.shift + n);
    for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(v[i], s[i], sh[i]);  // same rounding as a3d_epilogue_math below
    if (d.res) {
        const f32x4 r = *reinterpret_cast<const f32x4 *>(d.res + res_row * (size_t)d.Cout + n);
        v += r;
    }
    if (d.act == A3D_ACT_RELU) {
        for (int i = 0; i < 4; ++i) v[i] = v[i] <= 0.f ? 0.f : v[i];
    } else if (d.act == A3D_ACT_LEAKY) {
        for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : 0.01f * v[i];
    }
    return v;
}

// ---- fast epilogue ---------------------------------------------------------------------------------------------
// The per-quad form above issues three dependent global loads (scale, shift, residual) per output quad, and because
// y / res / gate may alias the compiler keeps every one of them in program order with the stores: 16 serialized
// round trips per wave.  The kernels therefore stage scale / shift of their N tile in LDS once (a3d_stage_scale_shift)
// and fetch all residual quads of an output row before its first store (one wait instead of eight).
__device__ __forceinline__ void a3d_stage_scale_shift(float *ss /*[2*BN]*/, const a3d_conv_desc &d, int n0, int BN, int tid) {
    if (tid < BN) {
        const int n = n0 + tid;
        const bool ok = n < d.Cout;
        ss[tid] = (d.scale && ok) ? d.scale[n] : 1.f;
        ss[BN + tid] = (d.shift && ok) ? d.shift[n] : 0.f;
    }
}
// v * scale + shift (+ residual), activation.  One rounding for the scale/shift pair (fused multiply-add), the same in
// every kernel, so a layer's result does not depend on which kernel variant ran it.
__device__ __forceinline__ f32x4 a3d_epilogue_math(const a3d_conv_desc &d, f32x4 v, const f32x4 s, const f32x4 sh, const bool has_res,
                                                    const f32x4 r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(v[i], s[i], sh[i]);
    if (has_res) v += r;
    if (d.act == A3D_ACT_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = v[i] <= 0.f ? 0.f : v[i];
    } else if (d.act == A3D_ACT_LEAKY) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : 0.01f * v[i];
    }
    return v;
}

// Output row addressing shared by the direct epilogue and the split-K reducer.
__device__ __forceinline__ void out_rows(const a3d_conv_desc &d, int m, size_t &res_row, int &b, int &oh, int &ow) {
    res_row = (size_t)m;
    b = oh = ow = 0;
    if (d.res_ups || d.pixshuf || d.phase) {
        const int hw = d.Ho * d.Wo;
        b = m / hw;
        const int r = m - b * hw;
        oh = r / d.Wo;
        ow = r - oh * d.Wo;
        if (d.res_ups) res_row = ((size_t)b * (d.Ho >> 1) + (oh >> 1)) * (size_t)(d.Wo >> 1) + (ow >> 1);
    }
}

__device__ __forceinline__ void store_out(const a3d_conv_desc &d, f32x4 v, int m, int n, int b, int oh, int ow) {
    if (d.phase) {
        const int dy = (d.phase - 1) >> 1, dx = (d.phase - 1) & 1;
        const size_t row = ((size_t)b * (2 * d.Ho) + (2 * oh + dy)) * (size_t)(2 * d.Wo) + (2 * ow + dx);
        *reinterpret_cast<f32x4 *>(d.y + row * d.Cout + n) = v;
    } else if (d.pixshuf) {
        const int co_n = d.Cout >> 2;  // real output channels
        const int q = n / co_n, co = n - q * co_n;
        const int dy = q >> 1, dx = q & 1;
        const size_t row = ((size_t)b * (2 * d.Ho) + (2 * oh + dy)) * (size_t)(2 * d.Wo) + (2 * ow + dx);
        *reinterpret_cast<f32x4 *>(d.y + row * co_n + co) = v;
    } else {
        if (d.gate) {
            const f32x4 g = *reinterpret_cast<const f32x4 *>(d.gate + (size_t)m * d.Cout + n);
            for (int i = 0; i < 4; ++i) v[i] = g[i] > 0.f ? v[i] : 0.f;
        }
        *reinterpret_cast<f32x4 *>(d.y + (size_t)m * d.Cout + n) = v;
    }
}


// ---- fp16x2 mode (a3d_conv_desc.precision == 3) and the per-image output maxima every split-operand kernel can record -------------
// power of two s with amax * s in [2^14, 2^15)  (amax == 0 or not finite -> 1)
__device__ __forceinline__ float a3d_pow2_scale(const float amax) {
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
    return ldexpf(1.f, min(14 - ilogbf(amax), 126));  // (amax below 2^-112: the largest finite power of two that keeps 1 / s normal)
}
// scale of the rows of input image b (both sources of a channel concat share it)
__device__ __forceinline__ float a3d_in_scale(const a3d_conv_desc &d, const int b) {
    float a = d.in_amax[b];
    if (d.in_amax2) a = fmaxf(a, d.in_amax2[b]);
    return a3d_pow2_scale(a);
}
// y_amax[b] = max(y_amax[b], v) for the lanes with `valid`; v >= 0.  One atomic per wave when the wave's valid lanes share b (the
// rule: 32 consecutive pixels of one image), and none at all once the slot already holds a larger value.
__device__ __forceinline__ void a3d_note_amax(float *y_amax, const int b, float v, const bool valid) {
    if (!valid) v = 0.f;
    const unsigned long long live = __ballot(valid);
    if (!live) return;
    const int b0 = __shfl(b, __ffsll((long long)live) - 1, 64);
    const bool uniform = __all(!valid || b == b0);
    if (uniform) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
        if ((threadIdx.x & 63) == 0 && v > y_amax[b0]) atomicMax(reinterpret_cast<int *>(y_amax + b0), __float_as_int(v));
    } else if (valid && v > y_amax[b]) {
        atomicMax(reinterpret_cast<int *>(y_amax + b), __float_as_int(v));
    }
}
// Every conv launcher records the kernel instantiation it dispatched (name + template arguments as they appear in a
// rocprofv3 kernel trace) in a per-thread slot; `a3d_last_conv_variant()` (include/a3d.h) reads it back, so measurement
// code labels launches with what the dispatcher DID, not with a host-side copy of its rules.
void a3d_note_variant(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

// v2 kernel family (conv_gemm_v2.hip): buffer-addressed, branch-free gather + software-pipelined main loop.
// Returns A3D_ERR_UNSUPPORTED when the descriptor needs the general kernel.
int a3d_conv_launch_v2(const a3d_conv_desc *d, hipStream_t s);
// fp16x2 pointwise layers with Cin <= 256, activations stationary in registers (conv_xs_h2.hip); tune 13 forces it on any eligible
// layer, tune 14 keeps it off.  A3D_ERR_UNSUPPORTED: not such a layer.
int a3d_conv_launch_xs_h2(const a3d_conv_desc *d, hipStream_t s);

// fp16x2 pointwise layers on SMALL grids (one frame): one wave per 32 x 32 output tile, operands straight into registers
// (conv_sg_h2.hip; the same bits); launches of up to A3D_SG_MAX_WAVES such tiles; tune 17 = whatever the size.
#define A3D_SG_MAX_WAVES 1280
int a3d_conv_launch_sg_h2(const a3d_conv_desc *d, hipStream_t s);

// Persistent pointwise kernel (conv_pw.hip): 1x1 stride-1 layers with K <= 2048 and enough tiles to keep a persistent
// grid busy.  Returns A3D_ERR_UNSUPPORTED otherwise.  `force` skips the grid-size heuristic (A/B measurements).
int a3d_conv_launch_pw(const a3d_conv_desc *d, hipStream_t s, int force);

// bf16-MFMA variant (conv_bf16.hip), selected by a3d_conv_desc.precision == 1.
int a3d_conv_launch_bf16(const a3d_conv_desc *d, hipStream_t s);
// its large-launch form with both operands by LDS-DMA (conv_bf16w.hip; needs a3d_conv_desc.w_bf16); A3D_ERR_UNSUPPORTED -> the kernel above
int a3d_conv_launch_bf16w(const a3d_conv_desc *d, hipStream_t s);
// its pointwise form with the activations stationary in registers (conv_bf16xs.hip; needs a3d_conv_desc.w_bf16); A3D_ERR_UNSUPPORTED -> the kernels above
int a3d_conv_launch_bf16xs(const a3d_conv_desc *d, hipStream_t s);
// fp32-grade 3-way bf16 split on the bf16 matrix pipe (conv_bf16x3.hip), selected by a3d_conv_desc.precision == 2.
int a3d_conv_launch_bf16x3(const a3d_conv_desc *d, hipStream_t s);
// its wide form (conv_bf16x3_wide.hip: 256 x 256 tiles, pre-split weights by LDS-DMA); A3D_ERR_UNSUPPORTED -> the kernel above
int a3d_conv_launch_bf16x3_wide(const a3d_conv_desc *d, hipStream_t s);
// the fused four-phase form with the input patch resident in LDS (conv_ph4p.hip); A3D_ERR_UNSUPPORTED -> the tap-outer form
int a3d_conv_launch_ph4p(const a3d_conv_desc *d, hipStream_t s);
// the same loop on a plain 3x3 stride-1 pad-1 fp16x2 layer (conv_ph4p.hip); A3D_ERR_UNSUPPORTED -> the tap-outer kernels
int a3d_conv_launch_c3p(const a3d_conv_desc *d, hipStream_t s);
// second launch of a split-K layer (conv_gemm_v2.hip): workspace [splitk][M][Cout] -> y with the fused epilogue
void a3d_launch_splitk_reduce(const a3d_conv_desc *d, int M, hipStream_t s);

// Winograd F(2x2,3x3) path (conv_wino.hip).  eligible() ignores the workspace pointer (used for sizing).
int a3d_wino_eligible(const a3d_conv_desc *d);
size_t a3d_wino_workspace_bytes(const a3d_conv_desc *d);
int a3d_conv_launch_wino(const a3d_conv_desc *d, hipStream_t s);
// One-launch Winograd (conv_wino_fused.hip): input transform inside the GEMM loader; needs a3d_conv_desc.w_wino_cm.
int a3d_wino_fused_eligible(const a3d_conv_desc *d);
int a3d_conv_launch_wino_fused(const a3d_conv_desc *d, hipStream_t s);
