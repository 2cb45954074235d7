// Shared pieces of the conv-GEMM kernels: tile constants and the fused epilogue.
#pragma once
#include "a3d_common.h"
#include "../../include/a3d.h"

#define BK 32
#define LDK 36

template <int ACT>
__device__ __forceinline__ float a3d_act(float v) {
    if (ACT == A3D_ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == A3D_ACT_LEAKY) return v > 0.f ? v : 0.01f * v;
    return v;
}

__device__ __forceinline__ f32x4 apply_epilogue(const a3d_conv_desc &d, f32x4 v, int n, size_t res_row) {
    if (d.scale) {
        const f32x4 s = *reinterpret_cast<const f32x4 *>(d.scale + n);
        v *= s;
    }
    if (d.shift) {
        const f32x4 s = *reinterpret_cast<const f32x4 *>(d.shift + n);
        v += s;
    }
    if (d.res) {
        const f32x4 r = *reinterpret_cast<const f32x4 *>(d.res + res_row * (size_t)d.Cout + n);
        v += r;
    }
    if (d.act == A3D_ACT_RELU) {
        for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
    } else if (d.act == A3D_ACT_LEAKY) {
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


// v2 kernel family (conv_gemm_v2.hip): buffer-addressed, branch-free gather + software-pipelined main loop.
// Returns A3D_ERR_UNSUPPORTED when the descriptor needs the general kernel.
int a3d_conv_launch_v2(const a3d_conv_desc *d, hipStream_t s);

// Winograd F(2x2,3x3) path (conv_wino.hip).  eligible() ignores the workspace pointer (used for sizing).
int a3d_wino_eligible(const a3d_conv_desc *d);
size_t a3d_wino_workspace_bytes(const a3d_conv_desc *d);
int a3d_conv_launch_wino(const a3d_conv_desc *d, hipStream_t s);
