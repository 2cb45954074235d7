// bf16 (a3d_conv_desc.precision == 1: the training step's autocast arithmetic) pointwise convolution for SMALL GRIDS: one wave per
// 32 x 32 output tile, operand fragments straight into a register ring, no LDS, no barrier -- conv_sg_h2.hip's form on the one-product
// bf16 pipe.
//
// What it is for.  At the reference's 2 images per GPU (BASELINE configs[4]) the deep 1x1 layers of res4 / res5, the top laterals and their
// data gradients are a few dozen 128 x 64 tiles each: conv_bf16_kernel<1> splits their reduction over workgroups and folds the partial
// sums in a second launch (conv_bf16_reduce_kernel) -- 43 second launches on the step's critical stream of 193.  Here a wave owns 32 pixels
// x 32 output channels for the whole reduction and keeps R chunks of both operands in flight (bf16-stored activations: 16 B per lane and
// chunk; the bf16 filter copy a3d_conv_desc.w_bf16 [Cout][Kpad]: 16 B), one MFMA per chunk; no partial sums, no second launch.
// Measured (tools/probes/rejected/bsg_probe.py, profiles/r06_bf16sg_probe.txt; kernel time at 2 images, tiled + fold | this kernel):
// res4 conv1 15.5 | 14.7 us, res5.0 conv1 14.7 | 10.4, lateral5 14.1 | 10.4, res5 conv1 15.4 | 16.6 -- the seven split layers 106 | 101 us:
// a tie in kernel time (a 32 x 32 tile re-reads its operands through the L2s), taken for the launch it removes from the chain.  On layers
// the tiled kernel runs UNSPLIT it loses (res4 conv3 11.7 | 16.2) and is not used.
//
// Per output element: the operands conv_bf16_kernel rounds (RNE) or reads, the same 16-deep products in ascending k into one fp32
// accumulator with the filter as operand A, the same epilogue (scale / shift, residual, activation, gate, fp32 or bf16 store): the bits
// of the UNSPLIT tiled launch.  (Against a split-K launch the sums differ in order, as any two split counts do.)
#include "conv_common.h"

namespace {
typedef __bf16 bs_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bs_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int bs_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t bs_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bs_widen4(const bs_u32x2 v) {
    f32x4 o;
    o[0] = __builtin_bit_cast(float, v[0] << 16);
    o[1] = __builtin_bit_cast(float, v[0] & 0xFFFF0000u);
    o[2] = __builtin_bit_cast(float, v[1] << 16);
    o[3] = __builtin_bit_cast(float, v[1] & 0xFFFF0000u);
    return o;
}
__device__ __forceinline__ f32x4 bs_read4(const float *base, size_t idx, bool is_bf16) {  // element index idx (multiple of 4)
    if (is_bf16) return bs_widen4(*reinterpret_cast<const bs_u32x2 *>(reinterpret_cast<const __bf16 *>(base) + idx));
    return *reinterpret_cast<const f32x4 *>(base + idx);
}

// R = chunks of 16 input channels in flight per wave (Cin / 16 is a multiple of R).  XB: the activations are stored as bf16.
template <int R, bool XB>
__global__ __launch_bounds__(64) void conv_bf16sg_kernel(const a3d_conv_desc d, const int M, const int ntn) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x;
    const int mt = tile / ntn, n0 = (tile - mt * ntn) * 32;
    const int pr = lane & 31, ph = lane >> 5;  // fragment row (pixel / filter row) and k half; also the accumulator layout
    const int m = mt * 32 + pr;
    const bool mok = m < M;
    const int hwo = d.Ho * d.Wo;
    const int mm = mok ? m : 0;
    const int b = mm / hwo, rr = mm - b * hwo;
    const int oh = rr / d.Wo, ow = rr - oh * d.Wo;
    const int nk = d.Cin >> 4;
    constexpr int XES = XB ? 2 : 4;  // bytes per stored activation

    const __amdgpu_buffer_rsrc_t rx = bs_rsrc(d.x, (unsigned)((size_t)d.B * d.H * d.W * d.Cin * XES));
    const __amdgpu_buffer_rsrc_t rw = bs_rsrc(d.w_bf16, (unsigned)((size_t)d.Cout * d.Kpad * 2));
    const int xoff = mok ? (((b * d.H + oh * d.stride) * d.W + ow * d.stride) * d.Cin + ph * 8) * XES : -1;  // (rows past M read as zeros)
    const int woff = ((n0 + pr) * d.Kpad + ph * 8) * 2;

    f32x4 xa[R][XB ? 1 : 2];
    bs_bf16x8 wf[R];
    auto issue = [&](const int slot, const int c) {
        xa[slot][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff, c * 16 * XES, 0));
        if constexpr (!XB) xa[slot][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff, c * 64 + 16, 0));
        wf[slot] = __builtin_bit_cast(bs_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, woff, c * 32, 0));
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto multiply = [&](const int slot) {
        bs_bf16x8 xf;
        if constexpr (XB) {
            xf = __builtin_bit_cast(bs_bf16x8, xa[slot][0]);
        } else {  // fp32 -> bf16, round to nearest even (what conv_bf16_kernel does on the way into LDS)
            const bs_bf16x4 lo = __builtin_convertvector(xa[slot][0], bs_bf16x4), hi = __builtin_convertvector(xa[slot][1], bs_bf16x4);
            xf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[slot], xf, acc, 0, 0, 0);
    };

#pragma unroll
    for (int j = 0; j < R; ++j) issue(j, j);
    int c0 = 0;
    for (; c0 + R < nk; c0 += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            multiply(j);
            issue(j, c0 + R + j);
            __builtin_amdgcn_sched_barrier(0);  // (a ring, not a block: see conv_sg_h2.hip)
        }
    }
    // what the epilogue reads is requested in front of the last R chunks' products
    const bool has_res = d.res != nullptr, has_gate = d.gate != nullptr;
    const bool yb = d.io_bf16 & 2, rb = d.io_bf16 & 4, gb = d.io_bf16 & 8;
    size_t res_row;
    int eb, eoh, eow;
    out_rows(d, mm, res_row, eb, eoh, eow);
    f32x4 sc[4], sh[4], rv[4] = {}, gv[4] = {};
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int n = n0 + rg * 8 + ph * 4;
        sc[rg] = d.scale ? *reinterpret_cast<const f32x4 *>(d.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
        sh[rg] = d.shift ? *reinterpret_cast<const f32x4 *>(d.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (has_res) rv[rg] = bs_read4(d.res, res_row * (size_t)d.Cout + n, rb);
        if (has_gate) gv[rg] = bs_read4(d.gate, (size_t)mm * d.Cout + n, gb);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) multiply(j);

    if (!mok) return;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int n = n0 + rg * 8 + ph * 4;
        f32x4 v = {acc[rg * 4 + 0], acc[rg * 4 + 1], acc[rg * 4 + 2], acc[rg * 4 + 3]};
        v = a3d_epilogue_math(d, v, sc[rg], sh[rg], has_res, rv[rg]);
        if (has_gate) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = gv[rg][i] > 0.f ? v[i] : 0.f;
        }
        const size_t o = (size_t)m * d.Cout + n;
        if (yb) *reinterpret_cast<bs_bf16x4 *>(reinterpret_cast<__bf16 *>(d.y) + o) = __builtin_convertvector(v, bs_bf16x4);
        else *reinterpret_cast<f32x4 *>(d.y + o) = v;
    }
}

template <int R>
int launch_bsg(const a3d_conv_desc *d, hipStream_t s) {
    const int M = d->B * d->Ho * d->Wo;
    const int ntn = d->Cout / 32, ntiles = ((M + 31) / 32) * ntn;
    a3d_note_variant("conv_bf16sg_kernel<%d>", R);
    if (d->io_bf16 & 1) hipLaunchKernelGGL((conv_bf16sg_kernel<R, true>), dim3(ntiles), dim3(64), 0, s, *d, M, ntn);
    else hipLaunchKernelGGL((conv_bf16sg_kernel<(R > 8 ? 8 : R), false>), dim3(ntiles), dim3(64), 0, s, *d, M, ntn);
    return a3d_check_launch();
}
}  // namespace

// A3D_ERR_UNSUPPORTED: not a layer / not a launch of this form (the caller goes on to the tiled kernels).  tune 0: the launches the caller
// asks to split (a3d_conv_desc.splitk > 1: the whole reduction then runs here in ONE launch, the workspace stays unused); tune 35: any
// eligible layer whatever the grid size; 36: never.
int a3d_conv_launch_bf16sg(const a3d_conv_desc *d, hipStream_t s) {
    if (d->precision != 1 || !d->w_bf16 || !(d->tune == 0 || d->tune == 35)) return A3D_ERR_UNSUPPORTED;
    if (d->KH != 1 || d->KW != 1 || d->pad != 0 || d->Kpad != d->Cin || d->stride < 1) return A3D_ERR_UNSUPPORTED;
    if (d->stem || d->ups || d->phase || d->pixshuf || d->x2 || d->Cin2 || d->m_dev || d->dot_w || d->splitk < 1) return A3D_ERR_UNSUPPORTED;
    if (d->splitk > 1 && d->res_ups) return A3D_ERR_UNSUPPORTED;
    if (d->io_bf16 & ~15) return A3D_ERR_UNSUPPORTED;
    if ((d->Cout & 31) || (d->Cin & 63)) return A3D_ERR_UNSUPPORTED;
    const size_t M = (size_t)d->B * d->Ho * d->Wo;
    if ((size_t)d->B * d->H * d->W * d->Cin * 4 >= ((size_t)1 << 31) || (size_t)d->Cout * d->Kpad * 2 >= ((size_t)1 << 31) || M >= ((size_t)1 << 26))
        return A3D_ERR_UNSUPPORTED;
    const size_t waves = ((M + 31) / 32) * (size_t)(d->Cout / 32);
    if (d->tune == 0 && (d->splitk == 1 || waves > A3D_BF16SG_MAX_WAVES)) return A3D_ERR_UNSUPPORTED;
    const int nk = d->Cin / 16;
    return nk % 16 == 0 ? launch_bsg<16>(d, s) : nk % 8 == 0 ? launch_bsg<8>(d, s) : launch_bsg<4>(d, s);
}
