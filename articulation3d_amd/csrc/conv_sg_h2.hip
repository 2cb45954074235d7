// fp16x2 (a3d_conv_desc.precision == 3) pointwise convolution for SMALL GRIDS: one wave per 32 x 32 output tile, no LDS, no barrier.
//
// What it is for.  One frame (the reference's own loop, tools/inference.py:215-228, calls the model once per frame) leaves the deep 1x1
// layers of the trunk with a handful of 128-wide tiles: 30x40 1024 -> 256 is 10 x 4 workgroups of conv_x3_kernel<1> on 256 CUs, each of
// them a k loop of 64 barrier-synchronised chunks whose operands are requested three chunks (~1.5 us of memory latency) ahead -- 34 us
// for 4 us of matrix work, and 29 + 18 such launches are a third of the single-frame pass (profiles/r06_loop_bench.txt).  The launch is
// bound by how many bytes the few busy CUs keep in flight, not by bandwidth or the matrix pipe.
//
// Here the unit of work is the MFMA's own tile: a wave owns 32 pixels x 32 output channels for the whole reduction, fetches its operand
// fragments straight into registers in the lane layout v_mfma_f32_32x32x16_f16 wants (activations: 32 B of fp32 per lane and chunk, split
// in registers; filter: 16 B per lane, chunk and plane of the pre-split a3d_conv_desc.w_x3) and keeps R chunks in flight in a register
// ring -- 4 KiB per wave and chunk, requested R chunks ahead, with nothing to synchronise.  30x40 1024 -> 256 becomes 304 independent
// one-wave workgroups instead of 40 four-wave ones: 13.9 us against 27.8; 15x20 2048 -> 512 17.2 against 46.6 (kernel trace, MI355X).
// What bounds it then is operand re-reads: a 32 x 32 tile fetches 8 bytes per multiply-add pair row, 82 MB through the L2s for the
// 6.4 MB of that layer -- so the form pays up to ~1200 tiles and loses to the 128-wide tiles above (A3D_SG_MAX_WAVES; measured at 1, 2
// and 4 frames, tools/probes/sg_probe.py).  Measured and not taken: a 16-chunk ring (348 registers; no faster alone, 25.6 against 23.6 us
// per launch inside the one-frame pass), two or four waves per workgroup (they share one L1 / texture path: 1.2 - 2.2 x slower).
//
// Per output element the operations are conv_x3_kernel's (same split, same k order, h.h + h.l + l.h per 16-deep chunk into ONE fp32
// accumulator of the same MFMA shape with the filter as operand A, same epilogue): bit-identical, so a frame's bits still do not
// depend on the batch it arrives in (tests/test_gpu_parity.py).
#include "conv_common.h"

namespace {
typedef _Float16 sg_h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 sg_h16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sg_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
// x * s = h + l (conv_bf16x3.hip split2h, element for element)
__device__ __forceinline__ void sg_split(const f32x4 v, const float s, sg_h16x4 &h, sg_h16x4 &l) {
    const f32x4 xs = v * s;
    h = __builtin_convertvector(xs, sg_h16x4);
    const f32x4 r = xs - __builtin_convertvector(h, f32x4);
    l = __builtin_convertvector(r, sg_h16x4);
}

// R = chunks of 16 input channels in flight per wave (Cin / 16 is a multiple of R).
template <int R>
__global__ __launch_bounds__(64) void conv_sg_kernel(const a3d_conv_desc d, const int M, const int ntn) {
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x;  // ONE wave per workgroup: two or four waves of a CU behind one L1 / texture path measured 1.2 - 2.2 x slower
    const int mt = tile / ntn, n0 = (tile - mt * ntn) * 32;
    const int pr = lane & 31, ph = lane >> 5;  // fragment row (pixel / filter row) and k half; also the accumulator layout
    const int m = mt * 32 + pr;
    const bool mok = m < M;
    const int hwo = d.Ho * d.Wo;
    const int mm = mok ? m : 0;
    const int b = mm / hwo, rr = mm - b * hwo;
    const int oh = rr / d.Wo, ow = rr - oh * d.Wo;
    const float sx = mok ? a3d_in_scale(d, b) : 1.f;
    const int nk = d.Cin >> 4;

    const __amdgpu_buffer_rsrc_t rx = sg_rsrc(d.x, (unsigned)((size_t)d.B * d.H * d.W * d.Cin * 4));
    const __amdgpu_buffer_rsrc_t rw = sg_rsrc(d.w_x3, (unsigned)((size_t)nk * d.Cout * 64));
    const int xoff = mok ? (((b * d.H + oh * d.stride) * d.W + ow * d.stride) * d.Cin + ph * 8) * 4 : -1;  // (rows past M read as zeros)
    const int woff = (n0 + pr) * 32 + ph * 16;  // w_x3 [Cin/16][2][Cout][16] fp16: row n of (chunk, plane) is 32 contiguous bytes
    const int wplane = d.Cout * 32;

    f32x4 xa[R][2];
    sg_h16x8 wf[R][2];
    auto issue = [&](const int slot, const int c) {
        xa[slot][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff, c * 64, 0));
        xa[slot][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, xoff, c * 64 + 16, 0));
        wf[slot][0] = __builtin_bit_cast(sg_h16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, woff, (2 * c) * wplane, 0));
        wf[slot][1] = __builtin_bit_cast(sg_h16x8, __builtin_amdgcn_raw_buffer_load_b128(rw, woff, (2 * c + 1) * wplane, 0));
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // One chunk = three dependent MFMAs on the one accumulator (32 matrix-pipe cycles each) and ~26 VALU instructions of split: the split
    // of chunk c + 1 is placed in the shadows of chunk c's products.
    sg_h16x8 xh, xl, nxh, nxl;
    auto split = [&](const int slot, sg_h16x8 &h, sg_h16x8 &l) {
        sg_h16x4 h0, l0, h1, l1;
        sg_split(xa[slot][0], sx, h0, l0);
        sg_split(xa[slot][1], sx, h1, l1);
        h = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
        l = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto step = [&](const int slot, const int nslot, const int cnext, const bool more, const bool fetch) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[slot][0], xh, acc, 0, 0, 0);
        if (more) split(nslot, nxh, nxl);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[slot][0], xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[slot][1], xh, acc, 0, 0, 0);
        if (fetch) issue(slot, cnext);
        if (more) {
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);  // <= 10 VALU
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // <= 2 buffer loads
            }
            xh = nxh;
            xl = nxl;
        }
        __builtin_amdgcn_sched_barrier(0);  // (left alone the scheduler gathers the R steps' loads behind their products: a block, not a ring)
    };

#pragma unroll
    for (int j = 0; j < R; ++j) issue(j, j);
    split(0, xh, xl);
    int c0 = 0;
    for (; c0 + R < nk; c0 += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) step(j, (j + 1) % R, c0 + R + j, true, true);
    }
    // what the epilogue reads is requested in front of the last R chunks' products
    const bool has_res = d.res != nullptr;
    size_t res_row;
    int eb, eoh, eow;
    out_rows(d, mm, res_row, eb, eoh, eow);
    f32x4 sc[4], sh[4], rv[4] = {};
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int n = n0 + rg * 8 + ph * 4;
        sc[rg] = d.scale ? *reinterpret_cast<const f32x4 *>(d.scale + n) : f32x4{1.f, 1.f, 1.f, 1.f};
        sh[rg] = d.shift ? *reinterpret_cast<const f32x4 *>(d.shift + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (has_res) rv[rg] = *reinterpret_cast<const f32x4 *>(d.res + res_row * (size_t)d.Cout + n);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) step(j, (j + 1) % R, 0, j + 1 < R, false);

    // (two exact factors, applied one after the other: conv_x3_kernel's epilogue)
    const float unx = 1.f / sx, unw = 1.f / d.w_scale;
    float vmax = 0.f;
    if (mok) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int n = n0 + rg * 8 + ph * 4;
            f32x4 v = {acc[rg * 4 + 0], acc[rg * 4 + 1], acc[rg * 4 + 2], acc[rg * 4 + 3]};
            v = (v * unx) * unw;
            v = a3d_epilogue_math(d, v, sc[rg], sh[rg], has_res, rv[rg]);
            vmax = fmaxf(vmax, a3d_absmax4(v));
            *reinterpret_cast<f32x4 *>(d.y + (size_t)m * d.Cout + n) = v;
        }
    }
    if (d.y_amax) a3d_note_amax(d.y_amax, b, vmax, mok);  // (every lane of the wave gets here)
}

template <int R>
int launch_sg(const a3d_conv_desc *d, hipStream_t s) {
    const int M = d->B * d->Ho * d->Wo;
    const int ntn = d->Cout / 32, ntiles = ((M + 31) / 32) * ntn;
    a3d_note_variant("conv_h2sg_kernel<%d>", R);
    hipLaunchKernelGGL((conv_sg_kernel<R>), dim3(ntiles), dim3(64), 0, s, *d, M, ntn);
    return a3d_check_launch();
}
}  // namespace

// A3D_ERR_UNSUPPORTED: not a layer / not a launch of this form (the caller goes on to the tiled kernels).  tune 17: whatever the grid size.
int a3d_conv_launch_sg_h2(const a3d_conv_desc *d, hipStream_t s) {
    if (d->tune != 0 && d->tune != 17) return A3D_ERR_UNSUPPORTED;
    if (d->precision != 3 || !d->w_x3 || !d->in_amax || d->in_amax2 || !(d->w_scale > 0.f)) return A3D_ERR_UNSUPPORTED;
    if (d->KH != 1 || d->KW != 1 || d->pad != 0 || d->Kpad != d->Cin || d->stride < 1) return A3D_ERR_UNSUPPORTED;
    if (d->stem || d->ups || d->phase || d->pixshuf || d->gate || d->x2 || d->Cin2 || d->splitk != 1 || d->m_dev) return A3D_ERR_UNSUPPORTED;
    if ((d->Cout & 31) || (d->Cin & 63)) return A3D_ERR_UNSUPPORTED;
    const size_t M = (size_t)d->B * d->Ho * d->Wo;
    if ((size_t)d->B * d->H * d->W * d->Cin * 4 >= ((size_t)1 << 31) || (size_t)d->Cout * d->Cin * 4 >= ((size_t)1 << 31) || M >= ((size_t)1 << 26))
        return A3D_ERR_UNSUPPORTED;
    const size_t waves = ((M + 31) / 32) * (size_t)(d->Cout / 32);
    if (d->tune == 0 && waves > A3D_SG_MAX_WAVES) return A3D_ERR_UNSUPPORTED;
    const int nk = d->Cin / 16;
    return nk % 8 == 0 ? launch_sg<8>(d, s) : launch_sg<4>(d, s);
}
