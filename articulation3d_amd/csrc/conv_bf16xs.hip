// bf16 (a3d_conv_desc.precision == 1: the training step's autocast arithmetic) pointwise convolution with the ACTIVATIONS STATIONARY
// IN REGISTERS -- conv_xs_h2.hip's decomposition on the one-product bf16 pipe.
//
// What it is for.  The 1x1 layers of the trainable trunk with Cin <= 512 (the bottleneck expansions 128 -> 512 / 256 -> 1024 with
// their residual, the same shapes as data gradients with the ReLU-backward gate and the shortcut's gradient as residual, the
// reductions 512 -> 128, the RPN predictors' data gradient 32 -> 256) are HBM-bound by construction: at 16 images per GPU a pixel's
// output, residual and gate rows are 3 - 12 x its input row and the whole reduction is 1 - 16 chunks.  conv_bf16_kernel runs them as
// 128 x 128 tiles whose 4 - 8 chunk k loop sits between a prologue and an epilogue that each wait a full memory round trip
// (measured per layer, round 5: 1.9 - 2.5 TB/s; the step's 1x1 launches 3.3 ms against 1.2 ms at 5 TB/s).
//
// Here a wave OWNS 32 pixels for the whole launch: their Cin channels are loaded once and kept as MFMA B fragments (Cin / 16 x 4
// VGPRs; fp32-stored activations are rounded to bf16 on the way, as conv_bf16_kernel rounds them into LDS).  The workgroup (4 waves
// = 128 pixels) walks ALL output channels: the bf16 filter copy (a3d_conv_desc.w_bf16, [Cout][Kpad]) streams global -> LDS by
// LDS-DMA through a ring of 8 KiB stages, six stages ahead of its use, and the stream of one N step continues into the next
// without a gap; the residual and gate rows of an N step are requested when its k loop starts.  The epilogue goes through a
// per-wave 4 KiB transposition so that a lane stores four consecutive channels of a row (8 lanes = one 64-byte run of bf16).
// Per output element: the same rounded operands, the same 16-deep products in the same order into one fp32 accumulator, the same
// epilogue arithmetic as conv_bf16_kernel -- bit-identical (tests/test_gpu_training.py).
#include "conv_common.h"

namespace {
typedef __bf16 bx_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bx_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int bx_u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t bx_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bx_widen4(const bx_u32x2 v) {
    f32x4 o;
    o[0] = __builtin_bit_cast(float, v[0] << 16);
    o[1] = __builtin_bit_cast(float, v[0] & 0xFFFF0000u);
    o[2] = __builtin_bit_cast(float, v[1] << 16);
    o[3] = __builtin_bit_cast(float, v[1] & 0xFFFF0000u);
    return o;
}
template <int N>
__device__ __forceinline__ void bx_wait_vm() {
    __asm__ volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int BX_NST = 7;         // ring stages of 8 KiB
constexpr int BX_D = BX_NST - 1;  // a stage's DMA is issued this many steps before its fragments are read
constexpr int BX_STAGE = 8192;    // bytes: eight [32 rows][16 k] bf16 pieces
constexpr int bx_lds_bytes(int bn) { return BX_NST * BX_STAGE + 4 * 4096 + 2 * 2 * bn * 4; }

// KC = Cin / 16.  NG = 32-channel groups per N step (BN = 32 NG), KS = 16-deep chunks per ring stage: NG * KS = 8.
// XB: the activations are stored as bf16 (io_bf16 bit 0).
template <int KC, int NG, int KS, bool XB>
__global__ __launch_bounds__(256, 2) void conv_bf16xs_kernel(const a3d_conv_desc d, const int M, const int full_tiles, const int ns_tail) {
    static_assert(NG * KS == 8 && KC % KS == 0, "a ring stage is 8 KiB: 8 (chunk, group) pieces of 1 KiB");
    constexpr int BN = 32 * NG;
    constexpr int SPT = KC / KS;  // ring steps per N step
    constexpr int R = 8 * NG;     // residual + gate loads (8 B per lane each) of an N step
    extern __shared__ __attribute__((aligned(16))) unsigned char bx_lds[];
    unsigned char *ring = bx_lds;
    float *Tall = reinterpret_cast<float *>(bx_lds + BX_NST * BX_STAGE);
    float *ssall = Tall + 4 * 1024;  // [2][2 * BN]: scale | shift of the N step, double-buffered

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int mt, nbeg, nsteps;
    if ((int)blockIdx.x < full_tiles) {
        mt = blockIdx.x;
        nbeg = 0;
        nsteps = d.Cout / BN;
    } else {  // the last partial round of pixel tiles: each split ns_tail ways along N so that the chip stays full
        const int t = blockIdx.x - full_tiles;
        mt = full_tiles + t / ns_tail;
        nsteps = (d.Cout / BN) / ns_tail;
        nbeg = (t % ns_tail) * nsteps * BN;
    }
    const int m0 = mt * 128 + wave * 32;  // this wave's 32 pixels
    const int Q = nsteps * SPT;

    // ---- filter stream: w_bf16 [Cout][Kpad].  Piece (16-deep chunk c, rows n .. n + 31) is 32 runs of 32 bytes; lane i of the
    // wave-instruction lands at LDS byte 16 i of the piece = row i / 2, half i % 2, and fetches the k half the image keeps there:
    // half ^ ((row >> 3) & 1) (the fragment reads below are then conflict-free).  A stage holds pieces [chunk-in-stage][group];
    // wave w moves pieces 2w and 2w + 1.
    const __amdgpu_buffer_rsrc_t rw = bx_rsrc(d.w_bf16, (unsigned)((size_t)d.Cout * d.Kpad * 2));
    const int wvoff = ((lane >> 1) * d.Kpad + (((lane & 1) ^ ((lane >> 4) & 1)) << 3)) * 2;
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    int dma_q = 0, dma_st = 0, rd_st = 0;  // next step to fetch; ring stage it goes to; ring stage the next fragment reads come from
    auto dma = [&]() {
        const int q = dma_q++;
        const int ns = q / SPT, t = q - ns * SPT;
        unsigned char *st = ring + dma_st * BX_STAGE;
        dma_st = dma_st == BX_NST - 1 ? 0 : dma_st + 1;
        const int voff = q < Q ? wvoff : -1;  // (past the last step: out of range -- zeros into a stage nobody reads; the op count per step stays fixed)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = uw * 2 + i;
            const int kl = j / NG, g = j % NG;
            const int c = t * KS + kl;
            const int soff = q < Q ? ((nbeg + ns * BN + g * 32) * d.Kpad + c * 16) * 2 : 0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void *)(st + j * 1024), 16, voff,
                                                     __builtin_amdgcn_readfirstlane(soff), 0, 0);
        }
    };
#pragma unroll
    for (int i = 0; i < BX_D; ++i) dma();

    // ---- the wave's activations: lane (pixel lane % 32, k group lane / 32) holds channels 16 c + 8 (lane / 32) .. + 7 of every chunk c
    const int mp = m0 + (lane & 31);
    const bool mok = mp < M;
    bx_bf16x8 xb[KC];
    {
        constexpr int XES = XB ? 2 : 4;
        const __amdgpu_buffer_rsrc_t rx = bx_rsrc(d.x, (unsigned)((size_t)M * d.Cin * XES));
        const int voff = mok ? (mp * d.Cin + (lane >> 5) * 8) * XES : -1;
        if constexpr (XB) {
#pragma unroll
            for (int c = 0; c < KC; ++c) xb[c] = __builtin_bit_cast(bx_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, c * 32, 0));
            bx_wait_vm<0>();
        } else {
            constexpr int CB = KC < 8 ? KC : 8;  // chunks per batch of fp32 loads (64 transient registers)
#pragma unroll
            for (int c0 = 0; c0 < KC; c0 += CB) {
                f32x4 raw[CB][2];
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    raw[c][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, (c0 + c) * 64, 0));
                    raw[c][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, (c0 + c) * 64 + 16, 0));
                }
                bx_wait_vm<0>();
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    const bx_bf16x4 lo = __builtin_convertvector(raw[c][0], bx_bf16x4), hi = __builtin_convertvector(raw[c][1], bx_bf16x4);
                    xb[c0 + c] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
        }
    }

    const int frow = lane & 31;
    const int frag_off = (frow * 16 + ((((lane >> 5) ^ (frow >> 3)) & 1) << 3)) * 2;  // bytes inside a 1 KiB piece
    const int pr = lane & 31, ph = lane >> 5;  // accumulator layout: pixel, channel quad half
    const int qr = lane >> 3, qc = lane & 7;   // row-major epilogue: row (+ 8 j), channel quad
    float *T = Tall + wave * 1024;
    const bool has_res = d.res != nullptr, has_gate = d.gate != nullptr;
    const bool yb = d.io_bf16 & 2;
    const bool side = has_res || has_gate;
    const unsigned obytes = (unsigned)((size_t)M * d.Cout * 2);
    const __amdgpu_buffer_rsrc_t rres = bx_rsrc(has_res ? (const void *)d.res : (const void *)d.w_bf16, has_res ? obytes : 0u);
    const __amdgpu_buffer_rsrc_t rgate = bx_rsrc(has_gate ? (const void *)d.gate : (const void *)d.w_bf16, has_gate ? obytes : 0u);

    for (int ns = 0, q = 0; ns < nsteps; ++ns) {
        const int n0 = nbeg + ns * BN;
        float *ss = ssall + (ns & 1) * 2 * BN;
        if (tid < BN) {
            ss[tid] = d.scale ? d.scale[n0 + tid] : 1.f;
            ss[BN + tid] = d.shift ? d.shift[n0 + tid] : 0.f;
        }
        // the residual and gate rows of this N step (stored bf16), in the row-major form the epilogue stores: requested now, used behind
        // the k loop.  A tensor that is absent has a zero-sized buffer: its loads return zeros without touching memory, and the count of
        // outstanding operations the waits below rely on does not depend on the layer.
        bx_u32x2 rvr[NG][4], gvr[NG][4];
        if (side) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + qr + 8 * j;
                const int voff = m < M ? (m * d.Cout + n0 + qc * 4) * 2 : -1;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    rvr[g][j] = __builtin_bit_cast(bx_u32x2, __builtin_amdgcn_raw_buffer_load_b64(rres, voff, g * 64, 0));
                    gvr[g][j] = __builtin_bit_cast(bx_u32x2, __builtin_amdgcn_raw_buffer_load_b64(rgate, voff, g * 64, 0));
                }
            }
        }
        f32x16 acc[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

#pragma unroll
        for (int t = 0; t < SPT; ++t, ++q) {
            // the DMA of step q (issued BX_D steps ago) has landed: younger than it are the DMAs of the BX_D - 1 steps since and, during
            // the first BX_D steps of an N step, that step's residual / gate loads (loads retire in order; stores only make the wait stricter)
            if (side && t < BX_D) bx_wait_vm<2 * (BX_D - 1) + R>();
            else bx_wait_vm<2 * (BX_D - 1)>();
            // (a bare barrier: __syncthreads() carries a workgroup fence, and the compiler completes every LDS-DMA in flight in front of it)
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const unsigned char *st = ring + rd_st * BX_STAGE + frag_off;
            rd_st = rd_st == BX_NST - 1 ? 0 : rd_st + 1;
            dma();  // step q + BX_D, into the stage read at step q - 1 (every wave is past that read: the barrier above)
#pragma unroll
            for (int kl = 0; kl < KS; ++kl) {
                const int c = t * KS + kl;
                bx_bf16x8 fa[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) fa[g] = *reinterpret_cast<const bx_bf16x8 *>(st + (kl * NG + g) * 1024);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g], xb[c], acc[g], 0, 0, 0);
            }
        }
        // ---- epilogue of the N step: a 32 x 32 tile goes through 4 KiB of LDS (XOR-swizzled) and comes back row-major
        if (side) {  // the residual / gate loads are older than the DMAs of the last min(SPT, BX_D) steps
            if constexpr (SPT < BX_D) bx_wait_vm<2 * SPT>();
            else bx_wait_vm<2 * BX_D>();
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 v = {acc[g][rg * 4 + 0], acc[g][rg * 4 + 1], acc[g][rg * 4 + 2], acc[g][rg * 4 + 3]};
                *reinterpret_cast<f32x4 *>(T + pr * 32 + (((rg * 2 + ph) ^ (pr & 7)) << 2)) = v;
            }
            const int nl = g * 32 + qc * 4;
            f32x4 tv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int qq = qr + 8 * j;
                tv[j] = *reinterpret_cast<const f32x4 *>(T + qq * 32 + ((qc ^ (qq & 7)) << 2));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + qr + 8 * j;
                if (m >= M) continue;
                f32x4 v = a3d_epilogue_math(d, tv[j], *reinterpret_cast<const f32x4 *>(ss + nl), *reinterpret_cast<const f32x4 *>(ss + BN + nl), has_res,
                                            bx_widen4(rvr[g][j]));
                if (has_gate) {
                    const f32x4 gq = bx_widen4(gvr[g][j]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = gq[i] > 0.f ? v[i] : 0.f;
                }
                const size_t o = (size_t)m * d.Cout + n0 + nl;
                if (yb) *reinterpret_cast<bx_bf16x4 *>(reinterpret_cast<__bf16 *>(d.y) + o) = __builtin_convertvector(v, bx_bf16x4);
                else *reinterpret_cast<f32x4 *>(d.y + o) = v;
            }
        }
    }
    bx_wait_vm<0>();  // the DMAs issued past the last step must not land in the LDS of the next workgroup
}

template <int KC, int NG, int KS>
int launch_bxs(const a3d_conv_desc *d, hipStream_t s) {
    constexpr int BN = 32 * NG;
    const int M = d->B * d->Ho * d->Wo;
    const int mtiles = (M + 127) / 128, nst = d->Cout / BN;
    // workgroup slots of the chip (2 per CU): whole rounds of pixel tiles walk all of N; the tiles of the last partial round are split
    // along N (a power of two that divides the N steps) so that they fill the slots once more instead of leaving most of them idle
    const int slots = 2 * 256;
    int full = (mtiles / slots) * slots, ns_tail = 1;
    const int rem = mtiles - full;
    if (rem == 0 || rem * 4 >= slots * 3) full = mtiles;  // (a last round >= 3/4 full stays whole)
    else
        while (ns_tail * 2 <= nst && nst % (ns_tail * 2) == 0 && rem * ns_tail * 2 <= slots + slots / 4) ns_tail *= 2;
    const int blocks = full + (mtiles - full) * ns_tail;
    const bool xb = d->io_bf16 & 1;
    constexpr bool F32X = KC < 32;  // (Cin 512 with fp32-stored activations: 128 fragment registers leave no room for the conversion batches -- not instantiated)
    if (!xb && !F32X) return A3D_ERR_UNSUPPORTED;
    static a3d_attr_once attr;
    if (attr.needed()) {
        if (hipFuncSetAttribute((const void *)conv_bf16xs_kernel<KC, NG, KS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, bx_lds_bytes(BN)) != hipSuccess) return A3D_ERR_LAUNCH;
        if constexpr (F32X)
            if (hipFuncSetAttribute((const void *)conv_bf16xs_kernel<KC, NG, KS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, bx_lds_bytes(BN)) != hipSuccess) return A3D_ERR_LAUNCH;
        attr.mark();
    }
    a3d_note_variant("conv_bf16xs_kernel<%d>", 16 * KC);
    if (xb) hipLaunchKernelGGL((conv_bf16xs_kernel<KC, NG, KS, true>), dim3(blocks), dim3(256), bx_lds_bytes(BN), s, *d, M, full, ns_tail);
    else if constexpr (F32X) hipLaunchKernelGGL((conv_bf16xs_kernel<KC, NG, KS, false>), dim3(blocks), dim3(256), bx_lds_bytes(BN), s, *d, M, full, ns_tail);
    return a3d_check_launch();
}
}  // namespace

// A3D_ERR_UNSUPPORTED: not a layer of this form (the caller goes on to the tiled kernels).  tune 33: this kernel on every layer it can
// run; tune 34: never.
int a3d_conv_launch_bf16xs(const a3d_conv_desc *d, hipStream_t s) {
    if (d->precision != 1 || !d->w_bf16 || d->tune == 34 || !(d->tune == 0 || d->tune == 33)) return A3D_ERR_UNSUPPORTED;
    if (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0 || d->Kpad != d->Cin) return A3D_ERR_UNSUPPORTED;
    if (d->stem || d->ups || d->phase || d->pixshuf || d->x2 || d->Cin2 || d->splitk != 1 || d->m_dev || d->res_ups || d->dot_w) return A3D_ERR_UNSUPPORTED;
    if (d->io_bf16 & ~15) return A3D_ERR_UNSUPPORTED;
    if ((d->res && !(d->io_bf16 & 4)) || (d->gate && !(d->io_bf16 & 8))) return A3D_ERR_UNSUPPORTED;  // residual / gate: stored bf16 (or absent)
    const size_t M = (size_t)d->B * d->Ho * d->Wo;
    if (M * d->Cin * 4 >= ((size_t)1 << 31) || M * d->Cout * 2 >= ((size_t)1 << 31) || (size_t)d->Cout * d->Kpad * 2 >= ((size_t)1 << 31)) return A3D_ERR_UNSUPPORTED;
    const int bn = d->Cin == 512 ? 64 : 128;
    if (d->Cout % bn) return A3D_ERR_UNSUPPORTED;
    // Where it pays (tools/bf16xs_ab.py, profiles/r05_bf16xs_ab.txt; tiled | this kernel, ms, 16 images per GPU): 128 -> 512 + residual
    // 0.073 | 0.049, the same shape as a data gradient (residual + gate) 0.105 | 0.057, lateral2's data gradient 256 -> 256 (fp32 in) 0.216 |
    // 0.154, the RPN predictors' 32 -> 256 on p2 0.146 | 0.087, 256 -> 1024 on 30 x 40 0.059 | 0.048.  Ties or small losses: under ~128 pixel
    // tiles (2 images per GPU below p2: every form is one launch latency there) and Cin 512 with nothing beside the output (lateral3
    // 0.050 | 0.053) -- those stay with the tiled kernels.
    if (d->tune != 33 && (M < 128 * 128 || (d->Cin == 512 && !d->res && !d->gate))) return A3D_ERR_UNSUPPORTED;
    switch (d->Cin) {
    case 32: return launch_bxs<2, 4, 2>(d, s);
    case 128: return launch_bxs<8, 4, 2>(d, s);
    case 256: return launch_bxs<16, 4, 2>(d, s);
    case 512: return launch_bxs<32, 2, 4>(d, s);
    default: return A3D_ERR_UNSUPPORTED;
    }
}
