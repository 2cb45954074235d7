// fp16x2 (a3d_conv_desc.precision == 3) pointwise convolution with the ACTIVATIONS STATIONARY IN REGISTERS.
//
// What it is for.  The 1x1 layers of the ResNet / FPN trunk with Cin <= 256 (the bottleneck expansions 64 -> 256, 128 -> 512,
// 256 -> 1024 with their residual adds, the FPN laterals, the 256 -> 64 reductions) are HBM-bound by construction: a pixel's output
// and residual rows are 4 .. 16 x its input row.  In the tiled kernel (conv_bf16x3.hip) a 128 x 128 tile is a 4 .. 16 chunk k loop
// between a prologue and an epilogue that each wait a full memory round trip: the loads of a chunk are issued two or three chunks
// (~0.5 us) ahead of their use against 1 - 2 us of latency, every N tile fetches and splits the same activation rows again, and the
// waves sit parked 60 % of their cycles (PMC).
//
// Here a wave OWNS 32 pixels for the whole launch: their Cin channels are loaded once -- one memory round trip per tile -- split once
// into the two fp16 planes, and kept as MFMA B fragments in registers (Cin / 16 x 8 VGPRs).  The workgroup (4 waves = 128 pixels)
// then walks over ALL output channels: the pre-split filter streams global -> LDS by LDS-DMA through a ring of 8 KiB stages, six
// stages ahead of its use (the stream of one N step continues into the next without a gap), a wave reads every A fragment of a
// stage and issues 12 MFMAs per stage.  The residual rows of an N step are requested when its k loop starts and arrive while it
// runs.  Per output element the operations are those of conv_x3_kernel (same split, same k order, h.h + h.l + l.h per 16-deep
// chunk into one fp32 accumulator, same epilogue): the two kernels agree bit for bit.
#include "conv_common.h"

namespace {
typedef _Float16 xs_h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 xs_h16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t xs_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
// x * s = h + l (conv_bf16x3.hip split2h, element for element)
__device__ __forceinline__ void xs_split(const f32x4 v, const float s, xs_h16x4 &h, xs_h16x4 &l) {
    const f32x4 xs = v * s;
    h = __builtin_convertvector(xs, xs_h16x4);
    const f32x4 r = xs - __builtin_convertvector(h, f32x4);
    l = __builtin_convertvector(r, xs_h16x4);
}
template <int N>
__device__ __forceinline__ void xs_wait_vm() {
    __asm__ volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int XS_NST = 7;             // ring stages of 8 KiB
constexpr int XS_D = XS_NST - 1;      // a stage's DMA is issued this many steps before its fragments are read
constexpr int XS_STAGE = 8192;        // bytes
constexpr int xs_lds_bytes(int bn) { return XS_NST * XS_STAGE + 4 * 4096 + 2 * 2 * bn * 4; }

// KC = Cin / 16.  NG = 32-channel groups per N step (BN = 32 NG), KS = 16-deep chunks per ring stage: NG * KS = 4.
template <int KC, int NG, int KS>
__global__ __launch_bounds__(256, 2) void conv_xs_kernel(const a3d_conv_desc d, const int M, const int full_tiles, const int ns_tail) {
    static_assert(NG * KS == 4 && KC % KS == 0, "a ring stage is 8 KiB: 4 (chunk, group) pairs x 2 planes x 1 KiB");
    constexpr int BN = 32 * NG;
    constexpr int SPT = KC / KS;   // ring steps per N step
    constexpr int R = 4 * NG;      // residual loads (16 B per lane) of an N step
    extern __shared__ __attribute__((aligned(16))) unsigned char xs_lds[];
    unsigned char *ring = xs_lds;
    float *Tall = reinterpret_cast<float *>(xs_lds + XS_NST * XS_STAGE);
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
    const int hwo = d.Ho * d.Wo;
    const int Q = nsteps * SPT;

    // ---- filter stream: w_x3 [Cin/16][2][Cout][16] fp16; piece (chunk c, plane p, rows n .. n+31) is 1 KiB contiguous.  A stage holds
    // pieces [chunk-in-stage][plane][group]; wave w moves pieces 2w and 2w+1.  Lane i lands at LDS byte 16 i of its piece = row i/2,
    // half i%2, and fetches the k half the image keeps there: half ^ ((row >> 3) & 1)  (the layout conv_x3_kernel reads).
    const __amdgpu_buffer_rsrc_t rw = xs_rsrc(d.w_x3, (unsigned)((size_t)KC * d.Cout * 64));
    const int wvoff = (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) << 4);
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    int dma_q = 0, dma_st = 0, rd_st = 0;  // next step to fetch; ring stage it goes to; ring stage the next fragment reads come from
    auto dma = [&]() {
        const int q = dma_q++;
        const int ns = q / SPT, t = q - ns * SPT;
        unsigned char *st = ring + dma_st * XS_STAGE;
        dma_st = dma_st == XS_NST - 1 ? 0 : dma_st + 1;
        const int voff = q < Q ? wvoff : -1;  // (past the last step: out of range -- zeros into a stage nobody reads; the op count per step stays fixed)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = uw * 2 + i;
            const int kl = j / (2 * NG), p = (j / NG) & 1, g = j % NG;
            const int c = t * KS + kl;
            const int soff = q < Q ? ((c * 2 + p) * d.Cout + nbeg + ns * BN + g * 32) * 32 : 0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void *)(st + j * 1024), 16, voff,
                                                     __builtin_amdgcn_readfirstlane(soff), 0, 0);
        }
    };
#pragma unroll
    for (int i = 0; i < XS_D; ++i) dma();

    // ---- the wave's activations: lane (pixel lane % 32, k group lane / 32) holds channels 16 c + 8 (lane / 32) .. + 7 of every chunk c
    const int mp = m0 + (lane & 31);
    const bool mok = mp < M;
    const float sx = mok ? a3d_in_scale(d, mp / hwo) : 1.f;
    xs_h16x8 xh[KC], xl[KC];
    {
        const __amdgpu_buffer_rsrc_t rx = xs_rsrc(d.x, (unsigned)((size_t)M * d.Cin * 4));
        const int voff = mok ? (mp * d.Cin + (lane >> 5) * 8) * 4 : -1;
        f32x4 raw[KC][2];
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            raw[c][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, c * 64, 0));
            raw[c][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, voff, c * 64 + 16, 0));
        }
        xs_wait_vm<0>();
#pragma unroll
        for (int c = 0; c < KC; ++c) {
            xs_h16x4 h0, l0, h1, l1;
            xs_split(raw[c][0], sx, h0, l0);
            xs_split(raw[c][1], sx, h1, l1);
            xh[c] = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
            xl[c] = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    }

    const int frow = lane & 31;
    const int frag_off = (frow * 16 + ((((lane >> 5) ^ (frow >> 3)) & 1) << 3)) * 2;  // bytes inside a 1 KiB piece
    const int pr = lane & 31, ph = lane >> 5;  // accumulator layout: pixel, channel quad half
    const int qr = lane >> 3, qc = lane & 7;   // row-major epilogue: row (+ 8 j), channel quad
    float *T = Tall + wave * 1024;
    const bool has_res = d.res != nullptr;
    const float unx = 1.f / sx, unw = 1.f / d.w_scale;
    const bool one_image = m0 < M && m0 / hwo == min(m0 + 31, M - 1) / hwo;
    float vmax[4] = {0.f, 0.f, 0.f, 0.f};  // running maxima of the wave's rows qr + 8 j over ALL its N steps (recorded once, at the end)

    int q = 0;
    for (int ns = 0; ns < nsteps; ++ns) {
        const int n0 = nbeg + ns * BN;
        float *ss = ssall + (ns & 1) * 2 * BN;
        if (tid < BN) {
            ss[tid] = d.scale ? d.scale[n0 + tid] : 1.f;
            ss[BN + tid] = d.shift ? d.shift[n0 + tid] : 0.f;
        }
        // the residual rows of this N step, in the row-major form the epilogue stores: requested now, used behind the k loop
        f32x4 rv[NG][4];
        if (has_res) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float *rrow = d.res + (size_t)min(m0 + qr + 8 * j, M - 1) * d.Cout + n0 + qc * 4;
#pragma unroll
                for (int g = 0; g < NG; ++g) rv[g][j] = *reinterpret_cast<const f32x4 *>(rrow + g * 32);
            }
        }
        f32x16 acc[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

#pragma unroll
        for (int t = 0; t < SPT; ++t, ++q) {
            // the DMA of step q (issued XS_D steps ago) has landed: younger than it are the DMAs of the XS_D - 1 steps since and,
            // during the first XS_D steps of an N step, that step's residual loads (loads retire in order)
            if (has_res && t < XS_D) xs_wait_vm<2 * (XS_D - 1) + R>();
            else xs_wait_vm<2 * (XS_D - 1)>();
            // (a bare barrier: __syncthreads() carries a workgroup fence, and the compiler completes every LDS-DMA in flight in front
            // of a fence -- vmcnt(0) -- which would put the whole ring's latency back into every step)
            __asm__ volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const unsigned char *st = ring + rd_st * XS_STAGE + frag_off;
            rd_st = rd_st == XS_NST - 1 ? 0 : rd_st + 1;
            dma();  // step q + XS_D, into the stage read at step q - 1 (every wave is past that read: the barrier above)
#pragma unroll
            for (int kl = 0; kl < KS; ++kl) {
                const int c = t * KS + kl;
                xs_h16x8 fa[2][NG];
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int g = 0; g < NG; ++g) fa[p][g] = *reinterpret_cast<const xs_h16x8 *>(st + ((kl * 2 + p) * NG + g) * 1024);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][g], xh[c], acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[0][g], xl[c], acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[1][g], xh[c], acc[g], 0, 0, 0);
            }
        }
        // ---- epilogue of the N step (conv_x3_kernel's row-major form: a 32 x 32 tile goes through 4 KiB of LDS, XOR-swizzled)
        if (has_res) {  // the residual loads are older than the DMAs of the last min(SPT, XS_D) steps
            if constexpr (SPT < XS_D) xs_wait_vm<2 * SPT>();
            else xs_wait_vm<2 * XS_D>();
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                f32x4 v = {acc[g][rg * 4 + 0], acc[g][rg * 4 + 1], acc[g][rg * 4 + 2], acc[g][rg * 4 + 3]};
                v = (v * unx) * unw;  // exact: powers of two
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
                const f32x4 v = a3d_epilogue_math(d, tv[j], *reinterpret_cast<const f32x4 *>(ss + nl), *reinterpret_cast<const f32x4 *>(ss + BN + nl), has_res, rv[g][j]);
                vmax[j] = fmaxf(vmax[j], a3d_absmax4(v));
                *reinterpret_cast<f32x4 *>(d.y + (size_t)m * d.Cout + n0 + nl) = v;
            }
        }
    }
    // The maxima go out ONCE per wave, behind its last store: an atomic on an image's slot queues behind every other workgroup's at the
    // L2, and a wave that goes on to another N step would wait for it at its next counted vmcnt (measured: 0.97 -> 0.43 ms on the
    // res2 64 -> 256 layer).
    if (d.y_amax) {
        if (one_image) {
            a3d_note_amax(d.y_amax, m0 / hwo, fmaxf(fmaxf(vmax[0], vmax[1]), fmaxf(vmax[2], vmax[3])), true);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + qr + 8 * j;
                float v = vmax[j];
                v = fmaxf(v, __shfl_xor(v, 1, 64));
                v = fmaxf(v, __shfl_xor(v, 2, 64));
                v = fmaxf(v, __shfl_xor(v, 4, 64));
                a3d_note_amax(d.y_amax, m < M ? m / hwo : 0, v, m < M && qc == 0);
            }
        }
    }
    xs_wait_vm<0>();  // the DMAs issued past the last step must not land in the LDS of the next workgroup
}

template <int KC, int NG, int KS>
int launch_xs(const a3d_conv_desc *d, hipStream_t s) {
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
    static a3d_attr_once attr;
    if (attr.needed()) {
        if (hipFuncSetAttribute((const void *)conv_xs_kernel<KC, NG, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, xs_lds_bytes(BN)) != hipSuccess) return A3D_ERR_LAUNCH;
        attr.mark();
    }
    a3d_note_variant("conv_h2xs_kernel<%d>", 16 * KC);
    hipLaunchKernelGGL((conv_xs_kernel<KC, NG, KS>), dim3(blocks), dim3(256), xs_lds_bytes(BN), s, *d, M, full, ns_tail);
    return a3d_check_launch();
}
}  // namespace

// A3D_ERR_UNSUPPORTED: not a layer of this form (the caller goes on to the tiled kernels).
int a3d_conv_launch_xs_h2(const a3d_conv_desc *d, hipStream_t s) {
    if (d->precision != 3 || !d->w_x3 || !d->in_amax || d->in_amax2 || !(d->w_scale > 0.f)) return A3D_ERR_UNSUPPORTED;
    if (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0 || d->Kpad != d->Cin) return A3D_ERR_UNSUPPORTED;
    if (d->stem || d->ups || d->phase || d->pixshuf || d->gate || d->x2 || d->Cin2 || d->splitk != 1 || d->m_dev || d->res_ups) return A3D_ERR_UNSUPPORTED;
    if (d->Cout % 128) return A3D_ERR_UNSUPPORTED;
    const size_t M = (size_t)d->B * d->Ho * d->Wo;
    if (M * d->Cin * 4 >= ((size_t)1 << 31) || (size_t)d->Cout * d->Cin * 4 >= ((size_t)1 << 31)) return A3D_ERR_UNSUPPORTED;
    // Where it pays (measured, tools/xs_check.py): Cin 64 and 128 once the pixel tiles fill the chip's 512 workgroup slots -- 64 -> 256 +
    // residual 0.68 -> 0.55 ms (5.1 TB/s), 128 -> 512 + residual 0.43 -> 0.34 ms.  At Cin 256 the 128 fragment registers leave 64-wide N
    // steps and the two forms tie on the layers without a residual (256 -> 256 at 120x160: 0.83 | 0.83 ms): those stay with the tiled kernel
    // unless tune 13 asks.
    // (Cin 256 with a residual and Cout >= 512 -- res4's expansions -- 0.256 -> 0.222 ms once the maxima moved behind the last store.)
    if (d->tune != 13 && (M < 128 * 512 || (d->Cin > 128 && !(d->res && d->Cout >= 512)))) return A3D_ERR_UNSUPPORTED;
    switch (d->Cin) {
    case 64: return launch_xs<4, 4, 1>(d, s);
    case 128: return launch_xs<8, 2, 2>(d, s);
    case 256: return launch_xs<16, 2, 2>(d, s);
    default: return A3D_ERR_UNSUPPORTED;
    }
}
