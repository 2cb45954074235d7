// Weight gradient in the bf16 arithmetic (a3d_wgrad_desc.precision == 1), second form (round 4): transposed LDS reads, taps shared.
//
//   dw[co][kh][kw][ci] = sum over output pixels p of dy[p][co] * x[p shifted by the tap][ci]
//
// The reduction index of both GEMM operands is the PIXEL, the slow index of an NHWC tensor, while v_mfma_f32_32x32x16_bf16 wants
// 8 consecutive k per lane.  conv_wgrad_bf16_kernel (conv_wgrad.hip) transposes on the way into LDS: a thread owns one channel and
// fetches it pixel by pixel -- 32 dword loads per thread and 32-pixel chunk beside 8 MFMAs per wave -- and every (128 x 128 tile, tap)
// workgroup streams both operands again: 11.3 GB through L2 for the 3x3 256 -> 256 layer of the p2 level at 16 images, 212 TFLOP/s.
// Here
//   * a chunk of 64 pixels goes into LDS AS IT LIES IN MEMORY, [pixel][channel] in bf16 (16-byte loads, 4 or 8 channels per lane,
//     rounded on the way in where the tensor is stored as fp32), and the fragments come out with ds_read_b64_tr_b16, the gfx950
//     transposing read (a 16-lane group reads 4 pixel rows x 16 channels and receives them channel-major): two reads per operand
//     fragment, conflict-free with the 16-byte chunks of a row XOR-ed by 4 (row & 3);
//   * a 3x3 stride-1 pad-1 layer runs the three taps of a filter ROW in one workgroup: the pixels are numbered along rows PADDED to
//     W + 2 (the two extra slots of a row carry dy = 0), so tap kw of pixel k reads the x row k + kw - 1 of the same numbering -- the
//     three taps are three shifted views (row offsets 0, 1, 2) of one 66-row patch, without any edge case: where a shifted row leaves
//     the image it lands on a padding slot (x = 0) or multiplies a padding pixel (dy = 0).  dy is loaded once for three taps, x once;
//   * 512 threads, 128 (co) x 128 (ci) x 3 taps or 128 x 256 x 1 tap per workgroup; loads of chunk c + 1 are in flight across the
//     MFMAs of chunk c (36 / 48 staging registers per thread = 72 / 96 KiB per CU); one barrier per chunk.
// The partial sums go to the same workspace layout [slice][co][tap][ci] as the first form's: the slice reduction (per launch or
// batched) is unchanged.  Products are the same bf16 roundings of the same values; the summation order over pixels differs from the
// first form's (other chunk and slice boundaries), i.e. results agree to fp32 rounding of the sums, not bit for bit.
#include "a3d_common.h"
#include "../../include/a3d.h"

namespace {
typedef short tr_s16x4 __attribute__((ext_vector_type(4)));
typedef short tr_s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 tr_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 tr_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned tr_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) tr_s16x4 *tr_lds_ptr;

constexpr int TR_CH = 64;  // pixels per chunk

constexpr int tr_stage_bytes(int NT, int TA, int TB) { return TR_CH * 2 * TA + (TR_CH + (NT == 3 ? 4 : 0)) * 2 * TB; }
constexpr int tr_lds_bytes(int NT, int TA, int TB) { return 2 * tr_stage_bytes(NT, TA, TB); }

// NT: taps (kw = 0 .. NT - 1 of filter row kh) per workgroup.  MB / NB: 32-channel blocks of co / ci per wave; WM x WN waves.
// XB / YB: x / dy stored as bf16.
template <int NT, int MB, int NB, int WM, int WN, bool XB, bool YB>
__global__ __launch_bounds__(64 * WM * WN, 1) void conv_wgrad_tr_kernel(const a3d_wgrad_desc d, const int Pp, const int Wp, const float invWp, const float invHo,
                                                                        const int mtiles, const int ntiles, const int chunk) {
    constexpr int NTH = 64 * WM * WN;
    constexpr int TA = 32 * MB * WM, TB = 32 * NB * WN;
    constexpr int RA = 2 * TA, RB = 2 * TB;                            // bytes of an LDS row
    constexpr int ROWS_B = TR_CH + (NT == 3 ? 2 : 0);                  // patch rows in use
    constexpr int SA = TR_CH * RA, STAGE = tr_stage_bytes(NT, TA, TB);
    constexpr int PPR_A = YB ? TA / 8 : TA / 4, RPP_A = NTH / PPR_A, NA = TR_CH / RPP_A;              // 16-byte pieces per row, rows per pass, passes
    constexpr int PPR_B = XB ? TB / 8 : TB / 4, RPP_B = NTH / PPR_B, NBP = (ROWS_B + RPP_B - 1) / RPP_B;
    static_assert(TR_CH % RPP_A == 0 && (RPP_A & 3) == 0 && (RPP_B & 3) == 0, "loader passes");
    extern __shared__ __attribute__((aligned(16))) unsigned char tr_lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    int t = blockIdx.x;
    const int nt = t % ntiles;
    t /= ntiles;
    const int mt = t % mtiles;
    const int kh = t / mtiles;  // filter row (NT == 3) or 0
    const int co0 = mt * TA, ci0 = nt * TB;
    const int p_begin = blockIdx.y * chunk, p_end = min(Pp, p_begin + chunk);
    const int nchunks = (p_end - p_begin + TR_CH - 1) / TR_CH;

    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(d.dy), 0, (int)(((size_t)d.B * d.Ho * d.Wo * d.Cout * 4) >> (YB ? 1 : 0)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(d.x), 0, (int)(((size_t)d.B * d.H * d.W * d.Cin * 4) >> (XB ? 1 : 0)), 0x00020000);

    // ---- loaders: thread = (row tid / PPR + RPP i, piece tid % PPR) of each operand
    const int pa = tid % PPR_A, ra0 = tid / PPR_A, pb = tid % PPR_B, rb0 = tid / PPR_B;
    const int cha = co0 + pa * (YB ? 8 : 4), chb = ci0 + pb * (XB ? 8 : 4);  // first channel of the piece
    const bool cva = cha < d.Cout, cvb = chb < d.Cin;
    // LDS byte of the piece inside its row: 16-byte chunk index ^ ((row & 3) << 2); (row & 3) is the same in every pass
    const int wa = YB ? 16 * (pa ^ ((ra0 & 3) << 2)) : 16 * ((pa >> 1) ^ ((ra0 & 3) << 2)) + 8 * (pa & 1);
    const int wb = XB ? 16 * (pb ^ ((rb0 & 3) << 2)) : 16 * ((pb >> 1) ^ ((rb0 & 3) << 2)) + 8 * (pb & 1);
    tr_u32x4 ga[NA], gb[NBP];
    auto load = [&](const int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int kap = k0 + ra0 + RPP_A * i;
            int off = -1;
            if (NT == 3) {
                const int q = (int)(((float)kap + 0.5f) * invWp), j = kap - q * Wp;
                if (kap < p_end && (unsigned)j < (unsigned)d.Wo && cva) off = ((q * d.Wo + j) * d.Cout + cha) * (YB ? 2 : 4);
            } else if (kap < p_end && cva) {
                off = (kap * d.Cout + cha) * (YB ? 2 : 4);
            }
            ga[i] = __builtin_bit_cast(tr_u32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, off, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < NBP; ++i) {
            const int row = rb0 + RPP_B * i;
            int off = -1;
            if (NT == 3) {
                const int kap = k0 - 1 + row;
                const int q = (int)(((float)kap + 0.5f) * invWp), j = kap - q * Wp;
                const int b = (int)(((float)q + 0.5f) * invHo), ih = q - b * d.Ho + kh - 1;
                if (row < ROWS_B && kap >= 0 && kap < Pp && (unsigned)j < (unsigned)d.W && (unsigned)ih < (unsigned)d.H && cvb) off = (((b * d.H + ih) * d.W + j) * d.Cin + chb) * (XB ? 2 : 4);
            } else {
                const int kap = k0 + row;
                if (kap < p_end && cvb) off = (kap * d.Cin + chb) * (XB ? 2 : 4);
            }
            gb[i] = __builtin_bit_cast(tr_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        }
    };
    auto store = [&](const int stage) {
        unsigned char *A = tr_lds + stage * STAGE, *Bp = A + SA;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            unsigned char *p = A + (ra0 + RPP_A * i) * RA + wa;
            if constexpr (YB) *reinterpret_cast<tr_u32x4 *>(p) = ga[i];
            else *reinterpret_cast<tr_bf16x4 *>(p) = __builtin_convertvector(__builtin_bit_cast(f32x4, ga[i]), tr_bf16x4);
        }
#pragma unroll
        for (int i = 0; i < NBP; ++i) {
            const int row = rb0 + RPP_B * i;
            if (NBP * RPP_B > ROWS_B && row >= ROWS_B) continue;
            unsigned char *p = Bp + row * RB + wb;
            if constexpr (XB) *reinterpret_cast<tr_u32x4 *>(p) = gb[i];
            else *reinterpret_cast<tr_bf16x4 *>(p) = __builtin_convertvector(__builtin_bit_cast(f32x4, gb[i]), tr_bf16x4);
        }
    };

    // ---- fragment addresses (ds_read_b64_tr_b16): lane = 16 g + 4 q + p supplies row q (+ 8 (g / 2): the k half of the fragment),
    // channels 16 (g % 2) + 4 p .. + 3 of its 32-channel block; block index blk -> chunk 4 blk + 2 (g % 2) + p / 2, byte 8 (p % 2)
    const int g = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    auto frag_base = [&](const int rowbytes, const int blk, const int shift) {
        const int row = shift + q4 + 8 * (g >> 1);
        return rowbytes * row + 16 * ((4 * blk + 2 * (g & 1) + (p4 >> 1)) ^ (((shift + q4) & 3) << 2)) + 8 * (p4 & 1);
    };
    int fa[MB], fb[NT][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i) fa[i] = frag_base(RA, wm * MB + i, 0);
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
        for (int j = 0; j < NB; ++j) fb[s][j] = SA + frag_base(RB, wn * NB + j, s);
    auto frag = [&](const int base, const int imm) {  // four pixel rows of the lane's k half
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_lds_ptr)(tr_lds + base + imm));
    };

    f32x16 acc[NT][MB][NB];
#pragma unroll
    for (int s = 0; s < NT; ++s)
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[s][i][j][r] = 0.f;

    if (nchunks > 0) {  // (an empty slice stores its zeros: the reduction reads every slice)
        load(p_begin);
        store(0);
        __syncthreads();
    }
    for (int c = 0; c < nchunks; ++c) {
        const int st = (c & 1) * STAGE;
        if (c + 1 < nchunks) load(p_begin + (c + 1) * TR_CH);
#pragma unroll
        for (int ks = 0; ks < TR_CH / 16; ++ks) {
            tr_bf16x8 a[MB], b[NT][NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const tr_s16x4 lo = frag(fa[i] + st, RA * (16 * ks)), hi = frag(fa[i] + st, RA * (16 * ks + 4));
                a[i] = __builtin_bit_cast(tr_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
#pragma unroll
            for (int s = 0; s < NT; ++s)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const tr_s16x4 lo = frag(fb[s][j] + st, RB * (16 * ks)), hi = frag(fb[s][j] + st, RB * (16 * ks + 4));
                    b[s][j] = __builtin_bit_cast(tr_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
            for (int s = 0; s < NT; ++s)
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) acc[s][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[s][j], acc[s][i][j], 0, 0, 0);
        }
        if (c + 1 < nchunks) store((c + 1) & 1);
        __syncthreads();
    }

    // partial[slice][co][tap][ci]; accumulator register r of lane l = row (r/4)*8 + (l/32)*4 + r%4, column l%32
    float *out = d.workspace + (size_t)blockIdx.y * d.Cout * d.KH * d.KW * d.Cin;
    const int taps = d.KH * d.KW;
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        const int tap = kh * d.KW + s;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int ci = ci0 + (wn * NB + j) * 32 + (lane & 31);
                if (ci >= d.Cin) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + (wm * MB + i) * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
                    if (co < d.Cout) out[((size_t)co * taps + tap) * d.Cin + ci] = acc[s][i][j][r];
                }
            }
    }
}

template <int NT, int MB, int NB, int WM, int WN, bool XB, bool YB>
int tr_launch(const a3d_wgrad_desc *d, hipStream_t s, const int Pp, const int Wp) {
    constexpr int TA = 32 * MB * WM, TB = 32 * NB * WN;
    const int mtiles = (d->Cout + TA - 1) / TA, ntiles = (d->Cin + TB - 1) / TB;
    int chunk = (Pp + d->splitk - 1) / d->splitk;
    chunk = (chunk + TR_CH - 1) / TR_CH * TR_CH;
    static a3d_attr_once attr;
    if (attr.needed()) {
        if (hipFuncSetAttribute((const void *)conv_wgrad_tr_kernel<NT, MB, NB, WM, WN, XB, YB>, hipFuncAttributeMaxDynamicSharedMemorySize, tr_lds_bytes(NT, TA, TB)) != hipSuccess)
            return A3D_ERR_LAUNCH;
        attr.mark();
    }
    hipLaunchKernelGGL((conv_wgrad_tr_kernel<NT, MB, NB, WM, WN, XB, YB>), dim3(mtiles * ntiles * (NT == 3 ? 3 : 1), d->splitk), dim3(64 * WM * WN), tr_lds_bytes(NT, TA, TB), s, *d,
                       Pp, Wp, 1.f / (float)Wp, 1.f / (float)d->Ho, mtiles, ntiles, chunk);
    return a3d_check_launch();
}
template <int NT, int MB, int NB, int WM, int WN>
int tr_launch_io(const a3d_wgrad_desc *d, hipStream_t s, const int Pp, const int Wp) {
    switch (d->io_bf16) {
    case 0: return tr_launch<NT, MB, NB, WM, WN, false, false>(d, s, Pp, Wp);
    case 1: return tr_launch<NT, MB, NB, WM, WN, true, false>(d, s, Pp, Wp);
    case 2: return tr_launch<NT, MB, NB, WM, WN, false, true>(d, s, Pp, Wp);
    default: return tr_launch<NT, MB, NB, WM, WN, true, true>(d, s, Pp, Wp);
    }
}
}  // namespace

// 0: not a layer of this form (the caller runs conv_wgrad_bf16_kernel); 3 / 1: taps per workgroup
int a3d_wgrad_tr_form(const a3d_wgrad_desc *d) {
    if (a3d_dev_knob("A3D_WGRAD_TR", 1) == 0 || d->precision != 1 || d->stride != 1 || d->H != d->Ho || d->W != d->Wo) return 0;
    if (((d->io_bf16 & 1) && (d->Cin & 7)) || ((d->io_bf16 & 2) && (d->Cout & 7))) return 0;
    if ((size_t)d->B * d->Ho * d->Wo * d->Cout * 4 >= ((size_t)1 << 31) || (size_t)d->B * d->H * d->W * d->Cin * 4 >= ((size_t)1 << 31)) return 0;
    if (d->KH == 3 && d->KW == 3 && d->pad == 1) return (size_t)d->B * d->Ho * (d->Wo + 2) < ((size_t)1 << 21) ? 3 : 0;  // (the float divisions of the loader are exact below 2^21)
    if (d->KH == 1 && d->KW == 1 && d->pad == 0) return 1;
    return 0;
}

// workgroup tiles of one pixel slice and the length of the reduction, for the caller's choice of `splitk`
extern "C" int a3d_wgrad_tiles(const a3d_wgrad_desc *d, int *tiles, int *reduction) {
    if (!d || !tiles || !reduction) return A3D_ERR_ARG;
    const int form = a3d_wgrad_tr_form(d);
    if (form == 3) {
        *tiles = ((d->Cout + 127) / 128) * ((d->Cin + 127) / 128) * 3;
        *reduction = d->B * d->Ho * (d->Wo + 2);
    } else if (form == 1) {
        *tiles = ((d->Cout + 127) / 128) * ((d->Cin + 255) / 256);
        *reduction = d->B * d->Ho * d->Wo;
    } else {
        *tiles = ((d->Cout + 127) / 128) * ((d->Cin + 127) / 128) * d->KH * d->KW;
        *reduction = d->B * d->Ho * d->Wo;
    }
    return form;
}

int a3d_wgrad_launch_tr(const a3d_wgrad_desc *d, hipStream_t s) {
    const int form = a3d_wgrad_tr_form(d);
    if (form == 3) return tr_launch_io<3, 2, 1, 2, 4>(d, s, d->B * d->Ho * (d->Wo + 2), d->Wo + 2);
    if (form == 1) return tr_launch_io<1, 2, 2, 2, 4>(d, s, d->B * d->Ho * d->Wo, d->Wo);
    return A3D_ERR_UNSUPPORTED;
}
