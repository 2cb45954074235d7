// Weight gradient of the fused convolution / linear layers (training step).
//
//   dw[co][kh][kw][ci] = scale[co] * sum_{b,oh,ow} dy[b,oh,ow,co] * x[b, oh*s + kh - pad, ow*s + kw - pad, ci]
//
// i.e. for every filter tap (kh,kw) a GEMM  dY^T [Cout x P] . X_tap [P x Cin]  whose reduction runs over the P = B*Ho*Wo
// output pixels.  Both operands are NHWC, so the reduction index (pixel) is the SLOW index of both: a k-chunk of 16 pixels
// is loaded as 16 rows of 128 consecutive channels (float4 per lane, fully coalesced) and kept in LDS as [k][channel].
// In that layout the fp32 MFMA fragments (v_mfma_f32_32x32x2: lane l holds A[m = l%32][k = l/32], B[k = l/32][n = l%32])
// are plain ds_read_b32 with consecutive lanes on consecutive words; the row pitch is 160 floats (= 32 mod 64 banks) so
// the two 32-lane halves of a wave hit disjoint banks.  One ds_read per operand fragment feeds 2 MFMAs (64x64 wave
// tile), i.e. 1 LDS word-read instruction per 64-cycle MFMA: far from the LDS limit.
// Workgroup = 128 (co) x 128 (ci) x one tap x one pixel slice; 4 waves as 2x2, each 64x64 (4 accumulators).  Pixel
// slices (split-K) write partials to the workspace; a second launch sums them in slice order (deterministic), applies
// the folded-BN scale and stores or accumulates into dw, which has the packed forward layout [Cout][KH][KW][Cin].
#include "a3d_common.h"
#include "../../include/a3d.h"

namespace {
constexpr int WG_BK = 16;    // pixels per chunk
constexpr int WG_LD = 160;   // LDS row pitch in floats

__global__ __launch_bounds__(256) void conv_wgrad_kernel(const a3d_wgrad_desc d, const int P, const int mtiles, const int ntiles,
                                                          const int chunk) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][WG_BK * WG_LD];  // [buffer][A|B][k][channel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int t = blockIdx.x;
    const int nt = t % ntiles;
    t /= ntiles;
    const int mt = t % mtiles;
    const int tap = t / mtiles;
    const int kh = tap / d.KW, kw = tap - kh * d.KW;
    const int co0 = mt * 128, ci0 = nt * 128;
    const int p_begin = blockIdx.y * chunk, p_end = min(P, p_begin + chunk);

    const int lrow = tid >> 5;          // 0..7 (+8 for the second row)
    const int lcol = (tid & 31) * 4;    // channel offset inside the 128-wide tile
    const bool a_ok = co0 + lcol < d.Cout, b_ok = ci0 + lcol < d.Cin;
    const int HoWo = d.Ho * d.Wo;

    f32x4 ra[2], rb[2];
    auto load = [&](int p0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = p0 + lrow + 8 * i;
            f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
            if (p < p_end) {
                if (a_ok) va = *reinterpret_cast<const f32x4 *>(d.dy + (size_t)p * d.Cout + co0 + lcol);
                const int b = p / HoWo, r = p - b * HoWo;
                const int oh = r / d.Wo, ow = r - oh * d.Wo;
                const int iy = oh * d.stride + kh - d.pad, ix = ow * d.stride + kw - d.pad;
                if (b_ok && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W)
                    vb = *reinterpret_cast<const f32x4 *>(d.x + (((size_t)b * d.H + iy) * d.W + ix) * d.Cin + ci0 + lcol);
            }
            ra[i] = va;
            rb[i] = vb;
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f32x4 *>(&lds[buf][0][(lrow + 8 * i) * WG_LD + lcol]) = ra[i];
            *reinterpret_cast<f32x4 *>(&lds[buf][1][(lrow + 8 * i) * WG_LD + lcol]) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = (p_end - p_begin + WG_BK - 1) / WG_BK;
    if (nchunks > 0) {
        load(p_begin);
        store(0);
        __syncthreads();
    }
    const int foff = (lane >> 5) * WG_LD + (lane & 31);
    for (int c = 0; c < nchunks; ++c) {
        const int cur = c & 1;
        if (c + 1 < nchunks) load(p_begin + (c + 1) * WG_BK);
        const float *A = &lds[cur][0][foff + wm * 64];
        const float *Bm = &lds[cur][1][foff + wn * 64];
#pragma unroll
        for (int k = 0; k < WG_BK; k += 2) {
            const float a0 = A[k * WG_LD], a1 = A[k * WG_LD + 32];
            const float b0 = Bm[k * WG_LD], b1 = Bm[k * WG_LD + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (c + 1 < nchunks) store(cur ^ 1);
        __syncthreads();
    }

    // partial[slice][co][tap][ci]; accumulator register r of lane l = row (r/4)*8 + (l/32)*4 + r%4, column l%32
    float *out = d.workspace + (size_t)blockIdx.y * d.Cout * d.KH * d.KW * d.Cin;
    const int taps = d.KH * d.KW;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = ci0 + wn * 64 + j * 32 + (lane & 31);
            if (ci >= d.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 64 + i * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
                if (co < d.Cout) out[((size_t)co * taps + tap) * d.Cin + ci] = acc[i][j][r];
            }
        }
}

__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const a3d_wgrad_desc d, const int slices) {
    const size_t row = (size_t)d.KH * d.KW * d.Cin;  // floats per output channel
    const size_t total4 = (size_t)d.Cout * row / 4;
    const size_t plane = (size_t)d.Cout * row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < slices; ++k) s += *reinterpret_cast<const f32x4 *>(d.workspace + (size_t)k * plane + i * 4);
        if (d.scale) s *= d.scale[(i * 4) / row];
        f32x4 *o = reinterpret_cast<f32x4 *>(d.dw + i * 4);
        *o = d.accumulate ? *o + s : s;
    }
}

// n layers in one launch: blockIdx.y = layer (a3d_wgrad_reduce_batch).  Per element the same operations as above.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_batch_kernel(const a3d_wgrad_desc *__restrict__ table) {
    const a3d_wgrad_desc d = table[blockIdx.y];
    const size_t row = (size_t)d.KH * d.KW * d.Cin;
    const size_t total4 = (size_t)d.Cout * row / 4;
    const size_t plane = (size_t)d.Cout * row;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < d.splitk; ++k) s += *reinterpret_cast<const f32x4 *>(d.workspace + (size_t)k * plane + i * 4);
        if (d.scale) s *= d.scale[(i * 4) / row];
        f32x4 *o = reinterpret_cast<f32x4 *>(d.dw + i * 4);
        *o = d.accumulate ? *o + s : s;
    }
}

// ------------------------------------------------------------------------------------------------
// bf16-MFMA variant (a3d_wgrad_desc.precision == 1; see conv_bf16.hip for the why).  The fragments of
// v_mfma_f32_32x32x16_bf16 want 8 CONSECUTIVE k (= pixels) per lane, but NHWC makes channels contiguous, so the transpose
// happens on the way into LDS: a thread owns ONE channel and loads it for 8 consecutive pixels (a wave still reads 256
// contiguous bytes per pixel), rounds the 8 values to bf16 and writes them as one 16-byte row segment of the LDS image
// [channel][32 pixels] (80-byte pitch: conflict-free ds_write_b128 / ds_read_b128).  Chunk = 32 pixels.
// ------------------------------------------------------------------------------------------------
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef float wg_f32x8 __attribute__((ext_vector_type(8)));

// x == h + m + l exactly (round-to-nearest-even at each level): the 3-way split of csrc/conv_bf16x3.hip, 8 wide
__device__ __forceinline__ void wg_split3(const wg_f32x8 v, wg_bf16x8 &h, wg_bf16x8 &m, wg_bf16x8 &l) {
    h = __builtin_convertvector(v, wg_bf16x8);
    const wg_f32x8 r1 = v - __builtin_convertvector(h, wg_f32x8);
    m = __builtin_convertvector(r1, wg_bf16x8);
    const wg_f32x8 r2 = r1 - __builtin_convertvector(m, wg_f32x8);
    l = __builtin_convertvector(r2, wg_bf16x8);
}

// X3 = false: precision 1 (operands rounded to bf16, chunk of 32 pixels).  X3 = true: precision 2, fp32-grade -- both
// operands split exactly into hi | mid | lo bf16 planes in LDS and six MFMAs per 16-pixel k step (conv_bf16x3.hip has the
// arithmetic); chunk of 16 pixels so that 3 planes x 2 operands x 2 buffers stay at 72 KB (2 workgroups per CU).
// IO (X3 = false only): a3d_wgrad_desc.io_bf16 -- bit 0: x, bit 1: dy stored as bf16 (a compile-time property: each form keeps only
// the staging registers it uses).
template <bool X3, int IO = 0>
__global__ __launch_bounds__(256, X3 ? 2 : 3) void conv_wgrad_bf16_kernel(const a3d_wgrad_desc d, const int P, const int mtiles, const int ntiles,
                                                                         const int chunk) {
    constexpr int BKP = X3 ? 16 : 32, LKB = BKP + 8, NPL = X3 ? 3 : 1, G = BKP / 16, S = BKP / 16;
    constexpr int PL = 128 * LKB;  // one operand plane: [channel][pixel]
    __shared__ __attribute__((aligned(16))) __bf16 lds[2][2][NPL * PL];  // [buffer][A|B][plane][channel][pixel]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int t = blockIdx.x;
    const int nt = t % ntiles;
    t /= ntiles;
    const int mt = t % mtiles;
    const int tap = t / mtiles;
    const int kh = tap / d.KW, kw = tap - kh * d.KW;
    const int co0 = mt * 128, ci0 = nt * 128;
    const int p_begin = blockIdx.y * chunk, p_end = min(P, p_begin + chunk);
    const int ch = tid & 127, kg = tid >> 7;  // channel inside the tile; pixel groups kg*G .. kg*G+G-1 (8 pixels each)
    const bool a_ok = co0 + ch < d.Cout, b_ok = ci0 + ch < d.Cin;
    const int HoWo = d.Ho * d.Wo;
    static_assert(!X3 || IO == 0, "bf16-stored operands belong to the bf16 arithmetic");
    constexpr bool xb = IO & 1, yb = IO & 2;  // operands stored as bf16
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(d.dy), 0, (int)(((size_t)P * d.Cout * 4) >> (yb ? 1 : 0)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(d.x), 0, (int)(((size_t)d.B * d.H * d.W * d.Cin * 4) >> (xb ? 1 : 0)), 0x00020000);
    // fp32-stored operand: a thread owns ONE channel (tid % 128) for G groups of 8 pixels: 8 G dword loads, rounded to bf16 on the way
    // into LDS.  bf16-stored operand (X3 = false only): a thread owns a channel PAIR (tid % 64) for ONE group of 8 pixels (tid / 64):
    // 8 dword loads -- half as many, each still a full dword -- whose low / high halves are the two channels' pixels; no conversion.
    const int cp = tid & 63, pg16 = tid >> 6;
    const bool a_ok2 = co0 + 2 * cp + 1 < d.Cout, b_ok2 = ci0 + 2 * cp + 1 < d.Cin;
    wg_f32x8 ra[G], rb[G];
    unsigned qa[8], qb[8];
    auto load = [&](int p0) {
        if constexpr (!yb || !xb || X3) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int pg = p0 + (kg * G + g) * 8;
                int b = pg / HoWo, r = pg - b * HoWo;
                int oh = r / d.Wo, ow = r - oh * d.Wo;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int p = pg + k;
                    const bool live = p < p_end;
                    const int iy = oh * d.stride + kh - d.pad, ix = ow * d.stride + kw - d.pad;
                    const bool in = live && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
                    if constexpr (!yb || X3) ra[g][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry, (live && a_ok) ? (p * d.Cout + co0 + ch) * 4 : -1, 0, 0));
                    if constexpr (!xb || X3)
                        rb[g][k] = __builtin_bit_cast(
                            float, __builtin_amdgcn_raw_buffer_load_b32(rx, (in && b_ok) ? (((b * d.H + iy) * d.W + ix) * d.Cin + ci0 + ch) * 4 : -1, 0, 0));
                    if (++ow == d.Wo) {  // next pixel of the run
                        ow = 0;
                        if (++oh == d.Ho) {
                            oh = 0;
                            ++b;
                        }
                    }
                }
            }
        }
        if constexpr (!X3) {
            if constexpr (yb || xb) {
                const int pg = p0 + pg16 * 8;
                int b = pg / HoWo, r = pg - b * HoWo;
                int oh = r / d.Wo, ow = r - oh * d.Wo;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int p = pg + k;
                    const bool live = p < p_end;
                    const int iy = oh * d.stride + kh - d.pad, ix = ow * d.stride + kw - d.pad;
                    const bool in = live && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
                    if constexpr (yb) qa[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(ry, (live && a_ok2) ? (p * d.Cout + co0 + 2 * cp) * 2 : -1, 0, 0);
                    if constexpr (xb) qb[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rx, (in && b_ok2) ? (((b * d.H + iy) * d.W + ix) * d.Cin + ci0 + 2 * cp) * 2 : -1, 0, 0);
                    if (++ow == d.Wo) {
                        ow = 0;
                        if (++oh == d.Ho) {
                            oh = 0;
                            ++b;
                        }
                    }
                }
            }
        }
    };
    auto store_pairs = [&](__bf16 *plane, const unsigned (&q)[8]) {  // [channel][pixel]: the pair's two channels, 8 pixels each
        typedef unsigned wg_u32x4 __attribute__((ext_vector_type(4)));
        wg_u32x4 lo, hi;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo[k] = (q[2 * k] & 0xFFFFu) | (q[2 * k + 1] << 16);
            hi[k] = (q[2 * k] >> 16) | (q[2 * k + 1] & 0xFFFF0000u);
        }
        *reinterpret_cast<wg_u32x4 *>(plane + (2 * cp) * LKB + pg16 * 8) = lo;
        *reinterpret_cast<wg_u32x4 *>(plane + (2 * cp + 1) * LKB + pg16 * 8) = hi;
    };
    auto store = [&](int buf) {
        if constexpr (!X3) {
            if constexpr (yb) store_pairs(&lds[buf][0][0], qa);
            if constexpr (xb) store_pairs(&lds[buf][1][0], qb);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int off = ch * LKB + (kg * G + g) * 8;
            if constexpr (X3) {
                wg_bf16x8 h, m, l;
                wg_split3(ra[g], h, m, l);
                *reinterpret_cast<wg_bf16x8 *>(&lds[buf][0][off]) = h;
                *reinterpret_cast<wg_bf16x8 *>(&lds[buf][0][PL + off]) = m;
                *reinterpret_cast<wg_bf16x8 *>(&lds[buf][0][2 * PL + off]) = l;
                wg_split3(rb[g], h, m, l);
                *reinterpret_cast<wg_bf16x8 *>(&lds[buf][1][off]) = h;
                *reinterpret_cast<wg_bf16x8 *>(&lds[buf][1][PL + off]) = m;
                *reinterpret_cast<wg_bf16x8 *>(&lds[buf][1][2 * PL + off]) = l;
            } else {
                if constexpr (!yb) *reinterpret_cast<wg_bf16x8 *>(&lds[buf][0][off]) = __builtin_convertvector(ra[g], wg_bf16x8);
                if constexpr (!xb) *reinterpret_cast<wg_bf16x8 *>(&lds[buf][1][off]) = __builtin_convertvector(rb[g], wg_bf16x8);
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nchunks = (p_end - p_begin + BKP - 1) / BKP;
    if (nchunks > 0) {
        load(p_begin);
        store(0);
        __syncthreads();
    }
    const int foff = (lane & 31) * LKB + (lane >> 5) * 8;
    for (int c = 0; c < nchunks; ++c) {
        const int cur = c & 1;
        if (c + 1 < nchunks) load(p_begin + (c + 1) * BKP);
        const __bf16 *A = &lds[cur][0][(wm * 64) * LKB + foff];
        const __bf16 *Bm = &lds[cur][1][(wn * 64) * LKB + foff];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            wg_bf16x8 a[NPL][2], b[NPL][2];
#pragma unroll
            for (int p = 0; p < NPL; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[p][i] = *reinterpret_cast<const wg_bf16x8 *>(A + p * PL + i * 32 * LKB + s * 16);
                    b[p][i] = *reinterpret_cast<const wg_bf16x8 *>(Bm + p * PL + i * 32 * LKB + s * 16);
                }
#define WG_TERM(PA, PB)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA][i], b[PB][j], acc[i][j], 0, 0, 0);
            WG_TERM(0, 0)
            if constexpr (X3) {
                WG_TERM(0, 1)
                WG_TERM(1, 0)
                WG_TERM(1, 1)
                WG_TERM(2, 0)
                WG_TERM(0, 2)
            }
#undef WG_TERM
        }
        if (c + 1 < nchunks) store(cur ^ 1);
        __syncthreads();
    }
    float *out = d.workspace + (size_t)blockIdx.y * d.Cout * d.KH * d.KW * d.Cin;
    const int taps = d.KH * d.KW;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = ci0 + wn * 64 + j * 32 + (lane & 31);
            if (ci >= d.Cin) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 64 + i * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
                if (co < d.Cout) out[((size_t)co * taps + tap) * d.Cin + ci] = acc[i][j][r];
            }
        }
}

int wgrad_check(const a3d_wgrad_desc *d) {
    if (!d || !d->x || !d->dy || !d->dw) return A3D_ERR_ARG;
    if (d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->Cin <= 0 || d->Cout <= 0) return A3D_ERR_ARG;
    if ((d->Cin & 3) || (d->Cout & 3) || d->KH < 1 || d->KW < 1 || d->stride < 1 || d->splitk < 1) return A3D_ERR_ARG;
    if ((size_t)d->B * d->Ho * d->Wo >= ((size_t)1 << 31)) return A3D_ERR_UNSUPPORTED;
    return A3D_OK;
}
}  // namespace

// conv_wgrad_tr.hip: the transposed-read form of the bf16 arithmetic (0 = not a layer of that form)
int a3d_wgrad_tr_form(const a3d_wgrad_desc *d);
int a3d_wgrad_launch_tr(const a3d_wgrad_desc *d, hipStream_t s);

extern "C" size_t a3d_wgrad_workspace_bytes(const a3d_wgrad_desc *d) {
    if (!d || d->splitk < 1) return 0;
    return (size_t)d->splitk * d->Cout * d->KH * d->KW * d->Cin * sizeof(float);
}

extern "C" int a3d_conv_wgrad_nhwc_f32(const a3d_wgrad_desc *d, void *stream) {
    const int rc = wgrad_check(d);
    if (rc != A3D_OK) return rc;
    if (!d->workspace) return A3D_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int P = d->B * d->Ho * d->Wo;
    const int mtiles = (d->Cout + 127) / 128, ntiles = (d->Cin + 127) / 128;
    int chunk = (P + d->splitk - 1) / d->splitk;
    chunk = (chunk + 31) / 32 * 32;  // a multiple of both kernels' k-chunk (16 / 32 pixels)
    if (d->defer_reduce && d->accumulate) return A3D_ERR_ARG;  // (a chain into one dw is ordered by its per-launch reduces)
    if (d->io_bf16 && (d->precision != 1 || (d->io_bf16 & ~3))) return A3D_ERR_ARG;  // bf16-stored operands: the bf16 arithmetic only
    // (the bf16-stored forms load channel PAIRS as dwords: an odd channel count would drop the last channel's gradient and misalign the loads)
    if (((d->io_bf16 & 1) && (d->Cin & 1)) || ((d->io_bf16 & 2) && (d->Cout & 1))) return A3D_ERR_ARG;
    a3d_begin();
    if (d->precision == 1 || d->precision == 2) {
        if ((size_t)P * d->Cout * 4 >= ((size_t)1 << 31) || (size_t)d->B * d->H * d->W * d->Cin * 4 >= ((size_t)1 << 31)) return A3D_ERR_UNSUPPORTED;
        const dim3 grid(mtiles * ntiles * d->KH * d->KW, d->splitk);
        if (d->precision == 1 && a3d_wgrad_tr_form(d)) {
            const int rt = a3d_wgrad_launch_tr(d, s);
            if (rt != A3D_OK) return rt;
        } else if (d->precision == 1) {
            switch (d->io_bf16) {
            case 1: hipLaunchKernelGGL((conv_wgrad_bf16_kernel<false, 1>), grid, dim3(256), 0, s, *d, P, mtiles, ntiles, chunk); break;
            case 2: hipLaunchKernelGGL((conv_wgrad_bf16_kernel<false, 2>), grid, dim3(256), 0, s, *d, P, mtiles, ntiles, chunk); break;
            case 3: hipLaunchKernelGGL((conv_wgrad_bf16_kernel<false, 3>), grid, dim3(256), 0, s, *d, P, mtiles, ntiles, chunk); break;
            default: hipLaunchKernelGGL((conv_wgrad_bf16_kernel<false, 0>), grid, dim3(256), 0, s, *d, P, mtiles, ntiles, chunk); break;
            }
        }
        else hipLaunchKernelGGL((conv_wgrad_bf16_kernel<true>), grid, dim3(256), 0, s, *d, P, mtiles, ntiles, chunk);
    } else
        hipLaunchKernelGGL(conv_wgrad_kernel, dim3(mtiles * ntiles * d->KH * d->KW, d->splitk), dim3(256), 0, s, *d, P, mtiles, ntiles, chunk);
    if (d->defer_reduce) return a3d_check_launch();  // the caller folds the slices later (a3d_wgrad_reduce_batch)
    const size_t total4 = (size_t)d->Cout * d->KH * d->KW * d->Cin / 4;
    int blocks = (int)((total4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, *d, d->splitk);
    return a3d_check_launch();
}

extern "C" int a3d_wgrad_reduce_batch(const a3d_wgrad_desc *table, int n, void *stream) {
    if (!table || n <= 0 || n > 65535) return A3D_ERR_ARG;
    a3d_begin();
    // 64 blocks of 256 threads per layer walk its float4s with a grid stride: 1-12 M floats per layer, slices summed per element
    hipLaunchKernelGGL(conv_wgrad_reduce_batch_kernel, dim3(64, n), dim3(256), 0, (hipStream_t)stream, table);
    return a3d_check_launch();
}
