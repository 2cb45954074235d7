// Persistent pointwise (1x1, stride 1) convolution / linear kernel.
//
// Why a second kernel for the 1x1 layers.  The bottleneck 1x1 convs of the backbone (64->256, 128->512, 256->1024,
// 512->2048 with the residual add, and their 256->64 ... counterparts) have K of 64-2048 only: one 128x128 output tile
// is a handful of k-chunks of MFMA work between a 64 KB operand fetch and a 64 KB (+64 KB residual) epilogue, and in
// conv_gemm_v2's one-tile-per-workgroup form those three phases run back to back -- the measured time of the 64->256
// layer (0.333 ms per 32 frames) is the SUM of its store phase (0.113 ms at 5.5 TB/s, tools/probes/store_pattern.hip),
// its MFMA phase (~0.16 ms) and its fetch, and neither more workgroups per CU nor a deeper load pipeline changes that
// (co-resident workgroups were launched together and stay in the same phase).  Measured here: 0.333 -> 0.277 ms for
// 64->256, 0.241 -> 0.225 for 128->512; ablations: without the stores 0.221 ms, with a quarter of the MFMAs 0.242 ms --
// what remains is the per-chunk barrier / LDS round trip of a K that is only 4 chunks deep.
// Here a workgroup is PERSISTENT: the grid is (CUs x workgroups per CU) and each workgroup walks tiles
// blockIdx.x, +grid, +2*grid, ...  Its k-chunk stream is continuous ACROSS tiles, so the operand loads of the next
// tile are issued (and land in LDS) while the current tile is still being multiplied, and the epilogue stores of tile t
// (fire-and-forget) drain under the MFMAs of tile t+1.  Same LDS image, fragment layout, k order and epilogue as
// conv_gemm_v2 -> results are bit-identical to it.
//
// Addressing: voffset carries the row (m or n) so the buffer range check zeroes rows past M / Cout; the scalar offset
// carries only the k position (soffset does not take part in the range check).
#include "conv_common.h"

namespace {
__device__ __forceinline__ f32x4 pw_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pw_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

template <int TM, int TN, int BKT>
__global__ __launch_bounds__(256, 3) void conv_pw_kernel(const a3d_conv_desc d, const int M, const int ntiles, const int total_tiles) {
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
    constexpr int LK = BKT + 4, TPR = BKT / 4, RPP = 256 / TPR;
    constexpr int XR = BM / RPP, WR = BN / RPP;
    constexpr int BUF = (BM + BN) * LK;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the loader pass");
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
    __shared__ __attribute__((aligned(16))) float ss[2 * BN];  // scale | shift of the current tile's N range

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = tid / TPR, lc = (tid % TPR) * 4;
    const int K = d.Kpad;  // == Cin for the layers routed here
    const int nk = K / BKT;
    const __amdgpu_buffer_rsrc_t rx = pw_rsrc(d.x, (unsigned)((size_t)M * K * 4));
    const __amdgpu_buffer_rsrc_t rw = pw_rsrc(d.w, (unsigned)((size_t)d.Cout * K * 4));
    int xbase[XR], wbase[WR];
#pragma unroll
    for (int i = 0; i < XR; ++i) xbase[i] = ((lr + RPP * i) * K + lc) * 4;
#pragma unroll
    for (int i = 0; i < WR; ++i) wbase[i] = ((lr + RPP * i) * K + lc) * 4;

    // loader state: (tile, chunk) of the next chunk to fetch -- runs two chunks ahead of the multiplier, across tiles
    int ld_tile = blockIdx.x, ld_k = 0;
    int ld_xrow = (ld_tile / ntiles) * BM * K * 4, ld_wrow = (ld_tile % ntiles) * BN * K * 4;  // byte offsets of the tile rows
    f32x4 xs[XR], ws[WR];
    auto load_chunk = [&]() {
        const int soff = ld_k * (BKT * 4);
        // tiles past the end have m0 >= M: the range check returns zeros, no branch needed
#pragma unroll
        for (int i = 0; i < XR; ++i) xs[i] = pw_load4(rx, xbase[i] + ld_xrow, soff);
#pragma unroll
        for (int i = 0; i < WR; ++i) ws[i] = pw_load4(rw, wbase[i] + ld_wrow, soff);
        if (++ld_k == nk) {
            ld_k = 0;
            ld_tile += gridDim.x;
            const int mt = ld_tile / ntiles;
            ld_xrow = mt * BM * K * 4;
            ld_wrow = (ld_tile - mt * ntiles) * BN * K * 4;
        }
    };
    auto store_chunk = [&](int buf) {
        float *X = lds + buf * BUF;
        float *Wt = X + BM * LK;
#pragma unroll
        for (int i = 0; i < XR; ++i) *reinterpret_cast<f32x4 *>(X + (lr + RPP * i) * LK + lc) = xs[i];
#pragma unroll
        for (int i = 0; i < WR; ++i) *reinterpret_cast<f32x4 *>(Wt + (lr + RPP * i) * LK + lc) = ws[i];
    };

    load_chunk();
    store_chunk(0);
    load_chunk();
    __syncthreads();

    const int frag_off = (lane & 31) * LK + (lane >> 5) * 4;
    int cur = 0;
    const bool has_res = d.res != nullptr;
    for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
        const int mt = tile / ntiles, nt = tile - mt * ntiles;
        const int m0 = mt * BM, n0 = nt * BN;
        // this tile's scale / shift: fetched now (lands long before the epilogue), parked in LDS after the k loop
        float es = 1.f, eh = 0.f;
        if (tid < BN && n0 + tid < d.Cout) {
            if (d.scale) es = d.scale[n0 + tid];
            if (d.shift) eh = d.shift[n0 + tid];
        }
        f32x16 acc[TN][TM];
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
        for (int it = 0; it < nk; ++it) {
            const float *X = lds + cur * BUF + (wm * TM * 32) * LK + frag_off;
            const float *Wt = lds + cur * BUF + BM * LK + (wn * TN * 32) * LK + frag_off;
            f32x4 fa[2][TN], fb[2][TM];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) fa[0][ni] = *reinterpret_cast<const f32x4 *>(Wt + ni * 32 * LK);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) fb[0][mi] = *reinterpret_cast<const f32x4 *>(X + mi * 32 * LK);
            store_chunk(cur ^ 1);  // the chunk fetched during the previous iteration (possibly the next tile's first)
#pragma unroll
            for (int q = 0; q < BKT / 8; ++q) {
                const int fc = q & 1, fn = fc ^ 1;
                if (q + 1 < BKT / 8) {
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) fa[fn][ni] = *reinterpret_cast<const f32x4 *>(Wt + ni * 32 * LK + (q + 1) * 8);
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) fb[fn][mi] = *reinterpret_cast<const f32x4 *>(X + mi * 32 * LK + (q + 1) * 8);
                }
                if (q == 0) load_chunk();
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                        for (int mi = 0; mi < TM; ++mi)
                            acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[fc][ni][j], fb[fc][mi][j], acc[ni][mi], 0, 0, 0);
            }
            __syncthreads();
            cur ^= 1;
        }
        // ---- epilogue of this tile (the next tile's first chunk is already in LDS, its second in flight) ----------
        if (tid < BN) {
            ss[tid] = es;
            ss[BN + tid] = eh;
        }
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int m = m0 + (wm * TM + mi) * 32 + (lane & 31);
            if (m >= M) continue;
            size_t res_row;
            int b, oh, ow;
            out_rows(d, m, res_row, b, oh, ow);
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                f32x4 rv[4];  // the residual quads of this 32-channel group, all in flight before its first store
                if (has_res) {
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        const int n = n0 + (wn * TN + ni) * 32 + rg * 8 + (lane >> 5) * 4;
                        rv[rg] = *reinterpret_cast<const f32x4 *>(d.res + res_row * (size_t)d.Cout + min(n, d.Cout - 4));
                    }
                }
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int nl = (wn * TN + ni) * 32 + rg * 8 + (lane >> 5) * 4;
                    const int n = n0 + nl;
                    if (n >= d.Cout) continue;
                    f32x4 v = {acc[ni][mi][rg * 4 + 0], acc[ni][mi][rg * 4 + 1], acc[ni][mi][rg * 4 + 2], acc[ni][mi][rg * 4 + 3]};
                    v = a3d_epilogue_math(d, v, *reinterpret_cast<const f32x4 *>(ss + nl), *reinterpret_cast<const f32x4 *>(ss + BN + nl),
                                          has_res, rv[rg]);
                    store_out(d, v, m, n, b, oh, ow);
                }
            }
        }
    }
}

template <int TM, int TN, int BKT>
void launch_pw(const a3d_conv_desc *d, hipStream_t s, int per_cu) {
    constexpr int BM = 2 * TM * 32, BN = 2 * TN * 32;
    const int M = d->B * d->Ho * d->Wo;
    const int mtiles = (M + BM - 1) / BM, ntiles = (d->Cout + BN - 1) / BN;
    const int total = mtiles * ntiles;
    int grid = 256 * per_cu;  // MI355X: 256 CUs
    if (grid > total) grid = total;
    a3d_note_variant("conv_pw_kernel<%d,%d,%d> %dx%d persistent", TM, TN, BKT, BM, BN);
    hipLaunchKernelGGL((conv_pw_kernel<TM, TN, BKT>), dim3(grid), dim3(256), 0, s, *d, M, ntiles, total);
}
}  // namespace

static int a3d_conv_pw_eligible(const a3d_conv_desc *d) {
    if (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad != 0) return 0;
    if (d->stem || d->ups || d->phase || d->pixshuf || d->x2 || d->Cin2 || d->splitk != 1 || d->m_dev) return 0;
    if (d->Kpad != d->Cin || (d->Cin & 31) || d->Ho != d->H || d->Wo != d->W) return 0;
    if (d->Kpad > 2048) return 0;  // deep GEMMs (box-head fc1) are MFMA-bound: conv_gemm_v2's BK=32 form is the better fit
    const size_t M = (size_t)d->B * d->Ho * d->Wo;
    if (M * d->Cin * 4 >= ((size_t)1 << 31) || (size_t)d->Cout * d->Kpad * 4 >= ((size_t)1 << 31)) return 0;
    if (d->Cout < 64) return 0;
    return 1;
}

int a3d_conv_launch_pw(const a3d_conv_desc *d, hipStream_t s, int force) {
    if (!a3d_conv_pw_eligible(d)) return A3D_ERR_UNSUPPORTED;
    const int M = d->B * d->Ho * d->Wo;
    const long n128 = (long)((M + 127) / 128) * ((d->Cout + 127) / 128);
    // persistence only pays when a workgroup gets several tiles; small grids keep the one-tile kernel
    if (d->Cout <= 64 || n128 <= 1000) {
        const long n64 = (long)((M + 127) / 128) * ((d->Cout + 63) / 64);
        if (n64 < 2 * 256 * 4 && !force) return A3D_ERR_UNSUPPORTED;
        launch_pw<2, 1, 16>(d, s, 4);
    } else {
        if (n128 < 2 * 256 * 3 && !force) return A3D_ERR_UNSUPPORTED;
        launch_pw<2, 2, 16>(d, s, 3);
    }
    return a3d_check_launch();
}
