// d2t_corr_fwd_mfma.hip -- PointwiseCorrelation forward OUTSIDE the tuned envelope on the f32 matrix pipe: any d_max <= 8, any
// stride, any map (reference layout).  Bit-identical to the thread-per-cell anchor kernel (d2t_generic.hip) and to the reference:
// v_mfma_f32_16x16x4_f32 is, per output element, the ascending-k chain of fmaf the reference's channel loop is
// (pointwise_correlation_cuda.cu:105-107; d2t_corr_tuned.hip states and tests the same for d_max = 8), successive MFMAs continue that
// chain in ascending channel order, and the zero channels that pad C to a multiple of 4 add fma(0, 0, acc) = acc exactly.
//
// Workgroup = 4 x 8 pixels = two 4 x 4 pixel tiles side by side, one wave each (two from d_max 3, four from d_max 5 up: they split the
// N-tiles).  Per chunk of 8 channels the union of their windows
// ((4 + 2d) rows x (16 + 2d) columns of FM1, zero outside the map) and the workgroup's pixels of FM0 are staged in LDS (the next chunk is in
// flight into registers meanwhile).  A wave multiplies its tile's 16 pixels (M) with ALL the positions of its tile's window, 16 at a
// time (N-tiles: the (4 + 2d) x WCL window, row-major with rows padded to a multiple of 4), 4 channels per MFMA (K): one LDS read
// per operand and MFMA.  Of the 16 x 16 products of an MFMA those whose position lies inside the pixel's own (2d + 1)^2 window are
// cells; the others (60 % useful at d_max = 7) are discarded.  The epilogue scatters the accumulators into an LDS image
// [pixel][cell] -- structural zeros included: cells the reference's loops do not visit (:88-93) are written as 0 -- and the workgroup
// stores whole pixel rows.  Measured against k_corr_fwd_tiled (d2t_corr_blocked.hip, the same tiling on the vector ALU) at B = 8,
// C = 256, 38 x 63: d_max 7 90 against 124 us, d_max 4 43 against 65 us, d_max 2 36 against 45 us.  What it took (docs/lab_notebook.md R4-16):
// small workgroups and few registers per wave -- with 23 N-tiles in one wave the kernel ran one wave per SIMD and 320 workgroups in two
// rounds (145 us).
#include "d2t_kernels.hpp"

namespace d2t {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kKC = 8;                                               // channels per staged chunk (two MFMA k-steps; 16: 114 against 118 us at
                                                                     // d_max 7 for 60 more registers -- not taken; forcing the
                                                                     // d_max 7 kernel into 128 registers for four waves per SIMD spills: 185 us)

#ifndef D2T_FWD_MFMA_W4
#define D2T_FWD_MFMA_W4 12                                           // N-tiles from which FOUR waves share a p-tile (d_max 5 up)
#endif
template <int D, int TP>                                          // TP: 4 x 4 p-tiles side by side per workgroup (4 or 2)
struct FwdMfma {
    static constexpr int CW = 2 * D + 1, CELLS = CW * CW;
    static constexpr int WR = 4 + 2 * D;                            // window rows
    static constexpr int WCL = (4 + 2 * D + 3) & ~3;                // a tile's window columns, padded
    static constexpr int NT = (WR * WCL + 15) / 16;                 // N-tiles of a tile's window
    static constexpr int WPT = NT >= D2T_FWD_MFMA_W4 ? 4 : NT >= 8 ? 2 : 1;   // waves per tile: they split the N-tiles (accumulators + operand offsets
    static constexpr int NTW = (NT + WPT - 1) / WPT;                //  of 23 N-tiles in one wave: 172 + 88 registers = one wave per SIMD at d_max 7;
                                                                    //  two waves: 184, two per SIMD, 120 us; four: 98, four per SIMD, 90 us)
    static constexpr int PX = 16 * TP;                             // pixels per workgroup
    static constexpr int THREADS = 64 * TP * WPT;
    static constexpr int WCP = (4 * TP + 2 * D + 3) & ~3;           // the workgroup's window columns, padded
    static constexpr int WIMG = WR * WCP;
    static constexpr int CST = ((WIMG + PX + 15) & ~15) + 16;       // floats per staged channel (+16: the four channels of a k-step on different banks)
    static constexpr int NPOS = (WIMG + THREADS - 1) / THREADS;     // window elements per thread and channel
    static constexpr size_t LDS_LOOP = (size_t)2 * kKC * CST * sizeof(float);      // two chunk images
    static constexpr size_t LDS_EPI = (size_t)PX * CELLS * sizeof(float);
    static constexpr size_t LDS = LDS_LOOP > LDS_EPI ? LDS_LOOP : LDS_EPI;
};

template <int D, int TP>
__global__ void __launch_bounds__((FwdMfma<D, TP>::THREADS))
k_corr_fwd_mfma(const float* __restrict__ fm0, const float* __restrict__ fm1, float* __restrict__ out,
                int B, int C, int H, int W, int s)
{
    using P = FwdMfma<D, TP>;
    constexpr int kThreads = P::THREADS, TW = 4 * TP;                // TW: pixel columns of the workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* img = reinterpret_cast<float*>(lds_raw);
    const int plane = H * W;
    const int tiles_j = (W + TW - 1) / TW, tiles_i = (H + 3) / 4;
    int tb = blockIdx.x;
    const int tj = tb % tiles_j; tb /= tiles_j;
    const int ti = tb % tiles_i, b = tb / tiles_i;
    const int i0 = 4 * ti, j0 = TW * tj;
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = wv % TP, nt0 = (wv / TP) * P::NTW;                 // this wave's tile and its first N-tile
    const float* f0b = fm0 + (size_t)b * C * plane;
    const float* f1b = fm1 + (size_t)b * C * plane;

    // what this thread stages per channel: NPOS window positions, and (channel, tile pixel) pairs 2 * (tid >> 6) + {0, 1}
    int src_off[P::NPOS], dst_off[P::NPOS];
#pragma unroll
    for (int e = 0; e < P::NPOS; ++e) {
        const int pos = tid + e * kThreads, row = pos / P::WCP, col = pos - row * P::WCP;
        const int gi = i0 - D + row, gj = j0 - D + col;
        dst_off[e] = pos < P::WIMG ? pos : -1;
        src_off[e] = (pos < P::WIMG && gi >= 0 && gi < H && gj >= 0 && gj < W) ? gi * W + gj : -1;
    }
    const int px = tid % P::PX, pxi = px / TW, pxj = px - pxi * TW;    // pixel (pxi, pxj) of the 4 x TW block; stored [tile][4 * pi + pj]
    const int src0 = (i0 + pxi < H && j0 + pxj < W) ? (i0 + pxi) * W + j0 + pxj : -1;
    const int dst0 = P::WIMG + (pxj >> 2) * 16 + 4 * pxi + (pxj & 3);
    const int k0 = tid / P::PX;                                        // channels k0, k0 + KS, ... of the chunk
    constexpr int KS = kThreads / P::PX, NP0 = (kKC + KS - 1) / KS;    // (KS = 4 / 8 / 16 with one / two / four waves per tile)

    float pre[P::NPOS][kKC], pre0[NP0];
    auto fetch = [&](int c0) {
#pragma unroll
        for (int e = 0; e < P::NPOS; ++e)
#pragma unroll
            for (int k = 0; k < kKC; ++k)
                pre[e][k] = (src_off[e] >= 0 && c0 + k < C) ? f1b[(size_t)(c0 + k) * plane + src_off[e]] : 0.f;
#pragma unroll
        for (int h = 0; h < NP0; ++h)
            pre0[h] = (src0 >= 0 && k0 + KS * h < kKC && c0 + k0 + KS * h < C) ? f0b[(size_t)(c0 + k0 + KS * h) * plane + src0] : 0.f;
    };

    // B operand of N-tile nt: window position p = 16 nt + n of this wave's tile (row-major, WCL per row), inside the workgroup's image
    int boff[P::NTW];
#pragma unroll
    for (int nt = 0; nt < P::NTW; ++nt) {
        const int p = 16 * (nt0 + nt) + n, row = p / P::WCL, col = p - row * P::WCL;
        boff[nt] = (row < P::WR ? row : P::WR - 1) * P::WCP + 4 * t + col + q * P::CST;    // (+ q: the lane's channel of the k-step)
    }
    const int aoff = P::WIMG + 16 * t + n + q * P::CST;                 // A operand: pixel n of the tile
    f32x4 acc[P::NTW];
#pragma unroll
    for (int nt = 0; nt < P::NTW; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Two chunk images: chunk n + 1 is written (from the registers it was fetched into during chunk n - 1) while other waves still
    // multiply chunk n -- one barrier per chunk.
    auto stash = [&](float* buf) {
#pragma unroll
        for (int e = 0; e < P::NPOS; ++e)
            if (dst_off[e] >= 0) {
#pragma unroll
                for (int k = 0; k < kKC; ++k) buf[k * P::CST + dst_off[e]] = pre[e][k];
            }
#pragma unroll
        for (int h = 0; h < NP0; ++h)
            if (k0 + KS * h < kKC) buf[(k0 + KS * h) * P::CST + dst0] = pre0[h];
    };
    constexpr int IMG = kKC * P::CST;                                 // floats per chunk image
    fetch(0);
    stash(img);
    if (kKC < C) fetch(kKC);
    __syncthreads();
    int cur = 0;
    for (int c0 = 0; c0 < C; c0 += kKC) {
        const float* now = img + cur * IMG;
        if (c0 + kKC < C) {
            stash(img + (cur ^ 1) * IMG);                            // chunk c0 + KC: its image was last read before the previous barrier
            if (c0 + 2 * kKC < C) fetch(c0 + 2 * kKC);               // in flight while this chunk is consumed
        }
#pragma unroll
        for (int ks = 0; ks < kKC / 4; ++ks) {                       // ascending channels: 4 per MFMA
            const float* ch = now + 4 * ks * P::CST;
            const float a = ch[aoff];
#pragma unroll
            for (int nt = 0; nt < P::NTW; ++nt)                       // (N-tiles past the window's end: clamped operands, results dropped)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ch[boff[nt]], acc[nt], 0, 0, 0);
        }
        __syncthreads();                                             // chunk c0 consumed, chunk c0 + KC published
        cur ^= 1;
    }
    __syncthreads();                                                 // the chunk image becomes the output image [tile][pixel][cell]

    // ---- epilogue: lane (n, q) holds, per N-tile, position p = 16 nt + n for the pixels m = 4 q + r: pixel (row q, column r) of the tile
    float* stage = img + (size_t)t * 16 * P::CELLS;
#pragma unroll
    for (int nt = 0; nt < P::NTW; ++nt) {
        const int p = 16 * (nt0 + nt) + n, row = p / P::WCL, col = p - row * P::WCL;
        const int ci = row - q;                                      // cell row for pixel row q
        const int i = i0 + q;
        if (row < P::WR && ci >= 0 && ci < P::CW) {
            const bool row_hit = i < H && corr_axis_hit(i, i - D + ci, H, D, s);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cj = col - r, j = j0 + 4 * t + r;
                if (cj >= 0 && cj < P::CW)
                    stage[(4 * q + r) * P::CELLS + ci * P::CW + cj] = row_hit && j < W && corr_axis_hit(j, j - D + cj, W, D, s) ? acc[nt][r] : 0.f;
            }
        }
    }
    __syncthreads();
    const int ncols = W - j0 < TW ? W - j0 : TW;
    for (int pi = 0; pi < 4; ++pi) {
        if (i0 + pi >= H) break;
        float* dst = out + ((size_t)(b * plane + (i0 + pi) * W + j0)) * P::CELLS;      // ncols * CELLS contiguous floats
        for (int e = tid; e < ncols * P::CELLS; e += kThreads) {
            const int jj = e / P::CELLS, cell = e - jj * P::CELLS;
            dst[e] = img[(size_t)((jj >> 2) * 16 + 4 * pi + (jj & 3)) * P::CELLS + cell];
        }
    }
}

template <int D, int TP>
int launch_fwd_mfma_tp(const float* fm0, const float* fm1, float* out, int B, int C, int H, int W, int s, hipStream_t st)
{
    using P = FwdMfma<D, TP>;
    const int grid = B * ((H + 3) / 4) * ((W + 4 * TP - 1) / (4 * TP));
    if (P::LDS > 64 * 1024) D2T_ENSURE_DYNAMIC_LDS((k_corr_fwd_mfma<D, TP>), P::LDS);
    hipLaunchKernelGGL((k_corr_fwd_mfma<D, TP>), dim3(grid), dim3(P::THREADS), P::LDS, st, fm0, fm1, out, B, C, H, W, s);
    return launch_status();
}

#ifndef D2T_FWD_MFMA_TP
#define D2T_FWD_MFMA_TP 0                                            // lab: 2 / 4 force the workgroup width
#endif
template <int D>
int launch_fwd_mfma(const float* fm0, const float* fm1, float* out, int B, int C, int H, int W, int s, hipStream_t st)
{
    // two p-tiles per workgroup (4 x 8 pixels).  B = 8, C = 256, 38 x 63, us, (two / four p-tiles): d 4 45.6 / 51.3, d 6 63.0 / 67.0, and with
    // four waves per p-tile d 7 90.3 / 103.5; d 7 at B = 1 41.9 / 61.0; B = 2, C = 1024, d 6 86.5 / 146 (three p-tiles at d 7: no better)
    const int tp = D2T_FWD_MFMA_TP ? D2T_FWD_MFMA_TP : 2;
    if (tp == 2) return launch_fwd_mfma_tp<D, 2>(fm0, fm1, out, B, C, H, W, s, st);
    return launch_fwd_mfma_tp<D, 4>(fm0, fm1, out, B, C, H, W, s, st);
}

}  // namespace

bool corr_fwd_mfma_supported(int B, int C, int H, int W, int d, int s)
{
    // measured against k_corr_fwd_tiled at B = 8, C = 256, 38 x 63 (us): d 2: 36 / 45, 4: 43 / 65, 5: 53 / 79, 6: 54 / 81, 7: 90 / 124, 8 (stride 2): 95 / 120
#ifndef D2T_FWD_MFMA_MAXD
#define D2T_FWD_MFMA_MAXD 8
#endif
    return B >= 1 && C >= 1 && H >= 1 && W >= 1 && d >= 0 && d <= D2T_FWD_MFMA_MAXD && s >= 1 && 1LL * B * ((H + 3) / 4) * ((W + 15) / 16) < 0x7fffffffLL &&
           fits_i32(1LL * B * H * W * (2 * d + 1) * (2 * d + 1)) && fits_i32(1LL * C * H * W);
}

int corr_fwd_mfma_f32(const float* fm0, const float* fm1, float* out, int B, int C, int H, int W, int d, int s, hipStream_t st)
{
    switch (d) {
        case 0: return launch_fwd_mfma<0>(fm0, fm1, out, B, C, H, W, s, st);
        case 1: return launch_fwd_mfma<1>(fm0, fm1, out, B, C, H, W, s, st);
        case 2: return launch_fwd_mfma<2>(fm0, fm1, out, B, C, H, W, s, st);
        case 3: return launch_fwd_mfma<3>(fm0, fm1, out, B, C, H, W, s, st);
        case 4: return launch_fwd_mfma<4>(fm0, fm1, out, B, C, H, W, s, st);
        case 5: return launch_fwd_mfma<5>(fm0, fm1, out, B, C, H, W, s, st);
        case 6: return launch_fwd_mfma<6>(fm0, fm1, out, B, C, H, W, s, st);
        case 7: return launch_fwd_mfma<7>(fm0, fm1, out, B, C, H, W, s, st);
        case 8: return launch_fwd_mfma<8>(fm0, fm1, out, B, C, H, W, s, st);
        default: return D2T_EINVAL;
    }
}

}  // namespace d2t
