// d2t_corr_blocked.hip -- PointwiseCorrelation OUTSIDE the tuned envelope (any d_max, any stride, f32 and f64),
// reference layout: the same arithmetic and the same summation ORDER as the thread-per-element kernels of
// d2t_generic.hip (so the results are bit-identical to them, forward and both gradients), with the memory
// behaviour fixed.  D2T_IMPL_GENERIC keeps the thread-per-element kernels as the correctness anchor; D2T_IMPL_AUTO
// takes these when the tuned gfx950 kernels do not apply (round 4: the cost of leaving the envelope was 15x forward
// and 35-150x backward, include/d2t_ops.h; now 2.7x / 4x at d_max = 7).
//
// Reference semantics: pointwise_correlation_cuda.cu:84-107 (forward), :145-171 (backward).
//
// f32, d_max <= 14 -- the tiled kernels (forward, d_max <= 8: the same tiling on the f32 matrix pipe, d2t_corr_fwd_mfma.hip):
//   forward   k_corr_fwd_tiled: workgroup = 4 x 8 pixels, the FM1 window and the tile's FM0 pixels in LDS per chunk of 8 / 16
//             channels (next chunk in flight into registers); thread (pixel column, window row) owns a cell row of four pixels,
//             one aligned 16-byte LDS read per 16 fused multiply-adds.  Each cell is still ONE ascending-channel fma chain (:105-107).
//   backward  k_corr_bwd_tiled: thread = four adjacent pixels x four channels; per window row the 2d + 4 values of S its windows
//             span come as aligned 16-byte loads from zero-padded row copies of the maps (k_corr_bwd_prepass, workspace); the
//             (2d+1)^2 gradOut cells of the row's pixels are staged in LDS; columns that do not exist for a pixel are masked out of
//             the execution (stride 1: no masks at all, see S1).  gradFM1 reads gradOut indexed by the DISPLACED pixel (workspace) with FM0
//             and mirrored offsets, centres in ascending (i, j) order as in d2t_generic.hip.
// everything else (f64, d_max > 14) -- the blocked kernels:
//   forward   k_corr_fwd_blocked: a thread owns FOUR adjacent cells (ci, cj .. cj+3) of one pixel: per channel one load of
//             FM0[c][i][j] and one 16-byte (f64: 32-byte) load of FM1[c][di][dj .. dj+3] feed four fused multiply-adds.  Groups
//             that touch the map's left / right edge take the per-cell path.  (f64 forward: the thread-per-cell kernel is faster.)
//   backward  k_corr_bwd_blocked: the thread-per-element kernel reads gradOut[b][pixel][cell] with the LANES along x, i.e.
//             4 (2d+1)^2 bytes apart: one cache line per lane and term.  Here a workgroup stages the cells of (b, row y, XT pixels)
//             in LDS once and every thread (channel, pixel) walks its window from there; FM is read along x (coalesced).
#include "d2t_kernels.hpp"

namespace d2t {

namespace {

constexpr int kBlk = 256;

template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef float type __attribute__((ext_vector_type(4), aligned(4))); };
template <> struct Vec4<double> { typedef double type __attribute__((ext_vector_type(4), aligned(8))); };

template <typename T>
__global__ void __launch_bounds__(kBlk)
k_corr_fwd_blocked(const T* __restrict__ fm0, const T* __restrict__ fm1, T* __restrict__ out,
                   int B, int C, int H, int W, int d, int s)
{
    typedef typename Vec4<T>::type V4;
    const int cw = 2 * d + 1, ng = (cw + 3) >> 2, plane = H * W;
    const long long total = 1LL * B * plane * cw * ng;
    for (long long i64 = (long long)blockIdx.x * kBlk + threadIdx.x; i64 < total; i64 += (long long)gridDim.x * kBlk) {
        const int gq = (int)(i64 % ng);
        const long long r1 = i64 / ng;
        const int ci = (int)(r1 % cw);
        const int pix = (int)(r1 / cw);
        const int j = pix % W, i = (pix / W) % H, b = pix / plane;
        const int di = i - d + ci, cj0 = 4 * gq, dj0 = j - d + cj0;
        const bool row_hit = corr_axis_hit(i, di, H, d, s);
        bool hit[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) hit[k] = row_hit && cj0 + k < cw && corr_axis_hit(j, dj0 + k, W, d, s);
        T acc[4] = {T(0), T(0), T(0), T(0)};
        if (hit[0] || hit[1] || hit[2] || hit[3]) {
            const T* a = fm0 + (size_t)b * C * plane + i * W + j;
            const T* q = fm1 + (size_t)b * C * plane + di * W;
            if (dj0 >= 0 && dj0 + 3 < W) {                            // the four columns lie inside the row: one vector load per channel
                q += dj0;
                for (int c = 0; c < C; ++c) {                         // ascending c, fused (:105-107)
                    const T av = a[(size_t)c * plane];
                    const V4 qv = *reinterpret_cast<const V4*>(q + (size_t)c * plane);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = fma_t(av, qv[k], acc[k]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = hit[k] ? acc[k] : T(0);      // (a cell the loops do not visit: its chain is discarded)
            } else {                                                  // at the map's edge: per cell, only the cells that exist
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!hit[k]) continue;
                    T v = T(0);
                    for (int c = 0; c < C; ++c) v = fma_t(a[(size_t)c * plane], q[(size_t)c * plane + dj0 + k], v);
                    acc[k] = v;
                }
            }
        }
        T* o = out + ((size_t)pix * cw + ci) * cw + cj0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (cj0 + k < cw) o[k] = acc[k];                          // structural zeros included
    }
}


// Forward, f32, LDS-tiled: workgroup = (b, tile of 4 x 8 pixels).  Per chunk of KC channels the tile's FM0 pixels and the FM1
// window (4 + 2d rows x 8 + 2d columns, zero outside the map) are staged in LDS -- the NEXT chunk is already on its way into
// registers while the current one is consumed.  Thread (pixel column pj, WINDOW row wrow) owns, for the four pixels (pi, pj) of
// its column, the cell row ci = wrow - pi -- the four pixels read the SAME window row -- and walks the channels in ascending
// order: one 16-byte LDS read of FM0[c][0..3][pj] and, per window column quad (16-byte aligned in the image whatever the pixel's
// column: the left half of the tile reads quads 0 .. NG-1, the right half 1 .. NG, and quad g holds the cells cj = 4g + x - (pj & 3)),
// one 16-byte LDS read for SIXTEEN fused multiply-adds.  Every cell is still one ascending-channel fma chain
// (pointwise_correlation_cuda.cu:105-107): bit-identical to the thread-per-cell kernel.  Chains of cells that do not exist (ci or
// cj outside the window, cells the reference's loops do not visit) are computed and discarded.
// NG = ceil((4 + 2d) / 4): column quads per thread, the compile-time bound of the accumulator array; KC channels per chunk.
constexpr int kTileH = 4, kTileW = 8;
constexpr int kTiledMaxD = (kBlk / kTileW - kTileH) / 2;              // window rows <= threads / 8: d <= 14
#ifndef TILED_ABL
#define TILED_ABL 0                                                  // lab: 1 no multiply-adds, 2 no staging loads
#endif

template <int NG, int KC>
__global__ void __launch_bounds__(kBlk)
k_corr_fwd_tiled(const float* __restrict__ fm0, const float* __restrict__ fm1, float* __restrict__ out,
                 int B, int C, int H, int W, int d, int s)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int wcp = 4 * (NG + 1);                                 // window columns 8 + 2d, padded to whole quads (zeros)
    constexpr int NPOS = (4 * NG * wcp + kBlk - 1) / kBlk;            // window elements per thread and channel (window rows <= 4 NG)
    constexpr int NP0 = KC * kTileH * kTileW / kBlk;                  // tile pixels per thread and chunk (1 or 2)
    const int cw = 2 * d + 1, plane = H * W;
    const int wr = kTileH + 2 * d;                                    // window rows (<= 4 NG <= 32)
    const int wimg = wr * wcp, cstride = wimg + kTileH * kTileW;      // floats per staged channel: window + the tile's pixels [pj][pi]
    float* img = reinterpret_cast<float*>(lds_raw);
    const int tiles_j = (W + kTileW - 1) / kTileW, tiles_i = (H + kTileH - 1) / kTileH;
    int t = blockIdx.x;
    const int tj = t % tiles_j; t /= tiles_j;
    const int ti = t % tiles_i, b = t / tiles_i;
    const int i0 = kTileH * ti, j0 = kTileW * tj;
    const int tid = threadIdx.x, pj = tid & 7, wrow = tid >> 3;
    const bool active = wrow < wr;
    const float* f0b = fm0 + (size_t)b * C * plane;
    const float* f1b = fm1 + (size_t)b * C * plane;

    // what this thread stages: NPOS window positions (every channel of the chunk) and NP0 (channel, tile pixel) pairs
    int src_off[NPOS], dst_off[NPOS];                                 // src < 0: outside the map (zero) or no such position
#pragma unroll
    for (int n = 0; n < NPOS; ++n) {
        const int pos = tid + n * kBlk, row = pos / wcp, col = pos - row * wcp;
        const int gi = i0 - d + row, gj = j0 - d + col;
        dst_off[n] = pos < wimg ? pos : -1;
        src_off[n] = (pos < wimg && gi >= 0 && gi < H && gj >= 0 && gj < W) ? gi * W + gj : -1;
    }
    const int q0 = tid & 31, q0i = q0 >> 3, q0j = q0 & 7;
    const int src0 = (i0 + q0i < H && j0 + q0j < W) ? (i0 + q0i) * W + j0 + q0j : -1;
    const int dst0 = wimg + q0j * kTileH + q0i, k0 = tid >> 5;        // channels k0, k0 + 8 of the chunk

    float pre[NPOS][KC], pre0[NP0];
    auto fetch = [&](int c0) {                                        // chunk c0 .. c0 + KC - 1 into registers (zeros past C)
#pragma unroll
        for (int n = 0; n < NPOS; ++n)
#pragma unroll
            for (int k = 0; k < KC; ++k)
                pre[n][k] = (src_off[n] >= 0 && c0 + k < C && !(TILED_ABL & 2)) ? f1b[(size_t)(c0 + k) * plane + src_off[n]] : 0.f;
#pragma unroll
        for (int n = 0; n < NP0; ++n)
            pre0[n] = (src0 >= 0 && c0 + k0 + 8 * n < C) ? f0b[(size_t)(c0 + k0 + 8 * n) * plane + src0] : 0.f;
    };
    f4 acc[kTileH][NG];
#pragma unroll
    for (int r = 0; r < kTileH; ++r)
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[r][g] = f4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    for (int c0 = 0; c0 < C; c0 += KC) {
        const int kc = C - c0 < KC ? C - c0 : KC;
        __syncthreads();                                              // previous chunk consumed
#pragma unroll
        for (int n = 0; n < NPOS; ++n)
            if (dst_off[n] >= 0) {
#pragma unroll
                for (int k = 0; k < KC; ++k) img[k * cstride + dst_off[n]] = pre[n][k];
            }
#pragma unroll
        for (int n = 0; n < NP0; ++n) img[(k0 + 8 * n) * cstride + dst0] = pre0[n];
        __syncthreads();
        if (c0 + KC < C) fetch(c0 + KC);                              // in flight while this chunk is consumed
        if (active && !(TILED_ABL & 1)) {
            const float* ch = img;
#pragma unroll 2
            for (int k = 0; k < kc; ++k, ch += cstride) {             // ascending c
                const f4 a = *reinterpret_cast<const f4*>(ch + wimg + pj * kTileH);
                const f4* rowq = reinterpret_cast<const f4*>(ch + wrow * wcp) + (pj >> 2);
                f4 q[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) q[g] = rowq[g];
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int r = 0; r < kTileH; ++r)
#pragma unroll
                        for (int x = 0; x < 4; ++x) acc[r][g][x] = fma_t(a[r], q[g][x], acc[r][g][x]);
            }
        }
    }
    const int j = j0 + pj;
    if (!active || j >= W) return;
#pragma unroll
    for (int r = 0; r < kTileH; ++r) {
        const int i = i0 + r, ci = wrow - r;                          // window row wrow is cell row wrow - r of pixel row r
        if (i < H && ci >= 0 && ci < cw) {
            float* o = out + ((size_t)(b * plane + i * W + j)) * cw * cw + ci * cw;
            const bool row_hit = corr_axis_hit(i, i - d + ci, H, d, s);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int cj = 4 * g + x - (pj & 3);              // the cell this window column is for this pixel column
                    if (cj >= 0 && cj < cw)
                        o[cj] = row_hit && corr_axis_hit(j, j - d + cj, W, d, s) ? acc[r][g][x] : 0.f;   // structural zeros included
                }
            }
        }
    }
}

// gradOut re-indexed by the displaced pixel: goutT[b][y][x][ci][cj] = gradOut[b][y - ci + d][x - cj + d][ci][cj] where that centre
// exists (elsewhere the value is never read: the consumer applies the same test).
template <typename T>
__global__ void __launch_bounds__(kBlk)
k_corr_gout_by_displaced(const T* __restrict__ gout, T* __restrict__ goutT, int B, int H, int W, int d)
{
    const int cw = 2 * d + 1, cells = cw * cw, plane = H * W;
    const long long total = 1LL * B * plane * cells;
    for (long long i64 = (long long)blockIdx.x * kBlk + threadIdx.x; i64 < total; i64 += (long long)gridDim.x * kBlk) {
        const int cell = (int)(i64 % cells);
        const int pix = (int)(i64 / cells);
        const int cj = cell % cw, ci = cell / cw;
        const int x = pix % W, y = (pix / W) % H, b = pix / plane;
        const int i = y - ci + d, j = x - cj + d;
        goutT[i64] = (i >= 0 && i < H && j >= 0 && j < W) ? gout[((size_t)(b * plane + i * W + j)) * cells + cell] : T(0);
    }
}

// One gradient.  MIRROR = false: gX[b][c][y][x] = sum over the window (di, dj) of (y, x), ascending, of G[(y,x)][cell] * S[c][di][dj]
// (gradFM0: G = gradOut, S = FM1; the reference's thread-owned order, :154-168).  MIRROR = true: gX[b][c][y][x] = sum over the
// centres (i, j) that reach (y, x), ascending, of G[(y,x)][cell(y-i+d, x-j+d)] * S[c][i][j] (gradFM1: G = gradOut by displaced pixel,
// S = FM0).  Workgroup = (b, y, XT pixels of the row, CB channels): the XT x (2d+1)^2 cells go to LDS once, then thread (channel
// slot, pixel) walks NP channel passes.
template <typename T, bool MIRROR>
__global__ void __launch_bounds__(kBlk)
k_corr_bwd_blocked(const T* __restrict__ G, const T* __restrict__ S, T* __restrict__ gx,
                   int B, int C, int H, int W, int d, int s, int XT, int CB)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* gl = reinterpret_cast<T*>(lds_raw);                            // [XT][cells]
    const int cw = 2 * d + 1, cells = cw * cw, plane = H * W;
    const int xtiles = (W + XT - 1) / XT, cblocks = (C + CB - 1) / CB;
    int t = blockIdx.x;
    const int cb = t % cblocks; t /= cblocks;
    const int xt = t % xtiles; t /= xtiles;
    const int y = t % H, b = t / H;
    const int x0 = xt * XT, nx = W - x0 < XT ? W - x0 : XT;
    const T* gsrc = G + ((size_t)(b * plane + y * W + x0)) * cells;   // nx * cells contiguous elements
    for (int e = threadIdx.x; e < nx * cells; e += kBlk) gl[e] = gsrc[e];
    __syncthreads();
    const int xl = threadIdx.x % XT, cl = threadIdx.x / XT, nch = kBlk / XT;
    const int x = x0 + xl;
    if (xl >= nx) return;
    const T* gc = gl + (size_t)xl * cells;
    const int c_end = (cb + 1) * CB < C ? (cb + 1) * CB : C;
    for (int c = cb * CB + cl; c < c_end; c += nch) {
        const T* sp = S + ((size_t)b * C + c) * plane;
        T a = T(0);
        if (!MIRROR) {                                                // centre (y,x): walk its window (ascending di, dj)
            const int lo_i = y - d > 0 ? y - d : 0, hi_i = y + d < H ? y + d : H;
            const int lo_j = x - d > 0 ? x - d : 0, hi_j = x + d < W ? x + d : W;
            for (int di = lo_i; di < hi_i; di += s)
                for (int dj = lo_j; dj < hi_j; dj += s)
                    a = fma_t(gc[(di - y + d) * cw + (dj - x + d)], sp[di * W + dj], a);
        } else {                                                      // displaced (y,x): the centres that reach it (ascending i, j)
            const int i_lo = y - d > 0 ? y - d : 0, i_hi = y + d < H - 1 ? y + d : H - 1;
            const int j_lo = x - d > 0 ? x - d : 0, j_hi = x + d < W - 1 ? x + d : W - 1;
            for (int i = i_lo; i <= i_hi; ++i) {
                if (!corr_axis_hit(i, y, H, d, s)) continue;
                for (int j = j_lo; j <= j_hi; ++j) {
                    if (!corr_axis_hit(j, x, W, d, s)) continue;
                    a = fma_t(gc[(y - i + d) * cw + (x - j + d)], sp[i * W + j], a);
                }
            }
        }
        gx[((size_t)b * C + c) * plane + y * W + x] = a;
    }
}


// Pre-passes of the tiled backward, ONE launch (blockIdx.y: 0 / 1 = FM0 / FM1 copied with d zero columns before and >= d + 6 after every row,
// row pitch Wp a multiple of 4 -- every 16-byte piece the kernel reads is aligned and inside a row, no edge cases; 2 = gradOut re-indexed by
// displaced pixel, see k_corr_gout_by_displaced).
__global__ void __launch_bounds__(kBlk)
k_corr_bwd_prepass(const float* __restrict__ fm0, const float* __restrict__ fm1, const float* __restrict__ gout,
                   float* __restrict__ pad0, float* __restrict__ pad1, float* __restrict__ goutT,
                   int rows, int B, int H, int W, int Wp, int d)
{
    // (32-bit index arithmetic: the launcher checks that both element counts fit.  The pass is 44 us at B = 8, C = 256, 38 x 63, d_max 7
    //  either way: the re-index reads one cache line per element)
    if (blockIdx.y < 2) {
        const float* S = blockIdx.y ? fm1 : fm0;
        float* P = blockIdx.y ? pad1 : pad0;
        const unsigned total = (unsigned)rows * (unsigned)Wp;
        for (unsigned i = blockIdx.x * kBlk + threadIdx.x; i < total; i += gridDim.x * kBlk) {
            const unsigned row = i / (unsigned)Wp;
            const int col = (int)(i - row * (unsigned)Wp) - d;
            P[i] = (col >= 0 && col < W) ? S[row * (unsigned)W + (unsigned)col] : 0.f;
        }
        return;
    }
    // gradOut by displaced pixel: goutT[b][y][x][ci][cj] = gradOut[b][y - ci + d][x - cj + d][ci][cj].  Work item = (b, y, ci, 64 columns): the
    // cell rows ci of the 64 + 2d source pixels (60-byte runs) go through LDS and leave as the cell rows ci of the 64 displaced pixels --
    // one cache line per source pixel instead of one per element (the element-wise form, k_corr_gout_by_displaced: 4.3 M requests).
    __shared__ float stage[(64 + 2 * kTiledMaxD) * (2 * kTiledMaxD + 1)];
    const int cw = 2 * d + 1, cells = cw * cw, plane = H * W, xt = (W + 63) / 64, nsrc = 64 + 2 * d;
    const int nitems = B * H * cw * xt;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {   // (uniform: the barriers are reached by every thread)
        int t = item;
        const int x0 = (t % xt) * 64; t /= xt;
        const int ci = t % cw; t /= cw;
        const int y = t % H, b = t / H;
        const int sy = y - ci + d;
        const bool row_ok = sy >= 0 && sy < H;
        for (int e = threadIdx.x; e < nsrc * cw; e += kBlk) {
            const int sp = e / cw, cj = e - sp * cw, sx = x0 - d + sp;
            stage[e] = (row_ok && sx >= 0 && sx < W) ? gout[((size_t)(b * plane + sy * W + sx)) * cells + ci * cw + cj] : 0.f;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * cw; e += kBlk) {
            const int xl = e / cw, cj = e - xl * cw;
            if (x0 + xl < W) goutT[((size_t)(b * plane + y * W + x0 + xl)) * cells + ci * cw + cj] = stage[(xl - cj + 2 * d) * cw + cj];
        }
        __syncthreads();
    }
}

// One gradient, f32, register-tiled (d <= 14).  Same sums in the same order as k_corr_bwd_blocked -- thread-owned, rows then columns
// ascending -- but a thread owns FOUR adjacent pixels of the row and CT channels: per window row it loads, per channel, the
// 4 NQ >= 2d + 4 values of S its four windows span with aligned 16-byte loads (S comes with zero-padded rows, k_corr_bwd_prepass), and every gradOut cell it reads from LDS feeds CT fused
// multiply-adds.  (k_corr_bwd_blocked issues one LDS read and one 4-byte global load per multiply-add.)  Which columns k of the
// window exist for pixel px (map edge, stride, the never-visited last column) does not depend on the row or the channel: one bit
// mask per pixel, and a term whose bit is clear is not executed at all -- non-finite values outside the window cannot leak in.
// S1 (stride 1): the only column inside the map that does not belong to a pixel's sum is the same for every pixel (the last one, 2d --
// MIRROR: the first), so it is skipped by a wave-uniform test; columns outside the map meet S = 0 from the padded copy and a gradOut
// cell that is zeroed while it is staged (MIRROR: by k_corr_gout_by_displaced), and fma(0, 0, acc) = acc exactly (acc, started at +0,
// is never -0).  No masks, no selects: the multiply-adds are unconditional.
template <bool MIRROR, int NQ, int CT, bool S1>
__global__ void __launch_bounds__(kBlk)
k_corr_bwd_tiled(const float* __restrict__ G, const float* __restrict__ S, float* __restrict__ gx,
                 int B, int C, int H, int W, int Wp, int d, int s, int XT, int CB)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    typedef float f4v __attribute__((ext_vector_type(4)));
    float* gl = reinterpret_cast<float*>(lds_raw) + 4;                // [XT][cells], four floats of padding either side
    const int cw = 2 * d + 1, cells = cw * cw, plane = H * W;
    const int xtiles = (W + XT - 1) / XT, cblocks = (C + CB - 1) / CB;
    int t = blockIdx.x;
    const int cb = t % cblocks; t /= cblocks;
    const int xt = t % xtiles; t /= xtiles;
    const int y = t % H, b = t / H;
    const int xb = xt * XT, nx = W - xb < XT ? W - xb : XT;
    const float* gsrc = G + ((size_t)(b * plane + y * W + xb)) * cells;   // nx * cells contiguous elements
    for (int e = threadIdx.x; e < nx * cells; e += kBlk) gl[e] = gsrc[e];
    if (S1 && !MIRROR && (xb < d || xb + nx + d > W)) {               // cells whose displaced column lies outside the map: never read by
        __syncthreads();                                              //  the reference, zeroed here (see S1 above); edge workgroups only
        for (int e = threadIdx.x; e < nx * cw; e += kBlk) {           // (pixel, cell row): columns [0, klo) and [khi, cw)
            const int px = e / cw, x = xb + px;
            const int klo = d - x > 0 ? d - x : 0, khi = W + d - x < cw ? W + d - x : cw;
            float* rowp = gl + (size_t)px * cells + (e - px * cw) * cw;
            for (int k = 0; k < klo; ++k) rowp[k] = 0.f;
            for (int k = khi; k < cw; ++k) rowp[k] = 0.f;
        }
    }
    __syncthreads();
    const int nxg = XT >> 2, xg = threadIdx.x % nxg, cl = threadIdx.x / nxg, nch = kBlk / nxg;
    const int x0 = xb + 4 * xg;
    if (4 * xg >= nx) return;
    unsigned mask[4];                                                 // bit k: column x - d + k belongs to the sum of pixel x = x0 + px
#pragma unroll
    for (int px = 0; px < 4; ++px) {
        const int x = x0 + px;
        unsigned m = 0;
        if (x < W) {
            const int lo_j = x - d > 0 ? x - d : 0, hi_j = x + d < W ? x + d : W;
            for (int k = 0; k < cw; ++k) {
                const int col = x - d + k;
                const bool ok = MIRROR ? (col >= 0 && col < W && corr_axis_hit(col, x, W, d, s))
                                       : (col >= lo_j && col < hi_j && (col - lo_j) % s == 0);
                m |= ok ? 1u << k : 0u;
            }
        }
        mask[px] = m;
    }
    const float* glx = gl + (size_t)(4 * xg) * cells;
    const int c_end = (cb + 1) * CB < C ? (cb + 1) * CB : C;
    // rows of the sum (the same for every pixel of the row y): first, last, step; MIRROR tests each row
    const int r_lo = y - d > 0 ? y - d : 0;
    const int r_hi = MIRROR ? (y + d < H - 1 ? y + d : H - 1) : (y + d < H ? y + d : H) - 1;
    const int r_step = MIRROR ? 1 : s;
    for (int c0 = cb * CB + cl * CT; c0 < c_end; c0 += nch * CT) {
        float acc[CT][4];
#pragma unroll
        for (int ch = 0; ch < CT; ++ch)
#pragma unroll
            for (int px = 0; px < 4; ++px) acc[ch][px] = 0.f;
        for (int row = r_lo; row <= r_hi; row += r_step) {
            if (MIRROR && !corr_axis_hit(row, y, H, d, s)) continue;
            const int rc = MIRROR ? y - row + d : row - y + d;        // cell row
            float seg[CT][4 * NQ];
#pragma unroll
            for (int ch = 0; ch < CT; ++ch) {
                const int cc = c0 + ch < C ? c0 + ch : C - 1;
                const f4v* sp = reinterpret_cast<const f4v*>(S + (((size_t)b * C + cc) * H + row) * Wp + x0);   // padded column x0 = column x0 - d
#pragma unroll
                for (int m = 0; m < NQ; ++m) {
                    const f4v v = sp[m];
#pragma unroll
                    for (int e = 0; e < 4; ++e) seg[ch][4 * m + e] = v[e];
                }
            }
            const float* gr = glx + rc * cw + (MIRROR ? 2 * d : 0);   // cell (rc, k) -- MIRROR: (rc, 2d - k)
            // The cells of column k + 1 are read (unconditionally: the image is padded) while the masked multiply-adds of column k
            // issue; sched_barrier keeps hipcc from hoisting ALL the reads to the top of the row (237 VGPRs, two waves per SIMD) or
            // sinking each into its masked block (an exposed LDS round trip per term).  With the reads out of the way hipcc turns the
            // masked blocks into unconditional multiply-adds + selects on row-invariant lane masks.  (A wave-uniform fast path for
            // columns every lane has -- a branch per term -- measured 404 against 268 us: the branches end the scheduling regions.)
            float gv[2][4];
#pragma unroll
            for (int px = 0; px < 4; ++px) gv[0][px] = gr[px * cells];
#pragma unroll
            for (int k = 0; k < 4 * NQ - 3; ++k) {                    // columns ascending
                if (k + 1 < 4 * NQ - 3) {
#pragma unroll
                    for (int px = 0; px < 4; ++px) gv[(k + 1) & 1][px] = gr[px * cells + (MIRROR ? -(k + 1) : k + 1)];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (S1) {
                    if (k < cw && k != (MIRROR ? 0 : 2 * d)) {        // wave-uniform
#pragma unroll
                        for (int px = 0; px < 4; ++px)
#pragma unroll
                            for (int ch = 0; ch < CT; ++ch) acc[ch][px] = fma_t(gv[k & 1][px], seg[ch][k + px], acc[ch][px]);
                    }
                } else {
#pragma unroll
                    for (int px = 0; px < 4; ++px)
                        if ((mask[px] >> k) & 1u) {
#pragma unroll
                            for (int ch = 0; ch < CT; ++ch) acc[ch][px] = fma_t(gv[k & 1][px], seg[ch][k + px], acc[ch][px]);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int ch = 0; ch < CT; ++ch)
            if (c0 + ch < c_end) {
                float* o = gx + ((size_t)b * C + c0 + ch) * plane + y * W + x0;
#pragma unroll
                for (int px = 0; px < 4; ++px)
                    if (x0 + px < W) o[px] = acc[ch][px];
            }
    }
}

// pixels of a row per workgroup: a power of two <= 64 whose cells fit 48 KB of LDS (0: the window is too large for this form)
template <typename T>
int blocked_xt(int W, int d)
{
    const long long cells = (2LL * d + 1) * (2LL * d + 1);
    int xt = 64;
    while (xt > 1 && (xt * cells * (long long)sizeof(T) > 48 * 1024 || xt / 2 >= W)) xt >>= 1;
    return xt * cells * (long long)sizeof(T) <= 48 * 1024 ? xt : 0;
}

}  // namespace

template <typename T>
bool corr_blocked_supported(int B, int C, int H, int W, int d, int s)
{
    if (B < 1 || C < 1 || H < 1 || W < 4 || d < 0 || s < 1) return false;
    const long long cw = 2LL * d + 1;
    if (blocked_xt<T>(W, d) < 4) return false;
    return fits_i32(1LL * B * H * W * cw * cw) && fits_i32(1LL * B * C * H * W) && 1LL * B * H * 64 * ((C + 63) / 64) < 0x7fffffffLL;
}

static int tiled_pitch(int W, int d) { return (W + 2 * d + 6 + 3) & ~3; }   // row pitch of the zero-padded maps

template <typename T>
size_t corr_bwd_blocked_ws_bytes(int B, int C, int H, int W, int d, int s)
{
    if (!corr_blocked_supported<T>(B, C, H, W, d, s)) return 0;
    const size_t cw = 2 * (size_t)d + 1;
    size_t bytes = ((size_t)B * H * W * cw * cw * sizeof(T) + 255) / 256 * 256;   // gradOut by displaced pixel
    if (sizeof(T) == 4 && d <= kTiledMaxD) bytes += 2 * (size_t)B * C * H * tiled_pitch(W, d) * sizeof(float);   // + both maps with padded rows
    return bytes;
}


static int corr_fwd_tiled_f32(const float* fm0, const float* fm1, float* out, int B, int C, int H, int W, int d, int s, hipStream_t st)
{
    const int ng = (kTileH + 2 * d + 3) / 4 < 2 ? 2 : (kTileH + 2 * d + 3) / 4;        // column quads per thread: 2 .. 8
    const int wr = kTileH + 2 * d, wcp = 4 * (ng + 1), cstride = wr * wcp + kTileH * kTileW;
    const int grid = B * ((H + kTileH - 1) / kTileH) * ((W + kTileW - 1) / kTileW);
    switch (ng) {
#define D2T_TILED(N, K) case N: hipLaunchKernelGGL((k_corr_fwd_tiled<N, K>), dim3(grid), dim3(kBlk), (size_t)K * cstride * sizeof(float), st, \
                                                   fm0, fm1, out, B, C, H, W, d, s); break;
        D2T_TILED(2, 16) D2T_TILED(3, 16) D2T_TILED(4, 16) D2T_TILED(5, 16) D2T_TILED(6, 8) D2T_TILED(7, 8) D2T_TILED(8, 8)
#undef D2T_TILED
        default: return D2T_EINVAL;
    }
    return launch_status();
}

template <typename T>
int corr_fwd_blocked(const T* fm0, const T* fm1, T* out, int B, int C, int H, int W, int d, int s, hipStream_t st)
{
    const long long cw = 2LL * d + 1, total = 1LL * B * H * W * cw * ((cw + 3) / 4);
    if (total == 0) return D2T_OK;
    if (sizeof(T) == 4 && corr_fwd_mfma_supported(B, C, H, W, d, s))                                             // d_max <= 8: the same tiling on the matrix pipe
        return corr_fwd_mfma_f32(reinterpret_cast<const float*>(fm0), reinterpret_cast<const float*>(fm1), reinterpret_cast<float*>(out), B, C, H, W, d, s, st);
    if (sizeof(T) == 4 && d <= kTiledMaxD && 1LL * B * ((H + 3) / 4) * ((W + 7) / 8) < 0x7fffffffLL)       // LDS-tiled form (f32)
        return corr_fwd_tiled_f32(reinterpret_cast<const float*>(fm0), reinterpret_cast<const float*>(fm1), reinterpret_cast<float*>(out), B, C, H, W, d, s, st);
    hipLaunchKernelGGL(k_corr_fwd_blocked<T>, dim3(grid_for(total, kBlk, 256 * 32)), dim3(kBlk), 0, st, fm0, fm1, out, B, C, H, W, d, s);
    return launch_status();
}

static int corr_bwd_tiled_f32(const float* gout, const float* fm0, const float* fm1, float* g0, float* g1, int B, int C, int H, int W,
                              int d, int s, int XT, float* goutT, hipStream_t st)
{
    const long long cw = 2LL * d + 1, cells = cw * cw;
    const int Wp = tiled_pitch(W, d);
    const long long rows = 1LL * B * C * H;
    float* pad0 = goutT + ((size_t)B * H * W * cells + 63) / 64 * 64;     // workspace: gradOut by displaced pixel | FM0 padded | FM1 padded
    float* pad1 = pad0 + (size_t)rows * Wp;
    {
        const long long most = rows * Wp > 1LL * B * H * W * cells ? rows * Wp : 1LL * B * H * W * cells;
        hipLaunchKernelGGL(k_corr_bwd_prepass, dim3(grid_for(most, kBlk, 256 * 16), 3), dim3(kBlk), 0, st, fm0, fm1, gout, pad0, pad1, goutT,
                           (int)rows, B, H, W, Wp, d);
    }
    const int nq = (int)(cw + 3 + 3) / 4;                             // 16-byte pieces spanning the four windows of a thread: 1 .. 8
    // channels per thread: four; two from d_max 7 up at stride 1 (20+ row values per channel in registers: 138 -> ~90 VGPRs, five waves per
    // SIMD).  B = 8, C = 256, 38 x 63, us (two / four): d 7 178 / 197, d 12 399 / 437, but d 6 163 / 142, d 5 126 / 120, d 8 stride 2 224 / 201
    const int CT = (nq >= 5 && s == 1) ? 2 : 4;
    const int CB = (kBlk / (XT / 4)) * CT < 64 ? 64 : (kBlk / (XT / 4)) * CT;   // one channel pass per workgroup where C allows
    const size_t lds = ((size_t)XT * cells + 8) * sizeof(float);
    const int grid = B * H * ((W + XT - 1) / XT) * ((C + CB - 1) / CB);
    int rc = D2T_OK;
    for (int pass = 0; pass < 2 && rc == D2T_OK; ++pass) {
        const float* G = pass ? goutT : gout;
        const float* S = pass ? pad0 : pad1;
        float* gx = pass ? g1 : g0;
        switch (nq) {
#define D2T_TILED_S(N, M, U, CTV) hipLaunchKernelGGL((k_corr_bwd_tiled<M, N, CTV, U>), dim3(grid), dim3(kBlk), lds, st, G, S, gx, B, C, H, W, Wp, d, s, XT, CB)
#define D2T_TILED(N) case N: if (pass) { if (s == 1) D2T_TILED_S(N, true, true, 4); else D2T_TILED_S(N, true, false, 4); } \
                             else { if (s == 1) D2T_TILED_S(N, false, true, 4); else D2T_TILED_S(N, false, false, 4); } break;
#define D2T_TILED2(N) case N: if (s == 1) { if (pass) D2T_TILED_S(N, true, true, 2); else D2T_TILED_S(N, false, true, 2); } \
                              else { if (pass) D2T_TILED_S(N, true, false, 4); else D2T_TILED_S(N, false, false, 4); } break;
            D2T_TILED(1) D2T_TILED(2) D2T_TILED(3) D2T_TILED(4) D2T_TILED2(5) D2T_TILED2(6) D2T_TILED2(7) D2T_TILED2(8)
#undef D2T_TILED2
#undef D2T_TILED
#undef D2T_TILED_S
            default: return D2T_EINVAL;
        }
        rc = launch_status();
    }
    return rc;
}

template <typename T>
int corr_bwd_blocked(const T* gout, const T* fm0, const T* fm1, T* g0, T* g1, int B, int C, int H, int W, int d, int s,
                     void* ws, hipStream_t st)
{
    const long long cw = 2LL * d + 1, cells = cw * cw;
    if (1LL * B * C * H * W == 0) return D2T_OK;
    const int XT = blocked_xt<T>(W, d), CB = 64;
    const size_t lds = (size_t)XT * cells * sizeof(T);
    const int grid = B * H * ((W + XT - 1) / XT) * ((C + CB - 1) / CB);
    T* goutT = static_cast<T*>(ws);
    if (sizeof(T) == 4 && d <= kTiledMaxD && XT >= 4 && 1LL * B * C * H * tiled_pitch(W, d) < 0x7fffffffLL)   // register-tiled form (f32)
        return corr_bwd_tiled_f32(reinterpret_cast<const float*>(gout), reinterpret_cast<const float*>(fm0), reinterpret_cast<const float*>(fm1),
                                  reinterpret_cast<float*>(g0), reinterpret_cast<float*>(g1), B, C, H, W, d, s, XT, reinterpret_cast<float*>(goutT), st);
    hipLaunchKernelGGL((k_corr_bwd_blocked<T, false>), dim3(grid), dim3(kBlk), lds, st, gout, fm1, g0, B, C, H, W, d, s, XT, CB);
    int rc = launch_status();
    if (rc != D2T_OK) return rc;
    hipLaunchKernelGGL(k_corr_gout_by_displaced<T>, dim3(grid_for(1LL * B * H * W * cells, kBlk, 256 * 32)), dim3(kBlk), 0, st, gout, goutT, B, H, W, d);
    rc = launch_status();
    if (rc != D2T_OK) return rc;
    hipLaunchKernelGGL((k_corr_bwd_blocked<T, true>), dim3(grid), dim3(kBlk), lds, st, goutT, fm0, g1, B, C, H, W, d, s, XT, CB);
    return launch_status();
}

#define D2T_INSTANTIATE_BLOCKED(T)                                                                      \
    template bool corr_blocked_supported<T>(int, int, int, int, int, int);                               \
    template size_t corr_bwd_blocked_ws_bytes<T>(int, int, int, int, int, int);                          \
    template int corr_fwd_blocked<T>(const T*, const T*, T*, int, int, int, int, int, int, hipStream_t); \
    template int corr_bwd_blocked<T>(const T*, const T*, const T*, T*, T*, int, int, int, int, int, int, void*, hipStream_t);
D2T_INSTANTIATE_BLOCKED(float)
D2T_INSTANTIATE_BLOCKED(double)

}  // namespace d2t
