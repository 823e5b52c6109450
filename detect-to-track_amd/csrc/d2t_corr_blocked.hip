// d2t_corr_blocked.hip -- PointwiseCorrelation OUTSIDE the tuned envelope (any d_max, any stride, f32 and f64),
// reference layout: the same arithmetic and the same summation ORDER as the thread-per-element kernels of
// d2t_generic.hip (so the results are bit-identical to them, forward and both gradients), with the memory
// behaviour fixed.  D2T_IMPL_GENERIC keeps the thread-per-element kernels as the correctness anchor; D2T_IMPL_AUTO
// takes these when the tuned gfx950 kernels do not apply (round 4: the cost of leaving the envelope was 15x forward
// and 35-150x backward, include/d2t_ops.h).
//
// Reference semantics: pointwise_correlation_cuda.cu:84-107 (forward), :145-171 (backward).
//
//   forward   a thread owns FOUR adjacent cells (ci, cj .. cj+3) of one pixel: per channel one load of FM0[c][i][j] and
//             one 16-byte (f64: 32-byte) load of FM1[c][di][dj .. dj+3] feed four fused multiply-adds -- the
//             thread-per-cell kernel issued two loads per multiply-add.  Each cell is still ONE ascending-channel fma
//             chain (:105-107).  Groups that touch the map's left / right edge take the per-cell path.
//   backward  the thread-per-element kernel reads gradOut[b][pixel][cell] with the LANES along x, i.e. 4 (2d+1)^2 bytes
//             apart: one cache line per lane and term.  Here a workgroup stages the cells of (b, row y, XT pixels) in
//             LDS once and every thread (channel, pixel) walks its window from there; FM is read along x (coalesced).
//             gradFM1 needs gradOut indexed by the DISPLACED pixel: a pre-pass writes that view into the caller's
//             workspace (goutT[b][y][x][ci][cj] = gradOut[b][y-ci+d][x-cj+d][ci][cj]), then the same kernel runs with
//             FM0 and mirrored offsets, centres in ascending (i, j) order as in d2t_generic.hip.
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

template <typename T>
size_t corr_bwd_blocked_ws_bytes(int B, int C, int H, int W, int d, int s)
{
    if (!corr_blocked_supported<T>(B, C, H, W, d, s)) return 0;
    const size_t cw = 2 * (size_t)d + 1;
    return ((size_t)B * H * W * cw * cw * sizeof(T) + 255) / 256 * 256;   // gradOut by displaced pixel
}

template <typename T>
int corr_fwd_blocked(const T* fm0, const T* fm1, T* out, int B, int C, int H, int W, int d, int s, hipStream_t st)
{
    const long long cw = 2LL * d + 1, total = 1LL * B * H * W * cw * ((cw + 3) / 4);
    if (total == 0) return D2T_OK;
    hipLaunchKernelGGL(k_corr_fwd_blocked<T>, dim3(grid_for(total, kBlk, 256 * 32)), dim3(kBlk), 0, st, fm0, fm1, out, B, C, H, W, d, s);
    return launch_status();
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
