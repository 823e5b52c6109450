// d2t_generic.hip -- type-generic kernels (f32 + f64) for the three ops.
//
// These are the correctness anchors of the library: one thread owns one output element
// and accumulates it in the reference's own order wherever the reference defines one, so
// forward results are bit-identical to a serial evaluation of the reference arithmetic
// (FMA-contracted as nvcc builds it).  D2T_IMPL_GENERIC selects them; under the default
// dispatch they serve what neither the tuned gfx950 kernels nor the second tier
// (d2t_corr_blocked.hip, d2t_pool_lists.hip: the same operations in the same order, bit-identical
// results, without the per-thread redundancy) take: maps narrower than 4 columns, d_max whose
// window does not fit LDS, fewer than 32 RoIs in the ROIPool forward, callers that pass no
// workspace.  All backward kernels are in gather form: no atomics, every output element written
// exactly once, deterministic.
//
// Reference semantics (paths relative to /root/reference/detect_to_track/models/):
//   correlation  pointwise_correlation/pointwise_correlation_cuda.cu:62-174
//   roipool      roipool/roipool_cuda.cu:5-127
//   psroipool    ps_roipool/ps_roipool_cuda.cu:9-141
#include "d2t_kernels.hpp"

namespace d2t {

constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------
// PointwiseCorrelation forward: thread per output cell (b,i,j,ci,cj).  Lanes run along
// cj, i.e. along W of FM1: coalesced.  Cells outside the reference's loop ranges are
// the structural zeros (its launcher pre-fills them, :192); here the kernel writes them.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_corr_fwd_generic(const T* __restrict__ fm0, const T* __restrict__ fm1, T* __restrict__ out,
                   int B, int C, int H, int W, int d, int s, int ps, int cs, long long bs)
{
    const int cw = 2 * d + 1;
    const int plane = H * W;
    const int total = B * plane * cw * cw;
    for (long long i64 = (long long)blockIdx.x * kBlock + threadIdx.x; i64 < total; i64 += (long long)gridDim.x * kBlock) {
        const int idx = static_cast<int>(i64);
        const int cj = idx % cw;
        const int ci = (idx / cw) % cw;
        const int pix = idx / (cw * cw);
        const int j = pix % W, i = (pix / W) % H, b = pix / plane;
        const int di = i - d + ci, dj = j - d + cj;
        T acc = T(0);
        if (corr_axis_hit(i, di, H, d, s) && corr_axis_hit(j, dj, W, d, s)) {
            const T* a = fm0 + (size_t)b * C * plane + i * W + j;
            const T* q = fm1 + (size_t)b * C * plane + di * W + dj;
            for (int c = 0; c < C; ++c)                       // ascending c, fused (:105-107)
                acc = fma_t(a[(size_t)c * plane], q[(size_t)c * plane], acc);
        }
        out[(size_t)b * bs + (size_t)(pix - b * plane) * ps + (size_t)(ci * cw + cj) * cs] = acc;   // CellLayout
    }
}

// ------------------------------------------------------------------------------------
// PointwiseCorrelation backward, gather form.  Thread per input element (b,c,y,x); it
// produces BOTH gradients of that element:
//   gfm0[b,c,y,x] = sum over the window of (y,x), ascending (di,dj), of g * FM1   (thread-
//                   owned in the reference too, :168, so this order IS the reference's)
//   gfm1[b,c,y,x] = sum over centres (i,j) whose window contains (y,x) of g * FM0 (:169,
//                   atomicAdd in the reference: order undefined there, ascending (i,j) here)
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_corr_bwd_generic(const T* __restrict__ gout, const T* __restrict__ fm0, const T* __restrict__ fm1,
                   T* __restrict__ g0, T* __restrict__ g1,
                   int B, int C, int H, int W, int d, int s, int ps, int cs, long long bs)
{
    const int cw = 2 * d + 1;
    const int plane = H * W;
    const int total = B * C * plane;
    for (long long i64 = (long long)blockIdx.x * kBlock + threadIdx.x; i64 < total; i64 += (long long)gridDim.x * kBlock) {
        const int idx = static_cast<int>(i64);
        const int x = idx % W, y = (idx / W) % H;
        const int b = idx / (C * plane);
        const size_t chan = (size_t)(idx / plane) * plane;       // offset of plane (b,c)
        const T* gb = gout + (size_t)b * bs;

        // d/dFM0: centre (y,x), walk its window.
        T a0 = T(0);
        {
            const int lo_i = y - d > 0 ? y - d : 0, hi_i = y + d < H ? y + d : H;
            const int lo_j = x - d > 0 ? x - d : 0, hi_j = x + d < W ? x + d : W;
            const T* gc = gb + (size_t)(y * W + x) * ps;
            for (int di = lo_i; di < hi_i; di += s)
                for (int dj = lo_j; dj < hi_j; dj += s)
                    a0 = fma_t(gc[(size_t)((di - y + d) * cw + (dj - x + d)) * cs], fm1[chan + di * W + dj], a0);
        }
        // d/dFM1: displaced pixel (y,x), walk the centres that reach it.
        T a1 = T(0);
        {
            const int i_lo = y - d > 0 ? y - d : 0, i_hi = y + d < H - 1 ? y + d : H - 1;
            const int j_lo = x - d > 0 ? x - d : 0, j_hi = x + d < W - 1 ? x + d : W - 1;
            for (int i = i_lo; i <= i_hi; ++i) {
                if (!corr_axis_hit(i, y, H, d, s)) continue;
                for (int j = j_lo; j <= j_hi; ++j) {
                    if (!corr_axis_hit(j, x, W, d, s)) continue;
                    a1 = fma_t(gb[(size_t)(i * W + j) * ps + (size_t)((y - i + d) * cw + (x - j + d)) * cs],
                               fm0[chan + i * W + j], a1);
                }
            }
        }
        g0[idx] = a0;
        g1[idx] = a1;
    }
}

__global__ void k_corr_mask(uint8_t* __restrict__ mask, int H, int W, int d, int s)
{
    const int cw = 2 * d + 1;
    const int total = H * W * cw * cw;
    for (long long i64 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i64 < total; i64 += (long long)gridDim.x * blockDim.x) {
        const int idx = static_cast<int>(i64);
        const int cj = idx % cw, ci = (idx / cw) % cw, pix = idx / (cw * cw);
        const int j = pix % W, i = pix / W;
        mask[idx] = (corr_axis_hit(i, i - d + ci, H, d, s) && corr_axis_hit(j, j - d + cj, W, d, s)) ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------
// ROIPool forward: thread per output (r,c,i,j); running sum over the bin in row-major
// pixel order (the reference's order, :56-60), then sum / n with no zero guard (:61).
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_roipool_fwd_generic(const T* __restrict__ fm, const T* __restrict__ rois, T* __restrict__ out,
                      int R, int C, int H, int W, int k)
{
    const int total = R * C * k * k;
    for (long long i64 = (long long)blockIdx.x * kBlock + threadIdx.x; i64 < total; i64 += (long long)gridDim.x * kBlock) {
        const int idx = static_cast<int>(i64);
        const int j = idx % k, i = (idx / k) % k, c = (idx / (k * k)) % C, r = idx / (k * k * C);
        const Bounds bb = roi_bin<T>(rois + 4 * r, i, j, H, W, k);
        const T* ch = fm + (size_t)c * H * W;
        T acc = T(0);
        for (int pI = bb.i0; pI < bb.i1; ++pI)
            for (int pJ = bb.j0; pJ < bb.j1; ++pJ)
                acc += ch[pI * W + pJ];
        const int n = (bb.i1 - bb.i0) * (bb.j1 - bb.j0);
        out[idx] = acc / static_cast<T>(n);          // n == 0 -> 0/0 = NaN, as the reference
    }
}

template <typename T>
__global__ void k_roipool_bins(const T* __restrict__ rois, int32_t* __restrict__ bounds,
                               int R, int H, int W, int k, int position_sensitive)
{
    const int total = R * k * k;
    for (long long i64 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i64 < total; i64 += (long long)gridDim.x * blockDim.x) {
        const int idx = static_cast<int>(i64);
        const int j = idx % k, i = (idx / k) % k, r = idx / (k * k);
        const Bounds bb = position_sensitive ? psroi_cell<T>(rois + 4 * r, i, j, H, W, k)
                                             : roi_bin<T>(rois + 4 * r, i, j, H, W, k);
        reinterpret_cast<int4*>(bounds)[idx] = make_int4(bb.i0, bb.i1, bb.j0, bb.j1);
    }
}

// ------------------------------------------------------------------------------------
// ROIPool backward, gather form: thread per input pixel (c,y,x).  `bins` is the (R,k,k,4)
// table written by k_roipool_bins on the same stream just before.  Row bounds of a bin
// depend only on (r,i) and column bounds only on (r,j) (roipool_cuda.cu:41-50), so the
// membership test is separable.  Ascending (r,i,j) accumulation; g / n as :123.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_roipool_bwd_generic(const T* __restrict__ gout, const int32_t* __restrict__ bins, T* __restrict__ gin,
                      int R, int C, int H, int W, int k, const unsigned long long* __restrict__ gate = nullptr, long long gate_cap = 0)
{
    if (gate && (long long)*gate <= gate_cap) return;                           // (d2t_pool_lists.hip: only when its lists overflowed)
    const int total = C * H * W;
    const int4* bt = reinterpret_cast<const int4*>(bins);
    for (long long i64 = (long long)blockIdx.x * kBlock + threadIdx.x; i64 < total; i64 += (long long)gridDim.x * kBlock) {
        const int idx = static_cast<int>(i64);
        const int x = idx % W, y = (idx / W) % H, c = idx / (H * W);
        T acc = T(0);
        for (int r = 0; r < R; ++r) {
            const int4* br = bt + (size_t)r * k * k;
            // quick reject on the RoI's overall extent.  Bin edges are monotone in the bin index but
            // DEcreasing for a RoI of negative height / width (its bins run in reverse order and
            // one-pixel bins stay non-empty, roipool_cuda.cu:41-50), so take min / max of both ends.
            const int4 bf = br[0], bl = br[k * k - 1];
            const int y_lo = bf.x < bl.x ? bf.x : bl.x, y_hi = bf.y > bl.y ? bf.y : bl.y;
            const int x_lo = bf.z < bl.z ? bf.z : bl.z, x_hi = bf.w > bl.w ? bf.w : bl.w;
            if (y < y_lo || y >= y_hi || x < x_lo || x >= x_hi) continue;
            const T* gr = gout + ((size_t)r * C + c) * k * k;
            for (int i = 0; i < k; ++i) {
                const int4 bi = br[i * k];
                if (y < bi.x || y >= bi.y) continue;
                for (int j = 0; j < k; ++j) {
                    const int4 bj = br[j];
                    if (x < bj.z || x >= bj.w) continue;
                    const int n = (bi.y - bi.x) * (bj.w - bj.z);
                    acc += gr[i * k + j] / static_cast<T>(n);
                }
            }
        }
        gin[idx] = acc;
    }
}

// ------------------------------------------------------------------------------------
// PSROIPool forward: thread per output (r,t,i,j), channel (t+1)*(i*k+j)
// (ps_roipool_cuda.cu:58), guarded divide (:67-69).
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_psroipool_fwd_generic(const T* __restrict__ fm, const T* __restrict__ rois, T* __restrict__ out,
                        int R, int nT, int H, int W, int k)
{
    const int total = R * nT * k * k;
    for (long long i64 = (long long)blockIdx.x * kBlock + threadIdx.x; i64 < total; i64 += (long long)gridDim.x * kBlock) {
        const int idx = static_cast<int>(i64);
        const int j = idx % k, i = (idx / k) % k, t = (idx / (k * k)) % nT, r = idx / (k * k * nT);
        const Bounds bb = psroi_cell<T>(rois + 4 * r, i, j, H, W, k);
        const T* ch = fm + (size_t)((t + 1) * (i * k + j)) * H * W;
        T acc = T(0);
        for (int pI = bb.i0; pI < bb.i1; ++pI)
            for (int pJ = bb.j0; pJ < bb.j1; ++pJ)
                acc += ch[pI * W + pJ];
        const int n = (bb.i1 - bb.i0) * (bb.j1 - bb.j0);
        if (n > 0) acc /= static_cast<T>(n);
        out[idx] = acc;
    }
}

__global__ void k_psroipool_channels(int32_t* __restrict__ ch, int nT, int k)
{
    const int total = nT * k * k;
    for (long long i64 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i64 < total; i64 += (long long)gridDim.x * blockDim.x) {
        const int idx = static_cast<int>(i64);
        const int j = idx % k, i = (idx / k) % k, t = idx / (k * k);
        ch[idx] = (t + 1) * (i * k + j);
    }
}

// ------------------------------------------------------------------------------------
// PSROIPool backward, gather form: thread per input pixel (ch,y,x).  The channel map
// (t+1)*bin is many-to-one: channel ch receives from every (t,bin) with (t+1)*bin == ch,
// i.e. bin | ch with ch/bin <= nT (and for ch == 0: bin 0 with every t).  Channels no
// pair maps to stay zero.  `cells` is the (R,k,k,4) table from k_roipool_bins(ps=1).
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_psroipool_bwd_generic(const T* __restrict__ gout, const int32_t* __restrict__ cells, T* __restrict__ gin,
                        int R, int nT, int H, int W, int k, const unsigned long long* __restrict__ gate = nullptr, long long gate_cap = 0)
{
    if (gate && (long long)*gate <= gate_cap) return;                // (d2t_pool_lists.hip: only when its lists overflowed)
    const int kk = k * k;
    const int total = nT * kk * H * W;
    const int4* ct = reinterpret_cast<const int4*>(cells);
    for (long long i64 = (long long)blockIdx.x * kBlock + threadIdx.x; i64 < total; i64 += (long long)gridDim.x * kBlock) {
        const int idx = static_cast<int>(i64);
        const int x = idx % W, y = (idx / W) % H, ch = idx / (H * W);
        T acc = T(0);
        const int bin_lo = ch == 0 ? 0 : 1;
        const int bin_hi = ch == 0 ? 0 : kk - 1;
        for (int bin = bin_lo; bin <= bin_hi; ++bin) {
            int t_lo, t_hi;
            if (ch == 0) { t_lo = 0; t_hi = nT - 1; }
            else {
                if (ch % bin != 0) continue;
                const int tp1 = ch / bin;
                if (tp1 > nT) continue;
                t_lo = t_hi = tp1 - 1;
            }
            for (int r = 0; r < R; ++r) {
                const int4 cb = ct[(size_t)r * kk + bin];
                if (y < cb.x || y >= cb.y || x < cb.z || x >= cb.w) continue;
                const int n = (cb.y - cb.x) * (cb.w - cb.z);      // > 0 here
                for (int t = t_lo; t <= t_hi; ++t)
                    acc += gout[((size_t)r * nT + t) * kk + bin] / static_cast<T>(n);
            }
        }
        gin[idx] = acc;
    }
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
template <typename T>
int corr_fwd_generic(const T* fm0, const T* fm1, T* out, int B, int C, int H, int W, int d, int s, hipStream_t st,
                     int ps, int cs, long long bs)
{
    if (ps == 0) { ps = (2 * d + 1) * (2 * d + 1); cs = 1; bs = 1LL * H * W * ps; }   // the reference's layout
    const long long total = 1LL * B * H * W * (2 * d + 1) * (2 * d + 1);
    if (total == 0) return D2T_OK;
    hipLaunchKernelGGL(k_corr_fwd_generic<T>, dim3(grid_for(total, kBlock, 256 * 32)), dim3(kBlock), 0, st,
                       fm0, fm1, out, B, C, H, W, d, s, ps, cs, bs);
    return launch_status();
}

template <typename T>
int corr_bwd_generic(const T* gout, const T* fm0, const T* fm1, T* g0, T* g1,
                     int B, int C, int H, int W, int d, int s, hipStream_t st, int ps, int cs, long long bs)
{
    if (ps == 0) { ps = (2 * d + 1) * (2 * d + 1); cs = 1; bs = 1LL * H * W * ps; }
    const long long total = 1LL * B * C * H * W;
    if (total == 0) return D2T_OK;
    hipLaunchKernelGGL(k_corr_bwd_generic<T>, dim3(grid_for(total, kBlock, 256 * 32)), dim3(kBlock), 0, st,
                       gout, fm0, fm1, g0, g1, B, C, H, W, d, s, ps, cs, bs);
    return launch_status();
}

int corr_mask(uint8_t* mask, int H, int W, int d, int s, hipStream_t st)
{
    const long long total = 1LL * H * W * (2 * d + 1) * (2 * d + 1);
    if (total == 0) return D2T_OK;
    hipLaunchKernelGGL(k_corr_mask, dim3(grid_for(total, 256)), dim3(256), 0, st, mask, H, W, d, s);
    return launch_status();
}

template <typename T>
int roipool_fwd_generic(const T* fm, const T* rois, T* out, int R, int C, int H, int W, int k, hipStream_t st)
{
    const long long total = 1LL * R * C * k * k;
    if (total == 0) return D2T_OK;
    hipLaunchKernelGGL(k_roipool_fwd_generic<T>, dim3(grid_for(total, kBlock, 256 * 32)), dim3(kBlock), 0, st,
                       fm, rois, out, R, C, H, W, k);
    return launch_status();
}

template <typename T>
int roipool_bins(const T* rois, int32_t* bounds, int R, int H, int W, int k, hipStream_t st)
{
    const long long total = 1LL * R * k * k;
    if (total == 0) return D2T_OK;
    hipLaunchKernelGGL(k_roipool_bins<T>, dim3(grid_for(total, 256)), dim3(256), 0, st, rois, bounds, R, H, W, k, 0);
    return launch_status();
}

template <typename T>
int psroipool_bins(const T* rois, int32_t* bounds, int R, int H, int W, int k, hipStream_t st)
{
    const long long total = 1LL * R * k * k;
    if (total == 0) return D2T_OK;
    hipLaunchKernelGGL(k_roipool_bins<T>, dim3(grid_for(total, 256)), dim3(256), 0, st, rois, bounds, R, H, W, k, 1);
    return launch_status();
}

template <typename T>
int roipool_bwd_generic(const T* gout, const T* rois, T* gin, int32_t* bins,
                        int R, int C, int H, int W, int k, hipStream_t st)
{
    const long long total = 1LL * C * H * W;
    if (total == 0) return D2T_OK;
    int rc = roipool_bins<T>(rois, bins, R, H, W, k, st);
    if (rc != D2T_OK) return rc;
    hipLaunchKernelGGL(k_roipool_bwd_generic<T>, dim3(grid_for(total, kBlock, 256 * 32)), dim3(kBlock), 0, st,
                       gout, bins, gin, R, C, H, W, k);
    return launch_status();
}

// the same kernel behind a device-side gate: it runs only if *gate > cap (d2t_pool_lists.hip); `bins` already written
template <typename T>
int roipool_bwd_generic_gated(const T* gout, const int32_t* bins, T* gin, const unsigned long long* gate, long long cap,
                              int R, int C, int H, int W, int k, hipStream_t st)
{
    const long long total = 1LL * C * H * W;
    if (total == 0) return D2T_OK;
    hipLaunchKernelGGL(k_roipool_bwd_generic<T>, dim3(grid_for(total, kBlock, 256 * 32)), dim3(kBlock), 0, st,
                       gout, bins, gin, R, C, H, W, k, gate, cap);
    return launch_status();
}

template <typename T>
int psroipool_bwd_generic_gated(const T* gout, const int32_t* cells, T* gin, const unsigned long long* gate, long long cap,
                                int R, int nT, int H, int W, int k, hipStream_t st)
{
    const long long total = 1LL * nT * k * k * H * W;
    if (total == 0) return D2T_OK;
    hipLaunchKernelGGL(k_psroipool_bwd_generic<T>, dim3(grid_for(total, kBlock, 256 * 32)), dim3(kBlock), 0, st,
                       gout, cells, gin, R, nT, H, W, k, gate, cap);
    return launch_status();
}

template <typename T>
int psroipool_fwd_generic(const T* fm, const T* rois, T* out, int R, int nT, int H, int W, int k, hipStream_t st)
{
    const long long total = 1LL * R * nT * k * k;
    if (total == 0) return D2T_OK;
    hipLaunchKernelGGL(k_psroipool_fwd_generic<T>, dim3(grid_for(total, kBlock, 256 * 32)), dim3(kBlock), 0, st,
                       fm, rois, out, R, nT, H, W, k);
    return launch_status();
}

int psroipool_channels(int32_t* ch, int nT, int k, hipStream_t st)
{
    const long long total = 1LL * nT * k * k;
    if (total == 0) return D2T_OK;
    hipLaunchKernelGGL(k_psroipool_channels, dim3(grid_for(total, 256)), dim3(256), 0, st, ch, nT, k);
    return launch_status();
}

template <typename T>
int psroipool_bwd_generic(const T* gout, const T* rois, T* gin, int32_t* cells,
                          int R, int nT, int H, int W, int k, hipStream_t st)
{
    const long long total = 1LL * nT * k * k * H * W;
    if (total == 0) return D2T_OK;
    int rc = psroipool_bins<T>(rois, cells, R, H, W, k, st);
    if (rc != D2T_OK) return rc;
    hipLaunchKernelGGL(k_psroipool_bwd_generic<T>, dim3(grid_for(total, kBlock, 256 * 32)), dim3(kBlock), 0, st,
                       gout, cells, gin, R, nT, H, W, k);
    return launch_status();
}

#define D2T_INSTANTIATE(T)                                                                              \
    template int corr_fwd_generic<T>(const T*, const T*, T*, int, int, int, int, int, int, hipStream_t, int, int, long long); \
    template int corr_bwd_generic<T>(const T*, const T*, const T*, T*, T*, int, int, int, int, int, int, hipStream_t, int, int, long long); \
    template int roipool_fwd_generic<T>(const T*, const T*, T*, int, int, int, int, int, hipStream_t);   \
    template int roipool_bwd_generic<T>(const T*, const T*, T*, int32_t*, int, int, int, int, int, hipStream_t); \
    template int roipool_bwd_generic_gated<T>(const T*, const int32_t*, T*, const unsigned long long*, long long, int, int, int, int, int, hipStream_t); \
    template int psroipool_bwd_generic_gated<T>(const T*, const int32_t*, T*, const unsigned long long*, long long, int, int, int, int, int, hipStream_t); \
    template int psroipool_fwd_generic<T>(const T*, const T*, T*, int, int, int, int, int, hipStream_t); \
    template int psroipool_bwd_generic<T>(const T*, const T*, T*, int32_t*, int, int, int, int, int, hipStream_t); \
    template int roipool_bins<T>(const T*, int32_t*, int, int, int, int, hipStream_t);                   \
    template int psroipool_bins<T>(const T*, int32_t*, int, int, int, int, hipStream_t);
D2T_INSTANTIATE(float)
D2T_INSTANTIATE(double)

}  // namespace d2t
