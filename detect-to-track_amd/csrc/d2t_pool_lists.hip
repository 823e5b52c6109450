// d2t_pool_lists.hip -- ROIPool backward OUTSIDE the tuned envelope (any bin count k, f32 and f64): the same sums in the
// same order as the thread-per-pixel kernel of d2t_generic.hip (k_roipool_bwd_generic: ascending (r, i, j), each term
// gradOut / n, roipool_cuda.cu:112-124 in gather form), so the result is bit-identical to it -- without its redundancy.
// Which bins contain a pixel does not depend on the channel; the thread-per-(c, y, x) kernel re-derives it per channel
// (C x H x W x R bin tests: 2.3 ms at R = 300, C = 1024, 38 x 63).  Here:
//
//   1. k_cover_count / k_cover_scan / k_cover_fill   per pixel, the list of the bins (r, i, j) that contain it, ascending, built
//      once (workgroup = pixel, thread = RoI; membership is separable: row bins of r that contain y times column bins that
//      contain x);
//   2. k_grad_by_bin   Q[(r, i, j)][c] = gradOut[r][c][i][j] / n(r, i, j): the division of :123 done once per gradOut element
//      instead of once per pixel of its bin, and the channel made the contiguous index;
//   3. k_roipool_bwd_lists   workgroup = pixel, thread = channel: gradIn[c][y][x] = sum over the pixel's list of Q[bin][c] --
//      the list entry is wave-uniform, the loads are coalesced along c.
//
// The lists live in the caller's workspace, sized for RoIs no larger than the map (R (H + 2k)(W + 2k) entries).  RoIs far larger
// than the map can exceed that (every bin of such a RoI covers the whole map): the scan notices, steps 1c-3 do nothing and the
// thread-per-pixel kernel, launched last and otherwise returning at once, does the work -- decided on the device, no host
// synchronisation.
#include "d2t_kernels.hpp"

namespace d2t {

namespace {

constexpr int kBlk = 256;

// Workgroups are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of pixels (bijective for any grid size).
__device__ __forceinline__ int xcd_run(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

__device__ __forceinline__ int axis_hits(const int4* __restrict__ br, int stride, int k, int p, bool rows)
{
    int n = 0;
    for (int i = 0; i < k; ++i) {
        const int4 b = br[i * stride];
        n += rows ? (p >= b.x && p < b.y) : (p >= b.z && p < b.w);
    }
    return n;
}

// exclusive scan of one int per thread over the workgroup; returns the thread's prefix, *total = the sum
__device__ __forceinline__ int block_exclusive_scan(int v, int* sh, int* total)
{
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int off = 1; off < kBlk; off <<= 1) {
        const int add = tid >= off ? sh[tid - off] : 0;
        __syncthreads();
        sh[tid] += add;
        __syncthreads();
    }
    const int incl = sh[tid];
    *total = sh[kBlk - 1];
    __syncthreads();
    return incl - v;
}

__global__ void __launch_bounds__(kBlk)
k_cover_count(const int32_t* __restrict__ bins, int32_t* __restrict__ counts, int R, int H, int W, int k)
{
    __shared__ int sh[kBlk];
    const int4* bt = reinterpret_cast<const int4*>(bins);
    const int p = blockIdx.x, y = p / W, x = p - y * W;
    int n = 0;
    for (int r = threadIdx.x; r < R; r += kBlk) {
        const int4* br = bt + (size_t)r * k * k;
        const int ni = axis_hits(br, k, k, y, true);
        if (ni) n += ni * axis_hits(br, 1, k, x, false);
    }
    int total;
    block_exclusive_scan(n, sh, &total);
    if (threadIdx.x == 0) counts[p] = total;
}

// offsets[p] = exclusive prefix of counts (in place), offsets[n] = total (saturating: a sum past INT_MAX stays there)
__global__ void __launch_bounds__(kBlk)
k_cover_scan(int32_t* __restrict__ offsets, int n)
{
    __shared__ int sh[kBlk];
    long long base = 0;
    for (int p0 = 0; p0 < n; p0 += kBlk) {
        const int p = p0 + threadIdx.x;
        const int v = p < n ? offsets[p] : 0;
        int total;
        const int pre = block_exclusive_scan(v, sh, &total);
        const long long o = base + pre;
        if (p < n) offsets[p] = o > 0x7fffffffLL ? 0x7fffffff : (int)o;
        base += total;
    }
    if (threadIdx.x == 0) offsets[n] = base > 0x7fffffffLL ? 0x7fffffff : (int)base;
}

__global__ void __launch_bounds__(kBlk)
k_cover_fill(const int32_t* __restrict__ bins, const int32_t* __restrict__ offsets, int32_t* __restrict__ entries, int cap,
             int R, int H, int W, int k)
{
    __shared__ int sh[kBlk];
    if (offsets[H * W] > cap) return;                                 // the lists do not fit: the thread-per-pixel kernel runs instead
    const int4* bt = reinterpret_cast<const int4*>(bins);
    const int p = blockIdx.x, y = p / W, x = p - y * W;
    int base = offsets[p];
    for (int r0 = 0; r0 < R; r0 += kBlk) {                            // RoIs ascending: 256 at a time, in thread order
        const int r = r0 + threadIdx.x;
        const int4* br = bt + (size_t)(r < R ? r : 0) * k * k;
        int ni = 0, nj = 0;
        if (r < R) {
            ni = axis_hits(br, k, k, y, true);
            if (ni) nj = axis_hits(br, 1, k, x, false);
        }
        int total;
        int pos = base + block_exclusive_scan(ni * nj, sh, &total);
        if (ni * nj) {
            for (int i = 0; i < k; ++i) {                             // (i, j) ascending
                const int4 bi = br[i * k];
                if (y < bi.x || y >= bi.y) continue;
                for (int j = 0; j < k; ++j) {
                    const int4 bj = br[j];
                    if (x < bj.z || x >= bj.w) continue;
                    entries[pos++] = (r * k + i) * k + j;
                }
            }
        }
        base += total;
    }
}

// Q[(r, bin)][c] = gradOut[r][c][bin] / n(r, bin).  Workgroup = (r, chunk of CH channels): CH * k^2 contiguous elements in,
// through LDS, rows of CH contiguous channels out.
template <typename T>
__global__ void __launch_bounds__(kBlk)
k_grad_by_bin(const T* __restrict__ gout, const int32_t* __restrict__ bins, const int32_t* __restrict__ gate, int cap,
              T* __restrict__ q, int R, int C, int kk, int k, int CH)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (*gate > cap) return;
    T* tile = reinterpret_cast<T*>(lds_raw);                          // [CH][kk + 1]
    const int4* bt = reinterpret_cast<const int4*>(bins);
    const int chunks = (C + CH - 1) / CH;
    const int r = blockIdx.x / chunks, c0 = (blockIdx.x % chunks) * CH;
    const int nc = C - c0 < CH ? C - c0 : CH;
    const T* src = gout + ((size_t)r * C + c0) * kk;
    for (int e = threadIdx.x; e < nc * kk; e += kBlk) tile[(e / kk) * (kk + 1) + e % kk] = src[e];
    __syncthreads();
    for (int e = threadIdx.x; e < kk * CH; e += kBlk) {
        const int bin = e / CH, cl = e - bin * CH;
        if (cl >= nc) continue;
        const int4 bi = bt[(size_t)r * kk + (bin / k) * k], bj = bt[(size_t)r * kk + bin % k];
        const int n = (bi.y - bi.x) * (bj.w - bj.z);
        q[((size_t)r * kk + bin) * C + c0 + cl] = tile[cl * (kk + 1) + bin] / static_cast<T>(n);    // (:123)
    }
}

// gradIn[c][y][x] = sum over the pixel's list, in list order, of Q[entry][c].  Workgroup = pixel; a thread owns the channels
// tid, tid + 256, tid + 512, tid + 768 of each group of 1024.
template <typename T>
__global__ void __launch_bounds__(kBlk)
k_roipool_bwd_lists(const T* __restrict__ q, const int32_t* __restrict__ offsets, const int32_t* __restrict__ entries, int cap,
                    T* __restrict__ gin, int C, int HW)
{
    if (offsets[HW] > cap) return;
    const int p = xcd_run(blockIdx.x, HW);                          // neighbouring pixels (nearly the same lists) share an L2
    const int beg = offsets[p], end = offsets[p + 1];
    for (int c0 = 0; c0 < C; c0 += 4 * kBlk) {
        const int c = c0 + threadIdx.x;
        T acc[4] = {T(0), T(0), T(0), T(0)};
        for (int e = beg; e < end; ++e) {
            const T* row = q + (size_t)entries[e] * C + c;
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (c + m * kBlk < C) acc[m] += row[m * kBlk];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (c + m * kBlk < C) gin[(size_t)(c + m * kBlk) * HW + p] = acc[m];
    }
}

inline size_t up256(size_t b) { return (b + 255) / 256 * 256; }

struct ListsLayout {
    size_t bins, offsets, entries, q, total;
    long long cap;
};

template <typename T>
ListsLayout lists_layout(int R, int C, int H, int W, int k)
{
    ListsLayout L;
    L.cap = 1LL * R * (H + 2LL * k) * (W + 2LL * k);
    L.bins = 0;
    L.offsets = up256((size_t)R * k * k * 16);
    L.entries = L.offsets + up256(((size_t)H * W + 1) * 4);
    L.q = L.entries + up256((size_t)L.cap * 4);
    L.total = L.q + up256((size_t)R * C * k * k * sizeof(T));
    return L;
}

}  // namespace

template <typename T>
bool roipool_bwd_lists_supported(int R, int C, int H, int W, int k)
{
    if (R < 1 || C < 1 || H < 1 || W < 1 || k < 1) return false;
    const long long cap = 1LL * R * (H + 2LL * k) * (W + 2LL * k);
    return cap < 0x4000000LL && 1LL * R * k * k < 0x7fffffffLL / 4 && fits_i32(1LL * R * C * k * k) && fits_i32(1LL * C * H * W) &&
           (size_t)k * k * sizeof(T) <= 16 * 1024;
}

template <typename T>
size_t roipool_bwd_lists_ws_bytes(int R, int C, int H, int W, int k)
{
    return roipool_bwd_lists_supported<T>(R, C, H, W, k) ? lists_layout<T>(R, C, H, W, k).total : 0;
}

template <typename T>
int roipool_bwd_lists(const T* gout, const T* rois, T* gin, void* ws, int R, int C, int H, int W, int k, hipStream_t st)
{
    const ListsLayout L = lists_layout<T>(R, C, H, W, k);
    unsigned char* base = static_cast<unsigned char*>(ws);
    int32_t* bins = reinterpret_cast<int32_t*>(base + L.bins);
    int32_t* offsets = reinterpret_cast<int32_t*>(base + L.offsets);
    int32_t* entries = reinterpret_cast<int32_t*>(base + L.entries);
    T* q = reinterpret_cast<T*>(base + L.q);
    const int HW = H * W, kk = k * k, cap = (int)L.cap;
    int rc = roipool_bins<T>(rois, bins, R, H, W, k, st);
    if (rc != D2T_OK) return rc;
    hipLaunchKernelGGL(k_cover_count, dim3(HW), dim3(kBlk), 0, st, bins, offsets, R, H, W, k);
    hipLaunchKernelGGL(k_cover_scan, dim3(1), dim3(kBlk), 0, st, offsets, HW);
    hipLaunchKernelGGL(k_cover_fill, dim3(HW), dim3(kBlk), 0, st, bins, offsets, entries, cap, R, H, W, k);
    int CH = 64;
    while (CH > 1 && (size_t)CH * (kk + 1) * sizeof(T) > 32 * 1024) CH >>= 1;
    hipLaunchKernelGGL(k_grad_by_bin<T>, dim3(R * ((C + CH - 1) / CH)), dim3(kBlk), (size_t)CH * (kk + 1) * sizeof(T), st,
                       gout, bins, offsets + HW, cap, q, R, C, kk, k, CH);
    hipLaunchKernelGGL(k_roipool_bwd_lists<T>, dim3(HW), dim3(kBlk), 0, st, q, offsets, entries, cap, gin, C, HW);
    rc = launch_status();
    if (rc != D2T_OK) return rc;
    return roipool_bwd_generic_gated<T>(gout, bins, gin, offsets + HW, cap, R, C, H, W, k, st);   // runs only if the lists overflowed
}

#define D2T_INSTANTIATE_LISTS(T)                                                  \
    template bool roipool_bwd_lists_supported<T>(int, int, int, int, int);         \
    template size_t roipool_bwd_lists_ws_bytes<T>(int, int, int, int, int);        \
    template int roipool_bwd_lists<T>(const T*, const T*, T*, void*, int, int, int, int, int, hipStream_t);
D2T_INSTANTIATE_LISTS(float)
D2T_INSTANTIATE_LISTS(double)

}  // namespace d2t
