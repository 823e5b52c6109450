// d2t_pool_lists.hip -- ROIPool backward OUTSIDE the tuned envelope (any bin count k, f32 and f64): the same sums in the
// same order as the thread-per-pixel kernel of d2t_generic.hip (k_roipool_bwd_generic: ascending (r, i, j), each term
// gradOut / n, roipool_cuda.cu:112-124 in gather form), so the result is bit-identical to it -- without its redundancy.
// Which bins contain a pixel does not depend on the channel; the thread-per-(c, y, x) kernel re-derives it per channel
// (C x H x W x R bin tests: 2.3 ms at R = 300, C = 1024, 38 x 63).  Here:
//
//   1. k_roi_row_build / k_roi_col_masks   membership is separable (row bins of r that contain y times column bins that contain x):
//      per map ROW the ascending list of the (RoI, bin row) pairs that contain it (workgroup = row, thread = RoI, block scans, ONE
//      atomic add on a counter for the workgroup's run of the entry array -- where a list lies is arbitrary, what it holds and in
//      which order is not), and per (RoI, column) a bit mask of the bin columns that contain it;
//   2. k_grad_by_bin   Q[(r, i, j)][c] = gradOut[r][c][i][j] / n(r, i, j): the division of :123 done once per gradOut element
//      instead of once per pixel of its bin, and the channel made the contiguous index;
//   3. k_roipool_bwd_rows   workgroup = pixel, thread = channel: the pixel's bins (r, i, j), ascending, come out of its row's list and
//      the column masks 256 entries at a time (into LDS); gradIn[c][y][x] = sum over them of Q[bin][c] -- the entry is uniform, the
//      loads are coalesced along c.  (First version: a list per PIXEL, R x H x W RoI tests to build -- 62 us of 166 at R = 300.)
//
// The lists live in the caller's workspace, sized for RoIs no larger than the map (R (H + 2k) entries; k <= 32).  RoIs far larger
// than the map can exceed that (every bin of such a RoI covers the whole map): the counter then ends above the capacity, steps
// 2-3 do nothing and the thread-per-pixel kernel, launched last and otherwise returning at once, does the work -- decided on
// the device, no host synchronisation.
//
// PSROIPool backward the same way (k_ps_row_build, k_psroipool_bwd_rows): there a list belongs to a (cell, map row) pair -- the RoIs
// whose cell (i, j) reaches that row, ascending, with the cell's column range -- and serves every pixel of the row and every channel
// (t + 1) * bin that reads the cell (ps_roipool_cuda.cu:58); k_psroipool_bwd_generic tested all R RoIs per (channel, pixel) and
// divisor of the channel.
#include "d2t_kernels.hpp"

namespace d2t {

namespace {

constexpr int kBlk = 256;
constexpr int kAhead = 8;                                              // list entries whose loads are in flight together (gather)

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

// exclusive scan of one int per thread over the workgroup; returns the thread's prefix, *total = the sum (sh: 4 ints)
__device__ __forceinline__ int block_exclusive_scan(int v, int* sh, int* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    if (lane == 63) sh[wave] = incl;
    __syncthreads();
    const int w0 = sh[0], w1 = sh[1], w2 = sh[2], w3 = sh[3];
    __syncthreads();
    *total = w0 + w1 + w2 + w3;
    return incl - v + (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0);
}

typedef unsigned long long u64;

// the workgroup's `total` entries: one atomic add, result broadcast (sh: 4 ints, reused)
__device__ __forceinline__ long long block_take(u64* counter, int total, int* sh)
{
    __shared__ long long start;
    if (threadIdx.x == 0) start = (long long)atomicAdd(counter, (u64)total);
    __syncthreads();
    const long long v = start;
    __syncthreads();
    return v;
}

// ROIPool: one list per map ROW -- the (RoI, bin row) pairs r * k + i whose bin row contains the row, ascending.  Workgroup = row,
// thread = RoI, ordered compaction by block scans, one atomic add for the workgroup's run of the entry array.
__global__ void __launch_bounds__(kBlk)
k_roi_row_build(const int32_t* __restrict__ bins, u64* __restrict__ counter, int2* __restrict__ heads, int32_t* __restrict__ entries,
                long long cap, int R, int k)
{
    __shared__ int sh[4];
    const int4* bt = reinterpret_cast<const int4*>(bins);
    const int y = blockIdx.x;
    int n = 0;
    for (int r = threadIdx.x; r < R; r += kBlk) n += axis_hits(bt + (size_t)r * k * k, k, k, y, true);
    int total;
    block_exclusive_scan(n, sh, &total);
    const long long start = block_take(counter, total, sh);
    const bool fits = start + total <= cap;
    if (threadIdx.x == 0) heads[y] = fits ? make_int2((int)start, total) : make_int2(0, 0);
    if (!fits || total == 0) return;                                  // (no list: the thread-per-pixel kernel runs instead)
    int base = (int)start;
    for (int r0 = 0; r0 < R; r0 += kBlk) {                            // RoIs ascending: 256 at a time, in thread order
        const int r = r0 + threadIdx.x;
        const int4* br = bt + (size_t)(r < R ? r : 0) * k * k;
        const int ni = r < R ? axis_hits(br, k, k, y, true) : 0;
        int pass;
        int pos = base + block_exclusive_scan(ni, sh, &pass);
        if (ni)
            for (int i = 0; i < k; ++i) {                             // bin rows ascending
                const int4 bi = br[i * k];
                if (y >= bi.x && y < bi.y) entries[pos++] = r * k + i;
            }
        base += pass;
    }
}

// colmask[r][x]: bit j set iff column x lies in bin column j of RoI r (k <= 32; column bounds depend on (r, j) only, roipool_cuda.cu:46-50)
__global__ void __launch_bounds__(kBlk)
k_roi_col_masks(const int32_t* __restrict__ bins, uint32_t* __restrict__ colmask, int R, int W, int k)
{
    const int4* bt = reinterpret_cast<const int4*>(bins);
    const int total = R * W;
    for (int id = blockIdx.x * kBlk + threadIdx.x; id < total; id += gridDim.x * kBlk) {
        const int r = id / W, x = id - r * W;
        const int4* br = bt + (size_t)r * k * k;
        uint32_t m = 0;
        for (int j = 0; j < k; ++j) {
            const int4 b = br[j];
            m |= (x >= b.z && x < b.w) ? 1u << j : 0u;
        }
        colmask[id] = m;
    }
}

// PSROIPool: one list per (cell, map ROW) -- the RoIs (ascending) whose cell's rows contain that row and whose column range is not
// empty, each with the cell's area and column range.  Workgroup = (cell, row), thread = RoI, ordered compaction by block scans; the
// workgroup takes its run of the entry array with one atomic add.  (Round 4 first kept a list per (cell, PIXEL): R k^2 H W cell tests
// to build, and a gather in which every lane walks its own list -- one cache line per lane and term: 130 us at R = 300, k = 6.)
__global__ void __launch_bounds__(kBlk)
k_ps_row_build(const int32_t* __restrict__ cells, u64* __restrict__ counter, int2* __restrict__ heads, int4* __restrict__ entries,
               long long cap, int R, int H, int kk)
{
    __shared__ int sh[4];
    const int4* ct = reinterpret_cast<const int4*>(cells);
    const int list = blockIdx.x, bin = list / H, y = list - bin * H;
    int n = 0;
    for (int r = threadIdx.x; r < R; r += kBlk) {
        const int4 cb = ct[(size_t)r * kk + bin];
        n += (y >= cb.x && y < cb.y && cb.w > cb.z);
    }
    int total;
    block_exclusive_scan(n, sh, &total);
    const long long start = block_take(counter, total, sh);
    const bool fits = start + total <= cap;
    if (threadIdx.x == 0) heads[list] = fits ? make_int2((int)start, total) : make_int2(0, 0);
    if (!fits || total == 0) return;
    int base = (int)start;
    for (int r0 = 0; r0 < R; r0 += kBlk) {                            // RoIs ascending: 256 at a time, in thread order
        const int r = r0 + threadIdx.x;
        const int4 cb = ct[(size_t)(r < R ? r : 0) * kk + bin];
        const int hit = r < R && y >= cb.x && y < cb.y && cb.w > cb.z;
        int pass;
        const int pos = base + block_exclusive_scan(hit, sh, &pass);
        if (hit) entries[pos] = make_int4(r, (cb.y - cb.x) * (cb.w - cb.z), cb.z, cb.w);
        base += pass;
    }
}

__device__ __forceinline__ float lane_value(float v, int u) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), u)); }
__device__ __forceinline__ double lane_value(double v, int u)
{
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffLL), u), hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), u);
    return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}

// gradIn[ch][y][x0 .. x0 + 63]: one WAVE per (channel, row, 64 columns), a lane per column.  Channel ch receives from every (t, bin) with
// (t + 1) * bin == ch (ch == 0: bin 0 with every t); bins ascending, then the row's list (RoIs ascending), then t -- the order of
// k_psroipool_bwd_generic; each term gradOut / n (ps_roipool_cuda.cu:118-127 in gather form).  The list is the same for the whole wave:
// 64 entries are loaded at once (lane = entry; its term gradOut / n is computed there) and handed round with v_readlane; a lane adds
// the terms whose column range contains its column.  kMaxT: targets of channel 0 held per entry (more: a scalar walk).
constexpr int kMaxT = 32;

template <typename T>
__global__ void __launch_bounds__(64)
k_psroipool_bwd_rows(const T* __restrict__ gout, const u64* __restrict__ counter, const int2* __restrict__ heads,
                     const int4* __restrict__ entries, long long cap, T* __restrict__ gin, int nT, int H, int W, int kk, int segs)
{
    if ((long long)*counter > cap) return;
    int wid = blockIdx.x;
    const int seg = wid % segs; wid /= segs;
    const int y = wid % H, ch = wid / H;
    const int lane = threadIdx.x, x = seg * 64 + lane;
    T acc = T(0);
    // bins that divide ch with ch / bin <= nT: only bins in [ceil(ch / nT), min(ch, k^2 - 1)] can
    const int bin_lo = ch == 0 ? 0 : (ch + nT - 1) / nT, bin_hi = ch == 0 ? 0 : (ch < kk - 1 ? ch : kk - 1);
    for (int bin = bin_lo; bin <= bin_hi; ++bin) {
        int t_lo, t_hi;
        if (ch == 0) { t_lo = 0; t_hi = nT - 1; }
        else {
            const int tp1 = ch / bin;
            if (tp1 * bin != ch) continue;
            t_lo = t_hi = tp1 - 1;
        }
        const int2 h = heads[(size_t)bin * H + y];
        const int end = h.x + h.y;
        for (int e0 = h.x; e0 < end; e0 += 64) {
            const int e = e0 + lane, cnt = end - e0 < 64 ? end - e0 : 64;
            const int4 ent = e < end ? entries[e] : make_int4(0, 1, 0, 0);
            const T dn = static_cast<T>(ent.y);
            if (t_lo == t_hi) {
                const T v = e < end ? gout[((size_t)ent.x * nT + t_lo) * kk + bin] / dn : T(0);
                for (int u = 0; u < cnt; ++u) {                       // list order
                    const int x0 = __builtin_amdgcn_readlane(ent.z, u), x1 = __builtin_amdgcn_readlane(ent.w, u);
                    const T vv = lane_value(v, u);
                    if (x >= x0 && x < x1) acc += vv;
                }
            } else if (nT <= kMaxT) {                                 // channel 0: every target of every listed RoI, RoI-major
                T v[kMaxT];
                const T* gp = gout + (size_t)ent.x * nT * kk + bin;
#pragma unroll
                for (int t = 0; t < kMaxT; ++t) v[t] = (e < end && t < nT) ? gp[(size_t)t * kk] / dn : T(0);
                for (int u = 0; u < cnt; ++u) {
                    const int x0 = __builtin_amdgcn_readlane(ent.z, u), x1 = __builtin_amdgcn_readlane(ent.w, u);
                    const bool hit = x >= x0 && x < x1;
#pragma unroll
                    for (int t = 0; t < kMaxT; ++t)
                        if (t < nT) {                                 // (uniform)
                            const T vv = lane_value(v[t], u);
                            if (hit) acc += vv;
                        }
                }
            } else {
                for (int u = 0; u < cnt; ++u) {
                    const int r = __builtin_amdgcn_readlane(ent.x, u), nn = __builtin_amdgcn_readlane(ent.y, u);
                    const int x0 = __builtin_amdgcn_readlane(ent.z, u), x1 = __builtin_amdgcn_readlane(ent.w, u);
                    for (int t = t_lo; t <= t_hi; ++t) {
                        const T vv = gout[((size_t)r * nT + t) * kk + bin] / static_cast<T>(nn);
                        if (x >= x0 && x < x1) acc += vv;
                    }
                }
            }
        }
    }
    if (x < W) gin[((size_t)ch * H + y) * W + x] = acc;
}

// Q[(r, bin)][c] = gradOut[r][c][bin] / n(r, bin).  Workgroup = (r, chunk of CH channels): CH * k^2 contiguous elements in,
// through LDS, rows of CH contiguous channels out.
template <typename T>
__global__ void __launch_bounds__(kBlk)
k_grad_by_bin(const T* __restrict__ gout, const int32_t* __restrict__ bins, const u64* __restrict__ counter, long long cap,
              T* __restrict__ q, int R, int C, int kk, int k, int CH)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if ((long long)*counter > cap) return;
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

// gradIn[c][y][x] = sum, ascending (r, i, j), of Q[(r, i, j)][c] over the bins that contain the pixel.  Workgroup = pixel; a thread owns
// the channels tid, tid + 256, tid + 512, tid + 768 of each group of 1024.  The pixel's bins come out of its ROW's list 256 entries at
// a time (thread = entry: the RoI's column mask at x says which bin columns j contain the pixel; a block scan orders the hits
// (r, i, j) into an LDS buffer), then every thread walks the buffer -- uniform entries, loads coalesced along c, kAhead in flight.
template <typename T>
__global__ void __launch_bounds__(kBlk)
k_roipool_bwd_rows(const T* __restrict__ q, const u64* __restrict__ counter, const int2* __restrict__ heads,
                   const int32_t* __restrict__ entries, const uint32_t* __restrict__ colmask, long long cap,
                   T* __restrict__ gin, int C, int H, int W, int k)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    __shared__ int sh[4];
    if ((long long)*counter > cap) return;
    int* hits = reinterpret_cast<int*>(lds_raw);                      // [kBlk * k]: Q row indices (r * k + i) * k + j of one chunk
    const int HW = H * W;
    const int p = xcd_run(blockIdx.x, HW), y = p / W, x = p - y * W;  // neighbouring pixels (nearly the same bins) share an L2
    const int2 h = heads[y];
    const int beg = h.x, end = h.x + h.y;
    for (int c0 = 0; c0 < C; c0 += 4 * kBlk) {
        const int c = c0 + threadIdx.x;
        T acc[4] = {T(0), T(0), T(0), T(0)};
        bool has[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) has[m] = c + m * kBlk < C;
        for (int e0 = beg; e0 < end; e0 += kBlk) {
            const int e = e0 + threadIdx.x;
            const int ent = e < end ? entries[e] : 0;
            uint32_t m = e < end ? colmask[(size_t)(ent / k) * W + x] : 0u;
            int total;
            int pos = block_exclusive_scan(__builtin_popcount(m), sh, &total);
            while (m) {                                               // bin columns ascending
                const int j = __builtin_ctz(m);
                m &= m - 1;
                hits[pos++] = ent * k + j;
            }
            __syncthreads();
            for (int t0 = 0; t0 < total; t0 += kAhead) {              // kAhead bins' loads in flight, added in (r, i, j) order
                T v[kAhead][4];
#pragma unroll
                for (int u = 0; u < kAhead; ++u) {
                    const T* row = q + (size_t)hits[t0 + u < total ? t0 + u : t0] * C + c;
#pragma unroll
                    for (int mm = 0; mm < 4; ++mm) v[u][mm] = has[mm] ? row[mm * kBlk] : T(0);
                }
#pragma unroll
                for (int u = 0; u < kAhead; ++u)
                    if (t0 + u < total) {
#pragma unroll
                        for (int mm = 0; mm < 4; ++mm) acc[mm] += v[u][mm];
                    }
            }
            __syncthreads();                                          // the buffer is free again
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (c + m * kBlk < C) gin[(size_t)(c + m * kBlk) * HW + p] = acc[m];
    }
}

inline size_t up256(size_t b) { return (b + 255) / 256 * 256; }

struct ListsLayout {
    size_t bins, counter, heads, entries, masks, q, total;
    long long cap;
};

template <typename T>
ListsLayout lists_layout(int R, int C, int H, int W, int k)          // ROIPool: bins | counter | heads[H] | entries | column masks | Q
{
    ListsLayout L;
    L.cap = 1LL * R * (H + 2LL * k);                                  // a RoI's bin rows cover <= H + 2k map rows between them
    L.bins = 0;
    L.counter = up256((size_t)R * k * k * 16);
    L.heads = L.counter + 256;
    L.entries = L.heads + up256((size_t)H * 8);
    L.masks = L.entries + up256((size_t)L.cap * 4);
    L.q = L.masks + up256((size_t)R * W * 4);
    L.total = L.q + up256((size_t)R * C * k * k * sizeof(T));
    return L;
}

inline ListsLayout ps_lists_layout(int R, int H, int W, int k)       // PSROIPool: cells | counter | heads[kk H] | entries (r, n, j0, j1)
{
    ListsLayout L;
    L.cap = 1LL * R * k * (H + 2LL * k);                              // per RoI and cell column the cell rows cover <= H + 2k map rows
    L.bins = 0;
    L.counter = up256((size_t)R * k * k * 16);
    L.heads = L.counter + 256;
    L.entries = L.heads + up256((size_t)k * k * H * 8);
    L.q = L.total = L.entries + up256((size_t)L.cap * 16);
    return L;
}

}  // namespace

template <typename T>
bool roipool_bwd_lists_supported(int R, int C, int H, int W, int k)
{
    if (R < 1 || C < 1 || H < 1 || W < 1 || k < 1) return false;
    return k <= 32 && 1LL * R * (H + 2LL * k) < 0x4000000LL && 1LL * R * k * k < 0x7fffffffLL / 4 && fits_i32(1LL * R * C * k * k) &&
           fits_i32(1LL * C * H * W) && fits_i32(1LL * R * W);
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
    u64* counter = reinterpret_cast<u64*>(base + L.counter);
    int2* heads = reinterpret_cast<int2*>(base + L.heads);
    int32_t* entries = reinterpret_cast<int32_t*>(base + L.entries);
    uint32_t* masks = reinterpret_cast<uint32_t*>(base + L.masks);
    T* q = reinterpret_cast<T*>(base + L.q);
    const int HW = H * W, kk = k * k;
    int rc = roipool_bins<T>(rois, bins, R, H, W, k, st);
    if (rc != D2T_OK) return rc;
    if (hipMemsetAsync(counter, 0, sizeof(u64), st) != hipSuccess) return launch_status();
    hipLaunchKernelGGL(k_roi_row_build, dim3(H), dim3(kBlk), 0, st, bins, counter, heads, entries, L.cap, R, k);
    hipLaunchKernelGGL(k_roi_col_masks, dim3(grid_for(1LL * R * W, kBlk)), dim3(kBlk), 0, st, bins, masks, R, W, k);
    int CH = 64;
    while (CH > 1 && (size_t)CH * (kk + 1) * sizeof(T) > 32 * 1024) CH >>= 1;
    hipLaunchKernelGGL(k_grad_by_bin<T>, dim3(R * ((C + CH - 1) / CH)), dim3(kBlk), (size_t)CH * (kk + 1) * sizeof(T), st,
                       gout, bins, counter, L.cap, q, R, C, kk, k, CH);
    hipLaunchKernelGGL(k_roipool_bwd_rows<T>, dim3(HW), dim3(kBlk), (size_t)kBlk * k * sizeof(int), st, q, counter, heads, entries, masks, L.cap,
                       gin, C, H, W, k);
    rc = launch_status();
    if (rc != D2T_OK) return rc;
    return roipool_bwd_generic_gated<T>(gout, bins, gin, counter, L.cap, R, C, H, W, k, st);   // runs only if the lists overflowed
}

bool psroipool_bwd_lists_supported(int R, int nT, int H, int W, int k)
{
    if (R < 1 || nT < 1 || H < 1 || W < 1 || k < 1) return false;
    return 1LL * R * k * (H + 2LL * k) < 0x4000000LL && 1LL * R * k * k < 0x7fffffffLL / 4 && fits_i32(1LL * R * nT * k * k) &&
           fits_i32(1LL * nT * k * k * H * W) && 1LL * k * k * H < 0x7fffffffLL && 1LL * nT * k * k * H * ((W + 63) / 64) < 0x7fffffffLL;
}

size_t psroipool_bwd_lists_ws_bytes(int R, int nT, int H, int W, int k)
{
    return psroipool_bwd_lists_supported(R, nT, H, W, k) ? ps_lists_layout(R, H, W, k).total : 0;
}

template <typename T>
int psroipool_bwd_lists(const T* gout, const T* rois, T* gin, void* ws, int R, int nT, int H, int W, int k, hipStream_t st)
{
    const ListsLayout L = ps_lists_layout(R, H, W, k);
    unsigned char* base = static_cast<unsigned char*>(ws);
    int32_t* cells = reinterpret_cast<int32_t*>(base + L.bins);
    u64* counter = reinterpret_cast<u64*>(base + L.counter);
    int2* heads = reinterpret_cast<int2*>(base + L.heads);
    int4* entries = reinterpret_cast<int4*>(base + L.entries);
    const int kk = k * k, segs = (W + 63) / 64;
    int rc = psroipool_bins<T>(rois, cells, R, H, W, k, st);
    if (rc != D2T_OK) return rc;
    if (hipMemsetAsync(counter, 0, sizeof(u64), st) != hipSuccess) return launch_status();
    hipLaunchKernelGGL(k_ps_row_build, dim3(kk * H), dim3(kBlk), 0, st, cells, counter, heads, entries, L.cap, R, H, kk);
    hipLaunchKernelGGL(k_psroipool_bwd_rows<T>, dim3(nT * kk * H * segs), dim3(64), 0, st,
                       gout, counter, heads, entries, L.cap, gin, nT, H, W, kk, segs);
    rc = launch_status();
    if (rc != D2T_OK) return rc;
    return psroipool_bwd_generic_gated<T>(gout, cells, gin, counter, L.cap, R, nT, H, W, k, st);
}

#define D2T_INSTANTIATE_LISTS(T)                                                  \
    template bool roipool_bwd_lists_supported<T>(int, int, int, int, int);         \
    template size_t roipool_bwd_lists_ws_bytes<T>(int, int, int, int, int);        \
    template int roipool_bwd_lists<T>(const T*, const T*, T*, void*, int, int, int, int, int, hipStream_t); \
    template int psroipool_bwd_lists<T>(const T*, const T*, T*, void*, int, int, int, int, int, hipStream_t);
D2T_INSTANTIATE_LISTS(float)
D2T_INSTANTIATE_LISTS(double)

}  // namespace d2t
