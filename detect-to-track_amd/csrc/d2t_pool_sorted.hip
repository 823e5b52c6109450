// d2t_pool_sorted.hip -- PSROIPool backward from SORTED CORNER LISTS (gfx950, f32, k = 7).
//
// gradIn of one channel is a sum of rectangles: every bin (cell) adds v = g/n to all its pixels
// (roipool_cuda.cu:119-125, ps_roipool_cuda.cu:131-139).  A rectangle is four signed deltas in a
// 2-D difference map -- +v at (i0,j0) and (i1,j1), -v at (i0,j1) and (i1,j0) -- and the gradient map
// is the 2-D prefix sum of the difference map: 4 updates per bin instead of one per pixel.
// Scattering those updates into LDS needs atomics or arbitration (round 2 measured a tag-arbitrated
// version: 2x SLOWER than per-pixel adds, every update a chain of dependent LDS round trips).  The
// geometry, however, does not depend on the target (the nT targets of a bin share its R cells), so
// the scatter is turned into a gather ONCE per call:
//
//   1. k_*_corner_lists: the corners of a list's rectangles are SORTED by map address (stable LSD
//      radix sort in LDS, ballot ranks + scanned digit counts, no atomics), packed as {sign, value
//      index}; a segment table gives every map address the start and length of its run.
//   2. the main kernels stage the channel's values v = g/n and the list in LDS; a thread owns one
//      map ADDRESS and adds up its run (f64: the few f32 terms sum exactly) into the f64 difference
//      map.  Every address has one owner: no conflicts, no atomics, no order dependence, any number
//      of waves.  (A first version walked the list 64 entries per wave with a segmented scan over
//      the lanes: 18 ds_bpermute per step made it LDS-bound, 244 us at R=3000, nT=31.)
//   3. prefix2d (f64) turns the difference map into the gradient map; +v / -v of a rectangle are the
//      same f32 number, so they cancel exactly outside it.
// Fixed summation order throughout: bitwise reproducible.
#include "d2t_kernels.hpp"
#include "d2t_tuned.hpp"
#include "d2t_pool_common.hpp"

namespace d2t { namespace tuned {

// packed list entry: bits 0-11 map address (row * LDW + column), bit 12 sign, bits 13-31 value index
constexpr int ADDR_BITS = 12, ADDR_MAX = 1 << ADDR_BITS;
__device__ __forceinline__ unsigned pack_entry(int addr, int neg, int vidx) { return (unsigned)addr | ((unsigned)neg << ADDR_BITS) | ((unsigned)vidx << (ADDR_BITS + 1)); }

// ---------------------------------------------------------------------------------------
// Stable LSD radix sort of n packed entries by their 12-bit address, in LDS, 1024 threads: two
// passes of 6 bits.  A wave step takes 64 consecutive entries; six ballots give every lane the set
// of lanes with its digit, hence its rank inside the step and the digit's count; the counts of all
// (digit, step) pairs are scanned in that order, which gives every entry its output slot (stable:
// steps in order, lanes in order).  No atomics, no serial chunk loops.
//   a: input / output, b: scratch (n words each); hist: 64 * nsteps u16 (+ scan scratch)
// ---------------------------------------------------------------------------------------
constexpr int CL_THREADS = 1024;
constexpr int CL_WAVES = CL_THREADS / 64;
constexpr int SCAN_PER = 16;                 // counters per thread and scan block

__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned* __restrict__ part /* LDS, CL_WAVES words */, int tid, unsigned& total)
{
    const int lane = tid & 63, wave = tid >> 6;
    unsigned inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) part[wave] = inc;
    __syncthreads();
    unsigned base = 0, tot = 0;
    for (int w = 0; w < CL_WAVES; ++w) { const unsigned pw = part[w]; base += w < wave ? pw : 0; tot += pw; }
    total = tot;
    __syncthreads();
    return base + inc - v;
}

// lanes of the wave whose 6-bit digit equals this lane's
__device__ __forceinline__ unsigned long long same_digit_mask(unsigned digit, bool on)
{
    unsigned long long m = __ballot(on);
#pragma unroll
    for (int bit = 0; bit < 6; ++bit) {
        const unsigned long long bm = __ballot((digit >> bit) & 1);
        m &= ((digit >> bit) & 1) ? bm : ~bm;
    }
    return m;
}

__device__ __forceinline__ void radix_sort_addr(unsigned* __restrict__ a, unsigned* __restrict__ b, unsigned short* __restrict__ hist,
                                                unsigned* __restrict__ part, int n, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int nsteps = (n + 63) >> 6, nh = 64 * nsteps;
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned* src = a;
    unsigned* dst = b;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        const int shift = 6 * pass;
        for (int e = tid; e < nh; e += CL_THREADS) hist[e] = 0;
        __syncthreads();
        for (int s = wave; s < nsteps; s += CL_WAVES) {              // counts: hist[digit][step]
            const int e = s * 64 + lane;
            const bool on = e < n;
            const unsigned d = on ? (src[e] >> shift) & 63 : 0;
            const unsigned long long m = same_digit_mask(d, on);
            if (on && (m & lt) == 0) hist[d * nsteps + s] = (unsigned short)__builtin_popcountll(m);   // the digit's first lane
        }
        __syncthreads();
        // exclusive scan of the counters in (digit, step) order: thread t owns a contiguous run
        // (blocks of SCAN_PER counters per thread, read into registers at once: independent LDS loads)
        unsigned carry_in = 0;
        for (int base = 0; base < nh; base += SCAN_PER * CL_THREADS) {
            const int lo = base + tid * SCAN_PER;
            unsigned c[SCAN_PER], sum = 0;
#pragma unroll
            for (int k = 0; k < SCAN_PER; ++k) { c[k] = lo + k < nh ? hist[lo + k] : 0; sum += c[k]; }
            unsigned total;
            unsigned run = carry_in + block_exclusive_scan(sum, part, tid, total);
#pragma unroll
            for (int k = 0; k < SCAN_PER; ++k) {
                if (lo + k < nh) hist[lo + k] = (unsigned short)run;
                run += c[k];
            }
            carry_in += total;
        }
        __syncthreads();
        for (int s = wave; s < nsteps; s += CL_WAVES) {              // scatter
            const int e = s * 64 + lane;
            const bool on = e < n;
            const unsigned v = on ? src[e] : 0;
            const unsigned d = (v >> shift) & 63;
            const unsigned long long m = same_digit_mask(d, on);
            if (on) dst[hist[d * nsteps + s] + __builtin_popcountll(m & lt)] = v;
        }
        __syncthreads();
        unsigned* t = src; src = dst; dst = t;
    }
    // two passes: the result is back in a
}

// ---------------------------------------------------------------------------------------
// PSROIPool: one list per bin.  Rectangle r of list `bin` = cell `bin` of RoI r, value index = r.
// lists[bin][0 .. counts[bin]) sorted; nn[bin][r] = pixels of the cell (0: empty, no entries).
// ---------------------------------------------------------------------------------------
constexpr int CL_PER = 4;                     // RoIs per thread of the list kernel: R <= 4096

__global__ void __launch_bounds__(CL_THREADS)
k_ps_corner_lists(const int4* __restrict__ cellsT, unsigned short* __restrict__ lists, unsigned* __restrict__ segs,
                  int* __restrict__ nn, int R, int LDW, int cap)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    unsigned* a = reinterpret_cast<unsigned*>(lds_raw);              // [cap]
    const int words = cap > ADDR_MAX ? cap : ADDR_MAX;
    unsigned* b = a + words;                                         // [words]
    unsigned* part = b + words;                                      // [CL_WAVES]
    unsigned short* cnt = reinterpret_cast<unsigned short*>(part + CL_WAVES);   // [64][steps]
    const int bin = blockIdx.x, tid = threadIdx.x;
    const int4* cells = cellsT + (size_t)bin * R;
    // valid cells in RoI order: thread t owns RoIs [4t, 4t+4); all four cells are requested at once
    int4 c[CL_PER];
    bool ok[CL_PER];
    unsigned nvalid = 0;
#pragma unroll
    for (int k = 0; k < CL_PER; ++k) {
        const int r = CL_PER * tid + k;
        c[k] = r < R ? cells[r] : make_int4(0, 0, 0, 0);
        ok[k] = c[k].y > c[k].x && c[k].w > c[k].z;
        nvalid += ok[k];
    }
#pragma unroll
    for (int k = 0; k < CL_PER; ++k) {
        const int r = CL_PER * tid + k;
        if (r < R) nn[(size_t)bin * R + r] = ok[k] ? (c[k].y - c[k].x) * (c[k].w - c[k].z) : 0;
    }
    unsigned total;
    unsigned pos = 4 * block_exclusive_scan(nvalid, part, tid, total);
#pragma unroll
    for (int k = 0; k < CL_PER; ++k) {
        if (!ok[k]) continue;
        const int r = CL_PER * tid + k;
        a[pos + 0] = pack_entry(c[k].x * LDW + c[k].z, 0, r);
        a[pos + 1] = pack_entry(c[k].x * LDW + c[k].w, 1, r);
        a[pos + 2] = pack_entry(c[k].y * LDW + c[k].z, 1, r);
        a[pos + 3] = pack_entry(c[k].y * LDW + c[k].w, 0, r);
        pos += 4;
    }
    const int n = 4 * (int)total;
    __syncthreads();
    radix_sort_addr(a, b, cnt, part, n, tid);
    // the list the main kernel reads: 16 bits per entry = sign << 15 | value index (the address is
    // implied by the segment table)
    unsigned short* out = lists + (size_t)bin * cap;
    for (int e = tid; e < n; e += CL_THREADS) out[e] = (unsigned short)(((a[e] >> ADDR_BITS) & 1) << 15 | (a[e] >> (ADDR_BITS + 1)));
    // segment table: seg[address] = start | length << 16 (0 where no corner falls).  An entry that opens
    // a run stores its index; the length is the distance to the next opener.
    unsigned* sg = segs + (size_t)bin * ADDR_MAX;
    unsigned* heads = b;                                             // b is free now
    for (int e = tid; e < ADDR_MAX; e += CL_THREADS) heads[e] = 0xffffffffu;
    __syncthreads();
    for (int e = tid; e < n; e += CL_THREADS) {                      // the entry that opens a run stores its index ...
        const int ad = a[e] & (ADDR_MAX - 1);
        const int prev = e > 0 ? (int)(a[e - 1] & (ADDR_MAX - 1)) : -1;
        if (prev != ad) heads[ad] = (unsigned)e;
    }
    __syncthreads();
    for (int e = tid; e < n; e += CL_THREADS) {                      // ... the one that closes it adds the length (one writer per
        const int ad = a[e] & (ADDR_MAX - 1);                        // address; a run at a map border can be hundreds long: no serial walk)
        const int next = e + 1 < n ? (int)(a[e + 1] & (ADDR_MAX - 1)) : -1;
        if (next != ad) heads[ad] |= (unsigned)(e + 1 - (int)heads[ad]) << 16;
    }
    __syncthreads();
    for (int e = tid; e < ADDR_MAX; e += CL_THREADS) sg[e] = heads[e] == 0xffffffffu ? 0u : heads[e];
}

// (rows = R, cols = nT*49) -> (cols, rows) transpose of gradOut fused with the division by the cell's
// pixel count: vT[t*49+bin][r] = gout[r][t][bin] / n[bin][r] (ps_roipool_cuda.cu:135), 0 for empty cells
__global__ void __launch_bounds__(256)
k_transpose_div(const float* __restrict__ in, const int* __restrict__ nn, float* __restrict__ out, int rows, int cols,
                int tiles_c, long long ntiles)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int c0 = (int)(t % tiles_c) * 32, r0 = (int)(t / tiles_c) * 32;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + ty + 8 * k, c = c0 + tx;
            if (r < rows && c < cols) tile[ty + 8 * k][tx] = in[(size_t)r * cols + c];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + ty + 8 * k, r = r0 + tx;
            if (r < rows && c < cols) {
                const int n = nn[(size_t)(c % KK) * rows + r];
                out[(size_t)c * rows + r] = n > 0 ? tile[tx][ty + 8 * k] / (float)n : 0.f;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// PSROIPool backward, main kernel.  Workgroup = one INPUT channel, 512 threads.  The planes (t, bin)
// that map to it ((t+1)*bin == channel, ps_roipool_cuda.cu:58,128) are accumulated into the same
// difference map, one list pass each; channel 0 (bin 0 of every target: one shared cell per RoI) adds
// the nT values first.  vT: gradOut / n transposed to (t*49+bin, r).
// A list pass: thread = map address; it adds up its run of the sorted list (f32 terms in f64, in RoI
// order) and adds the sum to the f64 difference map.  All global loads of a pass (list, values, this
// thread's segment words) are issued together before the first use.
// ---------------------------------------------------------------------------------------
constexpr int PS_THREADS = 512;
constexpr int PS_SEGS = 8;                    // addresses per thread: E <= 4096

struct PsLayout { int LDW, E, off_vals, off_listL, off_plist, off_scr; size_t bytes; };
inline PsLayout ps_layout(int R, int H, int W)
{
    PsLayout L;
    L.LDW = (W + 1) | 1;
    L.E = ((H + 1) * L.LDW + 3) & ~3;
    size_t o = (size_t)L.E * 8;
    L.off_vals = (int)o; o += ((size_t)R * 4 + 15) & ~(size_t)15;
    L.off_listL = (int)o; o += ((size_t)4 * R * 2 + 15) & ~(size_t)15;
    L.off_plist = (int)o; o += 256;
    L.off_scr = (int)o; o += prefix_scratch_bytes(1, H, W);
    L.bytes = o;
    return L;
}

__global__ void __launch_bounds__(PS_THREADS)
k_psroipool_bwd_sorted(const float* __restrict__ vT, const unsigned short* __restrict__ lists, const unsigned* __restrict__ segs,
                       float* __restrict__ gin, int R, int nT, int H, int W, int cap, PsLayout L)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double* D = reinterpret_cast<double*>(lds_raw);                  // [H+1][LDW]
    float* vals = reinterpret_cast<float*>(lds_raw + L.off_vals);    // [R]
    unsigned short* listL = reinterpret_cast<unsigned short*>(lds_raw + L.off_listL);   // [4R]
    int* plist = reinterpret_cast<int*>(lds_raw + L.off_plist);
    double* scr = reinterpret_cast<double*>(lds_raw + L.off_scr);
    const int ch = blockIdx.x, tid = threadIdx.x, HW = H * W;
    const int np = ch == 0 ? 1 : ps_channel_planes(ch, nT, plist, tid);
    float* dst = gin + (size_t)ch * HW;
    if (np == 0) {                                                   // nothing maps here: gradient is zero
        for (int e = tid; e < HW; e += PS_THREADS) dst[e] = 0.f;
        return;
    }
    for (int e = tid; e < L.E; e += PS_THREADS) D[e] = 0.0;
    const int n8 = (cap * 2 + 15) >> 4;                              // the list in 16-byte pieces (entries past its end are never referenced)
    for (int p = 0; p < np; ++p) {
        const int pl = ch == 0 ? 0 : plist[p], bin = pl % KK;
        const uint4* lg = reinterpret_cast<const uint4*>(lists + (size_t)bin * cap);
        const unsigned* sgp = segs + (size_t)bin * ADDR_MAX;
        unsigned sg[PS_SEGS];
#pragma unroll
        for (int k = 0; k < PS_SEGS; ++k) sg[k] = tid + k * PS_THREADS < L.E ? sgp[tid + k * PS_THREADS] : 0u;
        if (p > 0) __syncthreads();                                  // the previous pass has read vals / listL
        for (int e = tid; e < n8; e += PS_THREADS) reinterpret_cast<uint4*>(listL)[e] = lg[e];
        for (int r = tid; r < R; r += PS_THREADS) {
            float v;
            if (ch == 0) {
                v = 0.f;
                for (int t = 0; t < nT; ++t) v += vT[(size_t)t * KK * R + r];
            } else {
                v = vT[(size_t)pl * R + r];
            }
            vals[r] = v;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PS_SEGS; ++k) {
            const int start = sg[k] & 0xffff, len = sg[k] >> 16;
            if (len == 0) continue;
            double sum = 0.0;
            for (int q = 0; q < len; ++q) {
                const unsigned w = listL[start + q];
                const float v = vals[w & 0x7fff];
                sum += w & 0x8000 ? -(double)v : (double)v;
            }
            D[tid + k * PS_THREADS] += sum;
        }
    }
    __syncthreads();
    // difference map -> gradient map: column W and row H only hold closing deltas
    prefix2d(D, scr, 1, H, W, L.LDW, H * L.LDW, tid, PS_THREADS);
    for (int e = tid; e < HW; e += PS_THREADS) {
        const int y = e / W, x = e - y * W;
        dst[e] = (float)D[y * L.LDW + x];
    }
}

size_t corner_list_lds(int cap)
{
    const size_t words = (size_t)(cap > ADDR_MAX ? cap : ADDR_MAX);   // b doubles as the heads table
    return words * 8 + CL_WAVES * 4 + (size_t)64 * ((cap + 63) / 64) * 2 + 64;
}

bool psroipool_bwd_sorted_supported(int R, int nT, int H, int W, int k)
{
    if (!(k == KT && R >= 1 && nT >= 1 && H >= 1 && W >= 1)) return false;
    const PsLayout L = ps_layout(R, H, W);
    return L.E <= ADDR_MAX && L.E <= PS_SEGS * PS_THREADS && R <= CL_PER * CL_THREADS && R < 32768 && 4 * R < 65536 &&
           L.bytes <= (size_t)LDS_MAX && corner_list_lds(4 * R) <= (size_t)LDS_MAX && 1LL * nT * KK * R < 0x7fffffffLL;
}

// workspace: cellsT | vT | lists (49 x 4R u16) | segs (49 x 4096) | nn (49 x R)
size_t psroipool_bwd_sorted_ws_bytes(int R, int nT, int H, int W, int k)
{
    if (!psroipool_bwd_sorted_supported(R, nT, H, W, k)) return 0;
    return cellsT_bytes(R) + align256((size_t)nT * KK * R * 4) + align256((size_t)KK * ((4 * R + 7) & ~7) * 2) + (size_t)KK * ADDR_MAX * 4 +
           align256((size_t)KK * R * 4);
}

int psroipool_bwd_sorted_f32(const float* gout, const float* rois, float* gin, int R, int nT, int H, int W, int,
                             void* ws, hipStream_t st)
{
    char* w = static_cast<char*>(ws);
    int4* cellsT = reinterpret_cast<int4*>(w); w += cellsT_bytes(R);
    float* vT = reinterpret_cast<float*>(w); w += align256((size_t)nT * KK * R * 4);
    unsigned short* lists = reinterpret_cast<unsigned short*>(w); w += align256((size_t)KK * ((4 * R + 7) & ~7) * 2);
    unsigned* segs = reinterpret_cast<unsigned*>(w); w += (size_t)KK * ADDR_MAX * 4;
    int* nn = reinterpret_cast<int*>(w);
    const PsLayout L = ps_layout(R, H, W);
    const int cap = (4 * R + 7) & ~7;                                // 16-byte multiple of u16 entries per list
    int rc = ps_cells_T(rois, cellsT, R, H, W, st);
    if (rc != D2T_OK) return rc;
    D2T_ENSURE_DYNAMIC_LDS(k_ps_corner_lists, LDS_MAX);
    D2T_ENSURE_DYNAMIC_LDS(k_psroipool_bwd_sorted, LDS_MAX);
    hipLaunchKernelGGL(k_ps_corner_lists, dim3(KK), dim3(CL_THREADS), corner_list_lds(cap), st, cellsT, lists, segs, nn, R, L.LDW, cap);
    rc = launch_status();
    if (rc != D2T_OK) return rc;
    {
        const int rows = R, cols = nT * KK, tiles_c = (cols + 31) / 32;
        const long long ntiles = 1LL * tiles_c * ((rows + 31) / 32);
        const int grid = (int)(ntiles < 256 * 64 ? ntiles : 256 * 64);
        hipLaunchKernelGGL(k_transpose_div, dim3(grid), dim3(256), 0, st, gout, nn, vT, rows, cols, tiles_c, ntiles);
        rc = launch_status();
        if (rc != D2T_OK) return rc;
    }
    hipLaunchKernelGGL(k_psroipool_bwd_sorted, dim3(nT * KK), dim3(PS_THREADS), L.bytes, st,
                       vT, lists, segs, gin, R, nT, H, W, cap, L);
    return launch_status();
}

}}  // namespace d2t::tuned
