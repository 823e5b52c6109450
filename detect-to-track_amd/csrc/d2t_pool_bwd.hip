// d2t_pool_bwd.hip -- gfx950-tuned f32 ROIPool / PSROIPool BACKWARD kernels (k = 7), and the
// thread-per-output PSROIPool forward that serves small problems.
//
//  * ROIPool backward, round 2: a GEMM with an indicator matrix on the f32 matrix pipe (k_roi_rowlists +
//    k_roipool_bwd_gemm below; maps up to 128 columns).  The per-pixel kernel it replaces stays for wider maps:
//  * ROIPool backward (gather form): gradOut is transposed once to (R, 49, C); a workgroup owns
//    (64 channels, one map row), each of its waves walks a share of the RoIs -- work items come
//    from a row-mask table, 64 RoIs per instruction, several in flight -- and adds gradOut/n into
//    an LDS accumulator [W+1][65] that it alone touches (batched read-add-write, one channel per
//    lane); the partial rows are added in a fixed order.
//  * PSROIPool backward, phase 1: a workgroup owns one OUTPUT plane (t, bin) -- every plane has
//    exactly R cells, so the grid is balanced whatever the many-to-one channel map
//    ((t+1)*bin, ps_roipool_cuda.cu:58) does.  Its 4 waves own a quarter of the RoIs each and a
//    private copy of the plane in LDS (k_psroipool_bwd_plane_lds; maps above 4096 pixels: bands of
//    rows in registers, k_psroipool_bwd_plane).  64 RoIs are fetched per instruction (lane = RoI).
//    Phase 2 adds, per input channel, the planes that map to it in ascending (bin, t) order.
//  No accumulator is shared between waves, every gradIn element is written once, fixed summation
//  order: bitwise reproducible.
//
// Round 2 tried to replace both with 2-D DIFFERENCE PLANES (g/n added at the four corners of a bin's
// rectangle, then a 2-D prefix sum: 4 updates per bin instead of one per pixel; lanes whose corners
// coincide serialised through a tag plane, no atomics).  Correct and deterministic, but it needs f64
// planes (the +v / -v entries cancel only in the prefix sums; f32 planes missed the 1e-5 bar on small
// maps with many RoIs), which leaves room for 4-6 wave planes per CU, and every update is a chain of
// dependent LDS round trips (tag store, tag load, plane load, plane store): 346 us for ROIPool
// config 3 against 176 us here, 367 us against 278 us for PSROIPool R=3000 nT=31.  DESIGN.md keeps
// the numbers.
#include "d2t_kernels.hpp"
#include "d2t_tuned.hpp"
#include <type_traits>
#include "d2t_pool_common.hpp"
#include <cstdlib>
#include <cstring>

namespace d2t { namespace tuned {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access, dword aligned
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

D2T_KSTAMP_DEFINE(d2t_lab_pool_bwd_stamps)

inline size_t bins_bytes(int R) { return align256((size_t)R * KK * 4 * sizeof(int32_t)); }

// ---------------------------------------------------------------------------------------
// Per-RoI geometry record: 32 int32 = {i0[7], i1[7], j0[7], j1[7], top, bottom, left, right}.
// Row bounds of a bin depend on i only, column bounds on j only (roipool_cuda.cu:41-50), so 28
// numbers describe all 49 bins; one record is two s_load_dwordx16 for a wave.
// ---------------------------------------------------------------------------------------
constexpr int GEO = 32;
inline size_t geo_bytes(int R) { return align256((size_t)R * GEO * sizeof(int32_t)); }

__global__ void __launch_bounds__(64)
k_roi_geom(const float* __restrict__ rois, int32_t* __restrict__ geo, uint8_t* __restrict__ rowmask,
           float* __restrict__ rcp, int R, int H, int W)
{
    // one wave per RoI: lane i < 7 evaluates bin row i / bin column i, the tables are filled in parallel
    const int r = blockIdx.x, lane = threadIdx.x;
    int32_t* g = geo + (size_t)r * GEO;
    Bounds b{0, 0, 0, 0};
    if (lane < KT) b = roi_bin<float>(rois + 4 * r, lane, lane, H, W, KT);   // (i, i): row bounds of i, column bounds of j = i
    int i0[KT], i1[KT], j0[KT], j1[KT];
#pragma unroll
    for (int q = 0; q < KT; ++q) {
        i0[q] = __builtin_amdgcn_readlane(b.i0, q); i1[q] = __builtin_amdgcn_readlane(b.i1, q);
        j0[q] = __builtin_amdgcn_readlane(b.j0, q); j1[q] = __builtin_amdgcn_readlane(b.j1, q);
    }
    if (lane < KT) { g[lane] = b.i0; g[KT + lane] = b.i1; g[2 * KT + lane] = b.j0; g[3 * KT + lane] = b.j1; }
    if (lane == 0) { g[28] = i0[0]; g[29] = i1[KT - 1]; g[30] = j0[0]; g[31] = j1[KT - 1]; }
    if (rowmask) {                                                   // bit i of rowmask[r][y]: bin row i contains map row y
        for (int y = lane; y < H; y += 64) {
            int mk = 0;
#pragma unroll
            for (int i = 0; i < KT; ++i) mk |= (y >= i0[i] && y < i1[i]) ? 1 << i : 0;
            rowmask[(size_t)r * H + y] = (uint8_t)mk;
        }
    }
    if (rcp && lane < KK) {                                          // 1 / binNumel of the 49 bins (0 for empty bins)
        int n = 0;
#pragma unroll
        for (int i = 0; i < KT; ++i)
#pragma unroll
            for (int j = 0; j < KT; ++j) n = lane == i * KT + j ? (i1[i] - i0[i]) * (j1[j] - j0[j]) : n;
        rcp[(size_t)r * 64 + lane] = n > 0 ? 1.0f / static_cast<float>(n) : 0.f;
    }
}

static int roi_geom(const float* rois, int32_t* geo, uint8_t* rowmask, float* rcp, int R, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(k_roi_geom, dim3(R), dim3(64), 0, st, rois, geo, rowmask, rcp, R, H, W);
    return launch_status();
}

constexpr int RB_LD = 65;                                            // accumulator row stride (floats): lanes along x read it conflict-free


// batched (C x 49) -> (49 x C) transpose: block = one RoI x 64 channels
__global__ void __launch_bounds__(256)
k_transpose_gout(const float* __restrict__ in, float* __restrict__ out, int C)
{
    __shared__ float t[64 * KK + 8];
    const int r = blockIdx.x, c0 = blockIdx.y * 64;
    const int nch = C - c0 < 64 ? C - c0 : 64;
    const float* src = in + ((size_t)r * C + c0) * KK;
    for (int e = threadIdx.x; e < nch * KK; e += 256) t[e] = src[e];
    __syncthreads();
    float* dst = out + (size_t)r * KK * C + c0;
    for (int e = threadIdx.x; e < KK * 64; e += 256) {
        const int b = e >> 6, ch = e & 63;
        if (ch < nch) dst[(size_t)b * C + ch] = t[ch * KK + b];
    }
}

// ---------------------------------------------------------------------------------------
// ROIPool backward, channel-last gather.  gt: gradOut transposed to (R, 49, C).
// Workgroup = (map row y, 64 channels), one channel per lane; wave w walks RoIs [w*R/NW,
// (w+1)*R/NW) and adds into its own LDS accumulator [W+1][65] (row W is a dummy); the partial rows
// are added in a fixed order and stored as 64 rows of gradIn.  A first version walked the RoIs with
// scalar loads and added pixel by pixel (269 us at config 3: a chain of latencies); this one:
//  * work items (RoI r, bin row i containing y) come out of a row-mask table, 64 RoIs per
//    instruction (lane = RoI): no scalar load per RoI;
//  * RC_PF items are in flight: geometry record, 7 reciprocals and the 7 x 256 bytes of
//    (transposed) gradOut are fetched RC_PF items ahead of their use;
//  * 1/binNumel comes from a table written by k_roi_geom (one multiply instead of a divide per
//    bin: <= 1 ulp from gradOut/n, the reference's atomics leave the order undefined anyway);
//  * the pixels of the even column bins (disjoint when the RoI is >= 7 pixels wide) are read
//    together, added, written together, then the odd bins: 2 LDS round trips per item instead of
//    one per pixel (27 on average).  Slots past a bin's width go to the dummy row.
// Per pixel the order is ascending (r, i, j) within a wave; the waves' rows are added in order.
// ---------------------------------------------------------------------------------------
constexpr int RC_PF = 4;                                             // work items in flight per wave

__global__ void __launch_bounds__(256)
k_roipool_bwd_batched(const float* __restrict__ gt, const int32_t* __restrict__ geo, const uint8_t* __restrict__ rowmask,
                      const float* __restrict__ rcp, float* __restrict__ gin, int R, int C, int H, int W)
{
    extern __shared__ float lds[];                                   // [waves][W+1][65]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int y = blockIdx.x, c0 = blockIdx.y * 64;
    const int per = (W + 1) * RB_LD;
    float* acc = lds + (size_t)wave * per + lane;
    for (int x = 0; x <= W; ++x) acc[x * RB_LD] = 0.f;
    const int cl = c0 + lane < C ? c0 + lane : C - 1;                // clamped lane channel (never stored)
    const int r_lo = (int)((long long)R * wave / nw), r_hi = (int)((long long)R * (wave + 1) / nw);

    // work-item generator; state is wave-uniform
    int rb = r_lo - 64, cur = 0, mk = 0, bits = 0;
    unsigned long long m = 0;
    int nr = 0, ni = 0;
    auto advance = [&]() -> bool {
        for (;;) {
            if (bits) {
                ni = __builtin_ctz(bits);
                bits &= bits - 1;
                nr = rb + cur;
                return true;
            }
            if (!m) {
                if (rb + 64 >= r_hi) return false;
                rb += 64;
                mk = rb + lane < r_hi ? rowmask[(size_t)(rb + lane) * H + y] : 0;
                m = __ballot(mk != 0);
                continue;
            }
            cur = __builtin_ctzll(m);
            m &= m - 1;
            bits = __builtin_amdgcn_readlane(mk, cur);
        }
    };
    struct Item { int rec; float rq; float v[KT]; };
    auto fetch = [&](Item& it, int r, int i) {
        it.rec = geo[(size_t)r * GEO + (lane & (GEO - 1))];          // the record, one int per lane
        it.rq = rcp[(size_t)r * 64 + i * KT + (lane < KT ? lane : 0)];   // lanes 0..6: 1/n of the bin row's bins
        const float* gr = gt + ((size_t)r * KK + i * KT) * C + cl;
#pragma unroll
        for (int j = 0; j < KT; ++j) it.v[j] = gr[(size_t)j * C];    // 7 coalesced loads
    };
    auto add_item = [&](const Item& it) {
        int j0[KT], w[KT];
        float q[KT];
        int wmax = 0;
        bool regular = true;
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            j0[j] = __builtin_amdgcn_readlane(it.rec, 2 * KT + j);
            w[j] = __builtin_amdgcn_readlane(it.rec, 3 * KT + j) - j0[j];
            wmax = w[j] > wmax ? w[j] : wmax;
            q[j] = it.v[j] * __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, it.rq), j));
        }
#pragma unroll
        for (int j = 0; j + 2 < KT; ++j) regular = regular && j0[j + 2] >= j0[j] + w[j];
        auto batch = [&](auto sw_c, auto first_c) {                  // bins first, first+2, ..: SW slots each
            constexpr int SW = decltype(sw_c)::value, FIRST = decltype(first_c)::value;
            constexpr int NB = (KT - FIRST + 1) / 2;
            float t[NB][SW];
            int xo[NB][SW];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int s2 = 0; s2 < SW; ++s2) {
                    const int j = FIRST + 2 * b;
                    xo[b][s2] = (s2 < w[j] ? j0[j] + s2 : W) * RB_LD;   // wave-uniform; past the bin: dummy row
                    t[b][s2] = acc[xo[b][s2]];
                }
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int s2 = 0; s2 < SW; ++s2) acc[xo[b][s2]] = t[b][s2] + q[FIRST + 2 * b];
        };
        using std::integral_constant;
        if (regular && wmax <= 4) {
            batch(integral_constant<int, 4>{}, integral_constant<int, 0>{});
            batch(integral_constant<int, 4>{}, integral_constant<int, 1>{});
        } else if (regular && wmax <= 8) {
            batch(integral_constant<int, 8>{}, integral_constant<int, 0>{});
            batch(integral_constant<int, 8>{}, integral_constant<int, 1>{});
        } else {                                                     // tiny or very wide RoIs: pixel by pixel
#pragma unroll 1
            for (int j = 0; j < KT; ++j)
#pragma unroll 1
                for (int x = j0[j]; x < j0[j] + w[j]; ++x) acc[x * RB_LD] += q[j];
        }
    };
    Item it[RC_PF];
    bool ok[RC_PF];
#pragma unroll
    for (int s = 0; s < RC_PF; ++s) {
        ok[s] = advance();
        if (ok[s]) fetch(it[s], nr, ni);
    }
    while (ok[0]) {                                                  // slots are consumed round-robin = in item order
#pragma unroll
        for (int s = 0; s < RC_PF; ++s) {
            if (ok[s]) {
                add_item(it[s]);
                ok[s] = advance();
                if (ok[s]) fetch(it[s], nr, ni);
            }
        }
    }
    __syncthreads();
    // gin[c0+ch][y][0..W): lanes along x, fixed-order sum of the waves' partial rows
    const int nch = C - c0 < 64 ? C - c0 : 64;
    for (int e = threadIdx.x; e < nch * W; e += blockDim.x) {
        const int ch = e / W, x = e - ch * W;
        const int o = x * RB_LD + ch;
        float a = lds[o];
        for (int wv = 1; wv < nw; ++wv) a += lds[(size_t)wv * per + o];
        gin[((size_t)(c0 + ch) * H + y) * W + x] = a;
    }
}

// ---------------------------------------------------------------------------------------
// ROIPool backward as a GEMM with an indicator matrix (round 2).
//
// For one map row y the gradient is   gradIn[c][y][x] = sum over slots s = (r, i, j) with y in bin row i
//                                                        of  (gradOut[r][c][i][j] / n_ij)  *  [x in bin column j],
// i.e. D[c][x] = A[c][s] * B[s][x] with A the scaled gradients (read from gradOut in place, no workspace
// copy) and B a 0/1 matrix that is never stored: a lane builds
// its entries from the slot's column bounds.  The per-pixel version above does ~180 M LDS
// read-add-writes at config 3 and is bound by them (140 us); this form does ~2.5 M
// v_mfma_f32_16x16x4_f32 (~38 us of matrix pipe).
//   k_roi_rowlists      one workgroup per map row: evaluates the RoIs' bins itself (roipool_cuda.cu:32-51,
//                       the same roi_bin as every other kernel) and compacts the row's slot list --
//                       {gradOut offset, column bounds, 1/n} per (r, i, j), ascending -- without atomics.
//   k_roipool_bwd_gemm  one workgroup per TASK (row, NCT c-tiles of 16 channels).  The work of a row varies 40x
//                       between the middle of the map and its edges: tasks are numbered from the middle rows
//                       outwards and there are more workgroups than fit the chip, so the dispatcher hands the
//                       next task to whichever CU frees a slot.  (Measured on the way: a static grid of (row, 64
//                       channels) workgroups 148 us -- the middle rows' workgroups, co-resident with others, set
//                       the time; a software queue on an atomic counter 110 us -- ~3 us of atomic + broadcast
//                       latency per task.)  The NW waves of a workgroup take every NW-th k-step (= 4 slots) for
//                       all NCT c-tiles x XT column tiles: one entry read from the LDS copy of the list, NCT
//                       gradOut loads (RG_PF k-steps ahead), and for every column tile the k-step reaches (a mask
//                       the list carries: ~45 % of the tiles are skipped) an indicator and NCT MFMAs; the waves'
//                       partial sums are added through LDS in wave order.
// Every gradIn element is written once, in a fixed order: deterministic whichever workgroup runs a task.
// The products with B are exact (x * 1, x * 0); gradOut / n is gradOut * (1/n) as in the per-pixel kernel
// (<= 1 ulp; the reference's atomics leave the order open).  A slot of an empty bin has scale 0 and
// contributes an exact 0 whatever gradOut holds there.  Non-finite gradOut: 0 * Inf = NaN reaches the
// other columns of that channel's row -- the reference would only add it to the bin's own pixels: a task
// that stored a non-finite value is recomputed with exact membership (cold path at the end of the kernel).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool pool_nonfinite4(const f32x4& d)
{
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(d[0]), __builtin_fabsf(d[1])),
                                    __builtin_fmaxf(__builtin_fabsf(d[2]), __builtin_fabsf(d[3])));
    return !(m <= 3.4028234663852886e38f) || d[0] != d[0] || d[1] != d[1] || d[2] != d[2] || d[3] != d[3];
}
struct RmSlot { int goff; int jb; float scale; int pad; };           // 16 bytes
// every bin row of a RoI can contain a given map row (RoIs lower than 7 pixels): 7 pairs (RoI, bin row) per RoI and map row,
// each written as EIGHT entries = two k-steps: slots j = 0, 2, 4, 6, then j = 1, 3, 5 and a zero-scale pad.  Lane group g of the
// GEMM kernel then fetches slots 2g and 2g+1 of a pair with ONE 8-byte load per channel (the pair's 28-byte record of
// gradOut = one cache line touched once) and uses its two components in the two k-steps -- round 4; before, a pair's
// seven slots were packed 4 + 3 into k-steps that straddled pairs and every k-step fetched its 16 lines again.
constexpr int RM_PS = 8;                                               // list entries per (RoI, bin row) pair
inline int rm_cap(int R) { return R * KT * RM_PS; }
static size_t rm_lists_bytes(int R, int H) { return align256((size_t)H * rm_cap(R) * sizeof(RmSlot)) + align256((size_t)H * sizeof(int)); }

constexpr int RL_T = 512;                      // threads of a row-list workgroup = RoIs per pass (R = 300: ONE pass; 256 took two, the second with 44 RoIs)
__global__ void __launch_bounds__(RL_T)
k_roi_rowlists(const float* __restrict__ rois, RmSlot* __restrict__ rowslots, int* __restrict__ rownks,
               int R, int C, int H, int W, int cap)
{
    __shared__ unsigned pairs[RL_T * KT];                            // RoI - r0 | i << 10 | bin row height << 16
    __shared__ int colb[RL_T][KT];                                    // j0 | j1 << 16
    __shared__ int wsum[RL_T / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, y = blockIdx.x;
    RmSlot* out = rowslots + (size_t)y * cap;
    int nslots = 0;                                                  // uniform
    for (int r0 = 0; r0 < R; r0 += RL_T) {
        __syncthreads();                                             // previous chunk's tables consumed
        int mk = 0, hgt[KT];
        if (r0 + tid < R) {
#pragma unroll
            for (int q = 0; q < KT; ++q) {                           // (q, q): row bounds of bin row q, column bounds of bin column q
                const Bounds bq = roi_bin<float>(rois + 4 * (size_t)(r0 + tid), q, q, H, W, KT);
                mk |= (y >= bq.i0 && y < bq.i1) ? 1 << q : 0;
                hgt[q] = bq.i1 - bq.i0;
                colb[tid][q] = bq.j0 | (bq.j1 << 16);
            }
        }
        const int cnt = __builtin_popcount(mk);
        int inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int base = 0, npairs = 0;
#pragma unroll
        for (int w = 0; w < RL_T / 64; ++w) { const int v = wsum[w]; base += w < wave ? v : 0; npairs += v; }
        int pos = base + inc - cnt;
#pragma unroll
        for (int i = 0; i < KT; ++i)
            if (mk & (1 << i)) pairs[pos++] = (unsigned)tid | (unsigned)i << 10 | (unsigned)hgt[i] << 16;
        __syncthreads();
        for (int sidx = tid; sidx < npairs * RM_PS; sidx += RL_T) {
            const int p = sidx / RM_PS, q = sidx - p * RM_PS;
            const int j = q < 4 ? 2 * q : 2 * (q - 4) + 1;           // entry q of a pair: j = 0, 2, 4, 6, 1, 3, 5, (7 = pad)
            const unsigned pr = pairs[p];
            const int rl = pr & 1023, i = (pr >> 10) & 7, h = pr >> 16;
            const int cb = j < KT ? colb[rl][j] : 0, wd = (cb >> 16) - (cb & 0xffff), n = j < KT ? h * wd : 0;
            RmSlot e;
            e.goff = (r0 + rl) * C * KK + i * KT + (j < KT ? j : KT - 1);   // + channel * 49 (the pad points at slot 6: never dereferenced on its own)
            e.jb = n > 0 ? (cb & 0xffff) | (wd << 16) : 0;          // first column | width << 16; a slot that adds nothing (pad, empty bin) has width 0
            e.scale = n > 0 ? 1.0f / static_cast<float>(n) : 0.f;
            e.pad = 0;
            out[nslots + sidx] = e;
        }
        nslots += npairs * RM_PS;
    }
    const int nks = nslots >> 2;                                      // two k-steps per pair
    if (tid == 0) rownks[y] = nks;
    __syncthreads();                                                 // the row's list is complete (and visible to this workgroup)
    // per k-step (4 slots): which 16-column tiles does any of its slots reach?  Stored in all four `pad`
    // fields, so that the GEMM kernel skips the other tiles' MFMAs with a scalar test.
    for (int ks = tid; ks < nks; ks += RL_T) {
        int tm = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const RmSlot e = out[4 * ks + q];
            const int j0 = e.jb & 0xffff, j1 = j0 + (e.jb >> 16);
            if (j1 > j0) tm |= ((2 << ((j1 - 1) >> 4)) - 1) & ~((1 << (j0 >> 4)) - 1);   // tiles j0/16 .. (j1-1)/16
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) out[4 * ks + q].pad = tm;
    }
}

constexpr int RG_CH = 128;                    // k-steps (of 4 slots) per LDS chunk of the list: 8 KB
// lab overrides (make EXTRA=-D...), round 4 at config 3: 8 pairs in flight 67.1 us, 8 waves per task 72.2 us, both 72.5 us against
// 65.7 us for (4, 4) -- the kernel is bound by the instructions of a k-step (list entry, scale, indicator), not by its loads' latency.
// Round 5, with the lean k-step: pairs in flight per wave 1 / 2 / 3 / 4 / 8 -> 57.8 / 59.7 / 62.9 / 61.7 / 67.0 us (R = 300 C = 1891: 106.7 / 109 /
// 118 / 116 / 126); 8 or 2 waves per task at one pair in flight 61.2 / 76.7 us.  One pair ahead is enough (16 waves per CU hide the rest) and the
// shortest loop wins: vector and scalar instructions are what the k-step costs beside its MFMAs.
#ifndef D2T_RG_PF
#define D2T_RG_PF 1
#endif
#ifndef D2T_ROI_NW
#define D2T_ROI_NW 4
#endif
constexpr int RG_PF = D2T_RG_PF;              // pairs (two k-steps) of gradOut loads in flight per wave

// NCT: c-tiles (16 channels) per task, NW: waves that split a task's slots.  (NCT, NW) at config 3, whole op:
// (1,4) 158 us, (2,4) 91 us, (4,4) 97 us, (2,2) 128 us -- fewer channels per task repeat the per-k-step
// entry / indicator work, more make the middle rows' tasks too long.
template <int XT, int NCT, int NW>
__global__ void __launch_bounds__(NW * 64)
k_roipool_bwd_gemm(const float* __restrict__ gout, const RmSlot* __restrict__ rowslots, const int* __restrict__ rownks,
                   float* __restrict__ gin, int C, int H, int W, int cap, int ncb, int ntasks, unsigned gout_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    RmSlot* list = reinterpret_cast<RmSlot*>(lds_raw);               // [4 * RG_CH]
    f32x4* red = reinterpret_cast<f32x4*>(lds_raw + 4 * RG_CH * sizeof(RmSlot));   // [wave][c-tile * XT + x-tile][lane]
    constexpr int NACC = NCT * XT, NTHR = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gout), 0, (unsigned)gout_bytes, 0x00020000);
    // One task per workgroup; more workgroups than fit the chip, so the hardware dispatcher hands the next
    // task to whichever CU frees a slot -- a work queue without the ~3 us per task an atomic counter +
    // broadcast cost when tried.  Tasks are numbered from the MIDDLE rows outwards (mid, mid+1, mid-1, ...):
    // RoIs crowd the middle of the map, and the long tasks should start first.
    D2T_KSTAMP(0); D2T_KSTAMP_RT(14);
    const int dbg = D2T_KDBG;                                        // knob builds: 1 = no gradOut loads, 2 = no MFMA, 4 = no list reads in the k-steps
    for (int t = blockIdx.x; t < ntasks; t += gridDim.x) {
        const int p = t / ncb, d = (p + 1) >> 1, y = (p & 1) ? (H - 1) / 2 + d : (H - 1) / 2 - d;
        const int c0 = (t - p * ncb) * (16 * NCT);
        const RmSlot* sl = rowslots + (size_t)y * cap;
        const int nks = rownks[y];
        int gach[NCT];                                                // float offset of this lane's channel record (clamped: never stored)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) gach[ct] = (c0 + 16 * ct + n < C ? c0 + 16 * ct + n : C - 1) * KK;
        f32x4 acc[NCT][XT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int x = 0; x < XT; ++x) acc[ct][x] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int k0 = 0; k0 < nks; k0 += RG_CH) {
            const int kc = nks - k0 < RG_CH ? nks - k0 : RG_CH;      // k-steps in this chunk
            if (k0) __syncthreads();                                 // previous chunk consumed
            // the chunk's pairs are padded to a multiple of NW x RG_PF with entries that add nothing (width 0, scale 0, tile mask 0) and point
            // at the last real pair's record: the loops below run without tail guards
            const int kpp = ((kc >> 1) + NW * RG_PF - 1) / (NW * RG_PF) * (NW * RG_PF);
            for (int e = tid; e < 8 * kpp; e += NTHR) {
                RmSlot v = sl[4 * k0 + (e < 4 * kc ? e : 4 * kc - 8 + (e & 7))];
                if (e >= 4 * kc) { v.jb = 0; v.scale = 0.f; v.pad = 0; }
                list[e] = v;
            }
            __syncthreads();
            // wave w takes PAIRS (two k-steps = the 8 entries of one (RoI, bin row)) w, w+NW, ...; lane (n, g) loads the
            // pair's slots 2g, 2g+1 of channel n with one 8-byte load (range-checked: slot "7" of the last record of the
            // tensor lies behind it and reads as 0; its scale is 0 anyway) and multiplies component 0 in the pair's first
            // k-step (entries 8p + g: even slots) and component 1 in its second (entries 8p + 4 + g: odd slots, pad)
            const int mine = kpp / NW;                                // this wave's pairs of the padded chunk: a multiple of RG_PF
            f32x2 av[RG_PF][NCT];
            auto a_load = [&](int m, int ct) -> f32x2 {
                if (dbg & 1) return f32x2{1.f, 1.f};
                const int go = list[8 * (wave + NW * m) + g].goff;
                return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rg, (int)((unsigned)(gach[ct] + go) * 4u), 0, 0));
            };
#pragma unroll
            for (int q = 0; q < RG_PF; ++q)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) av[q][ct] = a_load(q, ct);
#pragma unroll 1
            for (int m0 = 0; m0 < mine; m0 += RG_PF) {
                const bool more = m0 + RG_PF < mine;                 // uniform: the next RG_PF pairs exist (all of them or none)
#pragma unroll
                for (int q = 0; q < RG_PF; ++q) {
                    const int m = m0 + q;                            // this wave's m-th pair of the chunk
                    f32x2 cur[NCT];
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) cur[ct] = av[q][ct];
                    if (more) {                                      // refill this register slot
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) av[q][ct] = a_load(m + RG_PF, ct);
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {                    // the pair's two k-steps
                        const RmSlot e = list[(dbg & 4) ? g : 8 * (wave + NW * m) + 4 * h + g];
                        // B = 1 / n where the lane's column lies in the slot's bin, else 0 (the scale sits in B: A is gradOut as loaded);
                        // (unsigned)(col - j0) < width is the whole membership test, false for every column of a slot of width 0
                        const int d0 = n - (e.jb & 0xffff);
                        const unsigned wd = (unsigned)e.jb >> 16;
                        const int tm = __builtin_amdgcn_readfirstlane(e.pad);   // column tiles this k-step reaches
#pragma unroll
                        for (int x = 0; x < XT; ++x) {
                            // none of the 4 slots reaches this tile: B = 0 (scalar test).  The test is not what the loop waits for: with tiles 1
                            // and 2 of every k-step multiplied unconditionally (same MFMA count, no tests) the op takes 64.3 us against 61.7
                            if (!(tm & (1 << x)) || (dbg & 2)) continue;
                            const float ind = (unsigned)(d0 + 16 * x) < wd ? e.scale : 0.f;
#pragma unroll
                            for (int ct = 0; ct < NCT; ++ct) acc[ct][x] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[ct][h], ind, acc[ct][x], 0, 0, 0);
                        }
                    }
                }
            }
        }
        D2T_KSTAMP(1);
        // partial sums of the NW waves -> LDS; wave w adds the accumulators (c-tile * XT + x) = w mod NW in wave order and stores them
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int x = 0; x < XT; ++x) red[(wave * NACC + ct * XT + x) * 64 + lane] = acc[ct][x];
        __syncthreads();
        for (int ai = wave; ai < NACC; ai += NW) {                   // uniform
            f32x4 v = red[(0 * NACC + ai) * 64 + lane];
            for (int w = 1; w < NW; ++w) {
                const f32x4 o = red[(w * NACC + ai) * 64 + lane];
                v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
            }
            const int ct = ai / XT, x = ai - ct * XT;
            const int col = 16 * x + n;                              // D[m = channel 4g + r][n = column]
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = c0 + 16 * ct + 4 * g + r;
                if (c < C && col < W) gin[((size_t)c * H + y) * W + col] = v[r];
            }
            // Cold path.  The matrix form multiplies every slot by a 0/1 weight for every column of a tile: an Inf / NaN in
            // gradOut then reaches (0 x Inf = NaN) the columns outside its bin, which the reference (roipool_cuda.cu:111-117)
            // never touches.  Whatever was poisoned is itself non-finite, so the wave that stored a non-finite value
            // recomputes that 16 x 16 tile in the reference's form: a pixel adds only the slots whose bin contains it.
            // (Wave-local on purpose: a workgroup-wide flag in static LDS pushed the kernel from 4 to 3 workgroups per CU.)
            if (__builtin_expect(__any(pool_nonfinite4(v)), 0)) {
                for (int e = lane; e < 256; e += 64) {
                    const int c = c0 + 16 * ct + (e >> 4), xx = 16 * x + (e & 15);
                    if (c >= C || xx >= W) continue;
                    const float* gc = gout + (size_t)c * KK;
                    float a = 0.f;
                    for (int k = 0; k < 4 * nks; ++k) {
                        const RmSlot sle = sl[k];
                        if ((unsigned)(xx - (sle.jb & 0xffff)) < ((unsigned)sle.jb >> 16)) a += gc[sle.goff] * sle.scale;
                    }
                    gin[((size_t)c * H + y) * W + xx] = a;
                }
            }
        }
        __syncthreads();                                             // red and list are free again
        D2T_KSTAMP(2); D2T_KSTAMP_RT(15);
        D2T_KSTAMP_PUT(3, (unsigned long long)nks);
    }
}

static bool roipool_bwd_mfma_supported(int R, int C, int H, int W, int k)
{
    return k == KT && R >= 1 && C >= 1 && H >= 1 && W >= 1 && W <= 128 && H <= 65535 && 1LL * R * C * KK * 4 < 0xfffffff0LL &&   // 32-bit byte offsets into gradOut (buffer loads)
          
           1LL * H * ((C + 15) / 16) < 0x7fffffffLL && 1LL * H * rm_cap(R) * (long long)sizeof(RmSlot) < 0x7fffffffLL;
}

// workspace of the GEMM form: row lists | k-steps per row
static int roipool_bwd_mfma_f32(const float* gout, const float* rois, float* gin, int R, int C, int H, int W, void* ws, hipStream_t st)
{
    char* w = static_cast<char*>(ws);
    const int cap = rm_cap(R);
    RmSlot* rowslots = reinterpret_cast<RmSlot*>(w); w += align256((size_t)H * cap * sizeof(RmSlot));
    int* rownks = reinterpret_cast<int*>(w);
    // A is read from gradOut IN PLACE (16 channels of a slot = 16 lines, a 28-byte run of each feeding two
    // k-steps): 83 us at config 3.  Reading an (R, 49, C) copy instead (one line per slot) makes the GEMM kernel
    // 18 us faster but the copy costs 26 us and 60 MB of workspace: 91 us.
    hipLaunchKernelGGL(k_roi_rowlists, dim3(H), dim3(RL_T), 0, st, rois, rowslots, rownks, R, C, H, W, cap);
    int rc = launch_status();
    if (rc != D2T_OK) return rc;
    const int xt = (W + 15) / 16;
    // (c-tiles per task, waves per task): D2T_ROI_CFG=1..4 -> (1,4) | (2,4) | (4,4) | (2,2); a lab knob, read once
    static const int cfg = [] { const int v = lab_env_int("D2T_ROI_CFG", 2); return v >= 1 && v <= 4 ? v : 2; }();   // -DD2T_LAB only
    const int nct = cfg == 1 ? 1 : cfg == 3 ? 4 : 2;
    const int ncb = (C + 16 * nct - 1) / (16 * nct), ntasks = H * ncb;
#define D2T_LAUNCH_GEMM(XTV, NCTV, NWV)                                                                        \
    {                                                                                                          \
        const size_t lds = 4 * RG_CH * sizeof(RmSlot) + (size_t)NWV * NCTV * XTV * 64 * sizeof(f32x4);         \
        D2T_ENSURE_DYNAMIC_LDS((k_roipool_bwd_gemm<XTV, NCTV, NWV>), 160 * 1024);                              \
        int per_cu = (int)((size_t)(160 * 1024) / (lds + 64));                                                 \
        const int by_waves = 16 / NWV;                                                                         \
        per_cu = per_cu > by_waves ? by_waves : per_cu;                                                        \
        (void)per_cu;                                                                                          \
        const int nwg = ntasks;            /* one task per workgroup: the dispatcher is the work queue */      \
        hipLaunchKernelGGL((k_roipool_bwd_gemm<XTV, NCTV, NWV>), dim3(nwg), dim3(NWV * 64), lds, st, gout, rowslots, rownks, gin, \
                           C, H, W, cap, ncb, ntasks, (unsigned)((size_t)R * C * KK * sizeof(float)));                 \
    }
#define D2T_LAUNCH_GEMM_X(NCTV, NWV) { if (xt <= 4) D2T_LAUNCH_GEMM(4, NCTV, NWV) else if (xt <= 5) D2T_LAUNCH_GEMM(5, NCTV, NWV) else D2T_LAUNCH_GEMM(8, NCTV, NWV) }
    if (cfg == 1) D2T_LAUNCH_GEMM_X(1, 4) else if (cfg == 2) D2T_LAUNCH_GEMM_X(2, D2T_ROI_NW) else if (cfg == 3) D2T_LAUNCH_GEMM_X(4, 4) else D2T_LAUNCH_GEMM_X(2, 2)
#undef D2T_LAUNCH_GEMM_X
#undef D2T_LAUNCH_GEMM
    return launch_status();
}

static bool roipool_bwd_pixel_supported(int R, int C, int H, int W, int k)
{
    return k == KT && R >= 1 && C >= 1 && H >= 1 && W >= 1 && (C + 63) / 64 <= 65535 &&
           (size_t)(W + 1) * RB_LD * sizeof(float) <= 64 * 1024;
}

// D2T_ROI_BWD=pixel|mfma forces one of the two designs (a lab knob for A/B measurements, read once)
static bool roipool_bwd_use_mfma(int R, int C, int H, int W, int k)
{
    static const int forced = [] { const char* e = lab_env_str("D2T_ROI_BWD"); return !e ? 0 : !strcmp(e, "pixel") ? 1 : !strcmp(e, "mfma") ? 2 : 0; }();   // -DD2T_LAB only
    const bool m = roipool_bwd_mfma_supported(R, C, H, W, k), p = roipool_bwd_pixel_supported(R, C, H, W, k);
    if (!m || !p) return m;
    return forced != 1;
}

bool roipool_bwd_supported(int R, int C, int H, int W, int k)
{
    return roipool_bwd_mfma_supported(R, C, H, W, k) || roipool_bwd_pixel_supported(R, C, H, W, k);
}

size_t roipool_bwd_ws_bytes(int R, int C, int H, int W, int k)
{
    if (!roipool_bwd_supported(R, C, H, W, k)) return 0;
    const size_t geom = geo_bytes(R) + align256((size_t)R * H) + align256((size_t)R * 64 * sizeof(float));
    return roipool_bwd_use_mfma(R, C, H, W, k) ? rm_lists_bytes(R, H) : align256((size_t)R * C * KK * sizeof(float)) + geom;
}

int roipool_bwd_f32(const float* gout, const float* rois, float* gin, int R, int C, int H, int W, int k,
                    void* ws, hipStream_t st)
{
    if (roipool_bwd_use_mfma(R, C, H, W, k)) return roipool_bwd_mfma_f32(gout, rois, gin, R, C, H, W, ws, st);
    float* gt = static_cast<float*>(ws);
    int32_t* geo = reinterpret_cast<int32_t*>(static_cast<char*>(ws) + align256((size_t)R * C * KK * sizeof(float)));
    hipLaunchKernelGGL(k_transpose_gout, dim3(R, (C + 63) / 64), dim3(256), 0, st, gout, gt, C);   // (R,C,49) -> (R,49,C)
    int rc = launch_status();
    if (rc != D2T_OK) return rc;
    uint8_t* rowmask = reinterpret_cast<uint8_t*>(geo) + geo_bytes(R);
    float* rcp = reinterpret_cast<float*>(rowmask + align256((size_t)R * H));
    rc = roi_geom(rois, geo, rowmask, rcp, R, H, W, st);
    if (rc != D2T_OK) return rc;
    const dim3 grid(H, (C + 63) / 64);
    const size_t acc_bytes = (size_t)(W + 1) * RB_LD * sizeof(float);
    int waves = (int)(64 * 1024 / acc_bytes);                        // private accumulators that fit 64 KB of LDS
    waves = waves > 4 ? 4 : waves;
    hipLaunchKernelGGL(k_roipool_bwd_batched, grid, dim3(64 * waves), waves * acc_bytes, st,
                       gt, geo, rowmask, rcp, gin, R, C, H, W);
    return launch_status();
}

// ---------------------------------------------------------------------------------------
// PSROIPool forward.  Workgroup = (RoI, 256 consecutive outputs (t, bin)).  The RoI's 49 cells
// are evaluated once per workgroup (lanes 0..48, double-precision bin centres as the reference)
// and shared through LDS instead of once per output; a thread then sums its cell row by row:
// the pixels of a row are fetched with 8 independent (clamped) loads and added in ascending x, so
// the dependent chain is the cell's rows, not its pixels.  The running sum sees the pixels in the
// reference's order (ps_roipool_cuda.cu:60-66) and the guarded IEEE divide is kept: bit-identical.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_psroipool_fwd_roi(const float* __restrict__ fm, const float* __restrict__ rois, float* __restrict__ out,
                    int nT, int H, int W)
{
    __shared__ int4 cells[KK];
    const int r = blockIdx.x;
    if (threadIdx.x < KK) {
        const int i = threadIdx.x / KT, j = threadIdx.x - i * KT;
        const Bounds c = psroi_cell<float>(rois + 4 * (size_t)r, i, j, H, W, KT);
        cells[threadIdx.x] = make_int4(c.i0, c.i1, c.j0, c.j1);
    }
    __syncthreads();
    const int e = blockIdx.y * 256 + threadIdx.x;                    // output (t, bin) of this RoI
    if (e >= nT * KK) return;
    const int t = e / KK, bin = e - t * KK;
    const int4 c = cells[bin];
    const float* ch = fm + (size_t)((t + 1) * bin) * H * W;          // ps_roipool_cuda.cu:58
    const int w = c.w - c.z;
    float acc = 0.f;
    for (int y = c.x; y < c.y; ++y) {
        const float* row = ch + y * W + c.z;
        for (int x0 = 0; x0 < w; x0 += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = row[x0 + k < w ? x0 + k : w - 1];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (x0 + k < w) acc += v[k];                         // ascending x
        }
    }
    const int n = (c.y - c.x) * w;
    if (n > 0) acc /= static_cast<float>(n);                         // guarded divide, :68
    out[(size_t)r * nT * KK + e] = acc;
}

int psroipool_fwd_small_f32(const float* fm, const float* rois, float* out, int R, int nT, int H, int W, int,
                      hipStream_t st)
{
    hipLaunchKernelGGL(k_psroipool_fwd_roi, dim3(R, (nT * KK + 255) / 256), dim3(256), 0, st, fm, rois, out, nT, H, W);
    return launch_status();
}

// ---------------------------------------------------------------------------------------
// Pixel ownership of the PSROIPool backward: 4 waves x bands of <= PX_ROWS rows, lane = column
// (two column groups for maps wider than 64).
// ---------------------------------------------------------------------------------------
constexpr int PX_MAXROWS = 16;               // rows per wave band: H <= 64
constexpr int PX_XG = 2;                     // column groups: W <= 128

// ---------------------------------------------------------------------------------------
// PSROIPool backward, phase 1.  Workgroup = output plane (t, bin).  cells: (R,7,7,4) int32.
// part[plane][y][x] = sum over RoIs r whose cell `bin` contains (y,x) of gout[r,t,bin] / n
// (ps_roipool_cuda.cu:131-139), ascending r.
// ---------------------------------------------------------------------------------------
template <int PX_ROWS>                       // band height the row loop is unrolled for (10: H <= 40)
__global__ void __launch_bounds__(256)
k_psroipool_bwd_plane(const float* __restrict__ gout, const int32_t* __restrict__ cells, float* __restrict__ part,
                      int R, int nT, int H, int W)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int plane = blockIdx.x, bin = plane % KK;                  // plane = t*49 + bin
    const int band = (H + 3) >> 2;
    const int y_lo = wave * band, y_hi = y_lo + band < H ? y_lo + band : H;
    float acc[PX_ROWS][PX_XG];
#pragma unroll
    for (int k = 0; k < PX_ROWS; ++k) { acc[k][0] = 0.f; acc[k][1] = 0.f; }

    if (y_lo < y_hi) {
        const int4* ct = reinterpret_cast<const int4*>(cells) + bin;
        const float* gp = gout + plane;
        for (int rb = 0; rb < R; rb += 64) {
            // 64 RoIs per instruction: lane l fetches and tests RoI rb+l
            const int rr = rb + lane;
            int4 cb = make_int4(0, 0, 0, 0);
            float v = 0.f;
            if (rr < R) {
                cb = ct[(size_t)rr * KK];
                v = gp[(size_t)rr * nT * KK];
            }
            const int n = (cb.y - cb.x) * (cb.w - cb.z);
            const bool hit = cb.y > cb.x && cb.w > cb.z && cb.y > y_lo && cb.x < y_hi;
            v = v / static_cast<float>(n > 0 ? n : 1);               // ps_roipool_cuda.cu:135
            unsigned long long m = __ballot(hit);
            while (m) {                                              // ascending r
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const int i0 = __builtin_amdgcn_readlane(cb.x, l), i1 = __builtin_amdgcn_readlane(cb.y, l);
                const int j0 = __builtin_amdgcn_readlane(cb.z, l), j1 = __builtin_amdgcn_readlane(cb.w, l);
                const float vv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
                const float v0 = (lane >= j0 && lane < j1) ? vv : 0.f;
                const float v1 = (lane + 64 >= j0 && lane + 64 < j1) ? vv : 0.f;
#pragma unroll
                for (int k = 0; k < PX_ROWS; ++k) {
                    const int y = y_lo + k;
                    if (y >= i0 && y < i1 && y < y_hi) { acc[k][0] += v0; acc[k][1] += v1; }   // wave-uniform
                }
            }
        }
    }
    float* dst = part + (size_t)plane * H * W;
#pragma unroll
    for (int k = 0; k < PX_ROWS; ++k) {
        const int y = y_lo + k;
        if (y < y_hi) {
            if (lane < W) dst[y * W + lane] = acc[k][0];
            if (lane + 64 < W) dst[y * W + lane + 64] = acc[k][1];
        }
    }
}

// ---------------------------------------------------------------------------------------
// PSROIPool backward, phase 1, LDS form (maps up to 4096 pixels).  Workgroup = output plane
// (t, bin); each of the 4 waves owns a quarter of the RoIs and a PRIVATE copy of the plane in LDS.
// 64 RoIs are fetched per instruction (lane = RoI) and their cells computed on the fly (no cell
// table, no extra launch).  Two ways to add them, chosen by the launcher:
//   SWEEP = false  every non-empty cell in turn (ascending r) by the lanes of an 8 x 8 grid laid
//                  over it: one LDS read-add-write of <= 64 pixels, ~25 instructions per cell (the
//                  register-band form above needs ~80: it tests all of its rows against every cell);
//   SWEEP = true   all 64 cells together, pixel (dy, dx) of every cell per step, with ds_add_f32
//                  on the wave-private plane: ~5 instructions per cell but ~3.4 cycles per lane in
//                  the LDS -- wins when there are too few planes to hide the per-cell latency.
// Each wave adds its RoIs in ascending order, the four copies are summed in wave order.
// ---------------------------------------------------------------------------------------
constexpr int PL_WAVES = 4;
constexpr int PL_MAXPIX = 4096;                                      // 4 planes x 16 KB = 64 KB of LDS

template <bool SWEEP>
__global__ void __launch_bounds__(PL_WAVES * 64)
k_psroipool_bwd_plane_lds(const float* __restrict__ gout, const float* __restrict__ rois, float* __restrict__ part,
                          int R, int nT, int H, int W)
{
    extern __shared__ float planes[];                                // [PL_WAVES][H*W]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int plane = blockIdx.x, bin = plane % KK, HW = H * W;      // plane = t*49 + bin
    float* mine = planes + wave * HW;
    for (int e = lane; e < HW; e += 64) mine[e] = 0.f;
    const int dy = lane >> 3, dx = lane & 7, loff = dy * W + dx;
    // gridDim.y workgroups share a plane (few planes, many RoIs: 196 workgroups of 3000 RoIs each leave a quarter of the
    // chip idle and the rest latency-bound): RoI slice blockIdx.y, then a quarter of it per wave; every slice writes its
    // own partial plane and k_psroipool_bwd_gather adds the slices in ascending order
    const int part_id = blockIdx.y * PL_WAVES + wave, parts = gridDim.y * PL_WAVES;
    const int r_lo = (int)((long long)R * part_id / parts), r_hi = (int)((long long)R * (part_id + 1) / parts);
    const int bi = bin / KT, bj = bin - bi * KT;
    const float* gp = gout + plane;
    for (int rb = r_lo; rb < r_hi; rb += 64) {
        const int rr = rb + lane;
        int4 cb = make_int4(0, 0, 0, 0);                             // (i0, i1, j0, j1) of this RoI's cell
        float v = 0.f;
        if (rr < r_hi) {
            const Bounds c = psroi_cell<float>(rois + 4 * (size_t)rr, bi, bj, H, W, KT);
            cb = make_int4(c.i0, c.i1, c.j0, c.j1);
            v = gp[(size_t)rr * nT * KK];
        }
        const int n = (cb.y - cb.x) * (cb.w - cb.z);
        const bool hit = cb.y > cb.x && cb.w > cb.z;
        v = v / static_cast<float>(n > 0 ? n : 1);                   // ps_roipool_cuda.cu:135
        if (SWEEP) {
            // lane = RoI: the 64 cells are swept together, pixel (dy, dx) of every cell per step, with
            // LDS float adds (the plane is private to the wave: no contention, fixed order)
            const int h = hit ? cb.y - cb.x : 0, w = cb.w - cb.z;
            float* p = mine + cb.x * W + cb.z;
            for (int sy = 0; __ballot(sy < h); ++sy)
                for (int sx = 0; __ballot(sy < h && sx < w); ++sx)
                    if (sy < h && sx < w)
                        __hip_atomic_fetch_add(p + sy * W + sx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            continue;
        }
        // cells are visited one at a time in ascending r; the lanes form an 8 x 8 grid laid over the
        // cell: one LDS read-add-write covers it (larger cells take more grid positions)
        const int base = cb.x * W + cb.z;
        const int hw = ((cb.y - cb.x) << 16) | (cb.w - cb.z);
        unsigned long long m = __ballot(hit);
        while (m) {
            const int l = __builtin_ctzll(m);
            m &= m - 1;
            const int sb = __builtin_amdgcn_readlane(base, l), shw = __builtin_amdgcn_readlane(hw, l);
            const float vv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
            const int h = shw >> 16, w = shw & 0xffff;
            float* p = mine + sb + loff;
            if (h <= 8 && w <= 8) {                                  // wave-uniform: the usual case
                if (dy < h && dx < w) *p += vv;
            } else {
                for (int ty = 0; ty < h; ty += 8)
                    for (int tx = 0; tx < w; tx += 8)
                        if (ty + dy < h && tx + dx < w) p[ty * W + tx] += vv;
            }
        }
    }
    __syncthreads();
    float* dst = part + ((size_t)blockIdx.y * gridDim.x + plane) * HW;
    for (int e = threadIdx.x; e < HW; e += PL_WAVES * 64)
        dst[e] = ((planes[e] + planes[HW + e]) + planes[2 * HW + e]) + planes[3 * HW + e];
}

// Phase 2: the planes that map to an input channel are (t, bin) with (t+1)*bin == ch, i.e. bin | ch
// with ch/bin <= nT (ps_roipool_cuda.cu:58); channel 0 collects bin 0 of every t.  Each workgroup
// finds its channel's planes itself (ascending bin) and adds them in that fixed order.

// phase 2: gin[ch] = sum of the planes that map to ch, ascending bin then t; channels nothing
// maps to are zero.
__global__ void __launch_bounds__(256)
k_psroipool_bwd_gather(const float* __restrict__ part, float* __restrict__ gin, int nT, int HW, int nslice = 1)
{
    // planes (t, bin) with (t+1)*bin == ch, ascending bin: lane b-1 of the first wave tests bin b
    __shared__ int32_t srcs[KK];
    __shared__ int nsrc;
    const int ch = blockIdx.y;
    if (threadIdx.x < 64) {
        const int bin = threadIdx.x + 1;
        const bool is_src = ch > 0 && bin < KK && ch % bin == 0 && ch / bin <= nT;
        const unsigned long long m = __ballot(is_src);
        if (is_src) srcs[__builtin_popcountll(m & ((1ull << threadIdx.x) - 1ull))] = (ch / bin - 1) * KK + bin;
        if (threadIdx.x == 0) nsrc = __builtin_popcountll(m);
    }
    __syncthreads();
    const int ns = nsrc;
    const size_t slice = (size_t)nT * KK * HW;                       // partial planes of one RoI slice (nslice = 1: the planes themselves)
    // the planes of a channel are read EIGHT at a time (all loads issued, then added in order): a thread that walks its <= 31 planes one
    // dependent load after the other spends a memory round trip on each (round 6: with one workgroup per channel instead of ten the op took 77 us)
    const int nsrc_all = ch == 0 ? nT : ns;
    auto plane_of = [&](int k) { return ch == 0 ? k * KK : srcs[k]; };   // bin 0 of every target / the listed planes
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        float a = 0.f;
        if (nslice == 1) {
            for (int k0 = 0; k0 < nsrc_all; k0 += 8) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = k0 + k < nsrc_all ? part[(size_t)plane_of(k0 + k) * HW + p] : 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) a = k0 + k < nsrc_all ? a + v[k] : a;
            }
        } else {                                                     // RoI slices: plane by plane, the slices of a plane ascending
            for (int k = 0; k < nsrc_all; ++k) {
                const size_t o = (size_t)plane_of(k) * HW + p;
                for (int s0 = 0; s0 < nslice; s0 += 4) {
                    float w[4];
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) w[sl] = s0 + sl < nslice ? part[(s0 + sl) * slice + o] : 0.f;
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) a = s0 + sl < nslice ? a + w[sl] : a;
                }
            }
        }
        gin[(size_t)ch * HW + p] = a;
    }
}

static bool psroipool_bwd_planes_supported(int R, int nT, int H, int W, int k)
{
    return k == KT && R >= 1 && nT >= 1 && H >= 1 && W >= 1 && H <= 4 * PX_MAXROWS && W <= 64 * PX_XG && nT * KK <= 65535;
}

// RoI slices per plane of the LDS form.  Measured in round 3 at R = 3000, nT = 4 (196 planes): 4 slices per plane (784
// workgroups instead of 196) leave k_psroipool_bwd_plane_lds<true> at 53.6 us (1 slice: 54-56) and the per-cell form at
// 65 us -- the kernel is not short of workgroups, its LDS read-add-write chain per cell is what takes the time.  So: 1.
static int ps_plane_slices(int, int, int, int) { return 1; }

static size_t psroipool_bwd_planes_ws_bytes(int R, int nT, int H, int W, int k)
{
    if (!psroipool_bwd_planes_supported(R, nT, H, W, k)) return 0;
    return bins_bytes(R) + align256((size_t)ps_plane_slices(R, nT, H, W) * nT * KK * H * W * sizeof(float));
}

static int psroipool_bwd_planes_f32(const float* gout, const float* rois, float* gin, int R, int nT, int H, int W, int k,
                                   void* ws, hipStream_t st)
{
    int32_t* cells = static_cast<int32_t*>(ws);                      // only the register-band form reads a cell table
    float* part = reinterpret_cast<float*>(static_cast<char*>(ws) + bins_bytes(R));
    int rc;
    const int nslice = ps_plane_slices(R, nT, H, W);
    if (H * W <= PL_MAXPIX) {
        // few planes (< 2 workgroups per CU): one wave per SIMD, latency-bound -> sweep 64 cells at a time;
        // many planes: the LDS float-add rate (~3.4 cycles per lane) would bound -> per-cell read-add-write
        if (nT * KK < 512)
            hipLaunchKernelGGL(k_psroipool_bwd_plane_lds<true>, dim3(nT * KK, nslice), dim3(PL_WAVES * 64),
                               (size_t)PL_WAVES * H * W * sizeof(float), st, gout, rois, part, R, nT, H, W);
        else
            hipLaunchKernelGGL(k_psroipool_bwd_plane_lds<false>, dim3(nT * KK, nslice), dim3(PL_WAVES * 64),
                               (size_t)PL_WAVES * H * W * sizeof(float), st, gout, rois, part, R, nT, H, W);
    } else {
        rc = psroipool_bins<float>(rois, cells, R, H, W, k, st);
        if (rc != D2T_OK) return rc;
        if (H <= 40)
            hipLaunchKernelGGL(k_psroipool_bwd_plane<10>, dim3(nT * KK), dim3(256), 0, st, gout, cells, part, R, nT, H, W);
        else
            hipLaunchKernelGGL(k_psroipool_bwd_plane<PX_MAXROWS>, dim3(nT * KK), dim3(256), 0, st, gout, cells, part, R, nT, H, W);
    }
    rc = launch_status();
    if (rc != D2T_OK) return rc;
    const int HW = H * W;
    hipLaunchKernelGGL(k_psroipool_bwd_gather, dim3((HW + 255) / 256, nT * KK), dim3(256), 0, st, part, gin, nT, HW, nslice);
    return launch_status();
}

// ---------------------------------------------------------------------------------------
// PSROIPool backward as a GEMM with an indicator matrix (round 2) -- the ROIPool scheme above with the
// TARGETS as the M dimension.  The nT output planes of one bin share their R cells, so for bin (i, j) and map
// row y
//     plane[t][y][x] = sum over the RoIs r whose cell row i contains y of  (gradOut[r][t][i][j] / n_r)  *  [x in cell column j of r],
// i.e. D[t][x] = A[t][r] * B[r][x]: 49 x H small GEMMs (K ~ R * 2.8 / H) on v_mfma_f32_16x16x4_f32.
//   k_ps_transpose_t   gradOut (R, nT, 49) -> (49, R, nTp): the nT targets of a slot are one or two 64-byte lines
//   k_ps_pairlists     workgroup (map row y, bin row i): the RoIs whose cell row i contains y, with the column bounds
//                      of their 7 cells -- compacted by wave scans, ascending RoI
//   k_ps_bwd_gemm      single-wave workgroups, task = (bin, y): the list chunk goes to LDS as {gradOut offset,
//                      column bounds, 1/n, column-tile mask of the k-step}; per k-step of 4 slots NCT loads, and
//                      for every column tile the k-step reaches an indicator and NCT MFMAs
//   k_psroipool_bwd_gather (above)  adds, per input channel, the planes that map to it.
// Every plane element is written once as an ascending-RoI f32 chain: deterministic.  gradOut / n is
// gradOut * (1/n) (<= 1 ulp; the reference's atomics leave the order open).
// ---------------------------------------------------------------------------------------
struct PgPair { unsigned rh; int jb[KT]; };                          // r | cell rows << 16, then j0 | j1 << 16 of the 7 cells of bin row i
constexpr int PG_CH = 256;                                           // pairs per LDS chunk

__global__ void __launch_bounds__(256)
k_ps_transpose_t(const float* __restrict__ gout, float* __restrict__ vt, int R, int nT, int nTp)
{
    __shared__ float t[32 * KK + 8];
    const int r = blockIdx.x;
    const float* src = gout + (size_t)r * nT * KK;
    for (int e = threadIdx.x; e < nT * KK; e += 256) t[e] = src[e];
    __syncthreads();
    for (int e = threadIdx.x; e < KK * nTp; e += 256) {
        const int bin = e / nTp, tt = e - bin * nTp;
        vt[((size_t)bin * R + r) * nTp + tt] = tt < nT ? t[tt * KK + bin] : 0.f;
    }
}

__global__ void __launch_bounds__(256)
k_ps_pairlists(const float* __restrict__ rois, PgPair* __restrict__ lists, int* __restrict__ counts, int R, int H, int W)
{
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, y = blockIdx.x, i = blockIdx.y;
    PgPair* out = lists + ((size_t)i * H + y) * R;
    int total = 0;                                                   // uniform
    for (int r0 = 0; r0 < R; r0 += 256) {
        const int r = r0 + tid;
        PgPair e{0u, {0, 0, 0, 0, 0, 0, 0}};
        bool in = false;
        if (r < R) {
#pragma unroll
            for (int q = 0; q < KT; ++q) {                           // cell (i, q): rows depend on i only, columns on q only
                const Bounds c = psroi_cell<float>(rois + 4 * (size_t)r, i, q, H, W, KT);
                if (q == 0) { in = y >= c.i0 && y < c.i1; e.rh = (unsigned)r | (unsigned)(c.i1 > c.i0 ? c.i1 - c.i0 : 0) << 16; }
                e.jb[q] = c.j0 | (c.j1 << 16);
            }
        }
        int inc = in ? 1 : 0;
        const int mine = inc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        __syncthreads();                                             // wsum of the previous chunk consumed
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int base = 0, n = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int v = wsum[w]; base += w < wave ? v : 0; n += v; }
        if (mine) out[total + base + inc - 1] = e;
        total += n;
    }
    if (tid == 0) counts[i * H + y] = total;
}

template <int XT, int NCT, int NW>
__global__ void __launch_bounds__(NW * 64)
k_ps_bwd_gemm(const float* __restrict__ vt, const PgPair* __restrict__ lists, const int* __restrict__ counts,
              float* __restrict__ part, int R, int nT, int nTp, int H, int W)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    RmSlot* slots = reinterpret_cast<RmSlot*>(lds_raw);              // [PG_CH]
    f32x4* red = reinterpret_cast<f32x4*>(lds_raw + PG_CH * sizeof(RmSlot));   // [wave][c-tile * XT + x-tile][lane]
    constexpr int NACC = NCT * XT, NTHR = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // one task per workgroup, numbered from the middle rows outwards (see k_roipool_bwd_gemm)
    const int task = blockIdx.x, p = task / KK, bin = task - p * KK, d = (p + 1) >> 1;
    const int y = (p & 1) ? (H - 1) / 2 + d : (H - 1) / 2 - d, i = bin / KT, j = bin - i * KT;
    const PgPair* src = lists + ((size_t)i * H + y) * R;
    const int cnt = counts[i * H + y];
    const float* va = vt + (size_t)bin * R * nTp + n;                // + r * nTp + 16 * c-tile
    f32x4 acc[NCT][XT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int x = 0; x < XT; ++x) acc[ct][x] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p0 = 0; p0 < cnt; p0 += PG_CH) {
        const int pc = cnt - p0 < PG_CH ? cnt - p0 : PG_CH;
        const int nksp = ((pc + 3) / 4 + NW * RG_PF - 1) / (NW * RG_PF) * (NW * RG_PF), nsl = 4 * nksp, nks = nksp;   // padded with zero-scale slots
        if (p0) __syncthreads();                                     // previous chunk consumed
        for (int e = tid; e < nsl; e += NTHR) {
            RmSlot sl{0, 0, 0.f, 0};
            if (e < pc) {
                const unsigned rh = src[p0 + e].rh;
                const int jb = src[p0 + e].jb[j];
                const int nn = (int)(rh >> 16) * ((jb >> 16) - (jb & 0xffff));
                sl.goff = (int)(rh & 0xffff) * nTp;
                sl.jb = jb;
                sl.scale = nn > 0 ? 1.0f / static_cast<float>(nn) : 0.f;
            }
            slots[e] = sl;
        }
        __syncthreads();
        for (int ks = tid; ks < nks; ks += NTHR) {                   // column tiles a k-step reaches
            int tm = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const RmSlot e = slots[4 * ks + q];
                const int j0 = e.jb & 0xffff, j1 = e.jb >> 16;
                if (e.scale != 0.f && j1 > j0) tm |= ((2 << ((j1 - 1) >> 4)) - 1) & ~((1 << (j0 >> 4)) - 1);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) slots[4 * ks + q].pad = tm;
        }
        __syncthreads();
        const int mine = nksp / NW;                                  // wave w takes k-steps w, w+NW, ... (padded: straight-line loop)
        const RmSlot* my = slots + 4 * wave + g;
        float av[RG_PF][NCT];
#pragma unroll
        for (int q = 0; q < RG_PF; ++q) {
            const int go = my[4 * NW * q].goff;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) av[q][ct] = va[go + 16 * ct];
        }
#pragma unroll 1
        for (int m0 = 0; m0 < mine; m0 += RG_PF) {
#pragma unroll
            for (int q = 0; q < RG_PF; ++q) {
                const RmSlot e = my[4 * NW * (m0 + q)];
                float a[NCT];                                        // A = gradOut as loaded; the scale sits in B (vector instructions are matrix time)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) a[ct] = av[q][ct];
                const int nxt = m0 + q + RG_PF < mine ? m0 + q + RG_PF : mine - 1;
                const int go = my[4 * NW * nxt].goff;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) av[q][ct] = va[go + 16 * ct];
                const int d0 = n - (e.jb & 0xffff);
                const unsigned wd = (unsigned)((e.jb >> 16) - (e.jb & 0xffff));   // (a slot with scale 0 multiplies by B = 0 whatever its range)
                const int tm = __builtin_amdgcn_readfirstlane(e.pad);
#pragma unroll
                for (int x = 0; x < XT; ++x) {
                    if (!(tm & (1 << x))) continue;
                    const float ind = (unsigned)(d0 + 16 * x) < wd ? e.scale : 0.f;
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) acc[ct][x] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ct], ind, acc[ct][x], 0, 0, 0);
                }
            }
        }
    }
    // partial sums of the NW waves -> LDS; wave w adds the accumulators (c-tile * XT + x) = w mod NW in wave order and stores them
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int x = 0; x < XT; ++x) red[(wave * NACC + ct * XT + x) * 64 + lane] = acc[ct][x];
    __syncthreads();
    for (int ai = wave; ai < NACC; ai += NW) {                       // uniform
        f32x4 v = red[(0 * NACC + ai) * 64 + lane];
        for (int w = 1; w < NW; ++w) {
            const f32x4 o = red[(w * NACC + ai) * 64 + lane];
            v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
        }
        const int ct = ai / XT, x = ai - ct * XT;
        const int col = 16 * x + n;                                  // D[m = target 16*ct + 4g + r][n = column]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = 16 * ct + 4 * g + r;
            if (t < nT && col < W) part[(((size_t)t * KK + bin) * H + y) * W + col] = v[r];
        }
        // Cold path, as in k_roipool_bwd_gemm: a non-finite gradOut value times a 0 weight poisons columns outside its cell
        // (the reference, ps_roipool_cuda.cu:131-139, only adds to the cell's own pixels); the wave recomputes the 16 x 16
        // tile it stored with exact membership.
        if (__builtin_expect(__any(pool_nonfinite4(v)), 0)) {
            for (int e = lane; e < 256; e += 64) {
                const int t = 16 * ct + (e >> 4), xx = 16 * x + (e & 15);
                if (t >= nT || xx >= W) continue;
                const float* vb = vt + (size_t)bin * R * nTp + t;
                float a = 0.f;
                for (int k = 0; k < cnt; ++k) {
                    const unsigned rh = src[k].rh;
                    const int jb = src[k].jb[j], j0 = jb & 0xffff, j1 = jb >> 16;
                    const int nn = (int)(rh >> 16) * (j1 - j0);
                    if (nn > 0 && xx >= j0 && xx < j1) a += vb[(size_t)(rh & 0xffff) * nTp] * (1.0f / static_cast<float>(nn));
                }
                part[(((size_t)t * KK + bin) * H + y) * W + xx] = a;
            }
        }
    }
}

static bool psroipool_bwd_gemm_supported(int R, int nT, int H, int W, int k)
{
    return k == KT && R >= 1 && R <= 65535 && nT >= 1 && nT <= 32 && H >= 1 && H <= 65535 && W >= 1 && W <= 128 &&
           1LL * KK * R * 32 < 0x7fffffffLL && 1LL * nT * KK * H * W < 0x7fffffffLL && 1LL * KK * H < 0x7fffffffLL;
}

// workspace: vt (49, R, nTp) | pair lists (7, H, R) | counts (7, H) | planes (nT*49, H*W)
static size_t psroipool_bwd_gemm_ws_bytes(int R, int nT, int H, int W, int k)
{
    if (!psroipool_bwd_gemm_supported(R, nT, H, W, k)) return 0;
    const int nTp = 32;
    return align256((size_t)KK * R * nTp * 4) + align256((size_t)KT * H * R * sizeof(PgPair)) + align256((size_t)KT * H * 4) +
           align256((size_t)nT * KK * H * W * 4);
}

static int psroipool_bwd_gemm_f32(const float* gout, const float* rois, float* gin, int R, int nT, int H, int W, void* ws, hipStream_t st)
{
    // always two c-tiles (targets padded to 32 with zeros): the one-c-tile instantiation compiles to a mess of
    // accumulator copies (356 VGPRs at 8 column tiles) and ran slower with 4 targets than this one with 31
    const int nTp = 32;
    char* w = static_cast<char*>(ws);
    float* vt = reinterpret_cast<float*>(w); w += align256((size_t)KK * R * nTp * 4);
    PgPair* lists = reinterpret_cast<PgPair*>(w); w += align256((size_t)KT * H * R * sizeof(PgPair));
    int* counts = reinterpret_cast<int*>(w); w += align256((size_t)KT * H * 4);
    float* part = reinterpret_cast<float*>(w);
    hipLaunchKernelGGL(k_ps_transpose_t, dim3(R), dim3(256), 0, st, gout, vt, R, nT, nTp);
    hipLaunchKernelGGL(k_ps_pairlists, dim3(H, KT), dim3(256), 0, st, rois, lists, counts, R, H, W);
    int rc = launch_status();
    if (rc != D2T_OK) return rc;
    const int xt = (W + 15) / 16, ntasks = KK * H;
#define D2T_LAUNCH_PG(XTV, NCTV, NWV)                                                                          \
    {                                                                                                          \
        const size_t lds = PG_CH * sizeof(RmSlot) + (size_t)NWV * NCTV * XTV * 64 * sizeof(f32x4);             \
        D2T_ENSURE_DYNAMIC_LDS((k_ps_bwd_gemm<XTV, NCTV, NWV>), 160 * 1024);                                   \
        hipLaunchKernelGGL((k_ps_bwd_gemm<XTV, NCTV, NWV>), dim3(ntasks), dim3(NWV * 64), lds, st, vt, lists, counts, part, R, nT, nTp, H, W); \
    }
#define D2T_LAUNCH_PG_X(NCTV, NWV) { if (xt <= 4) D2T_LAUNCH_PG(4, NCTV, NWV) else if (xt <= 5) D2T_LAUNCH_PG(5, NCTV, NWV) else D2T_LAUNCH_PG(8, NCTV, NWV) }
    D2T_LAUNCH_PG_X(2, 4)
#undef D2T_LAUNCH_PG_X
#undef D2T_LAUNCH_PG
    rc = launch_status();
    if (rc != D2T_OK) return rc;
    const int HW = H * W;
    hipLaunchKernelGGL(k_psroipool_bwd_gather, dim3((HW + 255) / 256, nT * KK), dim3(256), 0, st, part, gin, nT, HW);
    return launch_status();
}

// ---------------------------------------------------------------------------------------
// PSROIPool backward, ROW form (round 5): the GEMM above without its pair-list and transposing pre-passes.  TWO launches (rows, gather)
// instead of four, no (49, R, 32) copy of gradOut.  Workgroup = task (cell row i, map row y), 7 waves, wave j = bin (i, j):
//   1. all 448 threads evaluate the row bounds of cell row i of the RoIs (ps_roipool_cuda.cu:36-54 through the same bin_axis as every other
//      kernel; four RoIs per thread in flight, ONE barrier) and compact the RoIs whose cell row i contains y into an LDS hit list, ascending;
//   2. per chunk of 32 hits: 224 threads evaluate the hits' column bounds and 1 / n; the hits' gradOut runs gradOut[r][t][i][0..6] -- 28
//      contiguous bytes per target, so every fetched line serves all seven bins -- go to LDS as A[j][hit][target], two lanes per run
//      (16 bytes each); the next chunk's loads are issued before this chunk's MFMAs;
//   3. wave j: D_j[t][x] += A_j[t][hit] * B_j[hit][x] with B = 1 / n inside the cell's columns, 0 outside
//      (v_mfma_f32_16x16x4_f32: M = targets, N = 16 map columns, K = 4 hits; a column tile no hit of a k-step reaches is skipped).
// Every plane row is written once as an ascending-RoI chain fma(gradOut, 1 / n, acc): deterministic; <= 1 ulp per term from the
// reference's gradOut / n (its atomics leave the order open).
// Measured (bench_ops.py, us; the four-launch GEMM in brackets): R = 300 nT = 21: 25 (31); R = 300 nT = 31: 32 (41); R = 3000 nT = 4: 38
// (59); R = 3000 nT = 31: 87 (71) -- there the dispatch keeps the GEMM.  Where a task's time goes (lab/tools/kstamps.py,
// profiles/r05_c_kstamps_ps_rows_*.txt, R = 3000 nT = 31, 130 k cycles): the k-steps 59 k (740 cycles each: a cell is ~4 columns wide,
// an MFMA column tile 16, and 7 waves share 4 matrix pipes), issuing the run loads 37 k (a wave pays ~20 cycles per line its
// instruction touches, and every (hit, target) run is its own line), the hit list 11 k.  Measured and dropped: the accumulation as an
// LDS read-add-write of the cell's own columns, lane = (target, column) -- 650 cycles per hit and wave in dependent LDS round trips,
// 158 us (profiles/r05_c_kstamps_ps_rows_lds_rmw_lost.txt).
// ---------------------------------------------------------------------------------------
constexpr int PR_WAVES = KT, PR_THREADS = PR_WAVES * 64;             // 448
constexpr int PR_EC = 32;                                            // hits per chunk with two c-tiles per wave (one c-tile: 64)
constexpr int PR_TP = 32;                                            // targets padded to two c-tiles
constexpr int PR_MAXHITS = 2048;                                     // LDS hit list (ints); more RoIs in a workgroup's range: further rounds
constexpr int PR_SCAN = 4;                                           // passes of 448 RoIs per round (1,792): 28 (pass, wave) counters, one wave-scan;
                                                                     // (two workgroups per CU need the kernel's static LDS below 80 KB)
static_assert(PR_SCAN * PR_WAVES <= 63 && PR_SCAN * PR_THREADS <= PR_MAXHITS, "hit scan");

// The RoIs' row / column bounds are evaluated HERE, by the thread that needs them (one bin_axis per RoI and scan pass, two per staged hit and
// bin column): a pre-pass that tabulated them per axis (rounds of this file before) cost a launch and a dependency -- 2 us at R = 300, 1 us at
// R = 3000 (profiles/r05_l_ps_rows_inline_bounds.txt) -- although each of the 266 tasks now repeats the f64 arithmetic of its bin row.
template <int XT, int NCT>
__global__ void __launch_bounds__(PR_THREADS)
k_ps_bwd_rows(const float* __restrict__ gout, const float* __restrict__ rois, float* __restrict__ part, int R, int nT, int H, int W, int nseg, int rps)
{
    __shared__ int hits[PR_MAXHITS];
    __shared__ int wcnt[64];                                         // hits per (pass, wave) of a round's scan
    // a chunk holds EC hits x TP targets: 32 x 32 with two c-tiles per wave, 64 x 16 with one (up to 16 targets) -- the same 28 KB per buffer,
    // half the chunk barriers where thousands of RoIs make a task walk many chunks
    constexpr int EC = NCT == 1 ? 2 * PR_EC : PR_EC, TP = NCT == 1 ? PR_TP / 2 : PR_TP;
    __shared__ __attribute__((aligned(16))) float A[2][KT][EC][TP];  // 2 x 28 KB
    __shared__ int ejb[2][EC][KT + 1];
    __shared__ float esc[2][EC][KT + 1];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // one task per workgroup, numbered from the middle rows outwards: they carry the most hits and start first
    // a task (map row y, bin row i) is dealt to nseg workgroups by RoI RANGE -- workgroup `seg` walks RoIs [seg rps, seg rps + rps) and
    // writes its own partial planes; the gather adds the partials in ascending order.  (Thousands of RoIs: one workgroup per task was
    // 266 workgroups walking ~290 hits each, one after the other, with the chip a third full.)
    const int seg = blockIdx.x % nseg, task = blockIdx.x / nseg, p = task / KT, i = task - p * KT, d = (p + 1) >> 1;
    const int r_lo = seg * rps, r_hi = r_lo + rps < R ? r_lo + rps : R;
    part += (size_t)seg * nT * KK * H * W;
    const int y = (p & 1) ? (H - 1) / 2 + d : (H - 1) / 2 - d;
    const int j = wave;                                              // this wave's bin column
    D2T_KSTAMP(0); D2T_KSTAMP_RT(14);
    const int bin = i * KT + j;
    f32x4 acc[NCT][XT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int x = 0; x < XT; ++x) acc[ct][x] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned gout_bytes = (unsigned)((size_t)R * nT * KK * 4);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gout), 0, gout_bytes, 0x00020000);

    for (int rbase = r_lo; rbase < r_hi; rbase += PR_SCAN * PR_THREADS) {   // one round unless more than 1,792 RoIs in the range
        const int rend = r_hi - rbase < PR_SCAN * PR_THREADS ? r_hi : rbase + PR_SCAN * PR_THREADS;
        // ---- 1. hit list of RoIs [rbase, rend): ascending.  Thread tid tests RoIs rbase + 448 p + tid, p < 4, with all its loads in
        // flight together; the 4 x 7 (pass, wave) counts meet at ONE barrier and every wave scans the 28 of them by itself.  (A barrier
        // pair per pass of 448 RoIs, as at first, was 1.6 k cycles per pass: 11 k cycles for 3000 RoIs.)
        if (rbase != r_lo) __syncthreads();                          // the previous round's counts and hits have been consumed
        unsigned long long mk[PR_SCAN];
        int inb = 0;
#pragma unroll
        for (int ps = 0; ps < PR_SCAN; ++ps) {
            const int r = rbase + ps * PR_THREADS + tid;
            bool in = false;
            if (r < rend) {
                int a0, a1;                                          // rows of bin row i of RoI r (ps_roipool_cuda.cu:36-54 through bin_axis)
                bin_axis<float>(rois[4 * (size_t)r] - rois[4 * (size_t)r + 2] / 2.0f, rois[4 * (size_t)r + 2] / static_cast<float>(KT), i, H, a0, a1);
                in = y >= a0 && y < a1;
            }
            mk[ps] = __ballot(in);
            inb |= in ? 1 << ps : 0;
            if (lane == 0) wcnt[ps * PR_WAVES + wave] = __builtin_popcountll(mk[ps]);
        }
        __syncthreads();
        const int mine = lane < PR_SCAN * PR_WAVES ? wcnt[lane] : 0;
        int inc = mine;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const int o = __shfl_up(inc, dd, 64);
            if (lane >= dd) inc += o;
        }
        const int exc = inc - mine;
        const int total = __builtin_amdgcn_readlane(inc, 63);        // uniform
#pragma unroll
        for (int ps = 0; ps < PR_SCAN; ++ps) {
            const int base = __builtin_amdgcn_readlane(exc, ps * PR_WAVES + wave);
            if (inb & (1 << ps)) hits[base + __builtin_popcountll(mk[ps] & ((1ull << lane) - 1ull))] = rbase + ps * PR_THREADS + tid;
        }
        __syncthreads();
        D2T_KSTAMP(1);
        // ---- 2 + 3. chunks of 32 hits
        const int nchunk = (total + EC - 1) / EC;
        // What a thread stages per chunk.  (hit e, bin column q) for tid < 224: column bounds and 1 / n from the axis tables.  The
        // runs gradOut[r][t][i][0..6] -- 28 contiguous bytes -- by TWO lanes each (bytes 0..15 and 12..27, one 16-byte load per lane):
        // a vector-memory instruction costs its wave ~15-20 cycles per line it touches, and this way 64 lanes touch 32 lines once
        // (seven dword loads per run touched every line seven times: 20 k cycles per chunk).
        constexpr int NLD = (2 * EC * TP + PR_THREADS - 1) / PR_THREADS;   // 5
        f32x4 run[NLD];
        int gjb = 0; float gsc = 0.f;
        auto load_chunk = [&](int c) {                               // global -> registers
            const int e0 = c * EC;
            if (tid < EC * KT) {
                const int e = tid / KT, q = tid - e * KT;
                gjb = 0; gsc = 0.f;
                if (e0 + e < total) {
                    const int r = hits[e0 + e];
                    int a0, a1, b0, b1;
                    bin_axis<float>(rois[4 * (size_t)r] - rois[4 * (size_t)r + 2] / 2.0f, rois[4 * (size_t)r + 2] / static_cast<float>(KT), i, H, a0, a1);
                    bin_axis<float>(rois[4 * (size_t)r + 1] - rois[4 * (size_t)r + 3] / 2.0f, rois[4 * (size_t)r + 3] / static_cast<float>(KT), q, W, b0, b1);
                    const int rb = a0 | (a1 << 16), cb = b0 | (b1 << 16);
                    const int hh = (rb >> 16) - (rb & 0xffff), ww = (cb >> 16) - (cb & 0xffff);
                    // first column | width << 8 | column-tile mask << 16; a cell that adds nothing (no rows, no columns) has width 0 and no tiles
                    const int j0c = cb & 0xffff, j1c = cb >> 16;
                    const bool any = hh > 0 && ww > 0;
                    gjb = any ? j0c | (ww << 8) | ((((2 << ((j1c - 1) >> 4)) - 1) & ~((1 << (j0c >> 4)) - 1)) << 16) : 0;
                    gsc = any ? 1.0f / static_cast<float>(hh * ww) : 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int sl = tid + k * PR_THREADS, rn = sl >> 1, half = sl & 1, e = rn / TP, t = rn - e * TP;
                const bool on = rn < EC * TP && e0 + e < total && t < nT;
                const int off = on ? ((hits[e0 + (on ? e : 0)] * nT + t) * KK + i * KT) * 4 + 12 * half : 0x7ffffff0;   // out of range: zeros
                run[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0));
            }
        };
        auto store_chunk = [&](int buf) {                            // registers -> LDS
            if (tid < EC * KT) {
                const int e = tid / KT, q = tid - e * KT;
                ejb[buf][e][q] = gjb; esc[buf][e][q] = gsc;
            }
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const int sl = tid + k * PR_THREADS, rn = sl >> 1, half = sl & 1, e = rn / TP, t = rn - e * TP;
                if (rn < EC * TP) {
                    if (half == 0) { A[buf][0][e][t] = run[k][0]; A[buf][1][e][t] = run[k][1]; A[buf][2][e][t] = run[k][2]; A[buf][3][e][t] = run[k][3]; }
                    else { A[buf][4][e][t] = run[k][1]; A[buf][5][e][t] = run[k][2]; A[buf][6][e][t] = run[k][3]; }
                }
            }
        };
        if (nchunk > 0) { load_chunk(0); store_chunk(0); }
        __syncthreads();
        D2T_KSTAMP(2);
        D2T_KSTAMP_ONLY(unsigned long long k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0, tl = 0, tm = 0, ts = 0, tb = 0;)
        for (int c = 0; c < nchunk; ++c) {
            const int buf = c & 1;
            D2T_KCLK(k0);
            if (c + 1 < nchunk) load_chunk(c + 1);                   // in flight under the MFMAs
            D2T_KCLK(k1);
            const int ne = total - c * EC < EC ? total - c * EC : EC;
            const int nks = (ne + 3) >> 2;                           // uniform
            // k-steps, software-pipelined: the entry (column bounds, 1 / n, the A values of this lane's targets) of k-step ks+1 is read
            // from LDS before the MFMAs of ks; which column tiles a k-step reaches is ONE scalar mask -- the OR of its four hits' tile
            // ranges, read across the four lane groups -- instead of a ballot and a branch per tile
            struct KOp { int jb; float sc; float a[NCT]; };
            auto kfetch = [&](KOp& o, int ks) {
                const int e = 4 * ks + g;
                o.jb = ejb[buf][e][j];
                o.sc = esc[buf][e][j];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) o.a[ct] = A[buf][j][e][16 * ct + n];
            };
            // one k-step: the lane's membership test is (unsigned)(column - first column) < width (false for every column of a width-0 entry),
            // the scale sits in the B operand; the tile mask of the k-step is the OR of its four entries' masks, read across the lane groups
            // as scalars -- no vector instruction.  (Vector instructions are matrix time here: lab/csrc/mfma_valu_lab.)
            auto kstep = [&](const KOp& o) {
                const int d0 = n - (o.jb & 0xff);
                const unsigned wd = ((unsigned)o.jb >> 8) & 0xffu;
                const int tm2 = (__builtin_amdgcn_readlane(o.jb, 0) | __builtin_amdgcn_readlane(o.jb, 16) | __builtin_amdgcn_readlane(o.jb, 32) |
                                 __builtin_amdgcn_readlane(o.jb, 48)) >> 16;
#pragma unroll
                for (int x = 0; x < XT; ++x) {
                    if (!(tm2 & (1 << x))) continue;                 // no hit of this k-step reaches the column tile (scalar test)
                    const float b = (unsigned)(d0 + 16 * x) < wd ? o.sc : 0.f;
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) acc[ct][x] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[ct], b, acc[ct][x], 0, 0, 0);
                }
            };
            KOp ka, kb;                                              // two entries that swap roles: no register moves between k-steps
            kfetch(ka, 0);
            for (int ks = 0; ks < nks; ks += 2) {
                if (ks + 1 < nks) kfetch(kb, ks + 1);
                kstep(ka);
                if (ks + 1 >= nks) break;
                if (ks + 2 < nks) kfetch(ka, ks + 2);
                kstep(kb);
            }
            D2T_KCLK(k2);
            if (c + 1 < nchunk) store_chunk(buf ^ 1);
            D2T_KCLK(k3);
            __syncthreads();
            D2T_KCLK(k4);
            D2T_KSTAMP_ONLY(tl += k1 - k0; tm += k2 - k1; ts += k3 - k2; tb += k4 - k3;)
        }
        D2T_KSTAMP_PUT(5, tl); D2T_KSTAMP_PUT(6, tm); D2T_KSTAMP_PUT(7, ts); D2T_KSTAMP_PUT(8, tb); D2T_KSTAMP_PUT(9, (unsigned long long)nchunk);
    }
    D2T_KSTAMP(3);
    // ---- planes: D[m = target 16 ct + 4 g + r][n = column 16 x + n]
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int x = 0; x < XT; ++x) {
            const f32x4 v = acc[ct][x];
            const int col = 16 * x + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = 16 * ct + 4 * g + r;
                if (t < nT && col < W) part[(((size_t)t * KK + bin) * H + y) * W + col] = v[r];
            }
            // Cold path, as in k_roipool_bwd_gemm: a non-finite gradOut value times a 0 weight poisons columns outside its cell
            // (the reference, ps_roipool_cuda.cu:131-139, only adds to the cell's own pixels); the wave recomputes the 16 x 16
            // tile it stored with exact membership from the RoIs themselves.
            if (__builtin_expect(__any(pool_nonfinite4(v)), 0)) {
                for (int e = lane; e < 256; e += 64) {
                    const int t = 16 * ct + (e >> 4), xx = 16 * x + (e & 15);
                    if (t >= nT || xx >= W) continue;
                    float a = 0.f;
                    for (int r = r_lo; r < r_hi; ++r) {
                        const Bounds cb = psroi_cell<float>(rois + 4 * (size_t)r, i, j, H, W, KT);
                        const int nn = (cb.i1 - cb.i0) * (cb.j1 - cb.j0);
                        if (y >= cb.i0 && y < cb.i1 && xx >= cb.j0 && xx < cb.j1)
                            a = __builtin_fmaf(gout[((size_t)r * nT + t) * KK + bin], 1.0f / static_cast<float>(nn), a);
                    }
                    part[(((size_t)t * KK + bin) * H + y) * W + xx] = a;
                }
            }
        }
    D2T_KSTAMP(4); D2T_KSTAMP_RT(15);
}

static bool psroipool_bwd_rows_supported(int R, int nT, int H, int W, int k)
{
    return k == KT && R >= 1 && nT >= 1 && nT <= PR_TP && H >= 1 && H <= 32767 && W >= 1 && W <= 128 &&
           1LL * R * nT * KK * 4 < 0x7ffffff0LL && 1LL * nT * KK * H * W < 0x7fffffffLL && 1LL * KT * H < 0x7fffffffLL;
}

// RoI ranges per task.  Every workgroup pays ~20 k cycles whatever it walks (hit scan, first chunk, plane stores) and two fit a CU, so
// a range only pays where a task's walk is long: thousands of RoIs.  lab/tools/ps_segs_scan.py, 38 x 75, us by ranges 1 / 2 / 3 / 4 (scan
// build): R 3000 nT 4  58 / 44 / 51 / 49, nT 8  60 / 47 / 55 / 55, nT 16  66 / 52 / 65 / 64, nT 31  87 / 93 / 114 / 114;
// R 1000 nT 4  27 / 25 / 31 / 33, nT 16  35 / 36 / 44 / 49; R 300 nT 21  31 / 41 / 51 / 63 (profiles/r05_e_ps_rows_roi_ranges.txt).
static int ps_rows_segs(int R, int nT)
{
    const int forced = lab_env_int("D2T_PS_SEGS", 0);                // scan builds only
    if (forced > 0) return forced;
    return R >= 1400 && nT <= 16 ? 2 : 1;
}

// workspace: partial planes (segs, nT*49, H*W)
static size_t psroipool_bwd_rows_ws_bytes(int R, int nT, int H, int W, int k)
{
    return psroipool_bwd_rows_supported(R, nT, H, W, k) ? ps_rows_segs(R, nT) * align256((size_t)nT * KK * H * W * 4) : 0;
}

static int psroipool_bwd_rows_f32(const float* gout, const float* rois, float* gin, int R, int nT, int H, int W, void* ws, hipStream_t st)
{
    char* w = static_cast<char*>(ws);
    const int nseg = ps_rows_segs(R, nT), rps = (R + nseg - 1) / nseg;
    float* part = reinterpret_cast<float*>(w);
    const int xt = (W + 15) / 16, ntasks = KT * H;
#define D2T_LAUNCH_PR(XTV, NCTV) hipLaunchKernelGGL((k_ps_bwd_rows<XTV, NCTV>), dim3(ntasks * nseg), dim3(PR_THREADS), 0, st, gout, rois, part, R, nT, H, W, nseg, rps)
#define D2T_LAUNCH_PR_X(NCTV) { if (xt <= 4) D2T_LAUNCH_PR(4, NCTV); else if (xt <= 5) D2T_LAUNCH_PR(5, NCTV); else D2T_LAUNCH_PR(8, NCTV); }
    if (nT <= 16) D2T_LAUNCH_PR_X(1) else D2T_LAUNCH_PR_X(2)
#undef D2T_LAUNCH_PR_X
#undef D2T_LAUNCH_PR
    int rc = launch_status();
    if (rc != D2T_OK) return rc;
    const int HW = H * W;
    const int gx_all = (HW + 255) / 256, gx_knob = lab_env_int("D2T_PS_GATHER_GX", 0);   // scan builds: workgroups per channel
    const int gx = gx_knob > 0 && gx_knob < gx_all ? gx_knob : gx_all;
    hipLaunchKernelGGL(k_psroipool_bwd_gather, dim3(gx, nT * KK), dim3(256), 0, st, part, gin, nT, HW, nseg);
    return launch_status();
}

// The TILE form (one launch, the planes of a (map row, 16 columns) tile kept in LDS and gathered there: lab/csrc/d2t_ps_bwd_tiles.inc) was
// built and measured in round 6 and lost to the row form above (29.7 against 23.6 us at R = 300, nT = 21; profiles/r06_ps_bwd_tiles_lost.txt).
// Scan builds (-DD2T_ENV_KNOBS) still compile it: D2T_PS_BWD=tiles.
#ifdef D2T_ENV_KNOBS
#include "../../lab/csrc/d2t_ps_bwd_tiles.inc"
#else
static bool psroipool_bwd_tiles_supported(int, int, int, int, int) { return false; }
static int psroipool_bwd_tiles_f32(const float*, const float*, float*, int, int, int, int, hipStream_t) { return D2T_EINVAL; }
#endif

// Which of the three backward designs runs (ps_bwd_design below).  History of the sorted lists vs the planes:
// the sorted-corner-list kernels (d2t_pool_sorted.hip) do
// work proportional to the RoI corners per plane (4R) plus a fixed cost of three launches and a
// 49-workgroup sort; the plane kernels above walk every RoI's rows.  Measured crossover on MI355X
// (38x75 map, R in 300..3000 x nT in 4..31, lab/tools/ps_scan.py): the sorted design wins from 16 targets
// up once R * nT reaches ~16000 (R=1000 nT=16: 68 vs 89 us; R=3000 nT=31: 158 vs 277 us) and loses
// below 8 targets at every R (R=3000 nT=4: 94 vs 59 us).
// D2T_PS_BWD=planes|sorted|gemm overrides the choice (a lab knob for that measurement, read once).
constexpr long long PS_SORTED_MIN_WORK = 14000;
constexpr int PS_SORTED_MIN_TARGETS = 12;

static int ps_bwd_forced()
{
    static const int v = [] {
        const char* e = lab_env_str("D2T_PS_BWD");                       // -DD2T_LAB only
        if (!e) return 0;
        return !strcmp(e, "planes") ? 1 : !strcmp(e, "sorted") ? 2 : !strcmp(e, "gemm") ? 3 : !strcmp(e, "rows") ? 4 : !strcmp(e, "tiles") ? 5 : 0;
    }();
    return v;
}

// 0 = plane kernels, 1 = sorted corner lists, 2 = GEMM behind three pre-passes, 3 = row form (one launch + gather), 4 = tile form (ONE launch)
static int ps_bwd_design(int R, int nT, int H, int W, int k)
{
    const bool g = psroipool_bwd_gemm_supported(R, nT, H, W, k), s = psroipool_bwd_sorted_supported(R, nT, H, W, k),
               p = psroipool_bwd_planes_supported(R, nT, H, W, k), rw = psroipool_bwd_rows_supported(R, nT, H, W, k),
               tl = psroipool_bwd_tiles_supported(R, nT, H, W, k);
    const int f = ps_bwd_forced();
    if (f == 5 && tl) return 4;
    if (f == 4 && rw) return 3;
    if (f == 3 && g) return 2;
    if (f == 2 && s) return 1;
    if (f == 1 && p) return 0;
    // measured grid R in {300..3000} x nT in {4..31} on a 38x75 map (lab/tools/ps_scan.py, profiles/r02_b_ps_bwd_scan_*):
    // the GEMM wins from 12 targets up at every R (R=300 nT=16: 33 vs 40 us; R=3000 nT=31: 73 vs 148 sorted / 278 planes)
    // and from 8 targets at R >= 1000; the plane kernels keep the small shapes (R=300 nT=4: 15 vs 25 us)
    if (rw && f == 0 && !(g && nT > 16 && R >= 1500)) return 3;        // round 5: the row form -- except more than 16 targets (two c-tiles per wave) x thousands of RoIs
                                                                       // (R 3000 nT 31: 87 against 71 us, R 1500 nT 31: 57 against 53; R 3000 nT 16: 52 against 65)
    if (g && (nT >= 12 || (nT >= 8 && R >= 1000) || !p)) return 2;
    if (s && (!p || (nT >= PS_SORTED_MIN_TARGETS && 1LL * R * nT >= PS_SORTED_MIN_WORK))) return 1;
    return p ? 0 : (s ? 1 : 0);
}

bool psroipool_bwd_supported(int R, int nT, int H, int W, int k)
{
    return psroipool_bwd_tiles_supported(R, nT, H, W, k) || psroipool_bwd_rows_supported(R, nT, H, W, k) || psroipool_bwd_gemm_supported(R, nT, H, W, k) ||
           psroipool_bwd_sorted_supported(R, nT, H, W, k) || psroipool_bwd_planes_supported(R, nT, H, W, k);
}

size_t psroipool_bwd_ws_bytes(int R, int nT, int H, int W, int k)
{
    const int d = ps_bwd_design(R, nT, H, W, k);
    if (d == 4) return 0;
    if (d == 3) return psroipool_bwd_rows_ws_bytes(R, nT, H, W, k);
    return d == 2 ? psroipool_bwd_gemm_ws_bytes(R, nT, H, W, k) : d == 1 ? psroipool_bwd_sorted_ws_bytes(R, nT, H, W, k)
                                                                         : psroipool_bwd_planes_ws_bytes(R, nT, H, W, k);
}

int psroipool_bwd_f32(const float* gout, const float* rois, float* gin, int R, int nT, int H, int W, int k,
                      void* ws, hipStream_t st)
{
    const int d = ps_bwd_design(R, nT, H, W, k);
    if (d == 4) return psroipool_bwd_tiles_f32(gout, rois, gin, R, nT, H, W, st);
    if (d == 3) return psroipool_bwd_rows_f32(gout, rois, gin, R, nT, H, W, ws, st);
    if (d == 2) return psroipool_bwd_gemm_f32(gout, rois, gin, R, nT, H, W, ws, st);
    if (d == 1) return psroipool_bwd_sorted_f32(gout, rois, gin, R, nT, H, W, k, ws, st);
    return psroipool_bwd_planes_f32(gout, rois, gin, R, nT, H, W, k, ws, st);
}

}}  // namespace d2t::tuned
