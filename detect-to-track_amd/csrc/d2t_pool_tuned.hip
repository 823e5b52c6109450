// d2t_pool_tuned.hip -- gfx950-tuned f32 ROIPool / PSROIPool FORWARD kernels (k = 7).
//
// The pooling ops are byte movers (ROIPool config 3: 60 MB of output against 10 MB of input and
// ~0 flops; PSROIPool at the model's shapes: 17 MB of map against 2 MB of output).  Round 1 gathered
// every bin's pixels from L2 / L1 and scattered every gradient into its pixels; both were bound by
// the number of memory and LDS instructions, 10-60x off the HBM roof.  This version keeps the
// feature map of a few channels resident in LDS and touches every RoI bin a constant number of times:
//
//  * ROIPool forward: a workgroup builds the SUMMED-AREA TABLE of 1-4 channels in LDS (f64: exact to
//    2^-53 of the table's magnitude) and every output is four table look-ups, whatever the bin's
//    size.  The reference sums a bin in f32 in row-major order (roipool_cuda.cu:56-61); the table
//    gives the exactly rounded sum instead, so values agree to f32 rounding of the reference's own
//    running sum (tests: |delta| <= 1e-5 abs/rel, the contract of BASELINE.json) -- the type-generic
//    kernel remains the bit-exact form.  Empty bins give 0/0 = NaN like the reference; a non-finite
//    result (Inf / NaN somewhere in the channel poisons the table) is recomputed in the reference's
//    form from global memory.
//  * PSROIPool forward keeps the reference's running sum (ps_roipool_cuda.cu:60-66): a workgroup
//    stages the SEVEN input channels of one (target, bin row) in LDS, lane = RoI (RoIs ordered by cell
//    size), and writes out[r][t][7i..7i+6] directly (round 3; the round-2 form -- one channel per
//    workgroup, a (t*49+bin, RoI) buffer and a transposing pass -- remains for maps whose seven
//    channels do not fit in LDS and for more than 4096 RoIs).  Bit-identical to the reference.
//  The backward kernels live in d2t_pool_bwd.hip.
#include <cstdlib>
#include "d2t_kernels.hpp"
#include "d2t_tuned.hpp"
#include "d2t_pool_common.hpp"

namespace d2t { namespace tuned {

D2T_KSTAMP_DEFINE(d2t_lab_pool_fwd_stamps)

// ---------------------------------------------------------------------------------------
// (rows, cols) -> (cols, rows) transpose of a row-major f32 matrix, 32x32 tiles through LDS;
// the tiles are walked by a 1-D grid (no 65535 limit on either extent).
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_transpose(const float* __restrict__ in, float* __restrict__ out, int rows, int cols, int tiles_c, long long ntiles)
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
            if (r < rows && c < cols) out[(size_t)c * rows + r] = tile[tx][ty + 8 * k];
        }
        __syncthreads();
    }
}

int transpose(const float* in, float* out, int rows, int cols, hipStream_t st)
{
    if (rows == 0 || cols == 0) return D2T_OK;
    const int tiles_c = (cols + 31) / 32;
    const long long ntiles = 1LL * tiles_c * ((rows + 31) / 32);
    const int grid = (int)(ntiles < 256 * 64 ? ntiles : 256 * 64);
    hipLaunchKernelGGL(k_transpose, dim3(grid), dim3(256), 0, st, in, out, rows, cols, tiles_c, ntiles);
    return launch_status();
}

// ---------------------------------------------------------------------------------------
// ROIPool geometry record (LDS): per RoI 32 bytes = {(i0,i1)[7], (j0,j1)[7], 4 pad}: byte 2i = i0 of
// bin row i, 2i+1 = its i1, byte 14+2j = j0 of bin column j, 15+2j = its j1 (one 16-bit load per
// axis).  Row bounds of a bin depend on i only, column bounds on j only (roipool_cuda.cu:41-50), so
// 28 numbers describe all 49 bins; every bound lies in [0, 255] (maps above 255 rows or columns take
// the generic kernels).
// ---------------------------------------------------------------------------------------
constexpr int GEO8 = 32;

// ---------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------
// ROIPool forward: summed-area tables in LDS.
// Workgroup = (CG consecutive channels, a share of the RoIs), 1024 threads.
//   sat[c][y][x] (f64) = sum of FM[c][0..y-1][0..x-1]; row pitch LD is odd (rows land on different
//   banks), row 0 and column 0 are zero.  Built by a coalesced load and prefix2d.
//   An output is (S(i1,j1) - S(i0,j1)) - (S(i1,j0) - S(i0,j0)) rounded to f32 times 1/n (v_rcp_f32:
//   1 ulp; 0 * inf = NaN for an empty bin as the reference's 0/0, roipool_cuda.cu:61).  980 threads
//   work: thread t owns element t mod (CG*49) of a RoI's output run for good -- its (channel, bin
//   row, bin column) never change, no index arithmetic in the loop -- and walks the RoIs 980/(CG*49)
//   at a time.  Consecutive threads produce consecutive elements of out: contiguous runs of CG*196
//   bytes per RoI.
// ---------------------------------------------------------------------------------------
constexpr int RF_THREADS = 1024;
constexpr int RF_ACTIVE = 980;                // 20 * 49 = 10 * 98 = 5 * 196
constexpr int RF_RC = 240;                    // RoIs whose geometry is staged in LDS at a time (a multiple of 20)

struct SatLayout { int LD, plane; size_t bytes; };
inline SatLayout sat_layout(int CG, int H, int W, int geo = GEO8)
{
    SatLayout L;
    L.LD = (W + 1) | 1;
    L.plane = (H + 1) * L.LD;
    L.bytes = (size_t)CG * L.plane * 8 + prefix_scratch_bytes(CG, H, W) + (size_t)RF_RC * geo;
    return L;
}

template <int CG>
__global__ void __launch_bounds__(RF_THREADS)
k_roipool_fwd_sat(const float* __restrict__ fm, const float* __restrict__ rois, float* __restrict__ out,
                  int R, int C, int H, int W, int LD, int plane, int rois_per_wg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double* sat = reinterpret_cast<double*>(lds_raw);                // [CG][H+1][LD]
    double* scr = sat + (size_t)CG * plane;                          // prefix scratch
    uint32_t* geoL = reinterpret_cast<uint32_t*>(scr + prefix_scratch_doubles(CG, H, W));   // [RF_RC][8]
    const int tid = threadIdx.x, HW = H * W;
    const int c0 = blockIdx.x * CG;
    const int r_lo = blockIdx.y * rois_per_wg, r_hi = r_lo + rois_per_wg < R ? r_lo + rois_per_wg : R;

    // ---- load: sat[c][y+1][x+1] = FM, zero borders.  The CG channels are one contiguous run of
    // CG*H*W floats; every thread issues its (up to 4 x n) loads before the first use.
    for (int e = tid; e < CG * (H + 1); e += RF_THREADS) sat[(size_t)(e / (H + 1)) * plane + (e % (H + 1)) * LD] = 0.0;
    for (int e = tid; e < CG * LD; e += RF_THREADS) sat[(size_t)(e / LD) * plane + e % LD] = 0.0;
    {
        // a wave takes map rows (one coalesced run each), four rows' loads in flight; no per-element
        // index arithmetic
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nrow = CG * H;
        constexpr int NWV = RF_THREADS / 64;
        for (int row0 = wave; row0 < nrow; row0 += 4 * NWV) {
            for (int xb = 0; xb < W; xb += 64) {
                const int x = xb + lane;
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = row0 + k * NWV, c = row / H, y = row - c * H;
                    v[k] = row < nrow && x < W && c0 + c < C ? fm[(size_t)(c0 + c) * HW + y * W + x] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = row0 + k * NWV, c = row / H, y = row - c * H;
                    if (row < nrow && x < W) sat[(size_t)c * plane + (y + 1) * LD + x + 1] = (double)v[k];
                }
            }
        }
    }
    __syncthreads();
    prefix2d(sat + LD + 1, scr, CG, H, W, LD, plane, tid, RF_THREADS);

    // ---- outputs
    constexpr int TPR = CG * KK, RPP = RF_ACTIVE / TPR;              // threads per RoI, RoIs per pass
    const bool worker = tid < RF_ACTIVE;
    const int rr0 = tid / TPR, q = tid - rr0 * TPR, c = q / KK, bin = q - c * KK, i = bin / KT, j = bin - i * KT;
    const bool live = worker && c0 + c < C;
    const double* S = sat + (size_t)c * plane;
    const unsigned char* geoB = reinterpret_cast<const unsigned char*>(geoL) + 2 * i;
    const int joff = 2 * KT + 2 * j - 2 * i;
    for (int rb = r_lo; rb < r_hi; rb += RF_RC) {
        const int rc = r_hi - rb < RF_RC ? r_hi - rb : RF_RC;
        __syncthreads();                                             // previous records consumed
        unsigned char* geoW = reinterpret_cast<unsigned char*>(geoL);
        for (int e = tid; e < rc * KT; e += RF_THREADS) {            // bin row q and bin column q of RoI rb + e/7
            const int rr = e / KT, qq = e - rr * KT;
            const Bounds bq = roi_bin<float>(rois + 4 * (size_t)(rb + rr), qq, qq, H, W, KT);
            *reinterpret_cast<unsigned short*>(geoW + rr * GEO8 + 2 * qq) = (unsigned short)(bq.i0 | (bq.i1 << 8));
            *reinterpret_cast<unsigned short*>(geoW + rr * GEO8 + 2 * KT + 2 * qq) = (unsigned short)(bq.j0 | (bq.j1 << 8));
        }
        __syncthreads();
        if (!live) continue;
        float* dst = out + ((size_t)(rb + rr0) * C + c0) * KK + q;
        const size_t dstep = (size_t)RPP * C * KK;
        // two RoIs per iteration: two independent chains of LDS round trips per thread
        auto pool = [&](int rr, float* d, bool on) {
            const unsigned char* gr = geoB + (on ? rr : rr0) * GEO8;
            const unsigned pi = *reinterpret_cast<const unsigned short*>(gr);
            const unsigned pj = *reinterpret_cast<const unsigned short*>(gr + joff);
            const int i0 = pi & 255, i1 = pi >> 8, j0 = pj & 255, j1 = pj >> 8;
            double s = (S[i1 * LD + j1] - S[i0 * LD + j1]) - (S[i1 * LD + j0] - S[i0 * LD + j0]);
            if (i1 <= i0 || j1 <= j0) s = 0.0;                       // the reference's loops do not run
            const int n = (i1 - i0) * (j1 - j0);
            float res = (float)s * __builtin_amdgcn_rcpf((float)n);  // n == 0: 0 * inf = NaN, as 0/0 in roipool_cuda.cu:61
            if (on && n > 0 && !(__builtin_fabsf(res) <= 3.4028234663852886e38f)) {
                // non-finite: an Inf / NaN anywhere above-left of the bin poisons the table.  Redo
                // this bin in the reference's form (running f32 sum, row-major) from global memory.
                const float* chp = fm + (size_t)(c0 + c) * HW;
                float acc = 0.f;
                for (int y = i0; y < i1; ++y)
                    for (int x = j0; x < j1; ++x) acc += chp[y * W + x];
                res = acc / (float)n;
            }
            if (on) *d = res;
        };
        for (int rr = rr0; rr < rc; rr += 2 * RPP, dst += 2 * dstep) {
            pool(rr, dst, true);
            pool(rr + RPP, dst + dstep, rr + RPP < rc);
        }
    }
}

// ---------------------------------------------------------------------------------------
// The same kernel for TWO channels with their tables INTERLEAVED element by element: sat2[y][x][c].  A thread
// owns a bin and produces both channels: a corner is ONE ds_read_b128 for two outputs instead of two
// ds_read_b64 -- half the LDS instructions of the look-up phase, and 16-byte accesses at RoI-dependent
// addresses conflict less than 8-byte ones (the planar kernel's SQ_LDS_BANK_CONFLICT is 52 % of its LDS time).
// Round 5, what the workgroup's clocks say (lab/tools/kstamps.py, profiles/r05_c_kstamps_roipool_fwd_*.txt; config 3, cycles per workgroup):
// planes into LDS 8.6 k, prefix2d 8.0 k, geometry of 240 RoIs 4.1 k, look-ups 8.1 k, geometry of the last 60 5.4 k, look-ups 3.0 k =
// 37.7 k = 16 us, two workgroups per CU (512 in all).  The stamped build runs them one at a time, the product kernel (60 VGPRs) two at a time --
// lab/csrc/occ_lab: a CU holds two 1,024-thread workgroups of 56 KB up to 64 VGPRs -- which changes little: prefix and look-ups are LDS-bound and
// co-resident workgroups share the LDS pipe.  Measured and dropped: the geometry of all RoIs evaluated under the plane loads' latency (workgroup
// 37.7 k -> 29.7 k cycles, but the op 31.4 -> 34.2 us); the 2-D prefix as wave-wide f64 row scans (12.7 k cycles against 8.0 k); workgroups of
// 832 / 768 / 640 / 512 threads (33.4 / 31.4 / 33.1 / 35.5 us against 31.8).
// ---------------------------------------------------------------------------------------
typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int S2_THREADS = 1024;             // 56 KB of LDS per workgroup, 60 VGPRs: two workgroups per CU
constexpr int S2_ACTIVE = 980;               // 20 RoIs x 49 bins per pass
constexpr int S2_MAXK = 16;                  // run-time bin counts the interleaved kernel takes (4k bytes of geometry per RoI <= 64)

// KTT = 7: the model's bin count, the thread -> (RoI slot, bin) map in compile-time constants; KTT = 0: any k <= S2_MAXK, the same
// map computed once per thread from the run-time k (4k bytes of geometry per RoI instead of 32).
template <int KTT>
__global__ void __launch_bounds__(S2_THREADS)
k_roipool_fwd_sat2(const float* __restrict__ fm, const float* __restrict__ rois, float* __restrict__ out,
                   int R, int C, int H, int W, int LD, int plane, int rois_per_wg, int k_rt)
{
    const int kt = KTT ? KTT : k_rt, kk = kt * kt;
    const int geo = KTT ? GEO8 : 4 * kt;                             // bytes of geometry per RoI
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double* sat = reinterpret_cast<double*>(lds_raw);                // [H+1][LD][2]
    double* scr = sat + (size_t)2 * plane;                           // prefix scratch
    uint32_t* geoL = reinterpret_cast<uint32_t*>(scr + prefix_scratch_doubles(2, H, W));   // [RF_RC][8]
    const int tid = threadIdx.x, HW = H * W;
    const int c0 = blockIdx.x * 2;
    const int r_lo = blockIdx.y * rois_per_wg, r_hi = r_lo + rois_per_wg < R ? r_lo + rois_per_wg : R;
    D2T_KSTAMP(0);

    for (int e = tid; e < 2 * (H + 1); e += S2_THREADS) sat[(size_t)(e >> 1) * LD * 2 + (e & 1)] = 0.0;   // column 0
    for (int e = tid; e < 2 * LD; e += S2_THREADS) sat[e] = 0.0;                                           // row 0
    {
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nrow = 2 * H;
        constexpr int NWV = S2_THREADS / 64;
        for (int row0 = wave; row0 < nrow; row0 += 4 * NWV) {
            for (int xb = 0; xb < W; xb += 64) {
                const int x = xb + lane;
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = row0 + k * NWV, c = row / H, y = row - c * H;
                    v[k] = row < nrow && x < W && c0 + c < C ? fm[(size_t)(c0 + c) * HW + y * W + x] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = row0 + k * NWV, c = row / H, y = row - c * H;
                    if (row < nrow && x < W) sat[((size_t)(y + 1) * LD + x + 1) * 2 + c] = (double)v[k];
                }
            }
        }
    }
    __syncthreads();
    D2T_KSTAMP(1);
    prefix2d(sat + (size_t)(LD + 1) * 2, scr, 2, H, W, LD, 1, tid, S2_THREADS, 2);
    D2T_KSTAMP(2);

    // ---- outputs: thread t < 980 owns bin t mod 49 of RoI slot t / 49 (20 RoIs per pass), both channels
    const int RPP = KTT ? S2_ACTIVE / KK : S2_THREADS / kk;          // RoIs per pass
    const bool worker = tid < RPP * kk;
    const int rr0 = tid / kk, bin = tid - rr0 * kk, i = bin / kt, j = bin - i * kt;
    const bool live0 = worker && c0 < C, live1 = worker && c0 + 1 < C;
    const f64x2* S = reinterpret_cast<const f64x2*>(sat);
    const unsigned char* geoB = reinterpret_cast<const unsigned char*>(geoL) + 2 * i;
    const int joff = 2 * kt + 2 * j - 2 * i;
    for (int rb = r_lo; rb < r_hi; rb += RF_RC) {
        const int rc = r_hi - rb < RF_RC ? r_hi - rb : RF_RC;
        __syncthreads();                                             // previous records consumed
        unsigned char* geoW = reinterpret_cast<unsigned char*>(geoL);
        for (int e = tid; e < rc * kt; e += S2_THREADS) {            // bin row q and bin column q of RoI rb + e/k
            const int rr = e / kt, qq = e - rr * kt;
            const Bounds bq = roi_bin<float>(rois + 4 * (size_t)(rb + rr), qq, qq, H, W, kt);
            *reinterpret_cast<unsigned short*>(geoW + rr * geo + 2 * qq) = (unsigned short)(bq.i0 | (bq.i1 << 8));
            *reinterpret_cast<unsigned short*>(geoW + rr * geo + 2 * kt + 2 * qq) = (unsigned short)(bq.j0 | (bq.j1 << 8));
        }
        __syncthreads();
        D2T_KSTAMP(rb == r_lo ? 3 : 5);
        if (!live0) continue;
        float* dst = out + ((size_t)(rb + rr0) * C + c0) * kk + bin;
        const size_t dstep = (size_t)RPP * C * kk;
        auto pool = [&](int rr, float* d, bool on) {
            const unsigned char* gr = geoB + (on ? rr : rr0) * geo;
            const unsigned pi = *reinterpret_cast<const unsigned short*>(gr);
            const unsigned pj = *reinterpret_cast<const unsigned short*>(gr + joff);
            const int i0 = pi & 255, i1 = pi >> 8, j0 = pj & 255, j1 = pj >> 8;
            const f64x2 a = S[i1 * LD + j1], b2 = S[i0 * LD + j1], c2 = S[i1 * LD + j0], d2 = S[i0 * LD + j0];
            const bool empty = i1 <= i0 || j1 <= j0;                 // the reference's loops do not run
            const int n = (i1 - i0) * (j1 - j0);
            const float rn = __builtin_amdgcn_rcpf((float)n);        // n == 0: 0 * inf = NaN, as 0/0 in roipool_cuda.cu:61
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double s = empty ? 0.0 : (a[c] - b2[c]) - (c2[c] - d2[c]);
                float res = (float)s * rn;
                if (on && n > 0 && !(__builtin_fabsf(res) <= 3.4028234663852886e38f) && (c == 0 || live1)) {
                    // non-finite: an Inf / NaN anywhere above-left of the bin poisons the table.  Redo
                    // this bin in the reference's form (running f32 sum, row-major) from global memory.
                    const float* chp = fm + (size_t)(c0 + c) * HW;
                    float acc = 0.f;
                    for (int y = i0; y < i1; ++y)
                        for (int x = j0; x < j1; ++x) acc += chp[y * W + x];
                    res = acc / (float)n;
                }
                if (on && (c == 0 || live1)) d[c * kk] = res;
            }
        };
        for (int rr = rr0; rr < rc; rr += 2 * RPP, dst += 2 * dstep) {
            pool(rr, dst, true);
            pool(rr + RPP, dst + dstep, rr + RPP < rc);
        }
        D2T_KSTAMP(rb == r_lo ? 4 : 6);
    }
    D2T_KSTAMP(7);
}

static int sat_cg(int C, int H, int W)
{
    // Two channels per workgroup: 45 KB of LDS, so two workgroups share a CU and one's table build
    // overlaps the other's look-ups (config 3: 37.5 us; 4 channels = one workgroup per CU: 40.7; 1: 47.1).
    // D2T_SAT_CG=1|2|4 overrides (lab knob for that measurement, read once).
    static const int top = [] { const int v = lab_env_int("D2T_SAT_CG", 2); return v == 1 || v == 4 ? v : 2; }();   // -DD2T_LAB only
    for (int cg = top; cg >= 1; cg >>= 1)
        if (sat_layout(cg, H, W).bytes <= (size_t)LDS_MAX && (cg == 1 || C >= 2 * cg)) return cg;
    return sat_layout(1, H, W).bytes <= (size_t)LDS_MAX ? 1 : 0;
}

// k = 7: every channel grouping; other k <= S2_MAXK: the interleaved two-channel kernel with the bin count at run time
static bool sat2_anyk(int C, int H, int W, int k)
{
    return k >= 1 && k <= S2_MAXK && sat_layout(2, H, W, 4 * k > GEO8 ? 4 * k : GEO8).bytes <= (size_t)LDS_MAX;
}

// Round 5, measured and dropped: a forward for a HANDFUL of RoIs (the tracker pools ~8 boxes of a 1891-channel map) that stages the
// boxes of 16 channels through LDS with coalesced row loads and adds the bins from LDS in the reference's order (bit-identical): 25.2 us at
// R = 8 against 21.7 us for the thread-per-output kernel that capi.hip keeps for R < 32 (profiles/r05_e_roipool_few_rois_lost.txt: with every
// workgroup resident the box loads alone take 24 k cycles -- the op is bound by first-touch fills of partly used lines, not by its lanes).
// Nor did a thread-per-output kernel whose bin rows are fetched by eight independent loads before they are added (20.2 against 21.9 us at R = 8,
// 28.9 against 26.0 at R = 16): the generic kernel is not waiting on a dependent chain either.
#ifdef D2T_ENV_KNOBS
// scan builds: what the runtime says about the residency of the interleaved summed-area kernel (lab/tools/kstamps.py roipool_occupancy)
extern "C" int d2t_lab_roipool_fwd_occupancy(int H, int W, int threads, int* blocks, int* regs, int* static_lds, int* dyn_lds)
{
    const SatLayout L = sat_layout(2, H, W, GEO8);
    hipFuncAttributes fa;
    hipError_t e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_roipool_fwd_sat2<KT>));
    if (e != hipSuccess) return (int)e;
    *regs = fa.numRegs; *static_lds = (int)fa.sharedSizeBytes; *dyn_lds = (int)L.bytes;
    return (int)hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks, k_roipool_fwd_sat2<KT>, threads, L.bytes);
}
#endif

bool roipool_fwd_supported(int R, int C, int H, int W, int k)
{
    if (!(R >= 1 && C >= 1 && H >= 1 && W >= 1 && H <= 255 && W <= 255)) return false;
    return k == KT ? (roipool_fwd_direct_supported(R, C, H, W, k) || sat_cg(C, H, W) > 0) : sat2_anyk(C, H, W, k);
}

size_t roipool_fwd_ws_bytes(int, int, int, int, int) { return 0; }

int roipool_fwd_f32(const float* fm, const float* rois, float* out, int R, int C, int H, int W, int k,
                    void*, hipStream_t st)
{
    // round 6: k = 7 in the reference's own order (d2t_roipool_fwd_direct.hip: bit-identical, no tables); scan builds: D2T_ROI_FWD=sat keeps the tables
    if (k == KT && roipool_fwd_direct_supported(R, C, H, W, k) && lab_env_int("D2T_ROI_FWD_SAT", 0) == 0)
        return roipool_fwd_direct_f32(fm, rois, out, R, C, H, W, st);
    const int CG = k == KT ? sat_cg(C, H, W) : 2;
    const SatLayout L = sat_layout(CG, H, W, k == KT || 4 * k < GEO8 ? GEO8 : 4 * k);
    const int gx = (C + CG - 1) / CG;
    int split = (256 + gx - 1) / gx;                                 // enough workgroups for every CU
    const int max_split = (R + 63) / 64;
    split = split < 1 ? 1 : (split > max_split ? max_split : split);
    split = split > 65535 ? 65535 : split;
    const int per = (R + split - 1) / split;
    const dim3 grid(gx, (R + per - 1) / per);
#define D2T_LAUNCH_SAT(CGV)                                                                              \
    {                                                                                                    \
        D2T_ENSURE_DYNAMIC_LDS(k_roipool_fwd_sat<CGV>, LDS_MAX);                                        \
        hipLaunchKernelGGL(k_roipool_fwd_sat<CGV>, grid, dim3(RF_THREADS), L.bytes, st, fm, rois, out, R, C, H, W, \
                           L.LD, L.plane, per);                                                          \
    }
    static const int inter = lab_env_int("D2T_SAT_INTERLEAVED", 1);   // -DD2T_LAB only
    if (k != KT) {                                                   // any other bin count: the same kernel, k at run time
        D2T_ENSURE_DYNAMIC_LDS(k_roipool_fwd_sat2<0>, LDS_MAX);
        hipLaunchKernelGGL(k_roipool_fwd_sat2<0>, grid, dim3(S2_THREADS), L.bytes, st, fm, rois, out, R, C, H, W, L.LD, L.plane, per, k);
        return launch_status();
    }
    if (CG == 2 && inter) {
        D2T_ENSURE_DYNAMIC_LDS(k_roipool_fwd_sat2<KT>, LDS_MAX);
        hipLaunchKernelGGL(k_roipool_fwd_sat2<KT>, grid, dim3(S2_THREADS), L.bytes, st, fm, rois, out, R, C, H, W, L.LD, L.plane, per, KT);
        return launch_status();
    }
    if (CG == 4) D2T_LAUNCH_SAT(4) else if (CG == 2) D2T_LAUNCH_SAT(2) else D2T_LAUNCH_SAT(1)
#undef D2T_LAUNCH_SAT
    return launch_status();
}

// ---------------------------------------------------------------------------------------
// PSROIPool cells, bin-major: cellsT[bin][r] = {i0, i1, j0, j1} (ps_roipool_cuda.cu:45-54), so that
// 64 consecutive RoIs of one bin are one coalesced 1 KB load.
// ---------------------------------------------------------------------------------------

__global__ void __launch_bounds__(256)
k_ps_cells_T(const float* __restrict__ rois, int4* __restrict__ cellsT, int R, int H, int W)
{
    const int r = blockIdx.x * 256 + threadIdx.x, bin = blockIdx.y;
    if (r >= R) return;
    const Bounds c = psroi_cell<float>(rois + 4 * (size_t)r, bin / KT, bin % KT, H, W, KT);
    cellsT[(size_t)bin * R + r] = make_int4(c.i0, c.i1, c.j0, c.j1);
}

int ps_cells_T(const float* rois, int4* cellsT, int R, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(k_ps_cells_T, dim3((R + 255) / 256, KK), dim3(256), 0, st, rois, cellsT, R, H, W);
    return launch_status();
}

// ---------------------------------------------------------------------------------------
// PSROIPool forward.  Workgroup = (input channel, chunk of RoIs), 256 threads.  The channel is
// staged in LDS once; an item is (plane p of the channel, RoI r), lane = RoI.  Each item keeps the
// reference's running sum over its cell in row-major order and its guarded IEEE divide:
// bit-identical.  tmpT[(t*49+bin)][r] is transposed to out[r][t][bin] by the caller.
// ---------------------------------------------------------------------------------------
constexpr int PF_RC = 1024;                   // RoIs per workgroup (4096 -- the map staged once per channel at R = 3000 -- measured 2.2x SLOWER: channels that feed many planes become stragglers)

__global__ void __launch_bounds__(256)
k_psroipool_fwd_chan(const float* __restrict__ fm, const int4* __restrict__ cellsT, float* __restrict__ tmpT,
                     int R, int nT, int H, int W)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x, HW = H * W;
    float* map = reinterpret_cast<float*>(lds_raw);                  // [H*W]
    int* list = reinterpret_cast<int*>(lds_raw + (((size_t)HW * 4 + 15) & ~(size_t)15));   // [64]
    // blockIdx.x < nT: channel 0, target t = blockIdx.x alone (bin 0 of EVERY target reads channel 0:
    // one workgroup per target instead of one for all nT of them); otherwise channel blockIdx.x - nT + 1
    const int ch = blockIdx.x < nT ? 0 : blockIdx.x - nT + 1;
    const int r_lo = blockIdx.y * PF_RC, nr = R - r_lo < PF_RC ? R - r_lo : PF_RC;
    const int np = ch == 0 ? 1 : ps_channel_planes(ch, nT, list, tid);
    if (np == 0) return;                                             // no output reads this channel
    const float* src = fm + (size_t)ch * HW;
    for (int e = tid; e < HW; e += 256) map[e] = src[e];
    __syncthreads();
    const int items = np * nr;
    for (int e = tid; e < items; e += 256) {
        const int p = e / nr, r = r_lo + (e - p * nr);
        const int pl = ch == 0 ? (int)blockIdx.x * KK : list[p];     // plane index t*49 + bin
        const int bin = pl % KK;
        const int4 c = cellsT[(size_t)bin * R + r];
        float acc = 0.f;
        for (int y = c.x; y < c.y; ++y) {
            const float* row = map + y * W;
            for (int x0 = c.z; x0 < c.w; x0 += 4) {                  // 4 independent LDS loads, then the
                float v[4];                                          // adds in ascending x (:60-66)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = row[x0 + k < c.w ? x0 + k : c.w - 1];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (x0 + k < c.w) acc += v[k];
            }
        }
        const int n = (c.y - c.x) * (c.w - c.z);
        if (n > 0) acc /= static_cast<float>(n);                     // guarded divide, :67-69
        tmpT[(size_t)pl * R + r] = acc;
    }
}

// ---------------------------------------------------------------------------------------
// PSROIPool forward, bin-row form (round 3).  The channel-resident kernel above spends ~580 lane-instructions
// per output: every lane of a wave walks a cell of a different shape (the loops run to the largest), with clamped
// loads and predicates in the innermost loop, and its results need a transposing pass.  Here:
//  * k_ps_roi_order (one workgroup) orders the RoIs by the size of their cells (rows x columns, estimated from the
//    RoI's extent) so that the 64 RoIs of a wave walk about the same number of rows and columns: a counting sort on
//    LDS atomics with the RoIs held in registers (loads inside the counting loops made the first version 16 us, a
//    ballot loop per distinct key of a wave -- ~30 steps for 64 random RoIs -- the second 11.6 us).  The order
//    changes no value -- every output is computed on its own -- only which lanes share a wave.  (Exact per-bin-row shapes, evaluated and sorted by 7 workgroups, were measured too: the 24
//    bin axes per thread made that pre-pass 22 us and the main kernel no faster.)
//  * k_psroipool_fwd_rows: workgroup = (target t, bin row i, a share of the RoIs in that order).  The SEVEN channels
//    (t+1)(7i+j) it reads are staged in LDS once (80 KB at 38 x 75; all loads of a thread are issued before the
//    first LDS store -- staged batch by batch the workgroup spent 10 of its 14 us waiting for them) and the
//    workgroup then walks its RoIs 1024 at a time: the share is sized so that the grid is about one workgroup per CU
//    (LDS admits only one), i.e. the staging is paid once per (t, i), not once per 1024 RoIs.  A lane owns one RoI,
//    evaluates its eight bin axes itself and keeps seven running sums -- seven independent chains, each in the
//    reference's row-major order with its guarded divide (bit-identical; a masked step adds +0, which cannot change
//    a sum that started at +0).  The seven results of a RoI are one 28-byte run of out[r][t][7i .. 7i+6]: written
//    through a wave-private LDS patch, ~10 cache lines per store instruction instead of 64.  Workgroup ids are
//    dealt so that the seven bin rows of a (t, share) run on ONE XCD at the same time: its L2 merges their runs into
//    the 196 contiguous bytes of out[r][t][:].
// ---------------------------------------------------------------------------------------
constexpr int PR_T = 1024;                     // threads = RoIs per pass of a workgroup
constexpr int PR_Q = 4;                        // staging loads per thread and channel in flight (PR_T * PR_Q pixels per round)
constexpr int PR_MAXR = 4096;                  // most RoIs the sorted path takes (wave rows of the counting sort: 64)

__global__ void __launch_bounds__(1024)
k_ps_roi_order(const float* __restrict__ rois, int* __restrict__ perm, int R, int H, int W)
{
    __shared__ unsigned hist[256];
    const int tid = threadIdx.x;
    if (tid < 256) hist[tid] = 0;
    constexpr int NR = PR_MAXR / 1024;
    float4 roi[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) {                                   // every load in flight before anything waits
        const int r = q * 1024 + tid;
        roi[q] = r < R ? reinterpret_cast<const float4*>(rois)[r] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    int key[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) {
        float fh = roi[q].z * (float)H * (1.f / KT), fw = roi[q].w * (float)W * (1.f / KT);
        fh = fh == fh ? fh : 0.f; fw = fw == fw ? fw : 0.f;          // NaN extents: any bucket will do
        const int h = (int)fminf(fmaxf(fh, 0.f), 14.f) + 1, w = (int)fminf(fmaxf(fw, 0.f), 14.f) + 1;
        key[q] = q * 1024 + tid < R ? h * 16 + w : -1;               // (cell rows, cell columns), each 1..15
        if (key[q] >= 0) atomicAdd(&hist[key[q]], 1u);
    }
    __syncthreads();
    if (tid < 64) {                                                  // exclusive scan of the 256 counts by one wave
        unsigned c[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { c[k] = hist[4 * tid + k]; sum += c[k]; }
        unsigned incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned t = __shfl_up(incl, off, 64);
            if (tid >= off) incl += t;
        }
        unsigned run = incl - sum;
#pragma unroll
        for (int k = 0; k < 4; ++k) { hist[4 * tid + k] = run; run += c[k]; }
    }
    __syncthreads();
    unsigned pos[NR];
#pragma unroll
    for (int q = 0; q < NR; ++q) pos[q] = key[q] >= 0 ? atomicAdd(&hist[key[q]], 1u) : 0u;   // (order inside a bucket: arbitrary)
#pragma unroll
    for (int q = 0; q < NR; ++q)
        if (key[q] >= 0) perm[pos[q]] = q * 1024 + tid;
}

// LOCAL: one share and at most PR_T RoIs -- the workgroup orders the RoIs itself (the bucket sort of k_ps_roi_order in LDS: a thread per RoI,
// ~2 k cycles) instead of reading the order a one-workgroup launch in front of this one wrote: that launch and the dependency on it cost
// more than 147+ workgroups repeating the sort.  (With several shares the workgroups of a target would have to agree on the order inside
// a bucket, which the atomics leave open: the pre-pass stays there.)
template <bool LOCAL>
__global__ void __launch_bounds__(PR_T)
k_psroipool_fwd_rows(const float* __restrict__ fm, const float* __restrict__ rois, const int* __restrict__ perm,
                     float* __restrict__ out, int R, int nT, int H, int W, int nshare, int per, int HWp)
{
    __shared__ unsigned ohist[LOCAL ? 256 : 1];
    __shared__ int operm[LOCAL ? PR_T : 1];
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* map = reinterpret_cast<float*>(lds_raw);                  // [7][HWp]
    float* patch = map + KT * HWp;                                   // [waves][64 * 7]
    int* patch_r = reinterpret_cast<int*>(patch + (PR_T / 64) * 64 * KT);   // [waves][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, HW = H * W;
    // id -> (group = (t, share), bin row): the 7 rows of a group sit on one XCD (ids are dealt to the 8 XCDs round-robin)
    const int L = blockIdx.x, slot = L >> 3, g = (slot / KT) * 8 + (L & 7), i = slot % KT;
    if (g >= nT * nshare) return;
    const int t = g / nshare, share = g - t * nshare;
    if (LOCAL) {                                                     // the order of k_ps_roi_order, for R <= PR_T, into LDS
        if (tid < 256) ohist[tid] = 0;
        const float4 roi = tid < R ? reinterpret_cast<const float4*>(rois)[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        float fh = roi.z * (float)H * (1.f / KT), fw = roi.w * (float)W * (1.f / KT);
        fh = fh == fh ? fh : 0.f; fw = fw == fw ? fw : 0.f;
        const int bh = (int)fminf(fmaxf(fh, 0.f), 14.f) + 1, bw = (int)fminf(fmaxf(fw, 0.f), 14.f) + 1;
        const int key = tid < R ? bh * 16 + bw : -1;
        if (key >= 0) atomicAdd(&ohist[key], 1u);
        __syncthreads();
        if (tid < 64) {
            unsigned c[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { c[k] = ohist[4 * tid + k]; sum += c[k]; }
            unsigned incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned tt = __shfl_up(incl, off, 64);
                if (tid >= off) incl += tt;
            }
            unsigned run = incl - sum;
#pragma unroll
            for (int k = 0; k < 4; ++k) { ohist[4 * tid + k] = run; run += c[k]; }
        }
        __syncthreads();
        if (key >= 0) operm[atomicAdd(&ohist[key], 1u)] = tid;       // (order inside a bucket: arbitrary -- this workgroup takes every RoI)
    }
    for (int e0 = 0; e0 < HW; e0 += PR_T * PR_Q) {                   // (one round at 38 x 75)
        float v[KT][PR_Q];
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            const float* src = fm + (size_t)((t + 1) * (KT * i + j)) * HW;   // ps_roipool_cuda.cu:58
#pragma unroll
            for (int q = 0; q < PR_Q; ++q) {
                const int e = e0 + q * PR_T + tid;
                v[j][q] = e < HW ? src[e] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int q = 0; q < PR_Q; ++q) {
                const int e = e0 + q * PR_T + tid;
                if (e < HW) map[j * HWp + e] = v[j][q];
            }
    }
    __syncthreads();
    const int k_end = (share + 1) * per < R ? (share + 1) * per : R;
    float* mine = patch + wave * 64 * KT;
    const size_t plane0 = (size_t)t * KK + KT * i;
    for (int k0 = share * per; k0 < k_end; k0 += PR_T) {             // (per is a multiple of 64: waves stay whole)
        const int k = k0 + tid;
        if (k0 + (wave << 6) >= k_end) break;                        // wave-uniform; nothing below synchronises the workgroup
        const bool live = k < k_end;
        const int r = live ? (LOCAL ? operm[k] : perm[k]) : -1;
        int i0 = 0, i1 = 0, x0[KT], w[KT], wmax = 0;
#pragma unroll
        for (int j = 0; j < KT; ++j) { x0[j] = 0; w[j] = 0; }
        if (live) {                                                  // the cells of bin row i (ps_roipool_cuda.cu:45-54)
            const float4 roi = reinterpret_cast<const float4*>(rois)[r];
            bin_axis<float>(roi.x - roi.z / 2.f, roi.z / static_cast<float>(KT), i, H, i0, i1);
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                int j1;
                bin_axis<float>(roi.y - roi.w / 2.f, roi.w / static_cast<float>(KT), j, W, x0[j], j1);
                w[j] = j1 - x0[j];
                wmax = w[j] > wmax ? w[j] : wmax;
            }
        }
        const int h = i1 - i0;
        int hu = h, wu = h > 0 ? wmax : 0;                           // the wave's loop bounds
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int a = __shfl_xor(hu, off, 64), b = __shfl_xor(wu, off, 64);
            hu = a > hu ? a : hu;
            wu = b > wu ? b : wu;
        }
        hu = __builtin_amdgcn_readfirstlane(hu);
        wu = __builtin_amdgcn_readfirstlane(wu);
        float acc[KT];
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[j] = 0.f;
        for (int yy = 0; yy < hu; ++yy) {
            const bool yok = yy < h;
            const float* row = map + (i0 + yy) * W;                  // (a masked lane may point anywhere: its value is dropped)
            int lim[KT];
            const float* p[KT];
#pragma unroll
            for (int j = 0; j < KT; ++j) { lim[j] = yok ? w[j] : 0; p[j] = row + j * HWp + x0[j]; }
#pragma unroll 2
            for (int xx = 0; xx < wu; ++xx) {
                float v[KT];
#pragma unroll
                for (int j = 0; j < KT; ++j) v[j] = p[j][xx];
#pragma unroll
                for (int j = 0; j < KT; ++j)
                    if (xx < lim[j]) acc[j] += v[j];                   // ascending x inside ascending y (:60-66)
            }
        }
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            const int n = h * w[j];
            float a = acc[j];
            if (n > 0) a /= static_cast<float>(n);                   // guarded divide, :67-69
            mine[lane * KT + j] = a;
        }
        patch_r[wave * 64 + lane] = r;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < KT; ++q) {
            const int idx = q * 64 + lane, rl = idx / KT, j = idx - rl * KT;
            const int rr = patch_r[wave * 64 + rl];
            if (rr >= 0) out[(size_t)rr * nT * KK + plane0 + j] = mine[idx];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

static size_t ps_rows_lds(int H, int W) { return ((size_t)KT * (((size_t)H * W + 3) & ~(size_t)3) + (PR_T / 64) * 64 * (KT + 1)) * 4; }
static bool ps_rows_fit(int R, int nT, int H, int W)
{
    return R <= PR_MAXR && ps_rows_lds(H, W) <= (size_t)LDS_MAX - 2048 && nT <= (1 << 20);
}
// shares of the RoI list per (t, bin row): about one workgroup per CU, whole waves per share
static void ps_rows_shares(int R, int nT, int& nshare, int& per)
{
    nshare = 256 / (nT * KT);
    nshare = nshare < 1 ? 1 : nshare;
    per = ((R + nshare - 1) / nshare + 63) & ~63;
    nshare = (R + per - 1) / per;
}

// Below this many (RoI, target) pairs the single-launch thread-per-output kernel (d2t_pool_bwd.hip) is the faster one
// (R=300 nT=4: 5.9 us against 11.6 us for the two launches of the bin-row form); above it the bin-row form wins
// (R=300 nT=21: 13.3 / 14.2, R=300 nT=31: 14.0 / 18.3, R=3000 nT=4: 14.8 / 21.0, R=3000 nT=31: 37.5 us against 90 us
// for the channel-resident kernel + transpose and 129 us for the thread-per-output kernel).
static bool ps_fwd_small(int R, int nT) { return 1LL * R * nT < 6000; }

bool psroipool_fwd_supported(int R, int nT, int H, int W, int k)
{
    return k == KT && R >= 1 && nT >= 1 && H >= 1 && W >= 1 && (size_t)H * W * 4 <= (size_t)LDS_MAX - 1024 && H <= 32767 && W <= 32767 &&
           (R + PF_RC - 1) / PF_RC <= 65535 && 1LL * nT * KK * R < 0x7fffffffLL && (nT * KK + 255) / 256 <= 65535;
}

// The bin-row form as ONE launch (every workgroup orders the RoIs itself, in LDS): a target is one share of at most PR_T RoIs and
// the maps leave 8 KB for the histogram / order arrays.  Needs NO workspace.  One predicate for the workspace query and the launcher.
static bool ps_rows_local_order(int R, int nT, int H, int W)
{
    if (!ps_rows_fit(R, nT, H, W)) return false;
    int nshare, per;
    ps_rows_shares(R, nT, nshare, per);
    return nshare == 1 && R <= PR_T && ps_rows_lds(H, W) <= (size_t)LDS_MAX - 8192;
}

size_t psroipool_fwd_ws_bytes(int R, int nT, int H, int W, int k)
{
    if (!psroipool_fwd_supported(R, nT, H, W, k) || ps_fwd_small(R, nT)) return 0;
    if (ps_rows_local_order(R, nT, H, W)) return 0;
    if (ps_rows_fit(R, nT, H, W)) return align256((size_t)R * sizeof(int));
    return cellsT_bytes(R) + align256((size_t)nT * KK * R * sizeof(float));
}

int psroipool_fwd_f32(const float* fm, const float* rois, float* out, int R, int nT, int H, int W, int k,
                      void* ws, hipStream_t st)
{
    if (ps_fwd_small(R, nT)) return psroipool_fwd_small_f32(fm, rois, out, R, nT, H, W, k, st);
    if (ps_rows_fit(R, nT, H, W)) {
        int* perm = static_cast<int*>(ws);
        int nshare, per;
        ps_rows_shares(R, nT, nshare, per);
        if (ps_rows_local_order(R, nT, H, W)) {                      // one launch: every workgroup orders the RoIs itself (perm unused: no workspace)
            D2T_ENSURE_DYNAMIC_LDS(k_psroipool_fwd_rows<true>, LDS_MAX - 8192);          // 5 KB of static LDS (histogram, order) beside the maps
            hipLaunchKernelGGL(k_psroipool_fwd_rows<true>, dim3((nT * nshare + 7) / 8 * 8 * KT), dim3(PR_T), ps_rows_lds(H, W), st,
                               fm, rois, perm, out, R, nT, H, W, nshare, per, (H * W + 3) & ~3);
            return launch_status();
        }
        hipLaunchKernelGGL(k_ps_roi_order, dim3(1), dim3(1024), 0, st, rois, perm, R, H, W);
        D2T_ENSURE_DYNAMIC_LDS(k_psroipool_fwd_rows<false>, LDS_MAX);
        hipLaunchKernelGGL(k_psroipool_fwd_rows<false>, dim3((nT * nshare + 7) / 8 * 8 * KT), dim3(PR_T), ps_rows_lds(H, W), st,
                           fm, rois, perm, out, R, nT, H, W, nshare, per, (H * W + 3) & ~3);
        return launch_status();
    }
    // maps too large for seven resident channels, or more than 4096 RoIs: one channel per workgroup + transposing pass
    int4* cellsT = static_cast<int4*>(ws);
    float* tmpT = reinterpret_cast<float*>(static_cast<char*>(ws) + cellsT_bytes(R));
    int rc = ps_cells_T(rois, cellsT, R, H, W, st);
    if (rc != D2T_OK) return rc;
    const size_t lds = (((size_t)H * W * 4 + 15) & ~(size_t)15) + 256;
    D2T_ENSURE_DYNAMIC_LDS(k_psroipool_fwd_chan, LDS_MAX);
    hipLaunchKernelGGL(k_psroipool_fwd_chan, dim3(nT * KK + nT - 1, (R + PF_RC - 1) / PF_RC), dim3(256), lds, st,
                       fm, cellsT, tmpT, R, nT, H, W);
    rc = launch_status();
    if (rc != D2T_OK) return rc;
    return transpose(tmpT, out, nT * KK, R, st);                     // (t*49+bin, r) -> (r, t*49+bin)
}

}}  // namespace d2t::tuned
