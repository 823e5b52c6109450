// d2t_corr_tuned.hip -- gfx950-tuned f32 PointwiseCorrelation (d_max = 8, stride 1).
//
// The correlation is a banded GEMM: out[p][q] = sum_c FM0[c][p] * FM1[c][q] for q in the
// (2d x 2d) window of pixel p.  At the reference's shapes it is f32-FMA-bound, not HBM-bound
// (SURVEY.md F10), so the arithmetic runs on the exact-f32 matrix pipe:
// v_mfma_f32_16x16x4_f32 is bit-for-bit a k-ordered chain of fmaf (MI355X guide, "FP32-input
// MFMA"), i.e. the same ascending-channel FMA chain nvcc builds for the reference's inner loop
// (pointwise_correlation_cuda.cu:105-107).  Forward results are therefore BIT-IDENTICAL to the
// type-generic kernel and to the reference.
//
// Tiling.  M = 16 pixels p arranged 4x4 (a "p-tile"); their windows' union is 19 rows x 19
// columns of FM1.  That union is enumerated as groups of 4 consecutive columns (5 groups = 20
// columns per row, origin clamped so that all 20 columns lie inside the map): one lane loads one
// group with ONE 16-byte global load and feeds FOUR MFMAs (four N-tiles) with it.  16 groups x 4
// columns = one "tile-group" = 4 N-tiles of 16 columns, owned by one wave; a p-tile has at most
// 6 tile-groups (95 groups), rows of the window that fall outside the map are not enumerated at
// all.  K = channels, 4 per MFMA, ascending.  No LDS on the streamed operand: every FM1 element
// is used by exactly one MFMA of one wave.  The 16 x C FM0 tile is shared by the waves through
// LDS.  The epilogue stages the p-tile's 16 x 17 x 17 outputs (including the structural zeros
// the reference gets from at::zeros, :192) in LDS and writes 4 contiguous runs of 4 x 289 floats.
//
// One workgroup = one p-tile = 6 waves.  At B=8, 38x63 that is 1280 workgroups = 5 per CU, all
// resident at once; the XCD-aware block map puts a whole batch item (160 tiles) on one XCD so
// that its FM0/FM1 planes stream through that XCD's L2 once.
#include "d2t_tuned.hpp"

namespace d2t { namespace tuned {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load, dword aligned

#define D2T_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int TP = 4;                      // p-tile edge: 4x4 pixels = MFMA M = 16
constexpr int DT = 8;                      // d_max the tuned kernels are built for
constexpr int WR = TP + 2 * DT - 1;        // 19 window rows (and needed columns)
constexpr int NCG = (WR + 3) / 4;          // 5 column groups per window row
constexpr int WC = NCG * 4;                // 20 loaded columns
constexpr int CW = 2 * DT + 1;             // 17
constexpr int CELLS = CW * CW;             // 289
constexpr int FWD_WAVES = 6;               // >= max tile-groups = ceil(19*5/16)
constexpr int FWD_THREADS = FWD_WAVES * 64;
constexpr int CA = 256;                    // FM0 channels staged in LDS per pass

// Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of logical tiles
// (bijective for any grid size).  Placement only affects L2 reuse, never results.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

__global__ void __launch_bounds__(FWD_THREADS, 8)      // 8 waves/SIMD: 5 workgroups (30 waves) per CU
k_corr_fwd_mfma(const float* __restrict__ fm0, const float* __restrict__ fm1, float* __restrict__ out,
                int C, int H, int W, int tiles_i, int tiles_j)
{
    __shared__ float smem[16 * CELLS];                 // 18.5 KB: FM0 tile [CA][16], later the out tile
    static_assert(16 * CELLS >= CA * 16, "out tile must cover the FM0 staging tile");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tj = bid % tiles_j, ti = (bid / tiles_j) % tiles_i, b = bid / (tiles_j * tiles_i);
    const int i0 = ti * TP, j0 = tj * TP;
    const int HW = H * W;

    // window rows that exist in the map, and the clamped column origin (all 20 columns in-map)
    const int wr_lo = DT - i0 > 0 ? DT - i0 : 0;
    const int wr_hi = H + DT - i0 < WR ? H + DT - i0 : WR;
    const int NG = (wr_hi - wr_lo) * NCG;              // enumerated column groups
    const int ntg = (NG + 15) >> 4;                    // tile-groups (<= FWD_WAVES)
    int col0 = j0 - DT;
    col0 = col0 < 0 ? 0 : (col0 > W - WC ? W - WC : col0);

    // this lane's column group (clamped for the load; masked in the epilogue)
    const int gsel = 16 * wave + n;
    const int gi = gsel < NG ? gsel : NG - 1;
    const int wr = wr_lo + gi / NCG;
    const int di = i0 - DT + wr;
    const int djs = col0 + 4 * (gi % NCG);
    const int boff = g * HW + di * W + djs;            // lane part of the FM1 address (channel g)
    const int kstride = 4 * HW;                         // one k-step = 4 channels

    // FM0 pixel this thread stages (clamped into the map; rows of partial tiles are never stored)
    const int am = tid & 15;
    const int ai = i0 + (am >> 2) < H ? i0 + (am >> 2) : H - 1;
    const int aj = j0 + (am & 3) < W ? j0 + (am & 3) : W - 1;
    const float* ap = fm0 + (size_t)b * C * HW + ai * W + aj;

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const bool active = wave < ntg;

#define D2T_STEP(Q, KS)                                                                   \
    {                                                                                     \
        const float a_ = smem[((KS) * 4 + g) * 16 + n];                                   \
        acc0 = D2T_MFMA(a_, (Q).x, acc0);                                                 \
        acc1 = D2T_MFMA(a_, (Q).y, acc1);                                                 \
        acc2 = D2T_MFMA(a_, (Q).z, acc2);                                                 \
        acc3 = D2T_MFMA(a_, (Q).w, acc3);                                                 \
    }

    for (int c0 = 0; c0 < C; c0 += CA) {
        const int cc = C - c0 < CA ? C - c0 : CA;      // channels in this pass
        __syncthreads();
        for (int e = tid; e < CA * 16; e += FWD_THREADS) {
            const int c = e >> 4;                      // (e & 15) == am because FWD_THREADS % 16 == 0
            smem[e] = c < cc ? ap[(size_t)(c0 + c) * HW] : 0.f;
        }
        __syncthreads();
        if (active) {
            // Streamed operand: one 16-byte load per lane per k-step, kept PF = 4 k-steps ahead of
            // the MFMAs that consume it.  Every load is unconditional (indices past the end are
            // clamped to the last full k-step and simply re-read it) so that the compiler can
            // retire them with counted s_waitcnt vmcnt(3) instead of draining the queue.
            const float* bq = fm1 + ((size_t)b * C + c0) * HW;          // wave-uniform base
            const int nfull = cc >> 2;
#define D2T_LD(KS) (*reinterpret_cast<const f32x4u*>(bq + ((KS) < last ? (KS) : last) * kstride + boff))
            if (nfull > 0) {
                const int last = nfull - 1;
                f32x4 q0 = D2T_LD(0), q1 = D2T_LD(1), q2 = D2T_LD(2), q3 = D2T_LD(3);
                int ks = 0;
                for (; ks + 3 < nfull; ks += 4) {
                    D2T_STEP(q0, ks);     q0 = D2T_LD(ks + 4);
                    D2T_STEP(q1, ks + 1); q1 = D2T_LD(ks + 5);
                    D2T_STEP(q2, ks + 2); q2 = D2T_LD(ks + 6);
                    D2T_STEP(q3, ks + 3); q3 = D2T_LD(ks + 7);
                }
                if (ks < nfull)     D2T_STEP(q0, ks);
                if (ks + 1 < nfull) D2T_STEP(q1, ks + 1);
                if (ks + 2 < nfull) D2T_STEP(q2, ks + 2);
            }
#undef D2T_LD
            if (cc & 3) {                              // channel tail: lanes past C feed exact zeros
                f32x4 qt = {0.f, 0.f, 0.f, 0.f};
                if (nfull * 4 + g < cc) qt = *reinterpret_cast<const f32x4u*>(bq + nfull * kstride + boff);
                D2T_STEP(qt, nfull);
            }
        }
    }
#undef D2T_STEP

    // ---- epilogue: out tile [16 pixels][17][17] through LDS ----
    __syncthreads();
    for (int e = tid; e < 16 * CELLS; e += FWD_THREADS) smem[e] = 0.f;
    __syncthreads();
    if (active && gsel < NG) {
        // lane holds D[m = 4g + r][column n of N-tile t]: pixel (i0+g, j0+r), displaced (di, djs+t)
        const int ci = wr - g;                          // di - i + d
        if (ci >= 0 && ci < 2 * DT) {
            float* row = smem + (4 * g) * CELLS + ci * CW;
            const f32x4 a4[4] = {acc0, acc1, acc2, acc3};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cj = djs + t - (j0 + r) + DT;
                    if (cj >= 0 && cj < 2 * DT) row[r * CELLS + cj] = a4[t][r];
                }
            }
        }
    }
    __syncthreads();
    const int nj = W - j0 < TP ? W - j0 : TP;
    for (int pi = 0; pi < TP; ++pi) {
        const int i = i0 + pi;
        if (i >= H) break;
        float* dst = out + (((size_t)b * H + i) * W + j0) * CELLS;
        const float* src = smem + pi * 4 * CELLS;
        for (int e = tid; e < nj * CELLS; e += FWD_THREADS) dst[e] = src[e];
    }
}

bool corr_fwd_supported(int B, int C, int H, int W, int d, int s)
{
    if (d != DT || s != 1 || B < 1 || C < 1 || H < 1 || W < WC) return false;
    const long long blocks = 1LL * B * ((H + TP - 1) / TP) * ((W + TP - 1) / TP);
    return blocks <= 0x7fffffffLL;
}

size_t corr_fwd_ws_bytes(int, int, int, int, int, int) { return 0; }

int corr_fwd_f32(const float* fm0, const float* fm1, float* out, int B, int C, int H, int W, int, int,
                 void*, hipStream_t st)
{
    const int tiles_i = (H + TP - 1) / TP, tiles_j = (W + TP - 1) / TP;
    const int blocks = B * tiles_i * tiles_j;
    hipLaunchKernelGGL(k_corr_fwd_mfma, dim3(blocks), dim3(FWD_THREADS), 0, st,
                       fm0, fm1, out, C, H, W, tiles_i, tiles_j);
    return launch_status();
}

// ====================================================================================
// Backward.  Both gradients have the same shape of work:
//     gX[c][t] = sum over window slots w of  G[t][w] * S[c][w]
// role 0 (gradFM0): t = the 16 pixels (i,j) of a 4x4 tile, S = FM1, the window is the union of
//                   their displacement windows, G[t][w] = gradOut[b,i,j,di-i+d,dj-j+d];
// role 1 (gradFM1): t = the 16 displaced pixels (di,dj) of a 4x4 tile, S = FM0, the window is
//                   the set of centres (i,j) that reach them (rows di0-d+1 .. di0+3+d), same G.
// This is the gather form of pointwise_correlation_cuda.cu:154-171: no atomics, every output
// element written once, summation order fixed (deterministic).
//
// MFMA mapping: D[m = 16 channels][n = 16 tile pixels] += A[m][k] * B[k][n], k = window slots,
// 4 per MFMA.  The window is enumerated exactly as in the forward (rows inside the map only,
// 5 groups of 4 columns, column origin clamped into the map).  One lane owns channel m and
// k-slot g: ONE 16-byte load of group gamma = 4*kb + g gives its A operand for the four
// MFMAs of k-block kb (MFMA s of the block uses column s of every lane's group), and one
// ds_read_b128 of the G tile ([gamma][t][s] in LDS) gives the matching B operands.  Slots that
// are padding (gamma >= NG, the 20th column, cells outside a pixel's own window) have G = 0.
// A wave owns 4 c-tiles (64 channels) at a time: 4 loads + 1 LDS read feed 16 MFMAs.
// ====================================================================================
constexpr int BWD_WAVES = 4;
constexpr int BWD_THREADS = BWD_WAVES * 64;
constexpr int NGMAX = WR * NCG;                     // 95
constexpr int NKB = (NGMAX + 3) / 4;                // 24 k-blocks of 16 slots
constexpr int GT_FLOATS = NKB * 4 * 64;             // G tile [96 groups][16 t][4 s]

__global__ void __launch_bounds__(BWD_THREADS)
k_corr_bwd_mfma(const float* __restrict__ gout, const float* __restrict__ fm0, const float* __restrict__ fm1,
                float* __restrict__ g0, float* __restrict__ g1,
                int C, int H, int W, int tiles_i, int tiles_j, int tiles_total)
{
    __shared__ __attribute__((aligned(16))) float gs[GT_FLOATS];      // 24 KB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int role = bid >= tiles_total ? 1 : 0;                      // first all gradFM0 tiles, then gradFM1
    bid -= role * tiles_total;
    const int tj = bid % tiles_j, ti = (bid / tiles_j) % tiles_i, b = bid / (tiles_j * tiles_i);
    const int i0 = ti * TP, j0 = tj * TP, HW = H * W;
    const float* S = role ? fm0 : fm1;
    float* gx = role ? g1 : g0;

    // window geometry: role 0 rows i0-d .. i0+3+d-1, role 1 rows i0-d+1 .. i0+3+d (same for columns)
    const int wtop = i0 - DT + role, wleft = j0 - DT + role;
    const int row_first = wtop > 0 ? wtop : 0;
    const int row_end = wtop + WR < H ? wtop + WR : H;                // exclusive
    const int NG = (row_end - row_first) * NCG;
    const int nkb = (NG + 3) >> 2;
    const int col0 = wleft < 0 ? 0 : (wleft > W - WC ? W - WC : wleft);

    // ---- build the G tile in LDS: gs[(gamma*16 + t)*4 + s] ----
    const float* gb = gout + (size_t)b * HW * CELLS;
    if (role == 0) {
        for (int e = tid; e < GT_FLOATS; e += BWD_THREADS) gs[e] = 0.f;
        __syncthreads();
        for (int e = tid; e < 16 * CELLS; e += BWD_THREADS) {
            const int t = e / CELLS, cell = e - t * CELLS;
            const int ci = cell / CW, cj = cell - ci * CW;
            const int i = i0 + (t >> 2), j = j0 + (t & 3);
            const int di = i + ci - DT, dj = j + cj - DT;
            if (ci < 2 * DT && cj < 2 * DT && i < H && j < W && di >= 0 && di < H && dj >= 0 && dj < W) {
                const int wcc = dj - col0;                            // 0..19 by construction
                const int gamma = (di - row_first) * NCG + (wcc >> 2);
                gs[(gamma * 16 + t) * 4 + (wcc & 3)] = gb[(size_t)(i * W + j) * CELLS + cell];
            }
        }
    } else {
        for (int e = tid; e < GT_FLOATS; e += BWD_THREADS) {
            const int s = e & 3, t = (e >> 2) & 15, gamma = e >> 6;
            float v = 0.f;
            if (gamma < NG) {
                const int er = gamma / NCG, cg = gamma - er * NCG;
                const int i = row_first + er, j = col0 + 4 * cg + s;   // centre pixel, inside the map
                const int di = i0 + (t >> 2), dj = j0 + (t & 3);
                const int ci = di - i + DT, cj = dj - j + DT;
                if (ci >= 0 && ci < 2 * DT && cj >= 0 && cj < 2 * DT && di < H && dj < W)
                    v = gb[(size_t)(i * W + j) * CELLS + ci * CW + cj];
            }
            gs[e] = v;
        }
    }
    __syncthreads();

    // ---- main loop: passes of 64 channels per wave (4 c-tiles), 256 per workgroup ----
    const f32x4* gs4 = reinterpret_cast<const f32x4*>(gs);
    const int tpi = n >> 2, tpj = n & 3;                              // tile pixel of output column n
    const bool pix_ok = i0 + tpi < H && j0 + tpj < W;
    for (int cbase = wave * 64; cbase < C; cbase += BWD_WAVES * 64) {
        // lane's channel in each of the 4 c-tiles (clamped: rows past C are never stored)
        const float* sp[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int c = cbase + 16 * u + n;
            c = c < C ? c : C - 1;
            sp[u] = S + ((size_t)b * C + c) * HW + row_first * W + col0;
        }
        f32x4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int kb = 0; kb < nkb; ++kb) {
            int gamma = 4 * kb + g;
            const f32x4 bv = gs4[gamma * 16 + n];                     // G[t = n][slots of gamma], 0 if gamma >= NG
            gamma = gamma < NG ? gamma : NG - 1;
            const int er = gamma / NCG, cg = gamma - er * NCG;
            const int off = er * W + 4 * cg;
            f32x4 av[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) av[u] = *reinterpret_cast<const f32x4u*>(sp[u] + off);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = D2T_MFMA(av[u][s], bv[s], acc[u]);
            }
        }
        // D[m = 4g + r][n]: channel cbase + 16u + 4g + r, tile pixel n
        if (pix_ok) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = cbase + 16 * u + 4 * g + r;
                    if (c < C) gx[((size_t)b * C + c) * HW + (i0 + tpi) * W + j0 + tpj] = acc[u][r];
                }
            }
        }
    }
}

bool corr_bwd_supported(int B, int C, int H, int W, int d, int s)
{
    if (d != DT || s != 1 || B < 1 || C < 1 || H < 1 || W < WC) return false;
    const long long blocks = 2LL * B * ((H + TP - 1) / TP) * ((W + TP - 1) / TP);
    return blocks <= 0x7fffffffLL;
}

size_t corr_bwd_ws_bytes(int, int, int, int, int, int) { return 0; }

int corr_bwd_f32(const float* gout, const float* fm0, const float* fm1, float* g0, float* g1,
                 int B, int C, int H, int W, int, int, void*, hipStream_t st)
{
    const int tiles_i = (H + TP - 1) / TP, tiles_j = (W + TP - 1) / TP;
    const int tiles_total = B * tiles_i * tiles_j;
    hipLaunchKernelGGL(k_corr_bwd_mfma, dim3(2 * tiles_total), dim3(BWD_THREADS), 0, st,
                       gout, fm0, fm1, g0, g1, C, H, W, tiles_i, tiles_j, tiles_total);
    return launch_status();
}

}}  // namespace d2t::tuned
