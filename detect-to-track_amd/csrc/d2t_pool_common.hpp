// d2t_pool_common.hpp -- pieces shared by the pooling translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include "d2t_common.hpp"

namespace d2t { namespace tuned {

constexpr int KT = 7;                         // r_hw the tuned pooling kernels are built for
constexpr int KK = KT * KT;
constexpr int LDS_MAX = 160 * 1024;           // bytes of LDS a workgroup may use on gfx950

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }
inline size_t cellsT_bytes(int R) { return align256((size_t)R * KK * sizeof(int4)); }

// defined in d2t_pool_tuned.hip
int transpose(const float* in, float* out, int rows, int cols, hipStream_t st);        // (rows, cols) -> (cols, rows)
int ps_cells_T(const float* rois, int4* cellsT, int R, int H, int W, hipStream_t st);  // cellsT[bin][r] = {i0,i1,j0,j1}

// The output planes (t, bin) that read / feed input channel ch: (t+1)*bin == ch.  ch == 0: bin 0 of
// every target (handled by the callers); otherwise bin | ch, 1 <= bin <= 48, ch / bin <= nT.
// Lane b-1 of the first wave tests bin b; the list comes out in ascending bin order.
__device__ __forceinline__ int ps_channel_planes(int ch, int nT, int* __restrict__ list /* LDS, 64 ints */, int tid)
{
    if (tid < 64) {
        const int bin = tid + 1;
        const bool is_src = ch > 0 && bin < KK && ch % bin == 0 && ch / bin <= nT;
        const unsigned long long m = __ballot(is_src);
        if (is_src) list[__builtin_popcountll(m & ((1ull << tid) - 1ull))] = (ch / bin - 1) * KK + bin;
        if (tid == 0) list[63] = __builtin_popcountll(m);            // at most 48 entries: slot 63 is free
    }
    __syncthreads();
    return list[63];
}

// In-place inclusive 2-D prefix sum of nb f64 maps F[b][H][LDW] (columns 0..W-1), all threads of the
// workgroup: along x in segments of 16 plus a fix-up with the totals of the segments to the left,
// then along y in segments of 8 the same way (a thread's steps are independent LDS accesses, the
// running sum stays in a register).  scr: nb * (H*nsx + nsy*W) doubles.
constexpr int PX_SEG = 16, PY_SEG = 8;
__device__ __forceinline__ int prefix_scratch_doubles(int nb, int H, int W)
{
    return nb * (H * ((W + PX_SEG - 1) / PX_SEG) + ((H + PY_SEG - 1) / PY_SEG) * W);
}
inline size_t prefix_scratch_bytes(int nb, int H, int W)
{
    return (size_t)nb * ((size_t)H * ((W + PX_SEG - 1) / PX_SEG) + (size_t)((H + PY_SEG - 1) / PY_SEG) * W) * 8;
}

// es: doubles between horizontally adjacent elements of one map (1: planar maps; nb with mstride = 1: maps
// interleaved element by element).
__device__ __forceinline__ void prefix2d(double* __restrict__ F, double* __restrict__ scr, int nb, int H, int W, int LDW,
                                         int mstride /* doubles between maps */, int tid, int nthr, int es = 1)
{
    const int nsx = (W + PX_SEG - 1) / PX_SEG, nsy = (H + PY_SEG - 1) / PY_SEG;
    const int ntx = nb * H * nsx;
    // along x: a thread reads its segment into registers (independent loads: one LDS round trip),
    // prefixes it there, publishes the segment total, and after the barrier writes segment + the
    // totals of the segments to its left.  All threads run the same number of iterations.
    for (int t0 = 0; t0 < ntx; t0 += nthr) {
        const int t = t0 + tid;
        const bool on = t < ntx;
        const int row = on ? t / nsx : 0, sg = on ? t - row * nsx : 0, bb = row / H, y = row - bb * H;
        double* p = F + (size_t)bb * mstride + (size_t)(y * LDW + sg * PX_SEG) * es;
        const int len = W - sg * PX_SEG < PX_SEG ? W - sg * PX_SEG : PX_SEG;
        double v[PX_SEG];
#pragma unroll
        for (int k = 0; k < PX_SEG; ++k) v[k] = on && k < len ? p[k * es] : 0.0;
#pragma unroll
        for (int k = 1; k < PX_SEG; ++k) v[k] += v[k - 1];
        if (on) scr[t] = v[PX_SEG - 1];
        __syncthreads();
        double off = 0.0;
        for (int s2 = 0; s2 < sg; ++s2) off += scr[row * nsx + s2];
#pragma unroll
        for (int k = 0; k < PX_SEG; ++k)
            if (on && k < len) p[k * es] = v[k] + off;
        __syncthreads();
    }
    // along y the same way; consecutive threads own consecutive columns
    double* scy = scr + ntx;
    const int nty = nb * nsy * W;
    for (int t0 = 0; t0 < nty; t0 += nthr) {
        const int t = t0 + tid;
        const bool on = t < nty;
        const int bs = on ? t / W : 0, x = on ? t - bs * W : 0, bb = bs / nsy, sg = bs - bb * nsy;
        double* p = F + (size_t)bb * mstride + ((size_t)sg * PY_SEG * LDW + x) * es;
        const int len = H - sg * PY_SEG < PY_SEG ? H - sg * PY_SEG : PY_SEG;
        double v[PY_SEG];
#pragma unroll
        for (int k = 0; k < PY_SEG; ++k) v[k] = on && k < len ? p[(size_t)k * LDW * es] : 0.0;
#pragma unroll
        for (int k = 1; k < PY_SEG; ++k) v[k] += v[k - 1];
        if (on) scy[t] = v[PY_SEG - 1];
        __syncthreads();
        double off = 0.0;
        for (int s2 = 0; s2 < sg; ++s2) off += scy[(bb * nsy + s2) * W + x];
#pragma unroll
        for (int k = 0; k < PY_SEG; ++k)
            if (on && k < len) p[(size_t)k * LDW * es] = v[k] + off;
        __syncthreads();
    }
}

}}  // namespace d2t::tuned
