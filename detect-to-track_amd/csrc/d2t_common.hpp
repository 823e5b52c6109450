// d2t_common.hpp -- shared device helpers for libd2t_ops (gfx950 only).
//
// Index arithmetic that the parity contract requires bit-exact lives here, in ONE place,
// so that the pooling kernels, the backward kernels and the introspection entry points
// (d2t_*_bins_*, d2t_corr_mask) all run the same code.
//
// Reference semantics followed (paths relative to /root/reference/detect_to_track/models/):
//   clamp01      common/cuda_common.cuh:9-13
//   roi_bin      roipool/roipool_cuda.cu:32-51
//   psroi_cell   ps_roipool/ps_roipool_cuda.cu:36-54
//   corr window  pointwise_correlation/pointwise_correlation_cuda.cu:92-93
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include "../../include/d2t_ops.h"

namespace d2t {

constexpr int kWave = 64;  // CDNA4 wavefront

// This library is compiled with -ffp-contract=off: every fused multiply-add below is
// spelled out, so results do not depend on the optimiser.  The reference is built by
// nvcc, whose default fuses a*b+c; the explicit fma() calls mirror exactly those sites.
__device__ __forceinline__ float  fma_t(float a, float b, float c)   { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float  floor_t(float x)  { return __builtin_floorf(x); }
__device__ __forceinline__ double floor_t(double x) { return __builtin_floor(x); }
__device__ __forceinline__ float  ceil_t(float x)   { return __builtin_ceilf(x); }
__device__ __forceinline__ double ceil_t(double x)  { return __builtin_ceil(x); }

// max(0, min(1, x)) with CUDA's fmin/fmax NaN rule (a NaN operand yields the other one).
template <typename T>
__device__ __forceinline__ T clamp01(T x) {
    if (x != x) return T(1);
    const T m = x < T(1) ? x : T(1);
    return m > T(0) ? m : T(0);
}

struct Bounds { int i0, i1, j0, j1; };

// One axis of a bin: `corner` is the RoI's low edge on that axis (already clamped for
// ROIPool, raw for PSROIPool), `ext` the bin extent (roi extent / k), `n` the map size.
// The centre is evaluated in double ((T)idx + 0.5 is a double expression in the
// reference) with one fused multiply-add, then rounded once to T.
template <typename T>
__device__ __forceinline__ void bin_axis(T corner, T ext, int idx, int n, int& lo, int& hi) {
    const T ctr = static_cast<T>(__builtin_fma(static_cast<double>(static_cast<T>(idx)) + 0.5,
                                               static_cast<double>(ext),
                                               static_cast<double>(corner)));
    lo = static_cast<int>(floor_t(clamp01<T>(ctr - ext / T(2)) * static_cast<T>(n)));
    hi = static_cast<int>(ceil_t (clamp01<T>(ctr + ext / T(2)) * static_cast<T>(n)));
}

// ROIPool bin (r given by the 4 roi scalars): corner clamped to [0,1] first.
template <typename T>
__device__ __forceinline__ Bounds roi_bin(const T* __restrict__ roi, int i, int j, int H, int W, int k) {
    const T rI = roi[0], rJ = roi[1], rH = roi[2], rW = roi[3];
    Bounds b;
    bin_axis<T>(clamp01<T>(rI - rH / T(2)), rH / static_cast<T>(k), i, H, b.i0, b.i1);
    bin_axis<T>(clamp01<T>(rJ - rW / T(2)), rW / static_cast<T>(k), j, W, b.j0, b.j1);
    return b;
}

// PSROIPool cell: corner NOT clamped.
template <typename T>
__device__ __forceinline__ Bounds psroi_cell(const T* __restrict__ roi, int i, int j, int H, int W, int k) {
    const T rI = roi[0], rJ = roi[1], rH = roi[2], rW = roi[3];
    Bounds b;
    bin_axis<T>(rI - rH / T(2), rH / static_cast<T>(k), i, H, b.i0, b.i1);
    bin_axis<T>(rJ - rW / T(2), rW / static_cast<T>(k), j, W, b.j0, b.j1);
    return b;
}

// Is displacement target `t` (a row or column of FM1) visited from centre `p` on an axis
// of length n?  Loop of the reference: for (t = max(0,p-d); t < min(p+d, n); t += s).
__device__ __forceinline__ bool corr_axis_hit(int p, int t, int n, int d, int s) {
    const int lo = p - d > 0 ? p - d : 0;
    const int hi = p + d < n ? p + d : n;
    return t >= lo && t < hi && ((t - lo) % s) == 0;
}

inline int launch_status() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? D2T_OK : static_cast<int>(e);
}

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to a (function, device) pair.  Every launcher that needs more
// than 64 KB of dynamic LDS sets it for the calling thread's CURRENT device before its first launch there and
// returns the error if that fails; `done` (one per call site) is a write-once bit per device -- an idempotent
// cache, safe under concurrent callers (a race sets the attribute twice).
inline int ensure_dynamic_lds(const void* fn, int bytes, std::atomic<unsigned long long>& done)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return static_cast<int>(e);
    const unsigned long long bit = 1ull << (dev & 63);
    if (dev < 64 && (done.load(std::memory_order_acquire) & bit)) return D2T_OK;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return static_cast<int>(e);
    if (dev < 64) done.fetch_or(bit, std::memory_order_release);
    return D2T_OK;
}
#define D2T_ENSURE_DYNAMIC_LDS(fn, bytes)                                                                \
    do {                                                                                                 \
        static std::atomic<unsigned long long> done_{0};                                                 \
        const int rc_ = ::d2t::ensure_dynamic_lds(reinterpret_cast<const void*>(fn), (int)(bytes), done_); \
        if (rc_ != D2T_OK) return rc_;                                                                   \
    } while (0)

// Developer knobs (environment variables that force one of several measured designs) exist only in harness builds
// (-DD2T_LAB, or -DD2T_ENV_KNOBS: the knobs without the in-kernel stamps); the product library reads no environment.
#if defined(D2T_LAB) || defined(D2T_ENV_KNOBS)
inline int lab_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
inline const char* lab_env_str(const char* name) { return getenv(name); }
#else
inline int lab_env_int(const char*, int dflt) { return dflt; }
inline const char* lab_env_str(const char*) { return nullptr; }
#endif

// In-kernel clock reads of knob builds (-DD2T_ENV_KNOBS; lab/tools/kstamps.py): [workgroup][16] s_memtime values of thread 0.  A translation
// unit that stamps says D2T_KSTAMP_DEFINE(setter) once (the library is built without relocatable device code: the pointer is per unit) and
// D2T_KSTAMP(i) where it wants a clock read.  The product library is built without D2T_ENV_KNOBS: no stamp executes there.
#ifdef D2T_ENV_KNOBS
#define D2T_KSTAMP_DEFINE(setter)                                                                     \
    static __device__ unsigned long long* kstamps_;                                                   \
    static __device__ int kdbg_;                          /* ablation bits of a knob build (D2T_KDBG) */ \
    extern "C" int setter(void* p)                                                                    \
    {                                                                                                 \
        unsigned long long* q = static_cast<unsigned long long*>(p);                                  \
        return static_cast<int>(hipMemcpyToSymbol(HIP_SYMBOL(kstamps_), &q, sizeof(q)));              \
    }                                                                                                 \
    extern "C" int setter##_dbg(int bits)                                                             \
    {                                                                                                 \
        return static_cast<int>(hipMemcpyToSymbol(HIP_SYMBOL(kdbg_), &bits, sizeof(bits)));           \
    }
#define D2T_KSTAMP(i)                                                                                 \
    do {                                                                                              \
        if (kstamps_ && threadIdx.x == 0) {                                                           \
            unsigned long long t_;                                                                    \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            kstamps_[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = t_;                  \
        }                                                                                             \
    } while (0)
#define D2T_KSTAMP_RT(i)                                                                              \
    do {                                                                                              \
        if (kstamps_ && threadIdx.x == 0) {                                                           \
            unsigned long long t_;                                                                    \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
            kstamps_[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = t_;                  \
        }                                                                                             \
    } while (0)
#define D2T_KCLK(var) do { if (kstamps_) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); } while (0)
#define D2T_KSTAMP_PUT(i, val) do { if (kstamps_ && threadIdx.x == 0) kstamps_[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = (val); } while (0)
#define D2T_KSTAMP_ONLY(...) __VA_ARGS__
#define D2T_KDBG (kdbg_)
#else
#define D2T_KSTAMP_DEFINE(setter)
#define D2T_KSTAMP(i)
#define D2T_KSTAMP_RT(i)
#define D2T_KCLK(var)
#define D2T_KSTAMP_PUT(i, val)
#define D2T_KSTAMP_ONLY(...)
#define D2T_KDBG 0
#endif

inline bool fits_i32(long long v) { return v >= 0 && v <= 2147483647LL; }

inline int grid_for(long long work_items, int block, int cap = 256 * 16) {
    long long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return static_cast<int>(g);
}

}  // namespace d2t
