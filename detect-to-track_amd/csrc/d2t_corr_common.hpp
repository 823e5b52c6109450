// d2t_corr_common.hpp -- pieces shared by the gfx950-tuned correlation translation units
// (d2t_corr_tuned.hip: forward kernels + the 16-wave / narrow-grid backward; d2t_corr_bwd8.hip: the
// 8-wave backward).  Geometry of the MFMA tiling, the XCD-aware block map, the non-finite repair.
#pragma once
#include "d2t_tuned.hpp"

namespace d2t { namespace tuned {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load, dword aligned
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

#define D2T_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int TP = 4;                      // p-tile edge: 4x4 pixels = MFMA M = 16
constexpr int DT = 8;                      // d_max the tuned kernels are built for
constexpr int WR = TP + 2 * DT - 1;        // 19 window rows (and needed columns)
constexpr int NCG = (WR + 3) / 4;          // 5 column groups per window row
constexpr int WC = NCG * 4;                // 20 loaded columns
constexpr int CW = 2 * DT + 1;             // 17
constexpr int CELLS = CW * CW;             // 289
constexpr int NACT = 5;                    // backward: tiles alive during one super-step
constexpr int KB_SS = 5;                   // backward: k-blocks (16 window slots each) per super-step of 4 map rows

// Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of logical tiles
// (bijective for any grid size).  Placement only affects L2 reuse, never results.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// s_barrier that waits for this wave's LDS traffic and for all but its N youngest vector-memory
// operations (a __syncthreads() would drain vmcnt to 0 and serialise every load in flight).
template <int N>
__device__ __forceinline__ void dma_wait_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Buffer descriptor over [p, p + bytes) from values the compiler can SEE are wave-uniform: a descriptor it
// believes divergent is kept in VGPRs and every access through it becomes a waterfall loop (v_readfirstlane x4,
// compare, s_and_saveexec, ...).  p and bytes must really be the same in all lanes.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* p, unsigned bytes)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                             __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

__device__ __forceinline__ bool nonfinite4(const f32x4& d)
{
    const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(d[0]), __builtin_fabsf(d[1])),
                                    __builtin_fmaxf(__builtin_fabsf(d[2]), __builtin_fabsf(d[3])));
    return !(m <= 3.4028234663852886e38f) || d[0] != d[0] || d[1] != d[1] || d[2] != d[2] || d[3] != d[3];
}

// ------------------------------------------------------------------------------------
// Non-finite inputs (backward).  The MFMA form multiplies window slots a tile pixel does not own by
// an exact 0 weight; an Inf / NaN in the feature map S then turns that product into NaN and poisons
// the whole accumulator row (channel) of the tile, including pixels whose own window does not
// contain the bad value -- the reference (pointwise_correlation_cuda.cu:154-171) only ever touches a
// pixel's own window.  Every poisoned element is itself non-finite, so a wave that stored a
// non-finite value recomputes ITS region (16 channels x map rows [y0, y1) of the strip) in the
// reference's own form at the end of the kernel: gather loops, fused multiply-add chain, ascending
// order -- the same arithmetic as the type-generic kernel.  Cold code: never runs on finite inputs.
// ------------------------------------------------------------------------------------
static __device__ __attribute__((noinline)) void strip_repair(int role, int lane, const float* __restrict__ gb,
                                                              const float* __restrict__ Sb, float* __restrict__ gxb,
                                                              int cw, int C, int H, int W, int j0, int ps, int cs, int y0, int y1)
{
    const int HW = H * W, nr = y1 - y0;                              // map rows [y0, y1) of the strip
    for (int e = lane; e < 16 * nr * TP; e += 64) {
        const int c = cw + e / (nr * TP), rem = e % (nr * TP), y = y0 + rem / TP, x = j0 + rem % TP;
        if (c >= C || x >= W) continue;
        const float* sc = Sb + (size_t)c * HW;
        float a = 0.f;
        if (role == 0) {                                             // centre (y,x): walk its window of FM1
            const int lo_i = y - DT > 0 ? y - DT : 0, hi_i = y + DT < H ? y + DT : H;
            const int lo_j = x - DT > 0 ? x - DT : 0, hi_j = x + DT < W ? x + DT : W;
            const float* gc = gb + (size_t)(y * W + x) * ps;
            for (int di = lo_i; di < hi_i; ++di)
                for (int dj = lo_j; dj < hi_j; ++dj)
                    a = __builtin_fmaf(gc[(size_t)((di - y + DT) * CW + (dj - x + DT)) * cs], sc[di * W + dj], a);
        } else {                                                     // displaced (y,x): the centres that reach it
            const int i_lo = y - DT > 0 ? y - DT : 0, i_hi = y + DT < H - 1 ? y + DT : H - 1;
            const int j_lo = x - DT > 0 ? x - DT : 0, j_hi = x + DT < W - 1 ? x + DT : W - 1;
            for (int i = i_lo; i <= i_hi; ++i) {
                if (!corr_axis_hit(i, y, H, DT, 1)) continue;
                for (int j = j_lo; j <= j_hi; ++j) {
                    if (!corr_axis_hit(j, x, W, DT, 1)) continue;
                    a = __builtin_fmaf(gb[(size_t)(i * W + j) * ps + (size_t)((y - i + DT) * CW + (x - j + DT)) * cs], sc[i * W + j], a);
                }
            }
        }
        gxb[(size_t)c * HW + y * W + x] = a;
    }
}

// d2t_corr_fwd_band.hip: forward of small grids, a tile's window split over workgroups (bit-identical); cfg 0 = not taken
int  corr_fwd_band_config(int B, int H, int W);
int  corr_fwd_band_f32(int cfg, int nl, const float* const* fm0, const float* const* fm1, float* const* out, const int* C, int B, int H, int W,
                       CellLayout lay, hipStream_t st);

// d2t_corr_bwd8.hip
bool corr_bwd8_supported(int B, int C, int H, int W, int ps, int cs);
int  corr_bwd8_f32(const float* gout, const float* fm0, const float* fm1, float* g0, float* g1,
                   int B, int C, int H, int W, hipStream_t st, int variant = 0);

#ifdef D2T_LAB_KERNELS
// lab/d2t_corr_bwd8w.hip (strips 8 pixels wide x 128 channels: built, measured 81 against 70 us, lost -- notebook R4-3)
bool corr_bwd8w_supported(int B, int C, int H, int W, int ps, int cs);
long long corr_bwd8w_workgroups(int B, int C, int W);
int  corr_bwd8w_f32(const float* gout, const float* fm0, const float* fm1, float* g0, float* g1,
                    int B, int C, int H, int W, hipStream_t st);

// lab/d2t_corr_bwd8bf.hip (bf16 matrix pipe, operands split in three: notebook 4.3)
bool corr_bwd8bf_supported(int B, int C, int H, int W, int ps, int cs);
int  corr_bwd8bf_f32(const float* gout, const float* fm0, const float* fm1, float* g0, float* g1,
                     int B, int C, int H, int W, hipStream_t st);
#endif

}}  // namespace d2t::tuned
