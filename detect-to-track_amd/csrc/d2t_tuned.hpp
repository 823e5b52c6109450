// d2t_tuned.hpp -- interface of the gfx950-tuned f32 kernels.  Each op exposes
//   *_supported(...)  host predicate: do the kernel's assumptions hold for this shape?
//   *_ws_bytes(...)   device scratch it needs (0 = none)
//   *_f32(...)        async launcher; returns D2T_OK or a hipError_t
// The C ABI falls back to the type-generic kernels when *_supported is false.
#pragma once
#include <hip/hip_runtime.h>
#include "d2t_common.hpp"

namespace d2t { namespace tuned {

bool   corr_fwd_supported(int B, int C, int H, int W, int d, int s);
size_t corr_fwd_ws_bytes(int B, int C, int H, int W, int d, int s);
int    corr_fwd_f32(const float* fm0, const float* fm1, float* out,
                    int B, int C, int H, int W, int d, int s, void* ws, size_t ws_bytes, hipStream_t st);

// Cell (ci,cj) of pixel (i,j) of batch item b at b*bs + (i*W+j)*ps + (ci*17+cj)*cs floats (see d2t_corr_tuned.hip)
struct CellLayout { int ps, cs; long long bs; };
constexpr int MAXLV = 4;
size_t corr_fwd_levels_ws_bytes(int nl, const int* C, int B, int H, int W);
int    corr_fwd_levels_f32(int nl, const float* const* fm0, const float* const* fm1, float* const* out, const int* C,
                           int B, int H, int W, CellLayout lay, hipStream_t st, void* ws = nullptr, size_t ws_bytes = 0);
int    corr_bwd_levels_f32(int nl, const float* const* gout, const float* const* fm0, const float* const* fm1,
                           float* const* g0, float* const* g1, const int* C, int B, int H, int W, CellLayout lay, hipStream_t st,
                           int bwd_variant = 0, void* ws = nullptr, size_t ws_bytes = 0);
size_t corr_bwd_levels_ws_bytes(int nl, const int* C, int B, int H, int W, CellLayout lay);   // channel-major gradient: its re-laid copy

bool   corr_bwd_supported(int B, int C, int H, int W, int d, int s);
size_t corr_bwd_ws_bytes(int B, int C, int H, int W, int d, int s);
int    corr_bwd_f32(const float* gout, const float* fm0, const float* fm1, float* g0, float* g1,
                    int B, int C, int H, int W, int d, int s, void* ws, hipStream_t st, int bwd_variant = 0);

bool   roipool_fwd_supported(int R, int C, int H, int W, int k);
size_t roipool_fwd_ws_bytes(int R, int C, int H, int W, int k);
int    roipool_fwd_f32(const float* fm, const float* rois, float* out,
                       int R, int C, int H, int W, int k, void* ws, hipStream_t st);

// d2t_roipool_fwd_direct.hip: k = 7 in the reference's order (bit-identical)
bool   roipool_fwd_direct_supported(int R, int C, int H, int W, int k);
int    roipool_fwd_direct_f32(const float* fm, const float* rois, float* out, int R, int C, int H, int W, hipStream_t st);

bool   roipool_bwd_supported(int R, int C, int H, int W, int k);
size_t roipool_bwd_ws_bytes(int R, int C, int H, int W, int k);
int    roipool_bwd_f32(const float* gout, const float* rois, float* gin,
                       int R, int C, int H, int W, int k, void* ws, hipStream_t st);

bool   psroipool_fwd_supported(int R, int nT, int H, int W, int k);
size_t psroipool_fwd_ws_bytes(int R, int nT, int H, int W, int k);
int    psroipool_fwd_f32(const float* fm, const float* rois, float* out, int R, int nT, int H, int W, int k,
                         void* ws, hipStream_t st);
int    psroipool_fwd_small_f32(const float* fm, const float* rois, float* out, int R, int nT, int H, int W, int k,
                               hipStream_t st);
bool   psroipool_bwd_supported(int R, int nT, int H, int W, int k);
size_t psroipool_bwd_ws_bytes(int R, int nT, int H, int W, int k);
int    psroipool_bwd_f32(const float* gout, const float* rois, float* gin,
                         int R, int nT, int H, int W, int k, void* ws, hipStream_t st);

// d2t_pool_sorted.hip
bool   psroipool_bwd_sorted_supported(int R, int nT, int H, int W, int k);
size_t psroipool_bwd_sorted_ws_bytes(int R, int nT, int H, int W, int k);
int    psroipool_bwd_sorted_f32(const float* gout, const float* rois, float* gin,
                                int R, int nT, int H, int W, int k, void* ws, hipStream_t st);

}}  // namespace d2t::tuned
