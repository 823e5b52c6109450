// d2t_kernels.hpp -- internal launch interface between the C ABI (d2t_capi.hip) and the
// kernel translation units.  Every launcher is asynchronous on `stream`, allocates
// nothing, and returns D2T_OK or a hipError_t.
#pragma once
#include <hip/hip_runtime.h>
#include "d2t_common.hpp"

namespace d2t {

// ---- type-generic, reference-order kernels (d2t_generic.hip); f32 and f64 ----
// (ps, cs, bs): where cell c of pixel p of item b lives, b*bs + p*ps + c*cs; ps = 0 selects the reference's
// (B,H,W,2d+1,2d+1) layout
template <typename T> int corr_fwd_generic(const T* fm0, const T* fm1, T* out,
                                           int B, int C, int H, int W, int d, int s, hipStream_t st,
                                           int ps = 0, int cs = 0, long long bs = 0);
template <typename T> int corr_bwd_generic(const T* gout, const T* fm0, const T* fm1, T* g0, T* g1,
                                           int B, int C, int H, int W, int d, int s, hipStream_t st,
                                           int ps = 0, int cs = 0, long long bs = 0);
template <typename T> int roipool_fwd_generic(const T* fm, const T* rois, T* out,
                                              int R, int C, int H, int W, int k, hipStream_t st);
template <typename T> int roipool_bwd_generic(const T* gout, const T* rois, T* gin, int32_t* bins,
                                              int R, int C, int H, int W, int k, hipStream_t st);
template <typename T> int psroipool_fwd_generic(const T* fm, const T* rois, T* out,
                                                int R, int nT, int H, int W, int k, hipStream_t st);
template <typename T> int psroipool_bwd_generic(const T* gout, const T* rois, T* gin, int32_t* cells,
                                                int R, int nT, int H, int W, int k, hipStream_t st);
template <typename T> int roipool_bins(const T* rois, int32_t* bounds, int R, int H, int W, int k, hipStream_t st);
template <typename T> int psroipool_bins(const T* rois, int32_t* bounds, int R, int H, int W, int k, hipStream_t st);
int psroipool_channels(int32_t* ch, int nT, int k, hipStream_t st);
int corr_mask(uint8_t* mask, int H, int W, int d, int s, hipStream_t st);

// ---- correlation outside the tuned envelope, reference layout (d2t_corr_blocked.hip): the generic kernels' arithmetic and order
// (bit-identical results); f32 with d_max <= 14: LDS-tiled forward, register-tiled backward (ws = gradOut re-indexed by displaced
// pixel | both maps with zero-padded rows); otherwise four cells per thread (forward) / gradOut staged in LDS (backward)
template <typename T> bool   corr_blocked_supported(int B, int C, int H, int W, int d, int s);
template <typename T> size_t corr_bwd_blocked_ws_bytes(int B, int C, int H, int W, int d, int s);
template <typename T> int    corr_fwd_blocked(const T* fm0, const T* fm1, T* out, int B, int C, int H, int W, int d, int s, hipStream_t st);
template <typename T> int    corr_bwd_blocked(const T* gout, const T* fm0, const T* fm1, T* g0, T* g1,
                                              int B, int C, int H, int W, int d, int s, void* ws, hipStream_t st);

// ---- correlation forward outside the tuned envelope on the f32 matrix pipe (d2t_corr_fwd_mfma.hip): d_max <= 8, any stride / map; bit-identical
bool corr_fwd_mfma_supported(int B, int C, int H, int W, int d, int s);
int  corr_fwd_mfma_f32(const float* fm0, const float* fm1, float* out, int B, int C, int H, int W, int d, int s, hipStream_t st);

// ---- ROIPool backward outside the tuned envelope (d2t_pool_lists.hip; k <= 32): per-row lists of (RoI, bin row) pairs + per-(RoI, column)
// masks of bin columns, built once, shared by all channels; bit-identical to roipool_bwd_generic.
// ws: bins | counter | list heads | list entries | column masks | gradOut / n by (bin, channel)
template <typename T> bool   roipool_bwd_lists_supported(int R, int C, int H, int W, int k);
template <typename T> size_t roipool_bwd_lists_ws_bytes(int R, int C, int H, int W, int k);
template <typename T> int    roipool_bwd_lists(const T* gout, const T* rois, T* gin, void* ws, int R, int C, int H, int W, int k, hipStream_t st);
template <typename T> int    roipool_bwd_generic_gated(const T* gout, const int32_t* bins, T* gin, const unsigned long long* gate, long long cap,
                                                       int R, int C, int H, int W, int k, hipStream_t st);
// PSROIPool backward likewise: one list per (cell, map row).  ws: cells | counter | list heads | list entries (RoI, cell area, column range)
bool   psroipool_bwd_lists_supported(int R, int nT, int H, int W, int k);
size_t psroipool_bwd_lists_ws_bytes(int R, int nT, int H, int W, int k);
template <typename T> int    psroipool_bwd_lists(const T* gout, const T* rois, T* gin, void* ws, int R, int nT, int H, int W, int k, hipStream_t st);
template <typename T> int    psroipool_bwd_generic_gated(const T* gout, const int32_t* cells, T* gin, const unsigned long long* gate, long long cap,
                                                         int R, int nT, int H, int W, int k, hipStream_t st);

// ---- region proposals on the device (d2t_regions.hip): decode + confidence filter + top-k + greedy NMS
size_t region_filter_ws_bytes(int A, int max_dets);
int region_max_dets();
int region_filter_f32(const float* anchors, const float* offsets, const float* confs, int A,
                      float conf_thresh, int max_dets, float iou_thresh,
                      float* out_boxes, float* out_conf, int* out_idx, int* out_count, void* ws, hipStream_t st, int N = 1);

}  // namespace d2t
