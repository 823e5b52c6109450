/*
 * d2t_ops.h -- C ABI of libd2t_ops.so: the MI355X (gfx950) implementation of the
 * detect-to-track custom-op hot path (PointwiseCorrelation, ROIPool, PSROIPool;
 * forward + backward, f32 and f64).
 *
 * This is the drop-in boundary.  Each entry point replaces one function of the
 * reference's pybind module `_ext` (paths relative to
 * /root/reference/detect_to_track/models/):
 *
 *   d2t_corr_fwd_*       <- pointwise_correlation_forward   pointwise_correlation/pointwise_correlation.cpp:23-33
 *                           (launcher pointwise_correlation_cuda.cu:178-210)
 *   d2t_corr_bwd_*       <- pointwise_correlation_backward  pointwise_correlation/pointwise_correlation.cpp:36-48
 *                           (launcher pointwise_correlation_cuda.cu:214-249)
 *                           (d2t_corr_bwd_workspace_bytes is non-zero OUTSIDE the tuned envelope -- for the backward that includes
 *                           maps lower than 17 rows -- and for f64: with that
 *                           scratch the tiled / blocked kernels run, without it the thread-per-element ones -- same values)
 *   d2t_roipool_fwd_*    <- roipool_forward                 roipool/roipool.cpp:22-32   (roipool_cuda.cu:130-157)
 *   d2t_roipool_bwd_*    <- roipool_backward                roipool/roipool.cpp:35-45   (roipool_cuda.cu:160-190)
 *   d2t_psroipool_fwd_*  <- ps_roipool_forward              ps_roipool/ps_roipool.cpp:23-34 (ps_roipool_cuda.cu:144-174)
 *   d2t_psroipool_bwd_*  <- ps_roipool_backward             ps_roipool/ps_roipool.cpp:37-47 (ps_roipool_cuda.cu:177-204)
 *
 * Conventions
 *   - Plain pointers and sizes only; no torch / ATen types.  All pointers are DEVICE
 *     pointers on the current HIP device, contiguous row-major, naturally aligned.
 *   - The caller allocates every output; the library writes EVERY element of it (the
 *     reference's launchers pre-zero their outputs with at::zeros; here the structural
 *     zeros are written by the kernels, so torch.empty is enough).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls are
 *     asynchronous, allocate nothing, synchronise nothing and are graph-capturable.
 *   - `ws` is caller-provided device scratch of at least the matching
 *     d2t_*_workspace_bytes(...) bytes (may be NULL when that returns 0), 16-byte aligned
 *     (D2T_EINVAL otherwise; every device allocation is; not checked when ws_bytes is 0).
 *   - Return value: 0 on success; a negative D2T_E* code for argument errors; a
 *     positive value is a hipError_t from the launch.  The library keeps no pointers
 *     after return and is re-entrant (autograd calls backward from another thread).  Its only
 *     process state is an idempotent write-once flag per (kernel, device) recording that the
 *     kernel's dynamic-LDS limit has been raised on that device; a failure to raise it is
 *     returned as the hipError_t.  The library reads no environment variable.
 *   - All extents are int32, like the reference's index arithmetic
 *     (pointwise_correlation_cuda.cu:19-50); shapes whose element count would exceed
 *     2^31-1 are rejected with D2T_ETOOBIG.
 *
 * Layouts
 *   correlation:  fm0, fm1 (B,C,H,W); out / gout (B,H,W,2d+1,2d+1); gfm0, gfm1 (B,C,H,W)
 *   roipool:      fm (C,H,W); rois (R,4) = (centre_i, centre_j, height, width) as fractions
 *                 of the map; out / gout (R,C,k,k); gin (C,H,W)
 *   psroipool:    fm (nT*k*k,H,W); rois (R,4); out / gout (R,nT,k,k); gin (nT*k*k,H,W)
 */
#ifndef D2T_OPS_H
#define D2T_OPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* d2t_stream_t; /* hipStream_t */

enum {
    D2T_OK       = 0,
    D2T_EINVAL   = -1, /* null pointer, negative extent, stride < 1, k < 1 ... */
    D2T_ETOOBIG  = -2, /* an element count does not fit int32 */
    D2T_EWS      = -3  /* workspace missing or too small */
};

/* library / ABI version (major*100 + minor) and a static message for a return code */
int         d2t_version(void);
const char* d2t_error_string(int code);

/* The tuned envelope and what leaving it costs.  The gfx950-tuned kernels cover what the reference model uses
 * (cfg/default.yaml:48,50: D_MAX 8, K 7; correlation_tracker.py:26: stride 1): correlation d_max = 8, stride 1, W >= 20 (backward: H >= 17 too);
 * pooling k = 7; float32.  Everything else -- and all of float64 -- runs type-generic kernels in the reference's order whose
 * results are bit-identical to the thread-per-element kernels D2T_IMPL_GENERIC selects (tested) -- with ONE exception: the float32
 * ROIPool forward with k <= 16, k != 7 and at least 32 RoIs runs the f64 summed-area kernel (within 1e-5 of the reference, NaN pattern exact;
 * tests/test_roipool.py::test_forward_any_bin_count_summed_area_tables; k = 7 itself is bit-identical since ABI 1.07:
 * d2t_roipool_fwd_direct.hip) -- in two tiers: the default
 * dispatch takes kernels that share the work the anchor repeats per thread (correlation f32, d_max <= 14: d2t_corr_fwd_mfma.hip /
 * d2t_corr_blocked.hip -- forward tiles with the FM1 window in LDS (d_max <= 8: on the f32 matrix pipe), backward four pixels x four
 * channels per thread from zero-padded row copies; pooling backward, k <= 32: d2t_pool_lists.hip -- per map row the lists of the bin rows / cells
 * that reach it, built once and shared by the row's pixels and all channels), D2T_IMPL_GENERIC the thread-per-element anchors.  Measured on an MI355X (tools/envelope_cost.py, us forward /
 * backward; thread-per-element anchor in brackets):
 *   correlation B=8 C=256 38x63   tuned 46 / 74     d_max=7: 93 / 170 (668 / 6,743)    stride 2: 101 / 200 (787 / 2,539)
 *                                                   f64: 1,217 / 2,068 (1,230 / 11,216)
 *   ROIPool R=300 C=1024 38x63    tuned 31 / 65     k=6: 29 / 121 (168 / 2,346)        f64: 262 / 243 (270 / 3,105)
 *                                 (forward: the summed-area kernel takes any k <= 16, k != 7 -- within 1e-5 of the reference; k = 7 is bit-identical)
 *   PSROIPool R=300 nT=21 38x63   tuned 18 / 32     k=6: 18 / 84 (18 / 510)            f64: 25 / 125 (25 / 640)
 * (the Python wrappers warn once when a float32 call leaves the envelope under D2T_IMPL_AUTO).
 *
 * Implementation selector of the f32 entry points (per call, no global state):
 *   D2T_IMPL_AUTO    the tuned gfx950 path when its preconditions hold and it is the faster one
 *                    (the summed-area ROIPool forward -- k != 7 -- with fewer than 32 RoIs takes the generic kernel), else generic
 *   D2T_IMPL_GENERIC the type-generic reference-order kernels (also used for f64)
 *   D2T_IMPL_MFMA    the tuned path, demanded: correlation returns D2T_EINVAL when its preconditions
 *                    (d_max = 8, stride 1, W >= 20; backward: H >= 17 as well -- the strip kernel keeps five
 *                    4-row tiles alive) do not hold; the pooling ops fall back to generic */
enum { D2T_IMPL_AUTO = 0, D2T_IMPL_GENERIC = 1, D2T_IMPL_MFMA = 2, D2T_IMPL_FAST = 5 };
/*   D2T_IMPL_FAST    as D2T_IMPL_AUTO, and the correlation FORWARD may re-associate its channel sum: small grids with
 *                    many channels (the model's B = 1 pairs with 1024 / 2048 channels) split the channels of a level
 *                    over several workgroups and add the partial sums in a fixed order -- deterministic, within 1e-5
 *                    of the reference's single ascending-channel chain, NOT bit-identical to it (needs the workspace
 *                    of d2t_corr_fwd*_workspace_bytes; without it the call runs as D2T_IMPL_AUTO).  Every other
 *                    selector keeps the forward bit-identical to the reference: exactness is the default, speed the
 *                    opt-in.
 * Any other value is D2T_EINVAL.  (ABI 1.05 also enumerated 3, 4, 6, 7: one specific correlation-backward kernel each, for A/B
 * measurements.  Those kernels lost their measurements and left the product library in 1.06; they build into the LAB library only --
 * `make -C detect-to-track_amd/csrc lab`, selectors in lab/csrc/d2t_lab_selectors.h.) */

/* ---------------- PointwiseCorrelation ---------------- */
size_t d2t_corr_fwd_workspace_bytes(int B, int C, int H, int W, int d, int stride, int elem_size);
size_t d2t_corr_bwd_workspace_bytes(int B, int C, int H, int W, int d, int stride, int elem_size);

int d2t_corr_fwd_f32(const float* fm0, const float* fm1, float* out,
                     int B, int C, int H, int W, int d, int stride,
                     void* ws, size_t ws_bytes, int impl, d2t_stream_t stream);
int d2t_corr_fwd_f64(const double* fm0, const double* fm1, double* out,
                     int B, int C, int H, int W, int d, int stride,
                     void* ws, size_t ws_bytes, int impl, d2t_stream_t stream);
int d2t_corr_bwd_f32(const float* gout, const float* fm0, const float* fm1,
                     float* gfm0, float* gfm1,
                     int B, int C, int H, int W, int d, int stride,
                     void* ws, size_t ws_bytes, int impl, d2t_stream_t stream);
int d2t_corr_bwd_f64(const double* gout, const double* fm0, const double* fm1,
                     double* gfm0, double* gfm1,
                     int B, int C, int H, int W, int d, int stride,
                     void* ws, size_t ws_bytes, int impl, d2t_stream_t stream);

/* ---------------- tracker glue: several correlations, channel-major output ----------------
 * The reference's CorrelationTracker (correlation_tracker.py:64-83) correlates three pyramid levels
 * with three calls, turns each (1,H,W,2d+1,2d+1) result into ((2d+1)^2,H,W) with view + permute, and
 * torch.cat's them behind the two RPN feature maps before ROIPool.  These entry points run
 * n_levels <= 4 correlations that share (B,H,W,d,stride) -- level l has C[l] channels -- in one
 * call, and can write / read the output in the layout the
 * concatenation needs:
 *   D2T_LAYOUT_REFERENCE      out[l] is (B,H,W,2d+1,2d+1), batch_stride ignored
 *   D2T_LAYOUT_CHANNEL_MAJOR  cell (ci,cj) of pixel (i,j) of item b at
 *                             out[l][b*batch_stride + (ci*(2d+1)+cj)*H*W + i*W + j]:
 *                             out[l] may point into a wider (channels,H,W) buffer
 * Values equal those of d2t_corr_fwd_f32 / d2t_corr_bwd_f32 on the same inputs (forward: bit for bit unless
 * impl = D2T_IMPL_FAST splits channels, see d2t_corr_fwd_levels_workspace_bytes).
 * Arrays of pointers / channel counts are HOST arrays of n_levels entries.                      */
enum { D2T_LAYOUT_REFERENCE = 0, D2T_LAYOUT_CHANNEL_MAJOR = 1 };

/* Scratch that lets small-grid calls (the model's B = 1 pairs with 1024 / 2048 channels) split the channels of a level
 * over several workgroups and add the partial sums in a fixed order (deterministic; agrees with the unsplit result to
 * f32 rounding, not bit for bit).  Only impl = D2T_IMPL_FAST uses it; with any other selector, or ws = NULL / too
 * small, the unsplit kernels run (bit-identical to the reference).  0 when the call would not split under
 * D2T_IMPL_FAST.  C: HOST array of n_levels entries. */
size_t d2t_corr_fwd_levels_workspace_bytes(int n_levels, const int* C, int B, int H, int W, int d, int stride);

int d2t_corr_fwd_levels_f32(int n_levels, const float* const* fm0, const float* const* fm1, float* const* out,
                            const int* C, int B, int H, int W, int d, int stride,
                            int layout, long long batch_stride,
                            void* ws, size_t ws_bytes, int impl, d2t_stream_t stream);
/* Scratch of the backward levels call for D2T_LAYOUT_CHANNEL_MAJOR: with it the gradient is first re-laid into the reference's
 * layout and the kernels of d2t_corr_bwd_f32 run on that copy (the tracker's three B = 1 levels: 205 us instead of 304 us with the
 * layout-aware thread-per-element kernels, which run when ws is NULL / too small -- D2T_EWS under D2T_IMPL_MFMA; same 1e-5 contract).
 * 0 for the reference layout and for shapes outside the tuned envelope. */
size_t d2t_corr_bwd_levels_workspace_bytes(int n_levels, const int* C, int B, int H, int W, int d, int stride, int layout);

int d2t_corr_bwd_levels_f32(int n_levels, const float* const* gout, const float* const* fm0, const float* const* fm1,
                            float* const* gfm0, float* const* gfm1,
                            const int* C, int B, int H, int W, int d, int stride,
                            int layout, long long batch_stride,
                            void* ws, size_t ws_bytes, int impl, d2t_stream_t stream);

/* ---------------- ROIPool (average) ----------------
 * Backward outside the tuned envelope (k != 7, f64; k <= 32): d2t_roipool_bwd_workspace_bytes covers the per-row bin lists and the
 * (bin, channel) copy of gradOut / n of d2t_pool_lists.hip (about the size of gradOut); a caller that passes only the R k^2 16
 * bytes of the bin table gets the thread-per-pixel kernel -- same values, ~20x slower.  PSROIPool likewise. */
size_t d2t_roipool_fwd_workspace_bytes(int R, int C, int H, int W, int k, int elem_size);
size_t d2t_roipool_bwd_workspace_bytes(int R, int C, int H, int W, int k, int elem_size);

int d2t_roipool_fwd_f32(const float* fm, const float* rois, float* out,
                        int R, int C, int H, int W, int k,
                        void* ws, size_t ws_bytes, int impl, d2t_stream_t stream);
int d2t_roipool_fwd_f64(const double* fm, const double* rois, double* out,
                        int R, int C, int H, int W, int k,
                        void* ws, size_t ws_bytes, int impl, d2t_stream_t stream);
int d2t_roipool_bwd_f32(const float* gout, const float* rois, float* gin,
                        int R, int C, int H, int W, int k,
                        void* ws, size_t ws_bytes, int impl, d2t_stream_t stream);
int d2t_roipool_bwd_f64(const double* gout, const double* rois, double* gin,
                        int R, int C, int H, int W, int k,
                        void* ws, size_t ws_bytes, int impl, d2t_stream_t stream);

/* ---------------- PSROIPool (position-sensitive average) ---------------- */
size_t d2t_psroipool_fwd_workspace_bytes(int R, int nT, int H, int W, int k, int elem_size);
size_t d2t_psroipool_bwd_workspace_bytes(int R, int nT, int H, int W, int k, int elem_size);

int d2t_psroipool_fwd_f32(const float* fm, const float* rois, float* out,
                          int R, int nT, int H, int W, int k,
                          void* ws, size_t ws_bytes, int impl, d2t_stream_t stream);
int d2t_psroipool_fwd_f64(const double* fm, const double* rois, double* out,
                          int R, int nT, int H, int W, int k,
                          void* ws, size_t ws_bytes, int impl, d2t_stream_t stream);
int d2t_psroipool_bwd_f32(const float* gout, const float* rois, float* gin,
                          int R, int nT, int H, int W, int k,
                          void* ws, size_t ws_bytes, int impl, d2t_stream_t stream);
int d2t_psroipool_bwd_f64(const double* gout, const double* rois, double* gin,
                          int R, int nT, int H, int W, int k,
                          void* ws, size_t ws_bytes, int impl, d2t_stream_t stream);

/* ---------------- integer introspection (parity of index arithmetic) ----------------
 * These run the SAME device functions the pooling / correlation kernels use and write the
 * integer quantities the parity contract requires bit-exact:
 *   bins:     (R,k,k,4) int32 = {i0, i1, j0, j1} pixel bounds of every bin / cell
 *             (roipool_cuda.cu:47-50, ps_roipool_cuda.cu:51-54)
 *   channels: (nT,k,k) int32 position-sensitive channel of every output (ps_roipool_cuda.cu:58)
 *   mask:     (H,W,2d+1,2d+1) uint8, 1 where the correlation loops write
 *             (pointwise_correlation_cuda.cu:92-93)                                    */
int d2t_roipool_bins_f32(const float* rois, int32_t* bounds, int R, int H, int W, int k, d2t_stream_t stream);
int d2t_roipool_bins_f64(const double* rois, int32_t* bounds, int R, int H, int W, int k, d2t_stream_t stream);
int d2t_psroipool_bins_f32(const float* rois, int32_t* bounds, int R, int H, int W, int k, d2t_stream_t stream);
int d2t_psroipool_bins_f64(const double* rois, int32_t* bounds, int R, int H, int W, int k, d2t_stream_t stream);
int d2t_psroipool_channels(int32_t* channels, int nT, int k, d2t_stream_t stream);
int d2t_corr_mask(uint8_t* mask, int H, int W, int d, int stride, d2t_stream_t stream);

/* ---------------- region proposals: decode + confidence filter + top-k + NMS on the device ----------------
 * Replaces the host round trip between the RPN and the R-FCN heads (reference trainer.py:178-207,
 * inference.py:78-91): anchors + offsets -> boxes (data/encoding.py:182-206, frcnn_box_decode), then the three
 * `ml_utils` filters the reference composes -- ConfidenceFilter(conf_thresh), MaxDetFilter(max_dets),
 * NMSFilter(iou_thresh) (trainer.py:98-102) -- restated as: keep conf > conf_thresh; keep the max_dets highest
 * confidences (ties: lower anchor index first); greedy NMS in descending confidence, a kept box removes every later
 * box with IoU > iou_thresh.  (`ml_utils` is not vendored with the reference: parity for the filters is unpinned.)
 *   anchors, offsets (A,4); confs (A): device, float32; boxes are (centre_i, centre_j, height, width) fractions.
 *   anchors, offsets, out_boxes and ws are read / written as 16-byte vectors: they must be 16-byte aligned
 *   (D2T_EINVAL otherwise); confs, out_conf, out_idx and out_count need 4-byte alignment only.
 *   out_boxes (max_dets,4), out_conf (max_dets), out_idx (max_dets, anchor index or -1), out_count (1): device.
 *   Survivors come first, in descending confidence; the rest of the lists is padding (zero boxes, index -1), so the
 *   caller can keep static shapes and never reads the count back.  max_dets <= 4096.                            */
size_t d2t_region_filter_workspace_bytes(int A, int max_dets);
int d2t_region_filter_f32(const float* anchors, const float* offsets, const float* confs, int A,
                          float conf_thresh, int max_dets, float iou_thresh,
                          float* out_boxes, float* out_conf, int32_t* out_idx, int32_t* out_count,
                          void* ws, size_t ws_bytes, d2t_stream_t stream);
/* The same for the N frames of a step in ONE call (the reference runs the host pipeline once per frame, trainer.py:178-207):
 * anchors (A,4) shared; offsets (N,A,4), confs (N,A); out_boxes (N,max_dets,4), out_conf / out_idx (N,max_dets), out_count (N);
 * ws of N * d2t_region_filter_workspace_bytes(A, max_dets) bytes.  Frame f's results are those of a single-frame call on
 * frame f's slices, bit for bit.  (max_dets * 16 is a multiple of 16, so every frame's box list stays 16-byte aligned.) */
int d2t_region_filter_batched_f32(const float* anchors, const float* offsets, const float* confs, int N, int A,
                                  float conf_thresh, int max_dets, float iou_thresh,
                                  float* out_boxes, float* out_conf, int32_t* out_idx, int32_t* out_count,
                                  void* ws, size_t ws_bytes, d2t_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* D2T_OPS_H */
