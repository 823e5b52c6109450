// d2t_capi.hip -- the extern "C" surface declared in include/d2t_ops.h.
//
// Argument validation, int32 range guards, workspace accounting and dispatch between the
// tuned gfx950 kernels and the type-generic ones.  No allocation, no synchronisation, no
// environment reads; the only process state is the per-device dynamic-LDS flags of d2t_common.hpp.
#include "d2t_kernels.hpp"
#include "d2t_tuned.hpp"
#ifdef D2T_LAB_KERNELS
#include "../../lab/csrc/d2t_lab_selectors.h"
#endif

using namespace d2t;

namespace {

inline hipStream_t as_stream(d2t_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// common checks for the correlation entry points
int check_corr(const void* a, const void* b, const void* c, int B, int C, int H, int W, int d, int s)
{
    if (B < 0 || C < 0 || H < 0 || W < 0 || d < 0 || s < 1) return D2T_EINVAL;
    const long long cw = 2LL * d + 1;
    const long long out_n = 1LL * B * H * W * cw * cw;
    const long long in_n = 1LL * B * C * H * W;
    if (!fits_i32(cw * cw) || !fits_i32(1LL * B * H * W) || !fits_i32(out_n) || !fits_i32(in_n)) return D2T_ETOOBIG;
    if (in_n > 0 && (!a || !b)) return D2T_EINVAL;
    if (out_n > 0 && !c) return D2T_EINVAL;
    return D2T_OK;
}

int check_pool(const void* fm, const void* rois, const void* out, int R, int C, int H, int W, int k)
{
    if (R < 0 || C < 0 || H < 0 || W < 0 || k < 1) return D2T_EINVAL;
    const long long out_n = 1LL * R * C * k * k;
    const long long in_n = 1LL * C * H * W;
    if (!fits_i32(out_n) || !fits_i32(in_n) || !fits_i32(4LL * R * k * k)) return D2T_ETOOBIG;
    if (in_n > 0 && !fm) return D2T_EINVAL;
    if (R > 0 && !rois) return D2T_EINVAL;
    if (out_n > 0 && !out) return D2T_EINVAL;
    return D2T_OK;
}

// The four selectors of include/d2t_ops.h.  Values 3, 4, 6, 7 named lab kernels in ABI 1.05; the product library rejects them
// since 1.06 -- they exist in the lab build only (make lab: -DD2T_LAB_KERNELS, lab/csrc/d2t_lab_selectors.h).
#ifdef D2T_LAB_KERNELS
inline bool impl_ok(int impl) { return impl >= D2T_IMPL_AUTO && impl <= D2T_LAB_IMPL_STRIP4; }
// backward kernel of the tuned path: 0 the product's, 1 the 16-wave strip kernel, 3 bf16x3, 4 strips 8 pixels wide, 5 strips 4 pixels wide
inline int bwd_variant_of(int impl)
{
    return impl == D2T_LAB_IMPL_STRIP16 ? 1 : impl == D2T_LAB_IMPL_BF16X3 ? 3 : impl == D2T_LAB_IMPL_WIDE8 ? 4 : impl == D2T_LAB_IMPL_STRIP4 ? 5 : 0;
}
#else
inline bool impl_ok(int impl) { return impl == D2T_IMPL_AUTO || impl == D2T_IMPL_GENERIC || impl == D2T_IMPL_MFMA || impl == D2T_IMPL_FAST; }
inline int bwd_variant_of(int) { return 0; }
#endif
// D2T_IMPL_MFMA (and the lab selectors) demand the tuned kernels (an unsupported shape is an error); AUTO / FAST fall back
inline bool demands_tuned(int impl) { return impl >= D2T_IMPL_MFMA && impl != D2T_IMPL_FAST; }

inline size_t bins_bytes(int R, int k) { return align_up((size_t)R * k * k * 4 * sizeof(int32_t), 256); }

}  // namespace

extern "C" {

// workspace pointers: 16-byte aligned (the kernels read and write 16-byte vectors and a 64-bit counter in it; any device allocation is)
static inline bool ws_misaligned_(const void* ws, size_t ws_bytes) { return ws && ws_bytes && (reinterpret_cast<uintptr_t>(ws) & 15u) != 0; }
#define ws_misaligned(ws) ws_misaligned_(ws, ws_bytes)   /* checked only where a workspace is passed: ws_bytes = 0 never touches it */

int d2t_version(void) { return 107; }   // 1.07: round 6 -- ROIPool forward k = 7 in the reference's own order (bit-identical, any number of RoIs); PSROIPool forward needs no workspace where it is one launch

#ifdef D2T_LAB_KERNELS
int d2t_lab_build(void) { return 1; }   // present in the lab build only (lab/csrc/d2t_lab_selectors.h)
#endif

const char* d2t_error_string(int code)
{
    switch (code) {
        case D2T_OK: return "ok";
        case D2T_EINVAL: return "invalid argument (null pointer, negative extent, stride < 1 or k < 1)";
        case D2T_ETOOBIG: return "an element count does not fit int32";
        case D2T_EWS: return "workspace missing or too small";
        default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "unknown d2t error";
    }
}

// ------------------------------------------------------------------ correlation
size_t d2t_corr_fwd_workspace_bytes(int B, int C, int H, int W, int d, int stride, int elem_size)
{
    return elem_size == 4 ? tuned::corr_fwd_ws_bytes(B, C, H, W, d, stride) : 0;
}
size_t d2t_corr_bwd_workspace_bytes(int B, int C, int H, int W, int d, int stride, int elem_size)
{
    // inside the tuned envelope: what the tuned kernels need (nothing); outside it (and for f64): gradOut re-indexed by displaced
    // pixel for the blocked kernels of d2t_corr_blocked.hip -- optional: without it the thread-per-element kernels run
    if (elem_size == 4) return tuned::corr_bwd_supported(B, C, H, W, d, stride) ? tuned::corr_bwd_ws_bytes(B, C, H, W, d, stride)
                                                                                 : corr_bwd_blocked_ws_bytes<float>(B, C, H, W, d, stride);
    return elem_size == 8 ? corr_bwd_blocked_ws_bytes<double>(B, C, H, W, d, stride) : 0;
}

int d2t_corr_fwd_f32(const float* fm0, const float* fm1, float* out, int B, int C, int H, int W, int d, int stride,
                     void* ws, size_t ws_bytes, int impl, d2t_stream_t stream)
{
    if (!impl_ok(impl)) return D2T_EINVAL;
    if (ws_misaligned(ws)) return D2T_EINVAL;
    int rc = check_corr(fm0, fm1, out, B, C, H, W, d, stride);
    if (rc != D2T_OK) return rc;
    if (impl != D2T_IMPL_GENERIC && tuned::corr_fwd_supported(B, C, H, W, d, stride)) {
        // the channel-split path of small grids is an opt-in (D2T_IMPL_FAST + workspace); every other selector runs the
        // unsplit kernels, whose result is bit-identical to the reference's ascending-channel chain
        const bool split = impl == D2T_IMPL_FAST && ws && ws_bytes >= tuned::corr_fwd_ws_bytes(B, C, H, W, d, stride);
        return tuned::corr_fwd_f32(fm0, fm1, out, B, C, H, W, d, stride, split ? ws : nullptr, split ? ws_bytes : 0, as_stream(stream));
    }
    if (demands_tuned(impl)) return D2T_EINVAL;     // tuned path demanded but not applicable
    if (impl != D2T_IMPL_GENERIC && corr_blocked_supported<float>(B, C, H, W, d, stride))
        return corr_fwd_blocked<float>(fm0, fm1, out, B, C, H, W, d, stride, as_stream(stream));   // same values, bit for bit
    return corr_fwd_generic<float>(fm0, fm1, out, B, C, H, W, d, stride, as_stream(stream));
}

int d2t_corr_fwd_f64(const double* fm0, const double* fm1, double* out, int B, int C, int H, int W, int d, int stride,
                     void*, size_t, int impl, d2t_stream_t stream)
{
    if (!impl_ok(impl)) return D2T_EINVAL;
    int rc = check_corr(fm0, fm1, out, B, C, H, W, d, stride);
    if (rc != D2T_OK) return rc;
    if (demands_tuned(impl)) return D2T_EINVAL;
    // (f64 forward: the four-cells-per-thread kernel measured 1,326 us against 1,230 us for the thread-per-cell one at B=8 C=256 38x63 --
    // 32-byte vector loads at 8-byte alignment buy nothing; the f64 forward stays on the thread-per-cell kernel)
    return corr_fwd_generic<double>(fm0, fm1, out, B, C, H, W, d, stride, as_stream(stream));
}

int d2t_corr_bwd_f32(const float* gout, const float* fm0, const float* fm1, float* gfm0, float* gfm1,
                     int B, int C, int H, int W, int d, int stride,
                     void* ws, size_t ws_bytes, int impl, d2t_stream_t stream)
{
    if (!impl_ok(impl)) return D2T_EINVAL;
    if (ws_misaligned(ws)) return D2T_EINVAL;
    int rc = check_corr(fm0, fm1, gout, B, C, H, W, d, stride);
    if (rc != D2T_OK) return rc;
    if (1LL * B * C * H * W > 0 && (!gfm0 || !gfm1)) return D2T_EINVAL;
    if (impl != D2T_IMPL_GENERIC && tuned::corr_bwd_supported(B, C, H, W, d, stride)) {
        if (ws_bytes < tuned::corr_bwd_ws_bytes(B, C, H, W, d, stride)) return D2T_EWS;
        return tuned::corr_bwd_f32(gout, fm0, fm1, gfm0, gfm1, B, C, H, W, d, stride, ws, as_stream(stream),
                                   bwd_variant_of(impl));
    }
    if (demands_tuned(impl)) return D2T_EINVAL;
    if (impl != D2T_IMPL_GENERIC && corr_blocked_supported<float>(B, C, H, W, d, stride) && ws &&
        ws_bytes >= corr_bwd_blocked_ws_bytes<float>(B, C, H, W, d, stride))
        return corr_bwd_blocked<float>(gout, fm0, fm1, gfm0, gfm1, B, C, H, W, d, stride, ws, as_stream(stream));
    return corr_bwd_generic<float>(gout, fm0, fm1, gfm0, gfm1, B, C, H, W, d, stride, as_stream(stream));
}

int d2t_corr_bwd_f64(const double* gout, const double* fm0, const double* fm1, double* gfm0, double* gfm1,
                     int B, int C, int H, int W, int d, int stride,
                     void* ws, size_t ws_bytes, int impl, d2t_stream_t stream)
{
    if (!impl_ok(impl)) return D2T_EINVAL;
    if (ws_misaligned(ws)) return D2T_EINVAL;
    int rc = check_corr(fm0, fm1, gout, B, C, H, W, d, stride);
    if (rc != D2T_OK) return rc;
    if (1LL * B * C * H * W > 0 && (!gfm0 || !gfm1)) return D2T_EINVAL;
    if (demands_tuned(impl)) return D2T_EINVAL;
    if (impl != D2T_IMPL_GENERIC && corr_blocked_supported<double>(B, C, H, W, d, stride) && ws &&
        ws_bytes >= corr_bwd_blocked_ws_bytes<double>(B, C, H, W, d, stride))
        return corr_bwd_blocked<double>(gout, fm0, fm1, gfm0, gfm1, B, C, H, W, d, stride, ws, as_stream(stream));
    return corr_bwd_generic<double>(gout, fm0, fm1, gfm0, gfm1, B, C, H, W, d, stride, as_stream(stream));
}

// ------------------------------------------------------------------ correlation, several levels / channel-major
static int check_levels(int n, const void* const* a, const void* const* b, const void* const* c, const int* C,
                        int B, int H, int W, int d, int s, int layout, long long bstride)
{
    if (n < 1 || n > tuned::MAXLV || !a || !b || !c || !C) return D2T_EINVAL;
    if (layout != D2T_LAYOUT_REFERENCE && layout != D2T_LAYOUT_CHANNEL_MAJOR) return D2T_EINVAL;
    const long long cells = (2LL * d + 1) * (2LL * d + 1);
    if (layout == D2T_LAYOUT_CHANNEL_MAJOR && B > 1 && bstride < cells * H * W) return D2T_EINVAL;
    for (int l = 0; l < n; ++l) {
        const int rc = check_corr(a[l], b[l], c[l], B, C[l], H, W, d, s);
        if (rc != D2T_OK) return rc;
    }
    if (layout == D2T_LAYOUT_CHANNEL_MAJOR && !fits_i32((B > 0 ? B - 1 : 0) * bstride + cells * H * W)) return D2T_ETOOBIG;
    return D2T_OK;
}

size_t d2t_corr_fwd_levels_workspace_bytes(int n, const int* C, int B, int H, int W, int d, int stride)
{
    if (n < 1 || n > tuned::MAXLV || !C) return 0;
    for (int l = 0; l < n; ++l)
        if (!tuned::corr_fwd_supported(B, C[l], H, W, d, stride)) return 0;
    return tuned::corr_fwd_levels_ws_bytes(n, C, B, H, W);
}

int d2t_corr_fwd_levels_f32(int n, const float* const* fm0, const float* const* fm1, float* const* out, const int* C,
                            int B, int H, int W, int d, int stride, int layout, long long bstride,
                            void* ws, size_t ws_bytes, int impl, d2t_stream_t stream)
{
    if (!impl_ok(impl)) return D2T_EINVAL;
    if (ws_misaligned(ws)) return D2T_EINVAL;
    int rc = check_levels(n, (const void* const*)fm0, (const void* const*)fm1, (const void* const*)out, C, B, H, W, d, stride, layout, bstride);
    if (rc != D2T_OK) return rc;
    const int cells = (2 * d + 1) * (2 * d + 1), HW = H * W;
    const tuned::CellLayout lay = layout == D2T_LAYOUT_CHANNEL_MAJOR ? tuned::CellLayout{1, HW, bstride}
                                                                     : tuned::CellLayout{cells, 1, 1LL * HW * cells};
    bool tuned_ok = impl != D2T_IMPL_GENERIC;
    for (int l = 0; l < n; ++l) tuned_ok = tuned_ok && tuned::corr_fwd_supported(B, C[l], H, W, d, stride);
    if (tuned_ok) {
        const bool split = impl == D2T_IMPL_FAST && ws && ws_bytes >= tuned::corr_fwd_levels_ws_bytes(n, C, B, H, W);
        return tuned::corr_fwd_levels_f32(n, fm0, fm1, out, C, B, H, W, lay, as_stream(stream), split ? ws : nullptr, split ? ws_bytes : 0);
    }
    if (demands_tuned(impl)) return D2T_EINVAL;
    for (int l = 0; l < n; ++l) {
        rc = corr_fwd_generic<float>(fm0[l], fm1[l], out[l], B, C[l], H, W, d, stride, as_stream(stream), lay.ps, lay.cs, lay.bs);
        if (rc != D2T_OK) return rc;
    }
    return D2T_OK;
}

size_t d2t_corr_bwd_levels_workspace_bytes(int n, const int* C, int B, int H, int W, int d, int stride, int layout)
{
    if (n < 1 || n > tuned::MAXLV || !C || B < 1 || H < 1 || W < 1 || layout != D2T_LAYOUT_CHANNEL_MAJOR) return 0;
    for (int l = 0; l < n; ++l)
        if (!tuned::corr_bwd_supported(B, C[l], H, W, d, stride)) return 0;
    return tuned::corr_bwd_levels_ws_bytes(n, C, B, H, W, tuned::CellLayout{1, H * W, 0});
}

int d2t_corr_bwd_levels_f32(int n, const float* const* gout, const float* const* fm0, const float* const* fm1,
                            float* const* gfm0, float* const* gfm1, const int* C,
                            int B, int H, int W, int d, int stride, int layout, long long bstride,
                            void* ws, size_t ws_bytes, int impl, d2t_stream_t stream)
{
    if (!impl_ok(impl)) return D2T_EINVAL;
    if (ws_misaligned(ws)) return D2T_EINVAL;
    int rc = check_levels(n, (const void* const*)fm0, (const void* const*)fm1, (const void* const*)gout, C, B, H, W, d, stride, layout, bstride);
    if (rc != D2T_OK) return rc;
    if (!gfm0 || !gfm1) return D2T_EINVAL;
    for (int l = 0; l < n; ++l)
        if (1LL * B * C[l] * H * W > 0 && (!gfm0[l] || !gfm1[l])) return D2T_EINVAL;
    const int cells = (2 * d + 1) * (2 * d + 1), HW = H * W;
    const tuned::CellLayout lay = layout == D2T_LAYOUT_CHANNEL_MAJOR ? tuned::CellLayout{1, HW, bstride}
                                                                     : tuned::CellLayout{cells, 1, 1LL * HW * cells};
    bool tuned_ok = impl != D2T_IMPL_GENERIC;
    for (int l = 0; l < n; ++l) tuned_ok = tuned_ok && tuned::corr_bwd_supported(B, C[l], H, W, d, stride);
#ifndef D2T_LAB_KERNELS
    // channel-major gradient: the tuned kernels read a re-laid copy in the caller's workspace; without it the layout-aware anchor runs
    if (tuned_ok && layout == D2T_LAYOUT_CHANNEL_MAJOR && (!ws || ws_bytes < tuned::corr_bwd_levels_ws_bytes(n, C, B, H, W, lay))) {
        if (demands_tuned(impl)) return D2T_EWS;
        tuned_ok = false;
    }
#endif
    if (tuned_ok) return tuned::corr_bwd_levels_f32(n, gout, fm0, fm1, gfm0, gfm1, C, B, H, W, lay, as_stream(stream),
                                                    bwd_variant_of(impl), ws, ws ? ws_bytes : 0);
    if (demands_tuned(impl)) return D2T_EINVAL;
    for (int l = 0; l < n; ++l) {
        rc = corr_bwd_generic<float>(gout[l], fm0[l], fm1[l], gfm0[l], gfm1[l], B, C[l], H, W, d, stride, as_stream(stream),
                                     lay.ps, lay.cs, lay.bs);
        if (rc != D2T_OK) return rc;
    }
    return D2T_OK;
}

// ------------------------------------------------------------------ roipool
size_t d2t_roipool_fwd_workspace_bytes(int R, int C, int H, int W, int k, int elem_size)
{
    return elem_size == 4 ? tuned::roipool_fwd_ws_bytes(R, C, H, W, k) : 0;
}
size_t d2t_roipool_bwd_workspace_bytes(int R, int C, int H, int W, int k, int elem_size)
{
    const size_t generic = bins_bytes(R, k);
    const size_t t = elem_size == 4 ? tuned::roipool_bwd_ws_bytes(R, C, H, W, k) : 0;
    size_t lists = 0;                                                 // outside the tuned envelope: d2t_pool_lists.hip
    if (elem_size == 4 && !tuned::roipool_bwd_supported(R, C, H, W, k)) lists = roipool_bwd_lists_ws_bytes<float>(R, C, H, W, k);
    if (elem_size == 8) lists = roipool_bwd_lists_ws_bytes<double>(R, C, H, W, k);
    const size_t m = generic > t ? generic : t;
    return m > lists ? m : lists;
}

int d2t_roipool_fwd_f32(const float* fm, const float* rois, float* out, int R, int C, int H, int W, int k,
                        void* ws, size_t ws_bytes, int impl, d2t_stream_t stream)
{
    if (!impl_ok(impl)) return D2T_EINVAL;
    if (ws_misaligned(ws)) return D2T_EINVAL;
    int rc = check_pool(fm, rois, out, R, C, H, W, k);
    if (rc != D2T_OK) return rc;
    // The summed-area forward (k != 7, or maps too large for the direct kernel) builds a table of every channel: for a handful of RoIs the
    // thread-per-output kernel, which only touches the boxes' pixels, is the faster one.  The direct k = 7 kernel (round 6) is not: the
    // tracker's 8 boxes of a 1891-channel map take 9 us there against 21 us (profiles/r06_*).
    const bool direct = k == 7 && tuned::roipool_fwd_direct_supported(R, C, H, W, k);
    const bool few_rois = (impl == D2T_IMPL_AUTO || impl == D2T_IMPL_FAST) && R < 32 && !direct;
    if (impl != D2T_IMPL_GENERIC && !few_rois && tuned::roipool_fwd_supported(R, C, H, W, k)) {
        if (ws_bytes < tuned::roipool_fwd_ws_bytes(R, C, H, W, k) || (!ws && ws_bytes)) return D2T_EWS;
        return tuned::roipool_fwd_f32(fm, rois, out, R, C, H, W, k, ws, as_stream(stream));
    }
    return roipool_fwd_generic<float>(fm, rois, out, R, C, H, W, k, as_stream(stream));
}

int d2t_roipool_fwd_f64(const double* fm, const double* rois, double* out, int R, int C, int H, int W, int k,
                        void*, size_t, int impl, d2t_stream_t stream)
{
    if (!impl_ok(impl)) return D2T_EINVAL;
    int rc = check_pool(fm, rois, out, R, C, H, W, k);
    if (rc != D2T_OK) return rc;
    return roipool_fwd_generic<double>(fm, rois, out, R, C, H, W, k, as_stream(stream));
}

int d2t_roipool_bwd_f32(const float* gout, const float* rois, float* gin, int R, int C, int H, int W, int k,
                        void* ws, size_t ws_bytes, int impl, d2t_stream_t stream)
{
    if (!impl_ok(impl)) return D2T_EINVAL;
    if (ws_misaligned(ws)) return D2T_EINVAL;
    int rc = check_pool(gin, rois, gout, R, C, H, W, k);
    if (rc != D2T_OK) return rc;
    if (impl != D2T_IMPL_GENERIC && tuned::roipool_bwd_supported(R, C, H, W, k)) {
        if (!ws || ws_bytes < tuned::roipool_bwd_ws_bytes(R, C, H, W, k)) return D2T_EWS;
        return tuned::roipool_bwd_f32(gout, rois, gin, R, C, H, W, k, ws, as_stream(stream));
    }
    if (R > 0 && (!ws || ws_bytes < bins_bytes(R, k))) return D2T_EWS;
    if (impl != D2T_IMPL_GENERIC && roipool_bwd_lists_supported<float>(R, C, H, W, k) && ws_bytes >= roipool_bwd_lists_ws_bytes<float>(R, C, H, W, k))
        return roipool_bwd_lists<float>(gout, rois, gin, ws, R, C, H, W, k, as_stream(stream));
    return roipool_bwd_generic<float>(gout, rois, gin, static_cast<int32_t*>(ws), R, C, H, W, k, as_stream(stream));
}

int d2t_roipool_bwd_f64(const double* gout, const double* rois, double* gin, int R, int C, int H, int W, int k,
                        void* ws, size_t ws_bytes, int impl, d2t_stream_t stream)
{
    if (!impl_ok(impl)) return D2T_EINVAL;
    if (ws_misaligned(ws)) return D2T_EINVAL;
    int rc = check_pool(gin, rois, gout, R, C, H, W, k);
    if (rc != D2T_OK) return rc;
    if (R > 0 && (!ws || ws_bytes < bins_bytes(R, k))) return D2T_EWS;
    if (impl != D2T_IMPL_GENERIC && roipool_bwd_lists_supported<double>(R, C, H, W, k) && ws_bytes >= roipool_bwd_lists_ws_bytes<double>(R, C, H, W, k))
        return roipool_bwd_lists<double>(gout, rois, gin, ws, R, C, H, W, k, as_stream(stream));
    return roipool_bwd_generic<double>(gout, rois, gin, static_cast<int32_t*>(ws), R, C, H, W, k, as_stream(stream));
}

// ------------------------------------------------------------------ psroipool
size_t d2t_psroipool_fwd_workspace_bytes(int R, int nT, int H, int W, int k, int elem_size)
{
    return elem_size == 4 ? tuned::psroipool_fwd_ws_bytes(R, nT, H, W, k) : 0;
}
size_t d2t_psroipool_bwd_workspace_bytes(int R, int nT, int H, int W, int k, int elem_size)
{
    const size_t generic = bins_bytes(R, k);
    const size_t t = elem_size == 4 ? tuned::psroipool_bwd_ws_bytes(R, nT, H, W, k) : 0;
    size_t lists = 0;                                                 // outside the tuned envelope: d2t_pool_lists.hip
    if ((elem_size == 4 && !tuned::psroipool_bwd_supported(R, nT, H, W, k)) || elem_size == 8) lists = psroipool_bwd_lists_ws_bytes(R, nT, H, W, k);
    const size_t m = generic > t ? generic : t;
    return m > lists ? m : lists;
}

static int check_ps(const void* fm, const void* rois, const void* out, int R, int nT, int H, int W, int k)
{
    if (nT < 0 || k < 1) return D2T_EINVAL;
    if (!fits_i32(1LL * nT * k * k)) return D2T_ETOOBIG;
    int rc = check_pool(fm, rois, out, R, nT * k * k, H, W, 1);    // input side: nT*k*k channels
    if (rc != D2T_OK) return rc;
    if (!fits_i32(1LL * R * nT * k * k) || !fits_i32(4LL * R * k * k)) return D2T_ETOOBIG;
    return D2T_OK;
}

int d2t_psroipool_fwd_f32(const float* fm, const float* rois, float* out, int R, int nT, int H, int W, int k,
                          void* ws, size_t ws_bytes, int impl, d2t_stream_t stream)
{
    if (!impl_ok(impl)) return D2T_EINVAL;
    if (ws_misaligned(ws)) return D2T_EINVAL;
    int rc = check_ps(fm, rois, out, R, nT, H, W, k);
    if (rc != D2T_OK) return rc;
    if (impl != D2T_IMPL_GENERIC && R > 0 && tuned::psroipool_fwd_supported(R, nT, H, W, k)) {
        const size_t need = tuned::psroipool_fwd_ws_bytes(R, nT, H, W, k);
        if (need && (!ws || ws_bytes < need)) return D2T_EWS;
        return tuned::psroipool_fwd_f32(fm, rois, out, R, nT, H, W, k, ws, as_stream(stream));
    }
    return psroipool_fwd_generic<float>(fm, rois, out, R, nT, H, W, k, as_stream(stream));
}

int d2t_psroipool_fwd_f64(const double* fm, const double* rois, double* out, int R, int nT, int H, int W, int k,
                          void*, size_t, int impl, d2t_stream_t stream)
{
    if (!impl_ok(impl)) return D2T_EINVAL;
    int rc = check_ps(fm, rois, out, R, nT, H, W, k);
    if (rc != D2T_OK) return rc;
    return psroipool_fwd_generic<double>(fm, rois, out, R, nT, H, W, k, as_stream(stream));
}

int d2t_psroipool_bwd_f32(const float* gout, const float* rois, float* gin, int R, int nT, int H, int W, int k,
                          void* ws, size_t ws_bytes, int impl, d2t_stream_t stream)
{
    if (!impl_ok(impl)) return D2T_EINVAL;
    if (ws_misaligned(ws)) return D2T_EINVAL;
    int rc = check_ps(gin, rois, gout, R, nT, H, W, k);
    if (rc != D2T_OK) return rc;
    if (impl != D2T_IMPL_GENERIC && tuned::psroipool_bwd_supported(R, nT, H, W, k)) {
        if (!ws || ws_bytes < tuned::psroipool_bwd_ws_bytes(R, nT, H, W, k)) return D2T_EWS;
        return tuned::psroipool_bwd_f32(gout, rois, gin, R, nT, H, W, k, ws, as_stream(stream));
    }
    if (R > 0 && (!ws || ws_bytes < bins_bytes(R, k))) return D2T_EWS;
    if (impl != D2T_IMPL_GENERIC && psroipool_bwd_lists_supported(R, nT, H, W, k) && ws_bytes >= psroipool_bwd_lists_ws_bytes(R, nT, H, W, k))
        return psroipool_bwd_lists<float>(gout, rois, gin, ws, R, nT, H, W, k, as_stream(stream));
    return psroipool_bwd_generic<float>(gout, rois, gin, static_cast<int32_t*>(ws), R, nT, H, W, k, as_stream(stream));
}

int d2t_psroipool_bwd_f64(const double* gout, const double* rois, double* gin, int R, int nT, int H, int W, int k,
                          void* ws, size_t ws_bytes, int impl, d2t_stream_t stream)
{
    if (!impl_ok(impl)) return D2T_EINVAL;
    if (ws_misaligned(ws)) return D2T_EINVAL;
    int rc = check_ps(gin, rois, gout, R, nT, H, W, k);
    if (rc != D2T_OK) return rc;
    if (R > 0 && (!ws || ws_bytes < bins_bytes(R, k))) return D2T_EWS;
    if (impl != D2T_IMPL_GENERIC && psroipool_bwd_lists_supported(R, nT, H, W, k) && ws_bytes >= psroipool_bwd_lists_ws_bytes(R, nT, H, W, k))
        return psroipool_bwd_lists<double>(gout, rois, gin, ws, R, nT, H, W, k, as_stream(stream));
    return psroipool_bwd_generic<double>(gout, rois, gin, static_cast<int32_t*>(ws), R, nT, H, W, k, as_stream(stream));
}

// ------------------------------------------------------------------ introspection
#define D2T_BINS(NAME, T, FN)                                                                          \
    int NAME(const T* rois, int32_t* bounds, int R, int H, int W, int k, d2t_stream_t stream)           \
    {                                                                                                  \
        if (R < 0 || H < 0 || W < 0 || k < 1) return D2T_EINVAL;                                        \
        if (!fits_i32(4LL * R * k * k)) return D2T_ETOOBIG;                                             \
        if (R > 0 && (!rois || !bounds)) return D2T_EINVAL;                                             \
        return FN<T>(rois, bounds, R, H, W, k, as_stream(stream));                                      \
    }
D2T_BINS(d2t_roipool_bins_f32, float, roipool_bins)
D2T_BINS(d2t_roipool_bins_f64, double, roipool_bins)
D2T_BINS(d2t_psroipool_bins_f32, float, psroipool_bins)
D2T_BINS(d2t_psroipool_bins_f64, double, psroipool_bins)

int d2t_psroipool_channels(int32_t* channels, int nT, int k, d2t_stream_t stream)
{
    if (nT < 0 || k < 1) return D2T_EINVAL;
    if (!fits_i32(1LL * nT * k * k)) return D2T_ETOOBIG;
    if (nT > 0 && !channels) return D2T_EINVAL;
    return psroipool_channels(channels, nT, k, as_stream(stream));
}

int d2t_corr_mask(uint8_t* mask, int H, int W, int d, int stride, d2t_stream_t stream)
{
    if (H < 0 || W < 0 || d < 0 || stride < 1) return D2T_EINVAL;
    if (!fits_i32(1LL * H * W * (2LL * d + 1) * (2LL * d + 1))) return D2T_ETOOBIG;
    if (1LL * H * W > 0 && !mask) return D2T_EINVAL;
    return corr_mask(mask, H, W, d, stride, as_stream(stream));
}

// ------------------------------------------------------------------ region proposals
size_t d2t_region_filter_workspace_bytes(int A, int max_dets) { return region_filter_ws_bytes(A, max_dets); }

int d2t_region_filter_f32(const float* anchors, const float* offsets, const float* confs, int A,
                          float conf_thresh, int max_dets, float iou_thresh,
                          float* out_boxes, float* out_conf, int32_t* out_idx, int32_t* out_count,
                          void* ws, size_t ws_bytes, d2t_stream_t stream)
{
    return d2t_region_filter_batched_f32(anchors, offsets, confs, 1, A, conf_thresh, max_dets, iou_thresh,
                                         out_boxes, out_conf, out_idx, out_count, ws, ws_bytes, stream);
}

int d2t_region_filter_batched_f32(const float* anchors, const float* offsets, const float* confs, int N, int A,
                                  float conf_thresh, int max_dets, float iou_thresh,
                                  float* out_boxes, float* out_conf, int32_t* out_idx, int32_t* out_count,
                                  void* ws, size_t ws_bytes, d2t_stream_t stream)
{
    if (N < 0 || N > 65535) return D2T_EINVAL;
    if (N == 0) return D2T_OK;
    if (A < 0 || max_dets < 1 || max_dets > region_max_dets() || !out_boxes || !out_conf || !out_idx || !out_count) return D2T_EINVAL;
    if (A > 0 && (!anchors || !offsets || !confs)) return D2T_EINVAL;
    // boxes are moved as float4 / u32x4: a pointer that is only 4-byte aligned would fault on the device
    if (((reinterpret_cast<uintptr_t>(anchors) | reinterpret_cast<uintptr_t>(offsets) | reinterpret_cast<uintptr_t>(out_boxes) |
          reinterpret_cast<uintptr_t>(ws)) & 15) != 0) return D2T_EINVAL;
    if (!fits_i32(4LL * A) || !fits_i32(4LL * A * N) || !fits_i32(4LL * max_dets * N)) return D2T_ETOOBIG;
    if (A == 0) {                                                    // nothing to filter: all-padding lists
        hipError_t e = hipMemsetAsync(out_boxes, 0, (size_t)N * max_dets * 16, as_stream(stream));
        if (e == hipSuccess) e = hipMemsetAsync(out_conf, 0, (size_t)N * max_dets * 4, as_stream(stream));
        if (e == hipSuccess) e = hipMemsetAsync(out_idx, 0xff, (size_t)N * max_dets * 4, as_stream(stream));
        if (e == hipSuccess) e = hipMemsetAsync(out_count, 0, (size_t)N * 4, as_stream(stream));
        return e == hipSuccess ? D2T_OK : static_cast<int>(e);
    }
    if (!ws || ws_bytes < (size_t)N * region_filter_ws_bytes(A, max_dets)) return D2T_EWS;
    return region_filter_f32(anchors, offsets, confs, A, conf_thresh, max_dets, iou_thresh, out_boxes, out_conf, out_idx, out_count,
                             ws, as_stream(stream), N);
}

}  // extern "C"
