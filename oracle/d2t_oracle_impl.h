/*
 * d2t_oracle_impl.h -- type-generic body of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * Included twice by d2t_oracle.c, once with T=float and once with T=double.
 * Every function is a serial restatement, in this repo's own words, of what one of the
 * reference's CUDA kernels computes; the reference lines each one follows are cited.
 * Paths are relative to /root/reference/detect_to_track/models/.
 *
 * Floating-point contraction: the reference is built by nvcc, whose default (-fmad=true)
 * fuses `acc += a * b` into one FMA.  This file is compiled with -ffp-contract=off and
 * spells every such fusion explicitly with FMA(), so the arithmetic does not depend on
 * the host compiler's mood.  Places that are NOT a*b+c patterns stay unfused.
 *
 * Required macros: T (scalar type), SUF (name suffix), FMA(a,b,c), FLOOR(x), CEIL(x).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(CAT(d2t_oracle_, name), SUF)

/* clamp to [0,1] -- common/cuda_common.cuh:9-13: max(0, min(1, x)).
 * CUDA's max/min on floats are fmax/fmin: a NaN operand yields the other operand. */
static inline T CAT(clamp01, SUF)(T x) {
    if (x != x) return (T)1;                    /* fmin(1, NaN) = 1, then fmax(0, 1) = 1 */
    const T m = x < (T)1 ? x : (T)1;
    return m > (T)0 ? m : (T)0;
}

/* ------------------------------------------------------------------------------------
 * PointwiseCorrelation forward.
 * pointwise_correlation/pointwise_correlation_cuda.cu:84-107 (index maps :12-53).
 * out (B,H,W,2d+1,2d+1) is defined everywhere: cells the reference never visits are
 * the zeros its launcher pre-fills (:192).
 * ---------------------------------------------------------------------------------- */
void FN(corr_fwd)(const T* fm0, const T* fm1, T* out,
                  int B, int C, int H, int W, int d, int s)
{
    const int cw = 2 * d + 1;
    const long plane = (long)H * W;
    const long out_n = (long)B * H * W * cw * cw;
    for (long k = 0; k < out_n; ++k) out[k] = (T)0;

    #pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < H; ++i) {
            for (int j = 0; j < W; ++j) {
                const T* a = fm0 + (long)b * C * plane + (long)i * W + j;
                const int lo_i = i - d > 0 ? i - d : 0, hi_i = i + d < H ? i + d : H;
                const int lo_j = j - d > 0 ? j - d : 0, hi_j = j + d < W ? j + d : W;
                for (int di = lo_i; di < hi_i; di += s) {
                    for (int dj = lo_j; dj < hi_j; dj += s) {
                        const T* q = fm1 + (long)b * C * plane + (long)di * W + dj;
                        T acc = (T)0;
                        for (int c = 0; c < C; ++c)          /* :105-107, fused by nvcc */
                            acc = FMA(a[c * plane], q[c * plane], acc);
                        out[((((long)b * H + i) * W + j) * cw + (di - i + d)) * cw + (dj - j + d)] = acc;
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------
 * PointwiseCorrelation backward.  pointwise_correlation_cuda.cu:145-171.
 * gradFM0 is thread-owned in the reference (plain +=, :168) so its order is defined:
 * ascending (di,dj) per pixel.  gradFM1 is accumulated with atomicAdd (:169), order
 * undefined on the GPU; this restatement uses the order of a serial sweep of the
 * reference grid (ascending b,i,j then di,dj), which is one valid order.
 * ---------------------------------------------------------------------------------- */
void FN(corr_bwd)(const T* gout, const T* fm0, const T* fm1, T* g0, T* g1,
                  int B, int C, int H, int W, int d, int s)
{
    const int cw = 2 * d + 1;
    const long plane = (long)H * W;
    const long in_n = (long)B * C * plane;
    for (long k = 0; k < in_n; ++k) { g0[k] = (T)0; g1[k] = (T)0; }

    /* channel planes are independent, so (b,c) may run in parallel without changing any
     * element's accumulation order (ascending i,j then di,dj). */
    #pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int c = 0; c < C; ++c) {
            const long base = ((long)b * C + c) * plane;
            const T* gb = gout + (long)b * plane * cw * cw;
            for (int i = 0; i < H; ++i) {
                for (int j = 0; j < W; ++j) {
                    const long ctr = base + (long)i * W + j;
                    const int lo_i = i - d > 0 ? i - d : 0, hi_i = i + d < H ? i + d : H;
                    const int lo_j = j - d > 0 ? j - d : 0, hi_j = j + d < W ? j + d : W;
                    for (int di = lo_i; di < hi_i; di += s) {
                        for (int dj = lo_j; dj < hi_j; dj += s) {
                            const long dsp = base + (long)di * W + dj;
                            const T g = gb[(((long)i * W + j) * cw + (di - i + d)) * cw + (dj - j + d)];
                            g0[ctr] = FMA(g, fm1[dsp], g0[ctr]);      /* :168 */
                            g1[dsp] = FMA(g, fm0[ctr], g1[dsp]);      /* :169 */
                        }
                    }
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------------------
 * Bin geometry shared by ROIPool forward/backward.  roipool/roipool_cuda.cu:32-51.
 * The literal 0.5 in `(static_cast<scalar_t>(i) + 0.5) * bH` is a double, so for
 * T=float the centre is evaluated in double (clamped corner + product, fused by nvcc
 * into one double FMA) and rounded once to T.
 * bounds = {BI0, BI1, BJ0, BJ1}.
 * ---------------------------------------------------------------------------------- */
static inline void CAT(roi_bin, SUF)(const T* roi, int i, int j, int H, int W, int k, int* bounds)
{
    const T rI = roi[0], rJ = roi[1], rH = roi[2], rW = roi[3];
    const T bH = rH / (T)k, bW = rW / (T)k;
    const T cornerI = CAT(clamp01, SUF)(rI - rH / (T)2);
    const T cornerJ = CAT(clamp01, SUF)(rJ - rW / (T)2);
    const T bI = (T)fma((double)(T)i + 0.5, (double)bH, (double)cornerI);
    const T bJ = (T)fma((double)(T)j + 0.5, (double)bW, (double)cornerJ);
    bounds[0] = (int)FLOOR(CAT(clamp01, SUF)(bI - bH / (T)2) * (T)H);
    bounds[1] = (int)CEIL (CAT(clamp01, SUF)(bI + bH / (T)2) * (T)H);
    bounds[2] = (int)FLOOR(CAT(clamp01, SUF)(bJ - bW / (T)2) * (T)W);
    bounds[3] = (int)CEIL (CAT(clamp01, SUF)(bJ + bW / (T)2) * (T)W);
}

/* Cell geometry of PSROIPool.  ps_roipool/ps_roipool_cuda.cu:36-54: the RoI corner is
 * NOT clamped here (contrast roipool_cuda.cu:41-42). */
static inline void CAT(psroi_cell, SUF)(const T* roi, int i, int j, int H, int W, int k, int* bounds)
{
    const T rI = roi[0], rJ = roi[1], rH = roi[2], rW = roi[3];
    const T cH = rH / (T)k, cW = rW / (T)k;
    const T cornerI = rI - rH / (T)2;
    const T cornerJ = rJ - rW / (T)2;
    const T cI = (T)fma((double)(T)i + 0.5, (double)cH, (double)cornerI);
    const T cJ = (T)fma((double)(T)j + 0.5, (double)cW, (double)cornerJ);
    bounds[0] = (int)FLOOR(CAT(clamp01, SUF)(cI - cH / (T)2) * (T)H);
    bounds[1] = (int)CEIL (CAT(clamp01, SUF)(cI + cH / (T)2) * (T)H);
    bounds[2] = (int)FLOOR(CAT(clamp01, SUF)(cJ - cW / (T)2) * (T)W);
    bounds[3] = (int)CEIL (CAT(clamp01, SUF)(cJ + cW / (T)2) * (T)W);
}

/* integer bin bounds for every (r,i,j): out_bounds (R,k,k,4) int32 */
void FN(roipool_bins)(const T* rois, int* out_bounds, int R, int H, int W, int k)
{
    for (int r = 0; r < R; ++r)
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j)
                CAT(roi_bin, SUF)(rois + 4 * r, i, j, H, W, k, out_bounds + (((long)r * k + i) * k + j) * 4);
}

void FN(psroipool_bins)(const T* rois, int* out_bounds, int R, int H, int W, int k)
{
    for (int r = 0; r < R; ++r)
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j)
                CAT(psroi_cell, SUF)(rois + 4 * r, i, j, H, W, k, out_bounds + (((long)r * k + i) * k + j) * 4);
}

/* ------------------------------------------------------------------------------------
 * ROIPool forward.  roipool_cuda.cu:26-61.  Average over the bin, running sum in T in
 * row-major pixel order, then `sum / n` with NO zero guard: an empty bin gives 0/0=NaN.
 * ---------------------------------------------------------------------------------- */
void FN(roipool_fwd)(const T* fm, const T* rois, T* out, int R, int C, int H, int W, int k)
{
    #pragma omp parallel for schedule(static)
    for (int r = 0; r < R; ++r) {
        for (int i = 0; i < k; ++i) {
            for (int j = 0; j < k; ++j) {
                int bb[4];
                CAT(roi_bin, SUF)(rois + 4 * r, i, j, H, W, k, bb);
                const int n = (bb[1] - bb[0]) * (bb[3] - bb[2]);
                for (int c = 0; c < C; ++c) {
                    T acc = (T)0;
                    for (int pI = bb[0]; pI < bb[1]; ++pI)
                        for (int pJ = bb[2]; pJ < bb[3]; ++pJ)
                            acc += fm[((long)c * H + pI) * W + pJ];
                    out[(((long)r * C + c) * k + i) * k + j] = acc / (T)n;
                }
            }
        }
    }
}

/* ROIPool backward.  roipool_cuda.cu:88-125: every pixel of the bin receives g / n
 * (atomicAdd, order undefined; here ascending (r,i,j) per channel). */
void FN(roipool_bwd)(const T* gout, const T* rois, T* gin, int R, int C, int H, int W, int k)
{
    for (long q = 0; q < (long)C * H * W; ++q) gin[q] = (T)0;
    #pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        for (int r = 0; r < R; ++r)
            for (int i = 0; i < k; ++i)
                for (int j = 0; j < k; ++j) {
                    int bb[4];
                    CAT(roi_bin, SUF)(rois + 4 * r, i, j, H, W, k, bb);
                    const int n = (bb[1] - bb[0]) * (bb[3] - bb[2]);
                    const T g = gout[(((long)r * C + c) * k + i) * k + j];
                    for (int pI = bb[0]; pI < bb[1]; ++pI)
                        for (int pJ = bb[2]; pJ < bb[3]; ++pJ)
                            gin[((long)c * H + pI) * W + pJ] += g / (T)n;
                }
    }
}

/* ------------------------------------------------------------------------------------
 * PSROIPool forward.  ps_roipool_cuda.cu:30-69.  Channel map (t+1)*(i*k+j) (:58);
 * divide only when the cell is non-empty (:67-69), so an empty cell gives 0.
 * ---------------------------------------------------------------------------------- */
void FN(psroipool_fwd)(const T* fm, const T* rois, T* out, int R, int nT, int H, int W, int k)
{
    #pragma omp parallel for schedule(static)
    for (int r = 0; r < R; ++r)
        for (int t = 0; t < nT; ++t)
            for (int i = 0; i < k; ++i)
                for (int j = 0; j < k; ++j) {
                    int bb[4];
                    CAT(psroi_cell, SUF)(rois + 4 * r, i, j, H, W, k, bb);
                    const int ch = (t + 1) * (i * k + j);
                    const T* plane = fm + (long)ch * H * W;
                    T acc = (T)0;
                    for (int pI = bb[0]; pI < bb[1]; ++pI)
                        for (int pJ = bb[2]; pJ < bb[3]; ++pJ)
                            acc += plane[(long)pI * W + pJ];
                    const int n = (bb[1] - bb[0]) * (bb[3] - bb[2]);
                    if (n > 0) acc /= (T)n;
                    out[(((long)r * nT + t) * k + i) * k + j] = acc;
                }
}

/* PSROIPool backward.  ps_roipool_cuda.cu:97-139 (atomicAdd; here ascending (r,t,i,j)). */
void FN(psroipool_bwd)(const T* gout, const T* rois, T* gin, int R, int nT, int H, int W, int k)
{
    for (long q = 0; q < (long)nT * k * k * H * W; ++q) gin[q] = (T)0;
    for (int r = 0; r < R; ++r)
        for (int t = 0; t < nT; ++t)
            for (int i = 0; i < k; ++i)
                for (int j = 0; j < k; ++j) {
                    int bb[4];
                    CAT(psroi_cell, SUF)(rois + 4 * r, i, j, H, W, k, bb);
                    const int n = (bb[1] - bb[0]) * (bb[3] - bb[2]);
                    const int ch = (t + 1) * (i * k + j);
                    T* plane = gin + (long)ch * H * W;
                    T g = gout[(((long)r * nT + t) * k + i) * k + j];
                    if (n > 0) g /= (T)n;
                    for (int pI = bb[0]; pI < bb[1]; ++pI)
                        for (int pJ = bb[2]; pJ < bb[3]; ++pJ)
                            plane[(long)pI * W + pJ] += g;
                }
}

/* ------------------------------------------------------------------------------------
 * Wide-accumulator yardsticks for the gradients the reference sums with atomicAdd.
 * The reference's summation order is undefined there (pointwise_correlation_cuda.cu:169,
 * roipool_cuda.cu:123, ps_roipool_cuda.cu:137), and at thousands of terms per element the
 * T-precision sum of ANY order carries rounding noise above the 1e-5 contract.  These variants
 * keep every TERM exactly as the reference forms it (geometry and `g / n` in T, products as one
 * T-rounded... see each function) but add the terms in double and round once, so a kernel can be
 * held to 1e-5 of the element's magnitude scale: `mag` (optional, same shape as the gradient)
 * receives sum |term| in double.
 * ---------------------------------------------------------------------------------- */
void FN(roipool_bwd_acc64)(const T* gout, const T* rois, T* gin, double* mag, int R, int C, int H, int W, int k)
{
    #pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        double* acc = (double*)calloc((size_t)2 * H * W, sizeof(double));
        double* ab = acc + (size_t)H * W;
        for (int r = 0; r < R; ++r)
            for (int i = 0; i < k; ++i)
                for (int j = 0; j < k; ++j) {
                    int bb[4];
                    CAT(roi_bin, SUF)(rois + 4 * r, i, j, H, W, k, bb);
                    const int n = (bb[1] - bb[0]) * (bb[3] - bb[2]);
                    const T term = gout[(((long)r * C + c) * k + i) * k + j] / (T)n;   /* roipool_cuda.cu:123, in T */
                    for (int pI = bb[0]; pI < bb[1]; ++pI)
                        for (int pJ = bb[2]; pJ < bb[3]; ++pJ) {
                            acc[(long)pI * W + pJ] += (double)term;
                            ab[(long)pI * W + pJ] += fabs((double)term);
                        }
                }
        for (long q = 0; q < (long)H * W; ++q) {
            gin[(long)c * H * W + q] = (T)acc[q];
            if (mag) mag[(long)c * H * W + q] = ab[q];
        }
        free(acc);
    }
}

void FN(psroipool_bwd_acc64)(const T* gout, const T* rois, T* gin, double* mag, int R, int nT, int H, int W, int k)
{
    const long n_in = (long)nT * k * k * H * W;
    double* acc = (double*)calloc((size_t)2 * n_in, sizeof(double));
    double* ab = acc + n_in;
    for (int r = 0; r < R; ++r)
        for (int t = 0; t < nT; ++t)
            for (int i = 0; i < k; ++i)
                for (int j = 0; j < k; ++j) {
                    int bb[4];
                    CAT(psroi_cell, SUF)(rois + 4 * r, i, j, H, W, k, bb);
                    const int n = (bb[1] - bb[0]) * (bb[3] - bb[2]);
                    const long ch = (long)(t + 1) * (i * k + j);
                    T g = gout[(((long)r * nT + t) * k + i) * k + j];
                    if (n > 0) g /= (T)n;                                               /* ps_roipool_cuda.cu:131, in T */
                    for (int pI = bb[0]; pI < bb[1]; ++pI)
                        for (int pJ = bb[2]; pJ < bb[3]; ++pJ) {
                            acc[(ch * H + pI) * W + pJ] += (double)g;
                            ab[(ch * H + pI) * W + pJ] += fabs((double)g);
                        }
                }
    for (long q = 0; q < n_in; ++q) { gin[q] = (T)acc[q]; if (mag) mag[q] = ab[q]; }
    free(acc);
}

/* correlation: every term g * fm is formed exactly (the product of two T values is exact in double),
 * summed in double, rounded once.  mag0 / mag1 (optional): sum |g * fm| per element. */
void FN(corr_bwd_acc64)(const T* gout, const T* fm0, const T* fm1, T* g0, T* g1, double* mag0, double* mag1,
                        int B, int C, int H, int W, int d, int s)
{
    const int cw = 2 * d + 1;
    const long plane = (long)H * W;
    #pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int c = 0; c < C; ++c) {
            const long base = ((long)b * C + c) * plane;
            const T* gb = gout + (long)b * plane * cw * cw;
            double* acc = (double*)calloc((size_t)4 * plane, sizeof(double));
            double *a0 = acc, *a1 = acc + plane, *m0 = acc + 2 * plane, *m1 = acc + 3 * plane;
            for (int i = 0; i < H; ++i)
                for (int j = 0; j < W; ++j) {
                    const long ctr = (long)i * W + j;
                    const int lo_i = i - d > 0 ? i - d : 0, hi_i = i + d < H ? i + d : H;
                    const int lo_j = j - d > 0 ? j - d : 0, hi_j = j + d < W ? j + d : W;
                    for (int di = lo_i; di < hi_i; di += s)
                        for (int dj = lo_j; dj < hi_j; dj += s) {
                            const long dsp = (long)di * W + dj;
                            const double g = (double)gb[(((long)i * W + j) * cw + (di - i + d)) * cw + (dj - j + d)];
                            const double t0 = g * (double)fm1[base + dsp], t1 = g * (double)fm0[base + ctr];
                            a0[ctr] += t0; m0[ctr] += fabs(t0);
                            a1[dsp] += t1; m1[dsp] += fabs(t1);
                        }
                }
            for (long q = 0; q < plane; ++q) {
                g0[base + q] = (T)a0[q]; g1[base + q] = (T)a1[q];
                if (mag0) mag0[base + q] = m0[q];
                if (mag1) mag1[base + q] = m1[q];
            }
            free(acc);
        }
    }
}

#undef FN
#undef CAT
#undef CAT_
