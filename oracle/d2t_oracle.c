/*
 * d2t_oracle.c -- CPU oracle for the detect-to-track custom-op hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the shipped ops
 * (detect-to-track_amd/) never link, import or call it and have no CPU fallback.
 *
 * It restates, serially and in plain C, the arithmetic of the reference's six CUDA
 * kernels (reference paths relative to /root/reference/detect_to_track/models/):
 *   pointwise_correlation/pointwise_correlation_cuda.cu:62-174
 *   roipool/roipool_cuda.cu:5-127
 *   ps_roipool/ps_roipool_cuda.cu:9-141
 *   common/cuda_common.cuh:9-13 (clamp)
 *
 * Pinning (see DESIGN.md "Oracle"): the reference ships no golden vectors; its tests are
 * gradcheck self-consistency plus one known-answer case (tests/test_ps_roipool.py:33-44).
 * The oracle is pinned (a) by that known-answer case, (b) by the structural facts recorded
 * in SURVEY.md section 8c from an execution of the reference kernel bodies, and (c) by
 * fixtures under tests/golden/ produced on an MI355X by the reference's own kernels
 * compiled unmodified with hipcc (oracle/ref_build/, output in oracle/_ref/).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdlib.h>

/* ---- float instantiation ---- */
#define T float
#define SUF _f32
#define FMA(a, b, c) fmaf((a), (b), (c))
#define FLOOR(x) floorf(x)
#define CEIL(x) ceilf(x)
#include "d2t_oracle_impl.h"
#undef T
#undef SUF
#undef FMA
#undef FLOOR
#undef CEIL

/* ---- double instantiation ---- */
#define T double
#define SUF _f64
#define FMA(a, b, c) fma((a), (b), (c))
#define FLOOR(x) floor(x)
#define CEIL(x) ceil(x)
#include "d2t_oracle_impl.h"
#undef T
#undef SUF
#undef FMA
#undef FLOOR
#undef CEIL

/* Written-cell mask of the correlation output: 1 where the reference's displacement loops
 * (pointwise_correlation_cuda.cu:92-93) visit the cell, else 0.  Shape (H,W,2d+1,2d+1).
 * Note the exclusive upper bound min(i+d, H): displacement +d is never visited. */
void d2t_oracle_corr_mask(unsigned char* mask, int H, int W, int d, int s)
{
    const int cw = 2 * d + 1;
    for (long k = 0; k < (long)H * W * cw * cw; ++k) mask[k] = 0;
    for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
            const int lo_i = i - d > 0 ? i - d : 0, hi_i = i + d < H ? i + d : H;
            const int lo_j = j - d > 0 ? j - d : 0, hi_j = j + d < W ? j + d : W;
            for (int di = lo_i; di < hi_i; di += s)
                for (int dj = lo_j; dj < hi_j; dj += s)
                    mask[(((long)i * W + j) * cw + (di - i + d)) * cw + (dj - j + d)] = 1;
        }
}

/* PSROIPool channel map, ps_roipool_cuda.cu:58: ch[t,i,j] = (t+1)*(i*k+j).  (nT,k,k) int32 */
void d2t_oracle_psroipool_channels(int* ch, int nT, int k)
{
    for (int t = 0; t < nT; ++t)
        for (int i = 0; i < k; ++i)
            for (int j = 0; j < k; ++j)
                ch[(t * k + i) * k + j] = (t + 1) * (i * k + j);
}

int d2t_oracle_version(void) { return 1; }
