// d2t_roipool_fwd_direct.hip -- gfx950 f32 ROIPool FORWARD in the reference's own order (k = 7; round 6).
//
// out[r][c][i][j] = (sum of FM[c] over the bin's pixels, row-major, one running f32 sum) / n   -- roipool_cuda.cu:52-61.
//
// The summed-area kernel of rounds 2-5 (d2t_pool_tuned.hip, k_roipool_fwd_sat2) builds f64 tables of two channels per workgroup and
// answers every bin with four look-ups: within 1e-5 of the reference, not its bits, 31 us at BASELINE config 3 of which 16 k of a
// workgroup's 38 k cycles are table building and 9 k geometry that every one of 512 workgroups repeats.  A bin of the model's regions
// is ~11 pixels (2.8 rows x 3.9 columns), so walking it costs no more LDS bytes than four f64 look-ups -- and then no table, no f64, no
// prefix pass are needed and the result is the reference's, bit for bit:
//   * workgroup = (CG = 8 consecutive channels, a share of the RoIs); the 8 planes sit in LDS INTERLEAVED, px[y][x][8]: one pixel of all
//     channels is two ds_read_b128, the index arithmetic of a bin walk is shared by 8 outputs, geometry (7 + 7 bounds per RoI, once per
//     workgroup) by 8 channels instead of 2;
//   * thread = (RoI, bin): walks its bin row-major, 8 running sums (v_pk_add_f32: two IEEE adds per instruction), divides each by
//     float(n) -- correctly rounded like the IEEE division of the generic kernel, through one shared reciprocal (see below; n = 0:
//     0 / 0 = NaN as the reference) --, stores 8 floats;
//     consecutive lanes are consecutive bins of a RoI: their walks have the same length +-1 row / column, their stores are one
//     196-byte run per channel;
//   * non-finite feature values need no special path: a bin only ever adds its own pixels.
// Results are bit-identical to roipool_fwd_generic (tests/test_roipool.py) -- the tuned ROIPool forward is now exact like every other
// forward of the library.
#include "d2t_kernels.hpp"
#include "d2t_tuned.hpp"
#include "d2t_pool_common.hpp"

namespace d2t { namespace tuned {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access, dword aligned

D2T_KSTAMP_DEFINE(d2t_lab_roipool_direct_stamps)

constexpr int RD_CG = 8;                      // channels per workgroup
constexpr int RD_THREADS = 1024;
constexpr int RD_GEO = 32;                    // bytes of geometry per RoI: (i0 | i1 << 8)[7], (j0 | j1 << 8)[7], pad

__global__ void __launch_bounds__(RD_THREADS)
k_roipool_fwd_direct(const float* __restrict__ fm, const float* __restrict__ rois, float* __restrict__ out,
                     int R, int C, int H, int W, int rois_per_wg)
{
    extern __shared__ __attribute__((aligned(16))) float rd_lds[];
    const int tid = threadIdx.x, HW = H * W;
    float* px = rd_lds;                                              // [HW][8]
    unsigned short* geo = reinterpret_cast<unsigned short*>(px + (size_t)HW * RD_CG);   // [rois_per_wg][16]
    const int c0 = blockIdx.x * RD_CG;
    const int r_lo = blockIdx.y * rois_per_wg, r_hi = r_lo + rois_per_wg < R ? r_lo + rois_per_wg : R, nr = r_hi - r_lo;
    D2T_KSTAMP(0); D2T_KSTAMP_RT(14);

    // ---- planes -> LDS.  Lane = (channel, group of 4 pixels), channel fastest: the 8 lanes of a pixel group write 8 consecutive words
    // four times (conflict-free), a wave reads 8 runs of 128 bytes (one per plane).  Plane bases are only dword aligned (H*W odd): f32x4u.
    {
        const int nq = (HW + 3) >> 2, total = nq * RD_CG;
        for (int e0 = 0; e0 < total; e0 += 4 * RD_THREADS) {
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + k * RD_THREADS + tid, c = e & (RD_CG - 1), p = (e >> 3) << 2;
                v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (e < total && c0 + c < C) {
                    const float* src = fm + (size_t)(c0 + c) * HW + p;
                    if (p + 4 <= HW) v[k] = *reinterpret_cast<const f32x4u*>(src);
                    else
                        for (int q = 0; q < 4; ++q) v[k][q] = p + q < HW ? src[q] : 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = e0 + k * RD_THREADS + tid, c = e & (RD_CG - 1), p = (e >> 3) << 2;
                if (e < total)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (p + q < HW) px[(size_t)(p + q) * RD_CG + c] = v[k][q];
            }
        }
    }
    // ---- geometry: bin row q and bin column q of RoI r_lo + e / 7 (roipool_cuda.cu:41-50 through roi_bin, as every other kernel)
    for (int e = tid; e < nr * KT; e += RD_THREADS) {
        const int rr = e / KT, q = e - rr * KT;
        const Bounds b = roi_bin<float>(rois + 4 * (size_t)(r_lo + rr), q, q, H, W, KT);
        geo[rr * 16 + q] = (unsigned short)(b.i0 | (b.i1 << 8));
        geo[rr * 16 + KT + q] = (unsigned short)(b.j0 | (b.j1 << 8));
    }
    __syncthreads();
    D2T_KSTAMP(1);

    // ---- thread = (RoI, bin)
    const int items = nr * KK;
    for (int it = tid; it < items; it += RD_THREADS) {
        const int rr = it / KK, bin = it - rr * KK, i = bin / KT, j = bin - i * KT;
        const unsigned pi = geo[rr * 16 + i], pj = geo[rr * 16 + KT + j];
        const int i0 = pi & 255, i1 = pi >> 8, j0 = pj & 255, j1 = pj >> 8;
        const int h = i1 - i0, w = j1 - j0;
        f32x2 acc[RD_CG / 2];
#pragma unroll
        for (int c = 0; c < RD_CG / 2; ++c) acc[c] = f32x2{0.f, 0.f};
        // row-major: pI outer, pJ inner (roipool_cuda.cu:54-59), flattened to ONE loop of h x w steps (none for an extent <= 0, as there): the
        // kernel is bound by LDS bandwidth -- a ds_read_b128 costs its 8 cycles whatever the exec mask -- so what counts is how many steps
        // the longest lane of a wave takes.  Flattened: max(h w) over the wave's bins (17 on the bench's regions for a mean of 10.5); as
        // two nested loops: max(h) max(w) -- measured 41.7 k against 38.2 k cycles per workgroup (profiles/r06_roipool_direct_stamps.txt).
        // The pixel of step s+1 is fetched before the sums of step s are updated.
        const int cnt = h > 0 && w > 0 ? h * w : 0;
        const float* p = cnt ? px + (size_t)(i0 * W + j0) * RD_CG : px;   // (an empty bin may start behind the map: read pixel 0, add nothing)
        const int skip = (W - w) * RD_CG;
        int xx = 0;
        f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
        for (int s = 0; s < cnt; ++s) {
            ++xx;
            p += RD_CG;
            if (xx == w) { xx = 0; p += skip; }
            const float* pn = s + 1 < cnt ? p : px;
            f32x4 an = a, bn = b;
            if (!(D2T_KDBG & 2)) { an = *reinterpret_cast<const f32x4*>(pn); bn = *reinterpret_cast<const f32x4*>(pn + 4); }   // (scan builds: ablation bit 2 = no LDS reads in the walk)
            acc[0] += f32x2{a[0], a[1]}; acc[1] += f32x2{a[2], a[3]};
            acc[2] += f32x2{b[0], b[1]}; acc[3] += f32x2{b[2], b[3]};
            a = an; b = bn;
        }
        // sum / float(binNumel), :60-61 (a product of two negative extents is positive, as there; n == 0: 0 / 0 = NaN).  IEEE division is ~11
        // instructions; the 8 sums of a bin share the divisor, so: r = 1 / n once (IEEE), then q0 = a r, q = fma(fma(-n, q0, a), r, q0) --
        // the correctly rounded quotient whenever q0 is zero or 1e-30 < |q0| < 1e30 (lab/csrc/div_lab.hip: 0 mismatches against the
        // compiler's division in 1.4e11 pairs, every n in +-[1, 65025]; a = -0 is the one exception and cannot be a running sum that
        // started at +0).  Anything else -- denormal or huge quotients, Inf, NaN, n == 0 -- takes the plain division (cold).
        const float nf = static_cast<float>(h * w);
        const float rn = 1.0f / nf;
        float res[RD_CG];
        // safe for all 8 sums?  |a| (as an integer: NaN and Inf are the largest patterns) below 1e30, and every NON-zero |a| at least 1e-25
        // (then |q0| >= 1e-25 / 65025 > 1e-30): one max and one min over the patterns, zero wrapping to the largest value in the min
        unsigned big = 0u, small = 0xffffffffu;
#pragma unroll
        for (int c = 0; c < RD_CG; ++c) {
            const float a = acc[c >> 1][c & 1];
            const float q0 = a * rn;
            res[c] = __builtin_fmaf(__builtin_fmaf(-nf, q0, a), rn, q0);
            const unsigned t = __builtin_bit_cast(unsigned, a) & 0x7fffffffu;
            big = t > big ? t : big;
            small = t - 1u < small ? t - 1u : small;
        }
        const bool safe = big < 0x7149f2cau /* 1e30f */ && small >= 0x15f79688u - 1u /* 1e-25f */ && nf >= 1.0f;   // (n <= 0: the plain division)
        if (__builtin_expect(!safe, 0)) {
#pragma unroll
            for (int c = 0; c < RD_CG; ++c) res[c] = acc[c >> 1][c & 1] / nf;
        }
        float* dst = out + ((size_t)(r_lo + rr) * C + c0) * KK + bin;
        if (D2T_KDBG & 1) {                                          // (scan builds: ablation bit 1 = no stores)
            float sink = 0.f;
#pragma unroll
            for (int c = 0; c < RD_CG; ++c) sink += res[c];
            if (sink == 123.456f) dst[0] = sink;
        } else if (c0 + RD_CG <= C) {                                // uniform: a whole channel group
#pragma unroll
            for (int c = 0; c < RD_CG; ++c) dst[(size_t)c * KK] = res[c];
        } else {
#pragma unroll
            for (int c = 0; c < RD_CG; ++c)
                if (c0 + c < C) dst[(size_t)c * KK] = res[c];
        }
    }
    D2T_KSTAMP(2); D2T_KSTAMP_RT(15);
}

size_t direct_lds(int H, int W, int per) { return (size_t)H * W * RD_CG * 4 + (size_t)per * RD_GEO; }

// (16 channels per lane -- half the walk's bookkeeping per output, 16 planes = 153 KB at 38 x 63 -- measured 40.0 against 31.9 us at config 3:
// profiles/r06_ab_roipool_direct_cg16_lost.txt.)
// RoI shares per channel group: about one workgroup per CU (one round) -- or two where two workgroups fit a CU's LDS together (38 x 63:
// 79 KB each; config 3 30.6 against 31.4 us, while at 38 x 75, 91 KB, a second round of workgroups costs 64 against 58 us:
// profiles/r06_roipool_direct_ab.txt) --, never more shares than 32-RoI pieces
void direct_plan(int R, int C, int H, int W, int& gx, int& split, int& per)
{
    gx = (C + RD_CG - 1) / RD_CG;
    const int max_split = (R + 31) / 32;
    auto plan = [&](int wgs) {
        split = (wgs + gx / 2) / gx;
        split = split < 1 ? 1 : (split > max_split ? max_split : split);
        split = split > 65535 ? 65535 : split;
        per = (R + split - 1) / split;
        split = (R + per - 1) / per;
    };
    plan(512);
    if (2 * direct_lds(H, W, per) > (size_t)LDS_MAX) plan(256);
}

}  // namespace

bool roipool_fwd_direct_supported(int R, int C, int H, int W, int k)
{
    if (!(k == KT && R >= 1 && C >= 1 && H >= 1 && W >= 1 && H <= 255 && W <= 255)) return false;
    int gx, split, per;
    direct_plan(R, C, H, W, gx, split, per);
    return direct_lds(H, W, per) <= (size_t)LDS_MAX && 1LL * R * C * KK < 0x7fffffffLL;
}

int roipool_fwd_direct_f32(const float* fm, const float* rois, float* out, int R, int C, int H, int W, hipStream_t st)
{
    int gx, split, per;
    direct_plan(R, C, H, W, gx, split, per);
    D2T_ENSURE_DYNAMIC_LDS(k_roipool_fwd_direct, LDS_MAX);
    hipLaunchKernelGGL(k_roipool_fwd_direct, dim3(gx, split), dim3(RD_THREADS), direct_lds(H, W, per), st, fm, rois, out, R, C, H, W, per);
    return launch_status();
}

}}  // namespace d2t::tuned
