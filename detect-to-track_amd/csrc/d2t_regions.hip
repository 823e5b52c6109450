// d2t_regions.hip -- region proposals on the device: anchor-offset decoding, confidence filter, top-k, greedy NMS.
//
// SURVEY 8f-4.  Between the RPN and the R-FCN heads the reference copies the RPN's confidences and offsets to the
// host, decodes them with numpy (data/encoding.py:182-206, frcnn_box_decode), filters them with three `ml_utils`
// filters -- ConfidenceFilter(thresh), MaxDetFilter(max_dets), NMSFilter(iou) (trainer.py:98-102,178-190;
// inference.py:37-41,78-84) -- and copies the surviving boxes back (trainer.py:206-207): two device->host->device
// round trips per frame pair.  This file does the same chain in four small launches with no host involvement; the
// number of survivors stays on the device (out_count) and the box list is padded with zero boxes to max_dets.
//
// `ml_utils` (requirements.txt: ml-utils==3.0.0) is not vendored with the reference, so its three filters are
// restated from their names and call sites as the standard operations (PARITY UNPINNED for them):
//   confidence filter   keep conf > thresh
//   max-det filter      keep the max_dets highest confidences (ties: lower anchor index first)
//   NMS                 greedy, in descending confidence: a kept box removes every later box with IoU > iou_thresh
// Boxes are (centre_i, centre_j, height, width) fractions of the frame, as everywhere in the reference.
#include "d2t_kernels.hpp"

namespace d2t {

namespace {

constexpr int RG_MAXK = 4096;                 // most boxes that can survive the max-det filter
constexpr int RG_T = 1024;                    // threads of the single-workgroup kernels
constexpr int RG_W = RG_MAXK / 64;            // 64-bit words per NMS mask row

// monotone map float -> unsigned (larger float = larger key); 0 is reserved for "filtered out"
__device__ __forceinline__ unsigned order_key(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// data/encoding.py:182-206: b_ij = t_ij * a_hw + a_ij,  b_hw = exp(t_hw) * a_hw   (unfused, as numpy evaluates them)
__global__ void __launch_bounds__(256)
k_region_decode(const float* __restrict__ anchors, const float* __restrict__ offsets, const float* __restrict__ confs,
                float* __restrict__ boxes, unsigned* __restrict__ keys, int A, float thresh)
{
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= A) return;
    const float4 an = reinterpret_cast<const float4*>(anchors)[a], t = reinterpret_cast<const float4*>(offsets)[a];
    float4 b;
    b.x = t.x * an.z + an.x;
    b.y = t.y * an.w + an.y;
    b.z = expf(t.z) * an.z;
    b.w = expf(t.w) * an.w;
    reinterpret_cast<float4*>(boxes)[a] = b;
    const float c = confs[a];
    keys[a] = c > thresh ? order_key(c) : 0u;                        // NaN confidences are filtered out
}

// One workgroup: radix-select the max_dets-th largest key, compact the survivors in anchor order, sort them by
// (confidence descending, anchor index ascending) with a bitonic network in LDS, gather their boxes.
__global__ void __launch_bounds__(RG_T)
k_region_topk(const unsigned* __restrict__ keys, const float* __restrict__ boxes, const float* __restrict__ confs,
              float* __restrict__ sboxes, float* __restrict__ sconf, int* __restrict__ sidx, int* __restrict__ nsel, int A, int K, int NS)
{   // NS: the sorting network's size, the power of two >= K (<= RG_MAXK)
    __shared__ unsigned hist[256];
    __shared__ unsigned long long cand[RG_MAXK];
    __shared__ unsigned scan[RG_T];
    __shared__ unsigned s_prefix, s_need, s_total;
    const int tid = threadIdx.x;

    // ---- 4 passes of 8 bits, most significant first: `prefix` = the bits of the K-th largest key found so far
    unsigned prefix = 0, need = (unsigned)K;                          // need: rank still wanted among keys matching the prefix
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned hi_mask = pass == 0 ? 0u : 0xffffffffu << (shift + 8);
        for (int a = tid; a < A; a += RG_T) {
            const unsigned k = keys[a];
            if (k != 0 && (k & hi_mask) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1u);
        }
        __syncthreads();
        if (tid == 0) {                                              // walk the 256 bins from the top
            unsigned acc = 0, bin = 0, nd = need;
            bool found = false;
            for (int v = 255; v >= 0; --v) {
                if (acc + hist[v] >= nd) { bin = (unsigned)v; nd -= acc; found = true; break; }
                acc += hist[v];
            }
            if (!found) { bin = 0; nd = 0; }                         // fewer than K valid keys: everything valid is taken
            s_prefix = prefix | (bin << shift);
            s_need = found ? nd : 0;
            s_total = found ? 1u : 0u;
        }
        __syncthreads();
        prefix = s_prefix; need = s_need;
        if (!s_total) { prefix = 0; need = 0; break; }               // (uniform: s_total is shared)
        __syncthreads();
    }
    __syncthreads();
    // survivors: key > T, plus the first `need` keys == T in anchor order (T = 0: all valid keys)
    const unsigned T = prefix;
    const int per = (A + RG_T - 1) / RG_T, a0 = tid * per, a1 = a0 + per < A ? a0 + per : A;
    unsigned n_gt = 0, n_eq = 0;
    for (int a = a0; a < a1; ++a) {
        const unsigned k = keys[a];
        n_gt += (k != 0 && k > T) ? 1u : 0u;
        n_eq += (k != 0 && k == T && T != 0) ? 1u : 0u;
    }
    auto exclusive_scan = [&](unsigned v, unsigned& total) {         // over the 1024 threads
        scan[tid] = v;
        __syncthreads();
        for (int off = 1; off < RG_T; off <<= 1) {
            const unsigned add = tid >= off ? scan[tid - off] : 0u;
            __syncthreads();
            scan[tid] += add;
            __syncthreads();
        }
        total = scan[RG_T - 1];
        const unsigned ex = scan[tid] - v;
        __syncthreads();
        return ex;
    };
    unsigned tot_gt, tot_eq;
    const unsigned eq_before = exclusive_scan(n_eq, tot_eq);
    unsigned take_eq = 0;                                            // how many of this thread's ties are taken
    if (T != 0) {
        const unsigned room = eq_before < need ? need - eq_before : 0u;
        take_eq = n_eq < room ? n_eq : room;
    }
    unsigned tot;
    const unsigned base = exclusive_scan(n_gt + take_eq, tot);
    (void)tot_gt;
    unsigned w = base, eq_left = take_eq;
    for (int a = a0; a < a1; ++a) {
        const unsigned k = keys[a];
        bool take = k != 0 && k > T;
        if (k != 0 && k == T && T != 0 && eq_left) { take = true; --eq_left; }
        if (take && w < (unsigned)RG_MAXK) cand[w++] = ((unsigned long long)k << 32) | (unsigned)(0x7fffffff - a);   // ties: lower index = larger
    }
    const int n = (int)(tot < (unsigned)K ? tot : (unsigned)K);
    __syncthreads();
    for (int e = n + tid; e < NS; e += RG_T) cand[e] = 0ull;         // padding sorts to the end
    __syncthreads();
    // ---- bitonic sort, descending
    for (int k2 = 2; k2 <= NS; k2 <<= 1)
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int e = tid; e < NS; e += RG_T) {
                const int p = e ^ j;
                if (p > e) {
                    const unsigned long long x = cand[e], y = cand[p];
                    const bool desc = (e & k2) == 0;
                    if (desc ? x < y : x > y) { cand[e] = y; cand[p] = x; }
                }
            }
            __syncthreads();
        }
    for (int e = tid; e < n; e += RG_T) {
        const int a = 0x7fffffff - (int)(unsigned)(cand[e] & 0xffffffffu);
        reinterpret_cast<float4*>(sboxes)[e] = reinterpret_cast<const float4*>(boxes)[a];
        sconf[e] = confs[a];
        sidx[e] = a;
    }
    if (tid == 0) *nsel = n;
}

__device__ __forceinline__ float box_iou(const float4& a, const float4& b)
{
    const float ai0 = a.x - a.z / 2.f, ai1 = a.x + a.z / 2.f, aj0 = a.y - a.w / 2.f, aj1 = a.y + a.w / 2.f;
    const float bi0 = b.x - b.z / 2.f, bi1 = b.x + b.z / 2.f, bj0 = b.y - b.w / 2.f, bj1 = b.y + b.w / 2.f;
    const float ih = fminf(ai1, bi1) - fmaxf(ai0, bi0), iw = fminf(aj1, bj1) - fmaxf(aj0, bj0);
    const float inter = (ih > 0.f ? ih : 0.f) * (iw > 0.f ? iw : 0.f);
    const float uni = a.z * a.w + b.z * b.w - inter;
    return uni > 0.f ? inter / uni : 0.f;
}

// mask[i][w] bit j: box 64w+j comes later than box i and overlaps it by more than `iou`
__global__ void __launch_bounds__(64)
k_region_mask(const float* __restrict__ sboxes, const int* __restrict__ nsel, unsigned long long* __restrict__ mask, float iou)
{
    const int n = *nsel, rb = blockIdx.y, cb = blockIdx.x, lane = threadIdx.x;
    if (64 * rb >= n || 64 * cb >= n || cb < rb) return;             // (rows beyond n are never read; words left of the diagonal are zeroed below)
    __shared__ float4 col[64];
    col[lane] = 64 * cb + lane < n ? reinterpret_cast<const float4*>(sboxes)[64 * cb + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int i = 64 * rb + lane;
    if (i >= n) return;
    const float4 me = reinterpret_cast<const float4*>(sboxes)[i];
    unsigned long long m = 0;
    for (int j = 0; j < 64; ++j) {
        const int c = 64 * cb + j;
        if (c > i && c < n && box_iou(me, col[j]) > iou) m |= 1ull << j;
    }
    mask[(size_t)i * RG_W + cb] = m;
}

// One wave: greedy scan in blocks of 64 boxes.  Lane w owns word w of the `removed` bit set.  Inside a block the
// 64 x 64 diagonal sub-mask decides serially which boxes survive; the survivors' rows are then OR-ed in in parallel.
__global__ void __launch_bounds__(64)
k_region_nms(const float* __restrict__ sboxes, const float* __restrict__ sconf, const int* __restrict__ sidx,
             const int* __restrict__ nsel, const unsigned long long* __restrict__ mask,
             float* __restrict__ out_boxes, float* __restrict__ out_conf, int* __restrict__ out_idx, int* __restrict__ out_count, int K)
{
    __shared__ unsigned long long rows[64][RG_W + 1];
    const int lane = threadIdx.x, n = *nsel, nblk = (n + 63) / 64;
    unsigned long long removed = 0;                                  // word `lane`
    int kept = 0;
    for (int blk = 0; blk < nblk; ++blk) {
        const int nb = n - 64 * blk < 64 ? n - 64 * blk : 64;
        for (int e = lane; e < 64 * RG_W; e += 64) {                 // this block's rows, words >= blk (earlier ones are unset)
            const int r = e / RG_W, w = e - r * RG_W;
            rows[r][w] = (r < nb && w >= blk && 64 * w < n) ? mask[(size_t)(64 * blk + r) * RG_W + w] : 0ull;
        }
        __syncthreads();
        // serial part: the block's own word of `removed`
        unsigned long long rem = __shfl(removed, blk, 64), keep = 0;
        for (int r = 0; r < nb; ++r)
            if (!((rem >> r) & 1)) { keep |= 1ull << r; rem |= rows[r][blk]; }
        // parallel part: every later word
        if (lane > blk) {
            unsigned long long acc = removed;
            for (int r = 0; r < nb; ++r)
                if ((keep >> r) & 1) acc |= rows[r][lane];
            removed = acc;
        }
        if (lane == blk) removed = rem;
        // survivors of this block, in order
        const bool mine = lane < nb && ((keep >> lane) & 1);
        const int pos = kept + __popcll(keep & ((1ull << lane) - 1ull));
        if (mine && pos < K) {
            const int src = 64 * blk + lane;
            reinterpret_cast<float4*>(out_boxes)[pos] = reinterpret_cast<const float4*>(sboxes)[src];
            out_conf[pos] = sconf[src];
            out_idx[pos] = sidx[src];
        }
        kept += __popcll(keep);
        __syncthreads();
    }
    kept = kept < K ? kept : K;
    for (int e = kept + lane; e < K; e += 64) {                      // padding: zero boxes (PSROIPool pools them to 0), index -1
        reinterpret_cast<float4*>(out_boxes)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        out_conf[e] = 0.f;
        out_idx[e] = -1;
    }
    if (lane == 0) *out_count = kept;
}

inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

size_t region_filter_ws_bytes(int A, int max_dets)
{
    if (A < 1 || max_dets < 1 || max_dets > RG_MAXK) return 0;
    return al256((size_t)A * 16) + al256((size_t)A * 4) + al256((size_t)RG_MAXK * 16) + 2 * al256((size_t)RG_MAXK * 4) + 256 +
           al256((size_t)RG_MAXK * RG_W * 8);
}

int region_filter_f32(const float* anchors, const float* offsets, const float* confs, int A,
                      float conf_thresh, int max_dets, float iou_thresh,
                      float* out_boxes, float* out_conf, int* out_idx, int* out_count, void* ws, hipStream_t st)
{
    char* w = static_cast<char*>(ws);
    float* boxes = reinterpret_cast<float*>(w); w += al256((size_t)A * 16);
    unsigned* keys = reinterpret_cast<unsigned*>(w); w += al256((size_t)A * 4);
    float* sboxes = reinterpret_cast<float*>(w); w += al256((size_t)RG_MAXK * 16);
    float* sconf = reinterpret_cast<float*>(w); w += al256((size_t)RG_MAXK * 4);
    int* sidx = reinterpret_cast<int*>(w); w += al256((size_t)RG_MAXK * 4);
    int* nsel = reinterpret_cast<int*>(w); w += 256;
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(w);
    hipLaunchKernelGGL(k_region_decode, dim3((A + 255) / 256), dim3(256), 0, st, anchors, offsets, confs, boxes, keys, A, conf_thresh);
    int ns = 64;
    while (ns < max_dets) ns <<= 1;
    hipLaunchKernelGGL(k_region_topk, dim3(1), dim3(RG_T), 0, st, keys, boxes, confs, sboxes, sconf, sidx, nsel, A, max_dets, ns);
    const int nb = (max_dets + 63) / 64;
    hipLaunchKernelGGL(k_region_mask, dim3(nb, nb), dim3(64), 0, st, sboxes, nsel, mask, iou_thresh);
    hipLaunchKernelGGL(k_region_nms, dim3(1), dim3(64), 0, st, sboxes, sconf, sidx, nsel, mask, out_boxes, out_conf, out_idx, out_count, max_dets);
    return launch_status();
}

int region_max_dets() { return RG_MAXK; }

}  // namespace d2t
