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

#include <type_traits>
namespace d2t {

namespace {

constexpr int RG_MAXK = 4096;                 // most boxes that can survive the max-det filter
constexpr int RG_T = 1024;                    // threads of the single-workgroup kernels
constexpr int RG_W = RG_MAXK / 64;            // 64-bit words per NMS mask row
constexpr int RG_BINS = 4096;                 // counters of a radix-select pass (12-bit digits)
constexpr int RG_FUSE = 512;                  // most boxes the single-kernel path (top-k + IoU matrix + greedy scan) takes

// monotone map float -> unsigned (larger float = larger key); 0 is reserved for "filtered out"
__device__ __forceinline__ unsigned order_key(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Batched calls (several frames of one step, SURVEY 8f-1 / trainer.py:178-207 per frame): blockIdx.z = frame.  The anchors
// are shared; offsets / confidences / outputs of frame f follow those of frame f-1; every frame has its own copy of the
// workspace layout, ws_stride bytes apart.
template <typename T>
__device__ __forceinline__ T* frame_ws(T* p, size_t ws_stride, int f) { return reinterpret_cast<T*>(reinterpret_cast<char*>(const_cast<typename std::remove_const<T>::type*>(p)) + (size_t)f * ws_stride); }

// data/encoding.py:182-206: b_ij = t_ij * a_hw + a_ij,  b_hw = exp(t_hw) * a_hw   (unfused, as numpy evaluates them)
__global__ void __launch_bounds__(256)
k_region_decode(const float* __restrict__ anchors, const float* __restrict__ offsets, const float* __restrict__ confs,
                float* __restrict__ boxes, unsigned* __restrict__ keys, int A, float thresh, size_t ws_stride)
{
    const int f = blockIdx.z;
    offsets += (size_t)f * A * 4; confs += (size_t)f * A;
    boxes = frame_ws(boxes, ws_stride, f); keys = frame_ws(keys, ws_stride, f);
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= A) {
        if (a < ((A + 3) & ~3)) keys[a] = 0u;                        // the selection reads keys 16 bytes at a time
        return;
    }
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

// ---- helpers of the single-workgroup kernel
__device__ __forceinline__ unsigned wave_inclusive_scan(unsigned v, int lane)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}
// exclusive scan of v over the workgroup's RG_T threads (2 barriers); wtot: RG_T/64 words of LDS
__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned& total, unsigned* wtot, int lane, int wave)
{
    const unsigned incl = wave_inclusive_scan(v, lane);
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    unsigned base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < RG_T / 64; ++w) {
        const unsigned t = wtot[w];
        base += w < wave ? t : 0u;
        tot += t;
    }
    __syncthreads();
    total = tot;
    return base + incl - v;
}

__device__ __forceinline__ bool iou_exceeds(const float4& a, const float4& b, float thr);

// The greedy decision of one block of 64 boxes, by ONE wave.  `diag`: lane r holds row r's word of the block itself;
// rem: the block's word of the removed set.  Everything in the chain is wave-uniform, so it is kept in SGPRs
// (readfirstlane / readlane, scalar bit tests): 64 dependent steps of a few scalar instructions, no memory access.
__device__ __forceinline__ unsigned long long greedy_block(unsigned long long diag, unsigned long long& rem_io, int nb)
{
    const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
    unsigned rlo = __builtin_amdgcn_readfirstlane((unsigned)rem_io), rhi = __builtin_amdgcn_readfirstlane((unsigned)(rem_io >> 32));
    unsigned klo = 0, khi = 0;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)dlo, r), b = (unsigned)__builtin_amdgcn_readlane((int)dhi, r);
        if (r < nb && !((rlo >> r) & 1u)) { klo |= 1u << r; rlo |= a; rhi |= b; }
    }
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)dlo, 32 + r), b = (unsigned)__builtin_amdgcn_readlane((int)dhi, 32 + r);
        if (32 + r < nb && !((rhi >> r) & 1u)) { khi |= 1u << r; rlo |= a; rhi |= b; }
    }
    rem_io = (unsigned long long)rlo | ((unsigned long long)rhi << 32);
    return (unsigned long long)klo | ((unsigned long long)khi << 32);
}
// OR of the rows of the block's survivors (bits of the wave-uniform `keep`), word `col` of each: 8 LDS reads in flight
template <typename RowFn>
__device__ __forceinline__ unsigned long long or_kept_rows(unsigned long long keep, unsigned long long acc, RowFn row)
{
#pragma unroll 1
    for (int r0 = 0; r0 < 64; r0 += 8) {
        if (!((keep >> r0) & 0xffull)) continue;                     // wave-uniform
        unsigned long long v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = row(r0 + k);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc |= ((keep >> (r0 + k)) & 1ull) ? v[k] : 0ull;
    }
    return acc;
}

// One workgroup does everything behind the decode: radix-select the max_dets-th largest key, compact the survivors,
// order them by (confidence descending, anchor index ascending), gather their boxes and -- when max_dets <= RG_FUSE
// -- the IoU bit matrix and the greedy scan as well (FUSED), so a frame's proposals cost two launches.
//
// NV > 0: the thread keeps 4*NV keys in registers (ONE round of coalesced 16-byte loads for the whole selection;
// the first version re-read the keys from memory in each of the four passes, 35 dependent load latencies per pass
// for a 38 x 63 map, and was 0.26 ms).  NV == 0: any A, keys re-read per pass.
template <int NV, bool FUSED>
__global__ void __launch_bounds__(RG_T)
k_region_topk(const unsigned* __restrict__ keys, const float* __restrict__ boxes, const float* __restrict__ confs,
              float* __restrict__ sboxes, float* __restrict__ sconf, int* __restrict__ sidx, int* __restrict__ nsel,
              float* __restrict__ out_boxes, float* __restrict__ out_conf, int* __restrict__ out_idx, int* __restrict__ out_count,
              int A, int K, int NS, float iou, size_t ws_stride)
{   // NS: the sorting network's size, the power of two >= K (<= RG_MAXK); keys is padded with zeros to a multiple of 4
    __shared__ unsigned hist[RG_BINS];
    __shared__ unsigned long long cand[RG_MAXK];                     // candidates; later the fused path's bit matrix
    __shared__ unsigned wtot[RG_T / 64];
    __shared__ unsigned s_prefix[4], s_need[4], s_eq[4], s_found[4];
    __shared__ float4 sb[FUSED ? RG_FUSE : 1];
    __shared__ float sc[FUSED ? RG_FUSE : 1];
    __shared__ int si[FUSED ? RG_FUSE : 1];
    {
        const int f = blockIdx.z;                                    // frame of a batched call
        keys = frame_ws(keys, ws_stride, f); boxes = frame_ws(boxes, ws_stride, f); confs += (size_t)f * A;
        sboxes = frame_ws(sboxes, ws_stride, f); sconf = frame_ws(sconf, ws_stride, f); sidx = frame_ws(sidx, ws_stride, f);
        nsel = frame_ws(nsel, ws_stride, f);
        out_boxes += (size_t)f * K * 4; out_conf += (size_t)f * K; out_idx += (size_t)f * K; out_count += f;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int A4 = (A + 3) >> 2;                                     // 16-byte pieces of the key array

    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 kv[NV > 0 ? NV : 1];
    if constexpr (NV > 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int q = i * RG_T + tid;
            kv[i] = q < A4 ? reinterpret_cast<const u32x4*>(keys)[q] : u32x4{0u, 0u, 0u, 0u};
        }
    }
    // f(key, anchor index) over this thread's keys
    auto for_each_key = [&](auto f) {
        if constexpr (NV > 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const u32x4 v = kv[i];
                const int a = (i * RG_T + tid) * 4;
                f(v.x, a); f(v.y, a + 1); f(v.z, a + 2); f(v.w, a + 3);
            }
        } else {
            for (int q = tid; q < A4; q += RG_T) {
                const u32x4 v = reinterpret_cast<const u32x4*>(keys)[q];
                f(v.x, 4 * q); f(v.y, 4 * q + 1); f(v.z, 4 * q + 2); f(v.w, 4 * q + 3);
            }
        }
    };
    if (tid < 4) s_found[tid] = 0;

    // ---- radix select, most significant digit first: 12 + 12 + 8 bits.  `prefix` = the bits of the K-th largest key
    // found so far.  Plain LDS atomics: the keys that pass the confidence filter share two or three exponents, so an
    // 8-bit first digit would send them all to two or three counters (64 lanes serialising on one address), while
    // 12 bits spread them over dozens; a leader lane collecting the count of its bin per wave-instruction (the
    // first version) cost three times the instructions of the atomic itself.
    unsigned prefix = 0, need = (unsigned)K, eq = 0;                  // need: rank still wanted among the keys matching the prefix
    bool found = true;
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass == 0 ? 20 : (pass == 1 ? 8 : 0);
        const unsigned dmask = pass == 2 ? 0xffu : 0xfffu;
        const unsigned hi_mask = pass == 0 ? 0u : (pass == 1 ? 0xfff00000u : 0xffffff00u);
#pragma unroll
        for (int q = 0; q < RG_BINS / RG_T; ++q) hist[q * RG_T + tid] = 0;
        __syncthreads();
        for_each_key([&](unsigned k, int) {
            if (k != 0 && (k & hi_mask) == prefix) atomicAdd(&hist[(k >> shift) & dmask], 1u);
        });
        __syncthreads();
        unsigned c[RG_BINS / RG_T], sum = 0;                          // this thread's bins, from the top
#pragma unroll
        for (int q = 0; q < RG_BINS / RG_T; ++q) { c[q] = hist[RG_BINS - 1 - (tid * (RG_BINS / RG_T) + q)]; sum += c[q]; }
        unsigned total;
        unsigned run = block_exclusive_scan(sum, total, wtot, lane, wave);
#pragma unroll
        for (int q = 0; q < RG_BINS / RG_T; ++q) {                    // exactly one (thread, q) holds the K-th key, or none when total < need
            if (run < need && run + c[q] >= need) {
                s_prefix[pass] = prefix | ((unsigned)(RG_BINS - 1 - (tid * (RG_BINS / RG_T) + q)) << shift);
                s_need[pass] = need - run;
                s_eq[pass] = c[q];
                s_found[pass] = 1u;
            }
            run += c[q];
        }
        __syncthreads();
        if (!s_found[pass]) { found = false; break; }                // fewer than K valid keys: everything valid is taken
        prefix = s_prefix[pass]; need = s_need[pass]; eq = s_eq[pass];
    }
    // survivors: key > T, plus the first `need` keys == T in anchor order (not found: all valid keys)
    const unsigned T = found ? prefix : 0u;
    unsigned tot;
    if (NV > 0 && (!found || eq == need)) {
        // every key equal to T is taken (the usual case: T is the K-th key itself): no order among the ties needed
        unsigned cnt = 0;
        for_each_key([&](unsigned k, int) { cnt += (k != 0 && k >= T) ? 1u : 0u; });
        unsigned w = block_exclusive_scan(cnt, tot, wtot, lane, wave);
        for_each_key([&](unsigned k, int a) {
            if (k != 0 && k >= T && w < (unsigned)RG_MAXK) cand[w++] = ((unsigned long long)k << 32) | (unsigned)(0x7fffffff - a);
        });
    } else {
        // ties cut by the K-th rank: walk the anchors in order (threads own contiguous ranges; keys re-read)
        const int per = (A + RG_T - 1) / RG_T, a0 = tid * per, a1 = a0 + per < A ? a0 + per : A;
        unsigned n_gt = 0, n_eq = 0;
        for (int a = a0; a < a1; ++a) {
            const unsigned k = keys[a];
            n_gt += (k != 0 && k > T) ? 1u : 0u;
            n_eq += (k != 0 && k == T && T != 0) ? 1u : 0u;
        }
        unsigned tot_eq;
        const unsigned eq_before = block_exclusive_scan(n_eq, tot_eq, wtot, lane, wave);
        unsigned take_eq = 0;                                        // how many of this thread's ties are taken
        if (T != 0) {
            const unsigned room = eq_before < need ? need - eq_before : 0u;
            take_eq = n_eq < room ? n_eq : room;
        }
        unsigned w = block_exclusive_scan(n_gt + take_eq, tot, wtot, lane, wave), eq_left = take_eq;
        for (int a = a0; a < a1; ++a) {
            const unsigned k = keys[a];
            bool take = k != 0 && k > T;
            if (k != 0 && k == T && T != 0 && eq_left) { take = true; --eq_left; }
            if (take && w < (unsigned)RG_MAXK) cand[w++] = ((unsigned long long)k << 32) | (unsigned)(0x7fffffff - a);   // ties: lower index = larger
        }
    }
    const int n = __builtin_amdgcn_readfirstlane((int)(tot < (unsigned)K ? tot : (unsigned)K));   // (the same in every thread)
    __syncthreads();

    // ---- order: (key descending, anchor ascending).  Up to 1024 candidates: every thread ranks its own among all
    // (the composite values are distinct; LDS broadcast reads, no barrier); more: bitonic network.
    auto emit = [&](int pos, unsigned long long c) {
        const int a = 0x7fffffff - (int)(unsigned)(c & 0xffffffffu);
        const float4 bx = reinterpret_cast<const float4*>(boxes)[a];
        const float cf = confs[a];
        if constexpr (FUSED) { sb[pos] = bx; sc[pos] = cf; si[pos] = a; }
        else { reinterpret_cast<float4*>(sboxes)[pos] = bx; sconf[pos] = cf; sidx[pos] = a; }
    };
    if (n <= RG_T) {
        if (tid < n) {
            const unsigned long long x = cand[tid];
            int rank = 0;
#pragma unroll 4
            for (int j = 0; j < n; ++j) rank += cand[j] > x ? 1 : 0;
            emit(rank, x);
        }
    } else {
        for (int e = n + tid; e < NS; e += RG_T) cand[e] = 0ull;     // padding sorts to the end
        __syncthreads();
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
        for (int e = tid; e < n; e += RG_T) emit(e, cand[e]);
    }
    if constexpr (!FUSED) {
        if (tid == 0) *nsel = n;
        return;
    } else {
        // ---- fused tail (n <= RG_FUSE): bit matrix in LDS (over cand, which is dead), greedy scan by wave 0
        __syncthreads();
        constexpr int FW = RG_FUSE / 64;
        unsigned long long (*msk)[FW] = reinterpret_cast<unsigned long long (*)[FW]>(cand);
        const int nw = (n + 63) >> 6;
        for (int item = tid; item < n * nw; item += RG_T) {
            const int i = item / nw, w = item - i * nw;
            unsigned long long m = 0;
            if (w >= (i >> 6)) {
                const float4 me = sb[i];
                for (int j = 0; j < 64; ++j) {
                    const int c = 64 * w + j;
                    if (c > i && c < n && iou_exceeds(me, sb[c], iou)) m |= 1ull << j;
                }
            }
            msk[i][w] = m;
        }
        __syncthreads();
        if (wave != 0) return;
        unsigned long long removed = 0;                              // word `lane` of the removed set
        int kept = 0;
        for (int blk = 0; blk < nw; ++blk) {
            const int nb = n - 64 * blk < 64 ? n - 64 * blk : 64;
            const unsigned long long diag = lane < nb ? msk[64 * blk + lane][blk] : 0ull;   // lane r: row r's own-block word
            unsigned long long rem = __shfl(removed, blk, 64);
            const unsigned long long keep = greedy_block(diag, rem, nb);
            if (lane > blk && lane < nw)
                removed = or_kept_rows(keep, removed, [&](int r) { return msk[64 * blk + r][lane]; });
            if (lane == blk) removed = rem;
            const bool mine = lane < nb && ((keep >> lane) & 1);
            const int pos = kept + __popcll(keep & ((1ull << lane) - 1ull));
            if (mine && pos < K) {
                const int src = 64 * blk + lane;
                reinterpret_cast<float4*>(out_boxes)[pos] = sb[src];
                out_conf[pos] = sc[src];
                out_idx[pos] = si[src];
            }
            kept += __popcll(keep);
        }
        kept = kept < K ? kept : K;
        for (int e = kept + lane; e < K; e += 64) {                  // padding: zero boxes (PSROIPool pools them to 0), index -1
            reinterpret_cast<float4*>(out_boxes)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            out_conf[e] = 0.f;
            out_idx[e] = -1;
        }
        if (lane == 0) *out_count = kept;
    }
}

// IoU(a, b) > thr, decided as the float quotient inter / uni would decide it (oracle/regions.py: float32 throughout).
// The quotient is first estimated with v_rcp_f32 (error of a few 1e-7 for a value in [0, 1]); the IEEE division runs
// only when the estimate is within 1e-6 of the threshold or not a number (uni <= 0: the pair scores 0).
__device__ __forceinline__ bool iou_exceeds(const float4& a, const float4& b, float thr)
{
    const float ai0 = a.x - a.z / 2.f, ai1 = a.x + a.z / 2.f, aj0 = a.y - a.w / 2.f, aj1 = a.y + a.w / 2.f;
    const float bi0 = b.x - b.z / 2.f, bi1 = b.x + b.z / 2.f, bj0 = b.y - b.w / 2.f, bj1 = b.y + b.w / 2.f;
    const float ih = fminf(ai1, bi1) - fmaxf(ai0, bi0), iw = fminf(aj1, bj1) - fmaxf(aj0, bj0);
    const float inter = (ih > 0.f ? ih : 0.f) * (iw > 0.f ? iw : 0.f);
    const float uni = a.z * a.w + b.z * b.w - inter;
    const float q = inter * __builtin_amdgcn_rcpf(uni);
    if (uni > 0.f && fabsf(q - thr) > 1e-6f && q <= 2.f) return q > thr;   // (q <= 2: its error stays below the margin)
    return (uni > 0.f ? inter / uni : 0.f) > thr;
}

// mask[i][w] bit j: box 64w+j comes later than box i and overlaps it by more than `iou`
__global__ void __launch_bounds__(64)
k_region_mask(const float* __restrict__ sboxes, const int* __restrict__ nsel, unsigned long long* __restrict__ mask, float iou,
              size_t ws_stride)
{
    sboxes = frame_ws(sboxes, ws_stride, blockIdx.z); nsel = frame_ws(nsel, ws_stride, blockIdx.z); mask = frame_ws(mask, ws_stride, blockIdx.z);
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
        if (c > i && c < n && iou_exceeds(me, col[j], iou)) m |= 1ull << j;
    }
    mask[(size_t)i * RG_W + cb] = m;
}

// Greedy scan in blocks of 64 boxes.  Wave 0 decides: lane w owns word w of the `removed` bit set; inside a block the
// 64 x 64 diagonal sub-mask decides serially which boxes survive (the diagonal words sit in lanes and are read with
// v_readlane: no memory access in the serial chain); the survivors' rows are then OR-ed into the later words in
// parallel.  All 16 waves fetch: a block's rows (64 x up to 64 words) are one load per thread and register, issued
// two blocks ahead, so that the chain of blocks never waits for memory (the first version -- one wave loading each
// block's rows when it got there -- took 1.0 ms for 3000 boxes).
__global__ void __launch_bounds__(RG_T)
k_region_nms(const float* __restrict__ sboxes, const float* __restrict__ sconf, const int* __restrict__ sidx,
             const int* __restrict__ nsel, const unsigned long long* __restrict__ mask,
             float* __restrict__ out_boxes, float* __restrict__ out_conf, int* __restrict__ out_idx, int* __restrict__ out_count, int K,
             size_t ws_stride)
{
    {
        const int f = blockIdx.z;
        sboxes = frame_ws(sboxes, ws_stride, f); sconf = frame_ws(sconf, ws_stride, f); sidx = frame_ws(sidx, ws_stride, f);
        nsel = frame_ws(nsel, ws_stride, f); mask = frame_ws(mask, ws_stride, f);
        out_boxes += (size_t)f * K * 4; out_conf += (size_t)f * K; out_idx += (size_t)f * K; out_count += f;
    }
    __shared__ unsigned long long rows[64][RG_W + 1];
    __shared__ unsigned long long keepw[RG_W];                       // per block: its survivors ...
    __shared__ int keptb[RG_W + 1];                                  // ... and how many came before them
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = *nsel, nblk = (n + 63) / 64;
    constexpr int PER = 64 * RG_W / RG_T;                            // 4 words of a block per thread
    int er[PER], ew[PER];                                            // which (row, word) of a block: the same for every block
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int e = tid + RG_T * q;
        er[q] = nblk ? e / nblk : 64;                                // (rows >= 64: nothing to fetch)
        ew[q] = nblk ? e - er[q] * nblk : 0;
    }
    struct Pre { unsigned long long v[PER]; };
    auto fetch = [&](int blk) {                                      // words [blk, nblk) of the rows of block blk (earlier ones are unset)
        Pre p;
        const int nb = n - 64 * blk < 64 ? n - 64 * blk : 64;        // (<= 0 behind the last block)
#pragma unroll
        for (int q = 0; q < PER; ++q)
            p.v[q] = (er[q] < nb && ew[q] >= blk) ? mask[(size_t)(64 * blk + er[q]) * RG_W + ew[q]] : 0ull;
        return p;
    };
    auto publish = [&](const Pre& p) {
#pragma unroll
        for (int q = 0; q < PER; ++q)
            if (er[q] < 64) rows[er[q]][ew[q]] = p.v[q];
    };
    unsigned long long removed = 0;                                  // wave 0: word `lane`
    int kept = 0;
    auto decide = [&](int blk) {                                     // wave 0 only
        const int nb = n - 64 * blk < 64 ? n - 64 * blk : 64;
        const unsigned long long diag = lane < nb ? rows[lane][blk] : 0ull;
        unsigned long long rem = __shfl(removed, blk, 64);
        const unsigned long long keep = greedy_block(diag, rem, nb);
        if (lane > blk && lane < nblk)
            removed = or_kept_rows(keep, removed, [&](int r) { return rows[r][lane]; });
        if (lane == blk) removed = rem;
        if (lane == 0) { keepw[blk] = keep; keptb[blk] = kept; }     // the survivors are written out behind the scan: a gather
        kept += __popcll(keep);                                      // here would put memory latency (and, through the in-order
    };                                                               // vmcnt, the prefetches) into every step of the chain
    Pre p0 = fetch(0), p1 = fetch(1);
    for (int blk = 0; blk < nblk; blk += 2) {
        publish(p0);
        __syncthreads();
        p0 = fetch(blk + 2);
        if (wave == 0) decide(blk);
        __syncthreads();
        if (blk + 1 < nblk) {                                        // uniform
            publish(p1);
            __syncthreads();
            p1 = fetch(blk + 3);
            if (wave == 0) decide(blk + 1);
            __syncthreads();
        }
    }
    if (wave == 0 && lane == 0) keptb[RG_W] = kept;
    __syncthreads();
    const int total = keptb[RG_W] < K ? keptb[RG_W] : K;
    for (int src = tid; src < n; src += RG_T) {                      // survivors, in order
        const int blk = src >> 6, l = src & 63;
        const unsigned long long keep = keepw[blk];
        const int pos = keptb[blk] + __popcll(keep & ((1ull << l) - 1ull));
        if (((keep >> l) & 1) && pos < K) {
            reinterpret_cast<float4*>(out_boxes)[pos] = reinterpret_cast<const float4*>(sboxes)[src];
            out_conf[pos] = sconf[src];
            out_idx[pos] = sidx[src];
        }
    }
    for (int e = total + tid; e < K; e += RG_T) {                    // padding: zero boxes (PSROIPool pools them to 0), index -1
        reinterpret_cast<float4*>(out_boxes)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        out_conf[e] = 0.f;
        out_idx[e] = -1;
    }
    if (tid == 0) *out_count = total;
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
                      float* out_boxes, float* out_conf, int* out_idx, int* out_count, void* ws, hipStream_t st, int N)
{
    const size_t ws_stride = region_filter_ws_bytes(A, max_dets);    // frame f works in ws + f * ws_stride
    char* w = static_cast<char*>(ws);
    float* boxes = reinterpret_cast<float*>(w); w += al256((size_t)A * 16);
    unsigned* keys = reinterpret_cast<unsigned*>(w); w += al256((size_t)A * 4);
    float* sboxes = reinterpret_cast<float*>(w); w += al256((size_t)RG_MAXK * 16);
    float* sconf = reinterpret_cast<float*>(w); w += al256((size_t)RG_MAXK * 4);
    int* sidx = reinterpret_cast<int*>(w); w += al256((size_t)RG_MAXK * 4);
    int* nsel = reinterpret_cast<int*>(w); w += 256;
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(w);
    hipLaunchKernelGGL(k_region_decode, dim3((A + 255) / 256, 1, N), dim3(256), 0, st, anchors, offsets, confs, boxes, keys, A, conf_thresh, ws_stride);
    int ns = 64;
    while (ns < max_dets) ns <<= 1;
    const int A4 = (A + 3) / 4;
#define D2T_TOPK(NV, FUSED) hipLaunchKernelGGL((k_region_topk<NV, FUSED>), dim3(1, 1, N), dim3(RG_T), 0, st, keys, boxes, confs, sboxes, sconf, sidx, \
                                               nsel, out_boxes, out_conf, out_idx, out_count, A, max_dets, ns, iou_thresh, ws_stride)
    const bool fused = max_dets <= RG_FUSE;
    if (fused) {
        if (A4 <= 3 * RG_T) D2T_TOPK(3, true);
        else if (A4 <= 10 * RG_T) D2T_TOPK(10, true);
        else if (A4 <= 20 * RG_T) D2T_TOPK(20, true);
        else D2T_TOPK(0, true);
        return launch_status();
    }
    if (A4 <= 3 * RG_T) D2T_TOPK(3, false);
    else if (A4 <= 10 * RG_T) D2T_TOPK(10, false);
    else if (A4 <= 20 * RG_T) D2T_TOPK(20, false);
    else D2T_TOPK(0, false);
#undef D2T_TOPK
    const int nb = (max_dets + 63) / 64;
    hipLaunchKernelGGL(k_region_mask, dim3(nb, nb, N), dim3(64), 0, st, sboxes, nsel, mask, iou_thresh, ws_stride);
    hipLaunchKernelGGL(k_region_nms, dim3(1, 1, N), dim3(RG_T), 0, st, sboxes, sconf, sidx, nsel, mask, out_boxes, out_conf, out_idx, out_count, max_dets, ws_stride);
    return launch_status();
}

int region_max_dets() { return RG_MAXK; }

}  // namespace d2t
