// d2t_corr_bwd8.hip -- gfx950 f32 PointwiseCorrelation backward, 8-wave column-strip kernel
// (d_max = 8, stride 1, reference layout).  Gather form of pointwise_correlation_cuda.cu:145-171:
//     gX[c][t] = sum over window slots w of  G[t][w] * S[c][w]
// role 0 (gradFM0): t = centre pixels, S = FM1;  role 1 (gradFM1): t = displaced pixels, S = FM0;
// G = gradOut[b,i,j,di-i+d,dj-j+d] where the reference's loops visit that cell, else 0.  No atomics,
// every output element written once, fixed summation order.
//
// Same strip walk as k_corr_bwd_strip (d2t_corr_tuned.hip: one workgroup = (role, batch item, strip
// of 4 pixel columns, 256 channels), super-step = 4 map rows = 5 k-blocks of 16 window slots, 5
// tiles alive), rebuilt around what the round-2 instruments showed about that kernel:
//   * it sat at its 128-VGPR cap (16 waves), so the feature (S) fragment of a k-block was requested
//     less than one k-block before its MFMAs and every k-block opened with `s_waitcnt vmcnt(0)`:
//     the L2 round trip under load (1-2 us) was exposed 50 times per strip;
//   * all 16 waves produced G, stored tiles and hit the barrier at the same moment, with the matrix
//     pipe idle meanwhile.
// Here a workgroup is 8 waves of up to 256 VGPRs, a wave owns TWO c-tiles (32 channels):
//   * S fragments live in a static 5-entry register file a4[q][ct]: the piece of k-block q is
//     re-requested for the NEXT super-step as soon as its last MFMA has been issued -- a prefetch
//     distance of a whole super-step (about 5 us), through buffer loads whose range check returns
//     zeros for channels >= C (no clamps, no 64-bit address arithmetic);
//   * a G fragment read from LDS feeds both c-tiles (half the LDS read traffic per MFMA) and is
//     fetched one k-block ahead (two static register sets that swap roles every super-step);
//   * the one barrier of a super-step sits in front of its LAST k-block, whose G fragments are in
//     registers by then: barrier latency and the first LDS round trip of the next super-step hide
//     behind 40 MFMAs;
//   * G production (ring writes of super-step s+1, requests for s+2) and the tile stores are
//     spread over the k-blocks of a super-step instead of forming a burst at its end.
#include "d2t_corr_common.hpp"
#include <type_traits>

namespace d2t { namespace tuned {

// Per-wave clock reads for the developer harness lab/csrc/bwd8_stamp_lab.hip (a separate diagnostic build, see the
// MI355X guide "In-kernel stamps"); the product library is built without D2T_LAB: no stamp executes there.
#ifdef D2T_LAB
__device__ unsigned long long* lab8_stamps;                         // [workgroup][wave][16]
#define D2T_WCLK(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define D2T_WRT(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define D2T_LAB_ONLY(...) __VA_ARGS__
#else
#define D2T_WCLK(var)
#define D2T_WRT(var)
#define D2T_LAB_ONLY(...)
#endif

namespace {

constexpr int S8_CH = 256;                          // channels per workgroup: 16 / CT waves of CT c-tiles (16 channels) each
constexpr int S8_QUADS = KB_SS * NACT * 64;         // 1600 ring quads per super-step: [k-block][live tile][lane]
constexpr int S8_RING = 2048 * 4;                   // floats per ring buffer (32 KB): 2048 quad slots = 2048 / threads per thread
constexpr int S8_OOR = 0x7ffffff0;                  // byte offset that every buffer range check rejects

// One ring quad (k-block q of a super-step, live tile a, lane l) as a 16-byte run of gradOut: component c
// is cell cj + c of row ci of one centre pixel (role 0: the four cells feed four consecutive window
// slots of one tile pixel; role 1: one slot of four consecutive tile pixels, see g_put).
// off: byte offset of the run at super-step 0 (may be negative; only used for lo <= ss < hi);
// info: lo | hi << 8 | mask << 16, mask bit c = the reference's loops visit cell cj + c.  A run may
// start left of cell 0 or end right of cell 15 of its row: it then covers cells of the neighbouring
// row or pixel, inside gradOut[b] all the same (the first pixel's first row never starts left of cell 0,
// the last pixel's row 15 never ends past cell 288), and those components are masked off.
struct Quad8 { int off, info; };

// Which 4 window slots (one 16-byte piece of a map row) lane group gg of k-block q multiplies: map row 4*ss + xr,
// column group cg.  ROWKB = false: the 20 groups of a super-step in row-major order, four per k-block (a k-block
// straddles two rows).  ROWKB = true: k-blocks 0..3 are the first four groups of rows 0..3 (the four pieces of a
// channel are 64 contiguous bytes: 2 cache lines per channel and load instead of 3-4), k-block 4 collects the
// fifth group of the four rows.
template <bool ROWKB>
__device__ __forceinline__ void kb_slot(int q, int gg, int& xr, int& cg)
{
    if (ROWKB) {
        xr = q < 4 ? q : gg;
        cg = q < 4 ? gg : 4;
    } else {
        const int x = 4 * q + gg;
        xr = (x * 13) >> 6;                                                // x / 5 (x < 32)
        cg = x - xr * NCG;
    }
}

template <bool ROWKB>
__device__ __forceinline__ Quad8 quad8_desc(int role, int e, int H, int W, int tiles_i, int j0, int col0)
{
    const int q = e / (NACT * 64), r = e - q * (NACT * 64);
    const int a = r >> 6, l = r & 63, gg = l >> 4;
    int xr, cg;
    kb_slot<ROWKB>(q < KB_SS ? q : 0, gg, xr, cg);
    const int tpi = (l >> 2) & 3, lo2 = l & 3;                             // role 0: lo2 = tile column; role 1: slot column s
    const int ci = role ? 4 * a + tpi - xr : xr - 4 * a - tpi + 2 * DT;    // displaced - centre + d (constant)
    const int tj = j0 + lo2, sj = col0 + 4 * cg + (role ? lo2 : 0);
    const int cj = role ? j0 - sj + DT : sj - tj + DT;                     // component c reads cell cj + c
    int mask = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bool col_ok = role ? j0 + c < W : tj < W;
        mask |= (cj + c >= 0 && cj + c < 2 * DT && col_ok) ? (1 << c) : 0;
    }
    if (ci < 0 || ci >= 2 * DT || e >= S8_QUADS) mask = 0;
    const int pix0 = role ? xr * W + sj : (4 * (a - 2) + tpi) * W + tj;    // centre pixel at ss = 0
    int lo = 2 - a, hi = tiles_i + 2 - a;
    const int hi_t = (H - tpi + 3) / 4 + 2 - a;                            // 4(ss-2+a)+tpi < H
    const int hi_r = (H - xr + 3) / 4;                                     // 4ss+xr < H
    hi = hi < hi_t ? hi : hi_t;
    hi = hi < hi_r ? hi : hi_r;
    lo = lo < 0 ? 0 : lo;
    if (!mask || hi < lo) { lo = 0; hi = 0; }
    Quad8 d;
    d.off = (pix0 * CELLS + ci * CW + cj) * 4;
    d.info = lo | (hi << 8) | (mask << 16);
    return d;
}

// unvisited cells -> exact zeros.  Bit arithmetic on purpose (v_bfe_i32 + v_and): compares would be
// hoisted out of the strip loop into 16 SGPR pairs, and the kernel needs its SGPRs for three buffer
// descriptors (a descriptor that ends up in VGPRs turns every buffer access into a waterfall loop).
__device__ __forceinline__ f32x4 quad8_fix(const f32x4& v, int info)
{
    const u32x4 b = __builtin_bit_cast(u32x4, v);
    u32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = b[c] & (unsigned)__builtin_amdgcn_sbfe(info, 16 + c, 1);
    return __builtin_bit_cast(f32x4, o);
}

typedef std::integral_constant<int, 0> I0;
typedef std::integral_constant<int, 1> I1;
typedef std::integral_constant<int, 2> I2;
typedef std::integral_constant<int, 3> I3;
typedef std::integral_constant<int, 4> I4;
typedef std::integral_constant<int, 5> I5;

#define D2T_PIN() __builtin_amdgcn_sched_barrier(0)

// The strip walk for one role (compile-time: the two roles differ in the ring write pattern and in which
// k-block of a tile is structurally zero; a run-time role splits the pinned schedule into many basic blocks).
// FLEX: the workgroup owns the c-tiles [ct0, ct0 + nct) of its strip, nct <= 16, dealt to the waves round-robin (wave w:
// c-tiles w and w + 8, so that the two waves of a SIMD carry ceil(nct / 4) between them); a wave's missing c-tile is
// skipped behind wave-uniform branches.  That lets the host cut the channels of a SMALL grid (the model's B = 1 pairs:
// 32-38 strips) into as many blocks as fill the chip once, instead of 256-channel blocks that fill 60 % of it (C = 1024)
// or need a second, nearly empty round (C = 2048).  !FLEX: 256 channels, wave w = c-tiles 2w, 2w + 1, no branches.
#ifndef D2T_EXP_GCO
#define D2T_EXP_GCO 1
#endif
#ifndef D2T_EXP_SWZ
#define D2T_EXP_SWZ 1      // ring swizzle (round 4: 70.4 -> 70.2 us in two A/B rounds, LDS bank conflicts of the memory-order producer gone)
#endif
#ifndef D2T_EXP_PRIO
#define D2T_EXP_PRIO 0
#endif
#ifndef S8_ABL
#define S8_ABL 0      // timing ablations (lab builds only, results are wrong): 1 no S reloads, 2 no tile stores, 4 no G loads, 8 no ring writes, 32 role 0 alone, 64 role 1 alone
#endif
template <int role, int S8_CT, bool ROWKB, int ABL = 0, bool FLEX = false>    // ABL: ablation mask of lab/csrc/bwd8_stamp_lab (timing only)
__device__ __forceinline__ void strip8_body(float (&ring)[2][S8_RING], const float* __restrict__ gout,
                                            const float* __restrict__ fm0, const float* __restrict__ fm1,
                                            float* __restrict__ g0, float* __restrict__ g1,
                                            int b, int tj, int C, int H, int W, int tiles_i, int ct0, int nct)
{
    constexpr int S8_WAVES = 16 / S8_CT;
    constexpr int GT = S8_WAVES * 64, S8_NQ = 2048 / GT;             // threads, ring quads per thread
    constexpr int DEAD0 = ROWKB ? 3 : KB_SS - 1;                     // the k-block whose slots all lie below a tile's window (role 0)
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    D2T_LAB_ONLY(unsigned long long w_t0, w_t1, w_t2, w_t3, w_r1, w_r2, w_a, w_b, w_sl[3] = {0, 0, 0}, w_bar = 0, w_rot = 0;)
    D2T_WCLK(w_t0);
    const int j0 = tj * TP, HW = H * W;
    const int wleft = j0 - DT + role;                                // role 1 window is shifted by one
    const int col0 = wleft < 0 ? 0 : (wleft > W - WC ? W - WC : wleft);
    const float* S = (role ? fm0 : fm1) + (size_t)b * C * HW;
    float* gx = (role ? g1 : g0) + (size_t)b * C * HW;
    const float* gb = gout + (size_t)b * HW * CELLS;
    const unsigned plane_bytes = (unsigned)C * HW * 4u;
    const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(S, plane_bytes);
    const __amdgpu_buffer_rsrc_t rx = uniform_rsrc(gx, plane_bytes);
    const __amdgpu_buffer_rsrc_t rg = uniform_rsrc(gb, (unsigned)HW * CELLS * 4u);

    const int cw = FLEX ? (ct0 + wave) * 16 : ct0 * 16 + wave * (S8_CT * 16);   // first channel of this wave's first c-tile (!FLEX: ct0 = 16 x channel block)
    const bool on[2] = {!FLEX || wave < nct, !FLEX || wave + S8_WAVES < nct};             // which of its c-tiles exist (wave-uniform)
    // S piece of k-block q at super-step 0, c-tile 0 (bytes): lane (channel n, slot group 4q+g).  Channels
    // >= C lie behind the buffer: zeros.  Rows >= H (last super-step) read the next plane: their G is 0.
    int sv[KB_SS];
#pragma unroll
    for (int q = 0; q < KB_SS; ++q) {
        int xr, cg;
        kb_slot<ROWKB>(q, g, xr, cg);
        sv[q] = ((cw + n) * HW + xr * W + col0 + 4 * cg) * 4;
    }
    const int s_step = 4 * W * 4, ct_step = (FLEX ? S8_WAVES * 16 : 16) * HW * 4;
    auto s_load = [&](int ss, int q, int ct) -> f32x4 {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, sv[q] + ss * s_step + ct * ct_step, 0, 0);
        return __builtin_bit_cast(f32x4, v);
    };

    // ---- G production: ring quads e = tid + GT k, S8_NQ per thread.  (Measured and dropped: waves 0-3 -- which the
    // hardware favours at the matrix pipe, they reach every barrier a quarter of a super-step early -- producing all
    // of G in two batches: 72 -> 83 us; alternating s_setprio between the two waves of a SIMD: no change.)
    const int gtid = tid;
    // Which ring quad does production slot ep = tid + GT k fill?  Role 0: the 20 quads of one tile pixel and super-step
    // (4 window rows x 5 column groups) are 68 CONSECUTIVE floats of gradOut (rows ci .. ci+3 of the pixel's 17 x 17 cell
    // block), so consecutive lanes take the pieces of ONE pixel in memory order: a wave-instruction then touches the ~5
    // cache lines of each of its 3.2 pixels (about 17 lines) instead of 2 lines per (pixel, row) in 64 different cell blocks.
    // Role 1 keeps ring order (its 16-byte pieces lie 68 bytes apart whatever the order).
    constexpr bool GCO = D2T_EXP_GCO && ROWKB && role == 0;
    auto ring_index = [&](int ep) -> int {
        if (!GCO || ep >= S8_QUADS) return ep;
        const int p = (ep * 3277) >> 16, m = ep - p * 20;              // ep / 20 (ep < 1600): pixel of the 5 live tiles, piece of its 4 x 5
        const int a = p >> 4, pix = p & 15, xr = (m * 13) >> 6, cg = m - xr * 5;
        const int q = cg < 4 ? xr : 4, gg = cg < 4 ? cg : xr;
        return (q * NACT + a) * 64 + gg * 16 + pix;
    };
    Quad8 qd[S8_NQ];
    int er[S8_NQ];
#pragma unroll
    for (int k = 0; k < S8_NQ; ++k) {
        er[k] = ring_index(gtid + k * GT);
        qd[k] = quad8_desc<ROWKB>(role, er[k], H, W, tiles_i, j0, col0);
    }
    // (quads 1600..2047 do not exist: their descriptors are empty -- an out-of-range request, zeros into ring
    // slots nobody reads -- so that G production stays branch-free)
    const int g_step = 4 * W * CELLS * 4;                            // gradOut bytes per 4 map rows
    auto g_load = [&](int k, int ss) -> f32x4 {
        const int lo = qd[k].info & 255, hi = (qd[k].info >> 8) & 255;
        const int v = ss >= lo && ss < hi ? qd[k].off + ss * g_step : S8_OOR;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, v, 0, 0));
    };
    auto g_put = [&](float* rb, int k, const f32x4& raw) {
        const f32x4 v = quad8_fix(raw, qd[k].info);
        const int e = GCO ? er[k] : gtid + k * GT;
        // D2T_EXP_SWZ: inside its 64-quad block the quad of lane l = (gg, pixel) sits at position l ^ (gg & 3) -- the reader's
        // ds_read_b128 stays conflict-free (an XOR below 4 permutes inside aligned groups of four quads) and the role-0
        // producer, whose consecutive lanes write the pieces (k-block, gg) of ONE pixel, spreads over four bank groups
        const int f = D2T_EXP_SWZ ? (e >> 4) & 3 : 0;
        if (!role) { reinterpret_cast<f32x4*>(rb)[e ^ f] = v; return; }
        // role 1: component lo2 = s of the quads of lanes (tpi*4 + c, gg), c = 0..3
        float* w = rb + (((e & ~63) + (lane & 0x3c)) << 2) + (lane & 3);
        w[(0 ^ f) << 2] = v[0]; w[(1 ^ f) << 2] = v[1]; w[(2 ^ f) << 2] = v[2]; w[(3 ^ f) << 2] = v[3];
    };
    f32x4 gn[S8_NQ];
    auto g_load_all = [&](int ss) {
#pragma unroll
        for (int k = 0; k < S8_NQ; ++k) gn[k] = g_load(k, ss);
    };
    auto g_put_all = [&](float* rb) {
#pragma unroll
        for (int k = 0; k < S8_NQ; ++k) g_put(rb, k, gn[k]);
    };

    // ---- tile stores: lane (pixel n, channels 4g..4g+3 of each c-tile)
    unsigned long long badt = 0;                                     // tiles this lane stored a non-finite value for (bit u mod 64)
    const int x_lane = ((cw + 4 * g) * HW + (n >> 2) * W + j0 + (n & 3)) * 4;
    const bool col_ok = j0 + (n & 3) < W;
    auto store_tile = [&](const f32x4 (&d)[S8_CT], int u) {
        if (u < 0 || u >= tiles_i) return;                           // wave-uniform
        const int i = 4 * u + (n >> 2);
        const int base = col_ok && i < H ? x_lane + 4 * u * W * 4 : S8_OOR;
        bool bad = false;
#pragma unroll
        for (int ct = 0; ct < S8_CT; ++ct) bad = bad || (on[ct] && nonfinite4(d[ct]));
        badt |= bad && base != S8_OOR ? 1ull << (u & 63) : 0ull;
#pragma unroll
        for (int ct = 0; ct < S8_CT; ++ct) {
            if (FLEX && !on[ct]) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = d[ct][r];                            // (a bit_cast of the element lvalue d[ct][r] reads element 0)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rx,
                                                      base == S8_OOR ? S8_OOR : base + ct * ct_step + r * HW * 4, 0, 0);
            }
        }
    };

    f32x4 acc[S8_CT][NACT], a4[KB_SS][S8_CT], done[S8_CT];
#pragma unroll
    for (int ct = 0; ct < S8_CT; ++ct) {
        done[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < NACT; ++a) acc[ct][a] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- prologue: ring[0] <- G(0) (first: load -> LDS -> barrier -> fragment is the longest chain), S pieces of
    // super-step 0, registers <- G(1).  (Round 4, measured and dropped: the S pieces issued in front of the descriptor
    // arithmetic, so that their HBM round trip runs under it -- 70.5 us either way in tools/ab.sh, and 6 more spills.)
    g_load_all(0);
#pragma unroll
    for (int q = 0; q < KB_SS; ++q)
#pragma unroll
        for (int ct = 0; ct < S8_CT; ++ct) a4[q][ct] = !FLEX || on[ct] ? s_load(0, q, ct) : f32x4{0.f, 0.f, 0.f, 0.f};
    g_put_all(ring[0]);
    g_load_all(1);
    lds_barrier();

    const f32x4* lane_ring = reinterpret_cast<const f32x4*>(&ring[0][0]) + (D2T_EXP_SWZ ? lane ^ (g & 3) : lane);
    f32x4 bvP[NACT], bvQ[NACT];
    auto b_fetch = [&](f32x4 (&bv)[NACT], int buf, int q, auto lo_c, auto hi_c) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
#pragma unroll
        for (int a = LO; a < HI; ++a) bv[a] = lane_ring[buf * (S8_RING / 4) + (q * NACT + a) * 64];
    };

    // One super-step.  bvC holds the G fragments of its k-block 0 on entry, bvN is free; on exit the roles are
    // swapped (bvN holds k-block 0 of the next super-step).  LO/HI: live accumulators [LO, HI) -- the first two
    // super-steps carry tiles -2/-1 in acc[0..1], the last two carry tiles past the map in acc[3..4]; NLO/NHI:
    // the live range of the NEXT super-step (for the fragments fetched behind the barrier).
    auto super_step = [&](int ss, f32x4 (&bvC)[NACT], f32x4 (&bvN)[NACT], auto lo_c, auto hi_c, auto nlo_c, auto nhi_c) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        const int cur = ss & 1;
        auto mfma = [&](const f32x4 (&bv)[NACT], int q, int s_lo, int s_hi) {
            // the last super-step of a map whose height is not a multiple of 4: slot rows >= H carry G = 0
            if (ROWKB && HI == NACT - 2 && q < 4 && 4 * ss + q >= H) return;     // wave-uniform
            // One k-block of a tile's 25 is structurally zero: for role 0 the one on the row below its oldest live
            // tile's window, for role 1 the one on the row above its newest tile's window.
            const int A0 = (role == 0 && q == DEAD0 && LO == 0) ? 1 : LO;
            const int A1 = (role == 1 && q == 0 && HI == NACT) ? NACT - 1 : HI;
            if (!FLEX) {
#pragma unroll
                for (int s = s_lo; s < s_hi; ++s)
#pragma unroll
                    for (int a = A0; a < A1; ++a)
#pragma unroll
                        for (int ct = 0; ct < S8_CT; ++ct) acc[ct][a] = D2T_MFMA(a4[q][ct][s], bv[a][s], acc[ct][a]);
            } else {
#pragma unroll
                for (int ct = 0; ct < S8_CT; ++ct) {
                    if (!on[ct]) continue;                           // wave-uniform
#pragma unroll
                    for (int s = s_lo; s < s_hi; ++s)
#pragma unroll
                        for (int a = A0; a < A1; ++a) acc[ct][a] = D2T_MFMA(a4[q][ct][s], bv[a][s], acc[ct][a]);
                }
            }
        };
        auto kblock = [&](f32x4 (&bv)[NACT], f32x4 (&bvn)[NACT], auto q_c) {
            constexpr int q = decltype(q_c)::value;
            mfma(bv, q, 0, 1);
            D2T_PIN();
            if (q + 1 < KB_SS) b_fetch(bvn, cur, q + 1, lo_c, hi_c);  // next k-block's G fragments
            D2T_PIN();
            mfma(bv, q, 1, 2);
            D2T_PIN();
            // The super-step's non-MFMA work: tile store (k-block 0), ring writes of G(ss+1) (1), requests for G(ss+2) (2).
            // (Measured and dropped: placing it behind the 40th MFMA for waves 4-7 so that the two waves of a SIMD never
            // sit in it together -- no change, 72.1 us either way: the cost of these memory instructions is their wait
            // for the texture-address unit, lab/csrc/ta_lab, not the coincidence.)
            auto slice = [&]() {
                if (q <= 2) D2T_WCLK(w_a);
                if (q == 0) {                                        // complete since the end of the previous super-step
                    if (!(ABL & 2)) store_tile(done, ss - 3);
                    else asm volatile("" ::"v"(done[0]), "v"(done[S8_CT - 1]));
                }
                if (q == 1 && !(ABL & 8)) g_put_all(ring[cur ^ 1]);  // G(ss+1), requested a super-step ago; that buffer was last read in ss-1
                if (q == 2 && !(ABL & 4)) g_load_all(ss + 2);        // past the map: out of range, zeros
                if (q <= 2) { D2T_WCLK(w_b); D2T_LAB_ONLY(w_sl[q <= 2 ? q : 0] += w_b - w_a;) }
            };
            slice();
            D2T_PIN();
            mfma(bv, q, 2, 4);
            D2T_PIN();
#pragma unroll
            for (int ct = 0; ct < S8_CT; ++ct)
                if (!(ABL & 1) && (!FLEX || on[ct])) a4[q][ct] = s_load(ss + 1, q, ct);   // a whole super-step ahead
            if (q == KB_SS - 2) {
                // every wave has issued (and, lgkmcnt(0), received) its last fragments of ring[cur] and written its
                // part of ring[cur^1]: publish.  The k-block behind the barrier runs from registers.
                D2T_WCLK(w_a);
                lds_barrier();
                D2T_WCLK(w_b);
                D2T_LAB_ONLY(w_bar += w_b - w_a;)
                b_fetch(bv, cur ^ 1, 0, nlo_c, nhi_c);          // bv is free: its last MFMA has been issued
            }
            D2T_PIN();
        };
        kblock(bvC, bvN, I0{});
        kblock(bvN, bvC, I1{});
        kblock(bvC, bvN, I2{});
        kblock(bvN, bvC, I3{});                                      // ends with the barrier; refills bvN with (ss+1, k-block 0)
        kblock(bvC, bvN, I4{});
        // tile ss-2 is complete: keep it for the store in the next super-step, rotate
        D2T_WCLK(w_a);
#pragma unroll
        for (int ct = 0; ct < S8_CT; ++ct) {
            done[ct] = acc[ct][0];
#pragma unroll
            for (int a = 0; a + 1 < NACT; ++a) acc[ct][a] = acc[ct][a + 1];
            acc[ct][NACT - 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        D2T_PIN();
        D2T_WCLK(w_b);
        D2T_LAB_ONLY(w_rot += w_b - w_a;)
    };

    b_fetch(bvP, 0, 0, I2{}, I5{});
#if D2T_EXP_PRIO
    // the second-dispatched half of the workgroup loses the arbitration for the matrix pipe and the vector issue on every
    // super-step (waves 0-3 wait 3.6 k cycles per super-step at the barrier, waves 4-7 0.3 k): static priority for it.
    // Measured (round 4, tools/ab.sh, profiles/r04_d_ab_bwd_setprio.txt): 70.9 us against 70.2 without -- off.
    if (wave >= S8_WAVES / 2) __builtin_amdgcn_s_setprio(1);
#endif
    D2T_WCLK(w_t1); D2T_WRT(w_r1);
    // tiles_i >= 5 (host-checked): two leading, tiles_i - 4 full, two trailing super-steps
    super_step(0, bvP, bvQ, I2{}, I5{}, I1{}, I5{});
    {
        super_step(1, bvQ, bvP, I1{}, I5{}, I0{}, I5{});
        int ss = 2;
        const int last_full = tiles_i - 3;                           // full super-steps 2 .. tiles_i-3
        for (; ss + 1 < last_full; ss += 2) {
            super_step(ss, bvP, bvQ, I0{}, I5{}, I0{}, I5{});
            super_step(ss + 1, bvQ, bvP, I0{}, I5{}, I0{}, I5{});
        }
        if (ss < last_full) {                                        // two full ones left: ss, ss+1 = last_full
            super_step(ss, bvP, bvQ, I0{}, I5{}, I0{}, I5{});
            super_step(ss + 1, bvQ, bvP, I0{}, I5{}, I0{}, I4{});
            super_step(ss + 2, bvP, bvQ, I0{}, I4{}, I0{}, I3{});
            super_step(ss + 3, bvQ, bvP, I0{}, I3{}, I0{}, I3{});
        } else {                                                     // one full one left: ss = last_full
            super_step(ss, bvP, bvQ, I0{}, I5{}, I0{}, I4{});
            super_step(ss + 1, bvQ, bvP, I0{}, I4{}, I0{}, I3{});
            super_step(ss + 2, bvP, bvQ, I0{}, I3{}, I0{}, I3{});
        }
    }
    D2T_WCLK(w_t2); D2T_WRT(w_r2);
    store_tile(done, tiles_i - 3);
    {
        f32x4 t0[S8_CT], t1[S8_CT];
#pragma unroll
        for (int ct = 0; ct < S8_CT; ++ct) { t0[ct] = acc[ct][0]; t1[ct] = acc[ct][1]; }
        store_tile(t0, tiles_i - 2);                                 // their remaining super-steps lie below the map
        store_tile(t1, tiles_i - 1);
    }

#ifdef D2T_LAB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    D2T_WCLK(w_t3);
    if (lane == 0) {
        unsigned long long* o = lab8_stamps + ((size_t)(blockIdx.x + gridDim.x * blockIdx.y) * S8_WAVES + wave) * 16;
        o[0] = w_t0; o[1] = w_t1; o[2] = w_t2; o[3] = w_t3; o[4] = w_r1; o[5] = w_r2; o[6] = w_sl[0]; o[7] = w_sl[1]; o[8] = w_sl[2];
        o[9] = w_bar; o[10] = w_rot;
    }
#endif
    if (__builtin_expect(__any(badt != 0), 0)) {                     // cold: non-finite inputs only
        unsigned lo = (unsigned)badt, hi = (unsigned)(badt >> 32);
#pragma unroll
        for (int off = 32; off; off >>= 1) { lo |= __shfl_xor(lo, off, 64); hi |= __shfl_xor(hi, off, 64); }
        const unsigned long long m = ((unsigned long long)hi << 32) | lo;
        for (int ct = 0; ct < S8_CT; ++ct) {
            if (FLEX && !on[ct]) continue;
            if (tiles_i > 64) {
                strip_repair(role, lane, gb, S, gx, cw + (FLEX ? 16 * S8_WAVES : 16) * ct, C, H, W, j0, CELLS, 1, 0, H);
            } else {
                for (int u = 0; u < tiles_i; ++u)
                    if ((m >> u) & 1)
                        strip_repair(role, lane, gb, S, gx, cw + (FLEX ? 16 * S8_WAVES : 16) * ct, C, H, W, j0, CELLS, 1, 4 * u, 4 * u + 4 < H ? 4 * u + 4 : H);
            }
        }
    }
}
#undef D2T_PIN

template <int CT, bool ROWKB, int ABL = 0, bool FLEX = false>
__global__ void __launch_bounds__(1024 / CT)
k_corr_bwd_strip8(const float* __restrict__ gout, const float* __restrict__ fm0, const float* __restrict__ fm1,
                  float* __restrict__ g0, float* __restrict__ g1,
                  int B, int C, int H, int W, int tiles_i, int tiles_j)
{
    __shared__ __attribute__((aligned(16))) float ring[2][S8_RING];  // 64 KB
    // Workgroups go to the 8 XCDs round-robin in launch order (x fastest); an XCD should hold workgroups that read the
    // same lines: the strips of one (batch item, role, channel block) share the rows of S (each line is fetched by 5-6
    // neighbouring strips), the two roles of a batch item share gradOut[b].  Logical order: strip, channel block, role,
    // batch item -- with one channel block (the metric shape) an XCD holds one batch item, both roles; with several
    // (small grids) consecutive runs of strips, which launched as (strip, role, batch) x channel block would be dealt over
    // all eight L2s (measured for the one-block case: 133.8 us without the map against 71.7).
    const int nb = gridDim.y, lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * nb);
    const int tj = lid % tiles_j, yb = (lid / tiles_j) % nb, role = (lid / (tiles_j * nb)) & 1, b = lid / (2 * tiles_j * nb);
    const int Ct = (C + 15) / 16;                                    // FLEX: gridDim.y blocks of consecutive c-tiles, sizes differ by at most one
    const int ct0 = FLEX ? (int)((long long)yb * Ct / nb) : 16 * yb;
    const int nct = FLEX ? (int)((long long)(yb + 1) * Ct / nb) - ct0 : 16;
    if ((S8_ABL & 32) && role) return;
    if ((S8_ABL & 64) && !role) return;
    if (role) strip8_body<1, CT, ROWKB, ABL, FLEX>(ring, gout, fm0, fm1, g0, g1, b, tj, C, H, W, tiles_i, ct0, nct);
    else strip8_body<0, CT, ROWKB, ABL, FLEX>(ring, gout, fm0, fm1, g0, g1, b, tj, C, H, W, tiles_i, ct0, nct);
}

}  // namespace

bool corr_bwd8_supported(int B, int C, int H, int W, int ps, int cs)
{
    if (ps != CELLS || cs != 1 || B < 1 || C < 1 || W < WC) return false;
    const int tiles_i = (H + TP - 1) / TP;
    if (tiles_i < 5 || tiles_i > 250) return false;                  // super-step numbers are packed into 8 bits
    // 32-bit byte offsets inside one batch item: feature planes (incl. the c-tile behind C and the rows a
    // prefetch reaches past the map) and gradOut
    const bool fits = (C + 64LL) * H * W * 4 + 64LL * W < 0x7ffffff0LL && (1LL * H * W + 8LL * W) * CELLS * 4 < 0x7ffffff0LL;
    return fits && 2LL * B * ((W + TP - 1) / TP) * ((C + S8_CH - 1) / S8_CH) <= 0x7fffffffLL;
}

#ifdef D2T_LAB
template <int ABL>
void lab8_launch(const float* gout, const float* fm0, const float* fm1, float* g0, float* g1, int B, int C, int H, int W)
{
    const int tiles_i = (H + TP - 1) / TP, tiles_j = (W + TP - 1) / TP;
    hipLaunchKernelGGL((k_corr_bwd_strip8<2, true, ABL>), dim3(2 * B * tiles_j, (C + S8_CH - 1) / S8_CH), dim3(512), 0, 0,
                       gout, fm0, fm1, g0, g1, B, C, H, W, tiles_i, tiles_j);
}
#endif

// How many channel blocks per strip?  Model of a workgroup's time: a fixed part (prologue, G production, tails) plus the
// c-tiles of its busiest SIMD; of the launch: that, times the rounds of workgroups (one per CU).  Measured anchor: the
// metric shape, 16 c-tiles per workgroup (4 per SIMD) = 72 us.
int corr_bwd8_blocks(int B, int C, int W)
{
    const int Ct = (C + 15) / 16, strips = 2 * B * ((W + TP - 1) / TP);
    int best = (Ct + 15) / 16;
    double best_t = 1e30;
    for (int nb = (Ct + 15) / 16; nb <= Ct && nb <= 64; ++nb) {
        const int n = (Ct + nb - 1) / nb, per_simd = (n + 3) / 4;
        const long long wgs = 1LL * strips * nb;
        const double t = (double)((wgs + 255) / 256) * (12.0 + 15.0 * per_simd);
        if (t < best_t - 1e-9) { best_t = t; best = nb; }
    }
    return best;
}

int corr_bwd8_f32(const float* gout, const float* fm0, const float* fm1, float* g0, float* g1,
                  int B, int C, int H, int W, hipStream_t st, int variant)
{
    const int tiles_i = (H + TP - 1) / TP, tiles_j = (W + TP - 1) / TP;
    const int nb = corr_bwd8_blocks(B, C, W);
    if (variant == 1) {                                              // the row-major k-block enumeration (A/B measurements)
        hipLaunchKernelGGL((k_corr_bwd_strip8<2, false>), dim3(2 * B * tiles_j, (C + S8_CH - 1) / S8_CH), dim3(512), 0, st,
                           gout, fm0, fm1, g0, g1, B, C, H, W, tiles_i, tiles_j);
    } else if (C % S8_CH == 0 && nb == C / S8_CH) {                   // whole 256-channel blocks (the metric shape): no branches
        hipLaunchKernelGGL((k_corr_bwd_strip8<2, true, (S8_ABL & 15)>), dim3(2 * B * tiles_j, nb), dim3(512), 0, st,
                           gout, fm0, fm1, g0, g1, B, C, H, W, tiles_i, tiles_j);
    } else {
        hipLaunchKernelGGL((k_corr_bwd_strip8<2, true, 0, true>), dim3(2 * B * tiles_j, nb), dim3(512), 0, st,
                           gout, fm0, fm1, g0, g1, B, C, H, W, tiles_i, tiles_j);
    }
    return launch_status();
}

}}  // namespace d2t::tuned
