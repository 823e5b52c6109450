// d2t_corr_bwd8w.hip -- gfx950 f32 PointwiseCorrelation backward, 8-wave kernel on strips EIGHT pixels wide
// (d_max = 8, stride 1, reference layout).  Gather form of pointwise_correlation_cuda.cu:145-171, as in
// d2t_corr_bwd8.hip:
//     gX[c][t] = sum over window slots w of  G[t][w] * S[c][w]
// role 0 (gradFM0): t = centre pixels, S = FM1;  role 1 (gradFM1): t = displaced pixels, S = FM0.
//
// Why wider strips.  The 4-pixel strip kernel is co-limited by the matrix pipe and by the request rate between a
// CU's vector L1 and the L2 (lab/csrc/ta_lab + PMC, DESIGN 5: about 2.8 cycles per 128-byte request and CU; the
// 4-pixel kernel makes 10.6 M of them per launch, 0.65 of what the chip can serve in its 70 us).  Most are the
// feature (S) stream: a 4-pixel strip needs an 80-byte window row per channel, fetched as a 64-byte piece (1.47
// lines at 4-byte alignment) plus a 16-byte piece (1.09 lines) = 2.56 requests per (channel, row) for 4 pixels.
// Here a workgroup owns TWO 4 x 4 tile columns (A: strip columns 0-3, B: 4-7) and 128 channels.  The window row
// is 96 bytes = column groups g0 .. g5; A needs g0-g4, B needs g1-g5:
//   k-blocks 0-3   row q of the super-step, groups g1-g4 (64 contiguous bytes per channel): ONE S fragment feeds
//                  the MFMAs of both tile columns -- every slot is inside both windows' union, no padding added;
//   k-block  4     group g0 of the four rows: tile column A only;   k-block 5: group g5: tile column B only.
// Same MFMA count per tile as the 4-pixel kernel (5 k-blocks of 16 slots per super-step), 1.47 + 2 x 1.09 = 3.65
// requests per (channel, row) for 8 pixels (-29 %), and the tile stores / role-1 gradOut pieces of the two columns
// sit next to each other in memory.  The price: the G ring of a workgroup doubles (2 x 50 fragment blocks of 1 KB)
// while it serves half the channels, so G production per CU doubles; its requests are kept down by fetching in
// memory order (role 0: the 24 pieces of a pixel are 68 consecutive floats; role 1: the pieces of columns A and B
// for one (centre pixel, cell row) are 32 consecutive bytes).
// The strip window is NOT clamped to the map: slot columns left of column 0 / right of column W-1 exist in the
// enumeration with G = 0 (their S operand is whatever lies there -- the neighbouring row, or the range check's
// zeros); a non-finite value there is caught by the same repair path as any other (d2t_corr_common.hpp).  A piece
// that straddles the END of the last channel's last row keeps its in-range dwords (per-dword range check).
#include "../../detect-to-track_amd/csrc/d2t_corr_common.hpp"
#include <type_traits>

namespace d2t { namespace tuned {

namespace {

#ifndef W8_LOOP
#define W8_LOOP 0
#endif
#ifndef W8_ABL
#define W8_ABL 0      // timing ablations (lab builds only, results are wrong): 1 no G loads, 2 no ring writes, 4 no tile stores, 8 no S reloads, 16 no fragment reads
#endif
constexpr int W8_CH = 128;                          // channels per workgroup: 8 waves x 1 c-tile
constexpr int W8_KB = 6;                            // k-blocks per super-step (see above)
constexpr int W8_FB = 4 * 2 * NACT + 2 * NACT;      // 50 fragment blocks (64 quads = 1 KB each) per super-step
constexpr int W8_QUADS = W8_FB * 64;                // 3200 ring quads per super-step
constexpr int W8_T = 512;                           // threads
constexpr int W8_NQ = 7;                            // production slots per thread (3584 >= 3200; the rest are empty)
constexpr int W8_RING = W8_NQ * W8_T * 4;           // floats per ring buffer (57,344 bytes)
constexpr int W8_LDS = 2 * W8_RING * 4;             // 114,688 bytes
constexpr int W8_OOR = 0x7ffffff0;                  // byte offset that every buffer range check rejects

struct Quad8w { int off, info; };

// Inside its 64-quad fragment block the quad of lane l = (gg, pixel) of every k-block sits at position l ^ (gg & 3): the
// reader's ds_read_b128 stays conflict-free (an XOR below 4 permutes inside aligned groups of four quads), while the
// producer of role 0, whose consecutive lanes write the consecutive pieces (q, gg) of ONE pixel, spreads over four bank
// groups instead of hitting one eight times (SQ_LDS_BANK_CONFLICT was 53 % of SQ_LDS_IDX_ACTIVE without it).
__host__ __device__ constexpr int w8_swz(int gg) { return gg & 3; }

// fragment block of (k-block q, tile column t, live tile a); q = 4 exists for t = 0 only, q = 5 for t = 1 only
__host__ __device__ constexpr int w8_block(int q, int t, int a) { return q < 4 ? (q * 2 + t) * NACT + a : 8 * NACT + t * NACT + a; }

// Which 4 window slots lane group gg of k-block q multiplies: map row 4 ss + xr, column group cg (columns col0 + 4 cg ..)
__device__ __forceinline__ void w8_slot(int q, int gg, int& xr, int& cg)
{
    xr = q < 4 ? q : gg;
    cg = q < 4 ? gg + 1 : (q == 4 ? 0 : 5);
}

// One ring quad as a 16-byte run of gradOut (see Quad8 in d2t_corr_bwd8.hip): ring index e -> (block, lane) ->
// (k-block q, tile column t, live tile a, lane l).  Component c is cell cj + c of row ci of one centre pixel.
__device__ __forceinline__ Quad8w quad8w_desc(int role, int e, int H, int W, int tiles_i, int j0, int col0)
{
    const int blk = e >> 6, l = e & 63, gg = l >> 4;
    int q, t, a;
    if (blk < 8 * NACT) { const int qt = blk / NACT; a = blk - qt * NACT; q = qt >> 1; t = qt & 1; }
    else { const int r = blk - 8 * NACT; t = r >= NACT ? 1 : 0; a = r - t * NACT; q = 4 + t; }
    int xr, cg;
    w8_slot(q, gg, xr, cg);
    const int tpi = (l >> 2) & 3, lo2 = l & 3;                             // role 0: lo2 = tile column; role 1: slot column s
    const int ci = role ? 4 * a + tpi - xr : xr - 4 * a - tpi + 2 * DT;    // displaced - centre + d
    const int jt = j0 + 4 * t;                                             // first pixel column of this tile column
    const int tj = jt + lo2, sj = col0 + 4 * cg + (role ? lo2 : 0);
    const int cj = role ? jt - sj + DT : sj - tj + DT;                     // component c reads cell cj + c
    int mask = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        // role 0: tile pixel tj fixed, displaced column sj + c must lie in the map; role 1: centre column sj fixed and in
        // the map, displaced (tile) column jt + c must lie in the map
        const bool ok = role ? (jt + c < W && sj >= 0 && sj < W) : (tj < W && sj + c >= 0 && sj + c < W);
        mask |= (cj + c >= 0 && cj + c < 2 * DT && ok) ? (1 << c) : 0;
    }
    if (ci < 0 || ci >= 2 * DT || e >= W8_QUADS) mask = 0;
    const int pix0 = role ? xr * W + sj : (4 * (a - 2) + tpi) * W + tj;    // centre pixel at ss = 0
    int lo = 2 - a, hi = tiles_i + 2 - a;
    const int hi_t = (H - tpi + 3) / 4 + 2 - a;                            // 4(ss-2+a)+tpi < H
    const int hi_r = (H - xr + 3) / 4;                                     // 4ss+xr < H
    hi = hi < hi_t ? hi : hi_t;
    hi = hi < hi_r ? hi : hi_r;
    lo = lo < 0 ? 0 : lo;
    if (!mask || hi < lo) { lo = 0; hi = 0; }
    Quad8w d;
    d.off = (pix0 * CELLS + ci * CW + cj) * 4;
    d.info = lo | (hi << 8) | (mask << 16);
    return d;
}

__device__ __forceinline__ f32x4 quad8w_fix(const f32x4& v, int info)
{
    const u32x4 b = __builtin_bit_cast(u32x4, v);
    u32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = b[c] & (unsigned)__builtin_amdgcn_sbfe(info, 16 + c, 1);
    return __builtin_bit_cast(f32x4, o);
}

typedef std::integral_constant<int, 0> J0;
typedef std::integral_constant<int, 1> J1;
typedef std::integral_constant<int, 2> J2;
typedef std::integral_constant<int, 3> J3;
typedef std::integral_constant<int, 4> J4;
typedef std::integral_constant<int, 5> J5;

#define D2T_PIN() __builtin_amdgcn_sched_barrier(0)

template <int role>
__device__ __forceinline__ void strip8w_body(float* __restrict__ ring, const float* __restrict__ gout,
                                             const float* __restrict__ fm0, const float* __restrict__ fm1,
                                             float* __restrict__ g0, float* __restrict__ g1,
                                             int b, int tj8, int yb, int C, int H, int W, int tiles_i)
{
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j0 = tj8 * 2 * TP, HW = H * W;
    // First window column, NOT clamped to the map, the same for both roles: role 0 needs displaced columns j0-8 .. j0+14,
    // role 1 centre columns j0-7 .. j0+15 -- both inside the 24 columns from j0-8, and with j0 a multiple of 8 no 16-byte
    // piece straddles column 0 (a piece that starts in front of the buffer is dropped WHOLE by the range check -- its
    // dwords are checked one by one without wrapping -- so a straddling piece would lose the map's column 0).
    const int col0 = j0 - DT;
    const float* S = (role ? fm0 : fm1) + (size_t)b * C * HW;
    float* gx = (role ? g1 : g0) + (size_t)b * C * HW;
    const float* gb = gout + (size_t)b * HW * CELLS;
    const unsigned plane_bytes = (unsigned)C * HW * 4u;
    const __amdgpu_buffer_rsrc_t rs = uniform_rsrc(S, plane_bytes);
    const __amdgpu_buffer_rsrc_t rx = uniform_rsrc(gx, plane_bytes);
    const __amdgpu_buffer_rsrc_t rg = uniform_rsrc(gb, (unsigned)HW * CELLS * 4u);

    const int cw = yb * W8_CH + wave * 16;                           // first channel of this wave's c-tile
    // S piece of k-block q at super-step 0 (bytes): lane (channel n, lane group g).  Channels >= C lie behind the
    // buffer: zeros.  A negative offset (channel 0, columns left of the map in row 0) wraps to a huge unsigned one: zeros.
    int sv[W8_KB];
#pragma unroll
    for (int q = 0; q < W8_KB; ++q) {
        int xr, cg;
        w8_slot(q, g, xr, cg);
        sv[q] = ((cw + n) * HW + xr * W + col0 + 4 * cg) * 4;
    }
    const int s_step = 4 * W * 4;
    auto s_load = [&](int ss, int q) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, sv[q] + ss * s_step, 0, 0));
    };

    // ---- G production: slot ep = tid + 512 k fills ring quad ring_index(ep), in MEMORY order (see the file comment)
    auto ring_index = [&](int ep) -> int {
        if (ep >= W8_QUADS) return ep;
        if (role == 0) {
            // ep = pixel * 20 + piece; pixel = (t, a, 4 x 4 tile pixel), piece = (window row xr, k = 0..4 -> column group k + t)
            const int p = (ep * 3277) >> 16, m = ep - p * 20;        // ep / 20 (ep < 3200)
            const int ta = p >> 4, pix = p & 15, t = ta >= NACT ? 1 : 0, a = ta - t * NACT;
            const int xr = (m * 13) >> 6, cg = m - xr * 5 + t;
            const int q = cg == 0 ? 4 : (cg == 5 ? 5 : xr), gg = (cg == 0 || cg == 5) ? xr : cg - 1;
            return w8_block(q, t, a) * 64 + gg * 16 + pix;
        } else {
            // ep = 2 w + t: the pieces of tile columns A and B for one (centre pixel, cell row) are 32 consecutive bytes
            const int t = ep & 1, w = ep >> 1, l = w & 63, qa = w >> 6, qq = (qa * 13) >> 6, a = qa - qq * NACT;   // qa / 5 (qa < 25)
            const int q = qq < 4 ? qq : 4 + t;
            return w8_block(q, t, a) * 64 + l;
        }
    };
    Quad8w qd[W8_NQ];                                                // .info also carries the ring quad index (bits 20-31)
#pragma unroll
    for (int k = 0; k < W8_NQ; ++k) {
        const int e = ring_index(tid + k * W8_T);
        qd[k] = quad8w_desc(role, e, H, W, tiles_i, j0, col0);
        qd[k].info |= e << 20;
    }
    const int g_step = 4 * W * CELLS * 4;                            // gradOut bytes per 4 map rows
    auto g_load = [&](int k, int ss) -> f32x4 {
        const int lo = qd[k].info & 255, hi = (qd[k].info >> 8) & 255;
        const int v = ss >= lo && ss < hi ? qd[k].off + ss * g_step : W8_OOR;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, v, 0, 0));
    };
    auto g_put = [&](float* rb, int k, const f32x4& raw) {
        const f32x4 v = quad8w_fix(raw, qd[k].info);
        const int e = (unsigned)qd[k].info >> 20, f = w8_swz(e >> 4); // logical quad, the swizzle of its lane group
        if (!role) { reinterpret_cast<f32x4*>(rb)[e ^ f] = v; return; }
        // role 1: component s = e & 3 of the quads of lanes (tpi * 4 + c, gg), c = 0..3 (physical slot: ^ f)
        float* w = rb + ((e & ~3) << 2) + (e & 3);
        w[(0 ^ f) << 2] = v[0]; w[(1 ^ f) << 2] = v[1]; w[(2 ^ f) << 2] = v[2]; w[(3 ^ f) << 2] = v[3];
    };
    f32x4 gn[W8_NQ];
    auto g_load_all = [&](int ss) {
#pragma unroll
        for (int k = 0; k < W8_NQ; ++k) gn[k] = g_load(k, ss);
    };
    auto g_put_all = [&](float* rb) {
#pragma unroll
        for (int k = 0; k < W8_NQ; ++k) g_put(rb, k, gn[k]);
    };

    // ---- tile stores: lane (pixel n, channels 4g..4g+3), two tile columns
    unsigned long long badt[2] = {0, 0};                             // tiles this lane stored a non-finite value for (bit u mod 64)
    const int x_lane = ((cw + 4 * g) * HW + (n >> 2) * W + j0 + (n & 3)) * 4;
    auto store_tile = [&](const f32x4 (&d)[2], int u) {
        if (u < 0 || u >= tiles_i) return;                           // wave-uniform
        const int i = 4 * u + (n >> 2);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bool col_ok = j0 + 4 * t + (n & 3) < W;
            const int base = col_ok && i < H ? x_lane + 4 * u * W * 4 + 16 * t : W8_OOR;
            badt[t] |= nonfinite4(d[t]) && base != W8_OOR ? 1ull << (u & 63) : 0ull;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = d[t][r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rx, base == W8_OOR ? W8_OOR : base + r * HW * 4, 0, 0);
            }
        }
    };

    f32x4 acc[2][NACT], a4[W8_KB], done[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        done[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < NACT; ++a) acc[t][a] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- prologue: ring[0] <- G(0), S pieces of super-step 0, registers <- G(1)
    g_load_all(0);
#pragma unroll
    for (int q = 0; q < W8_KB; ++q) a4[q] = s_load(0, q);
    g_put_all(ring);
    g_load_all(1);
    lds_barrier();

    const f32x4* ring4 = reinterpret_cast<const f32x4*>(ring);
    const int lane_sw = lane ^ w8_swz(g);
    // A super-step is TEN halves of 20 MFMAs: (k-block 0, column A), (0, B), (1, A), ... (3, B), (4, A), (5, B).  A half needs
    // the five fragments (live tiles) of ONE tile column, so the two register sets that swap roles hold 5 quads each (with whole
    // k-blocks -- both columns at once -- they held 10 each and the kernel sat at the 256-register limit, spilling around
    // the steady state).  Ten halves: the sets end a super-step in the roles they started it with.
    f32x4 bvP[NACT], bvQ[NACT];
    auto half_q = [](int h) { return h < 8 ? h >> 1 : h - 4; };      // k-block of half h
    auto half_t = [](int h) { return h < 8 ? h & 1 : h - 8; };       // tile column of half h
    auto b_fetch = [&](f32x4 (&bv)[NACT], int buf, int h, auto lo_c, auto hi_c) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        const int q = half_q(h), t = half_t(h);
#pragma unroll
        for (int a = LO; a < HI; ++a) bv[a] = ring4[buf * (W8_RING / 4) + w8_block(q, t, a) * 64 + lane_sw];
    };

    auto super_step = [&](int ss, auto lo_c, auto hi_c, auto nlo_c, auto nhi_c) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        const int cur = ss & 1;
        auto mfma = [&](const f32x4 (&bv)[NACT], int q, int t, int s_lo, int s_hi) {
            // the last super-step of a map whose height is not a multiple of 4: slot rows >= H carry G = 0
            if (HI == NACT - 2 && q < 4 && 4 * ss + q >= H) return;  // wave-uniform
            // One k-block of a tile is structurally zero: role 0 the row-3 block of its oldest live tile (that row lies
            // below the tile's window), role 1 the row-0 block of its newest tile (above its window).
            const int A0 = (role == 0 && q == 3 && LO == 0) ? 1 : LO;
            const int A1 = (role == 1 && q == 0 && HI == NACT) ? NACT - 1 : HI;
#pragma unroll
            for (int s = s_lo; s < s_hi; ++s)
#pragma unroll
                for (int a = A0; a < A1; ++a) acc[t][a] = D2T_MFMA(a4[q][s], bv[a][s], acc[t][a]);
        };
        auto half = [&](f32x4 (&bv)[NACT], f32x4 (&bvn)[NACT], auto h_c) {
            constexpr int h = decltype(h_c)::value, q = h < 8 ? h >> 1 : h - 4, t = h < 8 ? h & 1 : h - 8;
            mfma(bv, q, t, 0, 1);
            D2T_PIN();
            if (h + 1 < 10 && !(W8_ABL & 16)) b_fetch(bvn, cur, h + 1, lo_c, hi_c);    // the next half's G fragments
            D2T_PIN();
            mfma(bv, q, t, 1, 2);
            D2T_PIN();
            if (h == 0 && !(W8_ABL & 4)) store_tile(done, ss - 3);   // complete since the end of the previous super-step
            if (h == 0 && (W8_ABL & 4)) asm volatile("" ::"v"(done[0]), "v"(done[1]));
            if (h == 2 && !(W8_ABL & 2)) g_put_all(ring + (cur ^ 1) * W8_RING);       // G(ss+1), requested a super-step ago; that buffer was last read in ss-1
            if (h == 4 && !(W8_ABL & 1)) g_load_all(ss + 2);         // past the map: out of range, zeros
            D2T_PIN();
            mfma(bv, q, t, 2, 4);
            D2T_PIN();
            if ((h >= 8 || t == 1) && !(W8_ABL & 8)) a4[q] = s_load(ss + 1, q);         // this k-block's last MFMA is issued: its piece for the next super-step
            if (h == 8) {
                // every wave has issued (and, lgkmcnt(0), received) its last fragments of ring[cur] and written its part
                // of ring[cur^1]: publish.  The half behind the barrier runs from registers.
                lds_barrier();
                b_fetch(bv, cur ^ 1, 0, nlo_c, nhi_c);               // bv is free: its last MFMA has been issued
            }
            D2T_PIN();
        };
        half(bvP, bvQ, J0{});
        half(bvQ, bvP, J1{});
        half(bvP, bvQ, J2{});
        half(bvQ, bvP, J3{});
        half(bvP, bvQ, J4{});
        half(bvQ, bvP, J5{});
        half(bvP, bvQ, std::integral_constant<int, 6>{});
        half(bvQ, bvP, std::integral_constant<int, 7>{});
        half(bvP, bvQ, std::integral_constant<int, 8>{});           // ends with the barrier; refills bvP with (ss+1, half 0)
        half(bvQ, bvP, std::integral_constant<int, 9>{});
        // tile ss-2 is complete: keep it for the store in the next super-step, rotate
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            done[t] = acc[t][0];
#pragma unroll
            for (int a = 0; a + 1 < NACT; ++a) acc[t][a] = acc[t][a + 1];
            acc[t][NACT - 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        D2T_PIN();
    };

    b_fetch(bvP, 0, 0, J2{}, J5{});
    // tiles_i >= 5 (host-checked): two leading, tiles_i - 4 full, two trailing super-steps.  ONE loop with a wave-uniform
    // switch over the live ranges instead of peeled straight-line code: with the super-step number a run-time value the
    // compiler cannot fold it into the addresses of the leading / trailing super-steps (the peeled form did, kept them all
    // live and spilled 75 registers there: 59 MB written per launch against 39 MB of gradients), and every phase is
    // register-allocated against the same loop header as the steady state, which does not spill.
#if W8_LOOP
    {
        const int last_full = tiles_i - 3;                           // full super-steps 2 .. tiles_i-3
#pragma unroll 1
        for (int ss = 0; ss < tiles_i; ++ss) {
            const int ph = ss < 2 ? ss : (ss < last_full ? 2 : ss - last_full + 3);
            switch (ph) {
                case 0: super_step(ss, J2{}, J5{}, J1{}, J5{}); break;
                case 1: super_step(ss, J1{}, J5{}, J0{}, J5{}); break;
                case 2: super_step(ss, J0{}, J5{}, J0{}, J5{}); break;
                case 3: super_step(ss, J0{}, J5{}, J0{}, J4{}); break;
                case 4: super_step(ss, J0{}, J4{}, J0{}, J3{}); break;
                default: super_step(ss, J0{}, J3{}, J0{}, J3{}); break;
            }
        }
    }
#else
    super_step(0, J2{}, J5{}, J1{}, J5{});
    super_step(1, J1{}, J5{}, J0{}, J5{});
    {
        int ss = 2;
        const int last_full = tiles_i - 3;                           // full super-steps 2 .. tiles_i-3
#pragma unroll 1
        for (; ss < last_full; ++ss) super_step(ss, J0{}, J5{}, J0{}, J5{});
        super_step(ss, J0{}, J5{}, J0{}, J4{});
        super_step(ss + 1, J0{}, J4{}, J0{}, J3{});
        super_step(ss + 2, J0{}, J3{}, J0{}, J3{});
    }
#endif
    store_tile(done, tiles_i - 3);
    {
        f32x4 t0[2], t1[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) { t0[t] = acc[t][0]; t1[t] = acc[t][1]; }
        store_tile(t0, tiles_i - 2);                                 // their remaining super-steps lie below the map
        store_tile(t1, tiles_i - 1);
    }

    if (__builtin_expect(__any((badt[0] | badt[1]) != 0), 0)) {      // cold: non-finite inputs only
#pragma unroll 1
        for (int t = 0; t < 2; ++t) {
            unsigned lo = (unsigned)badt[t], hi = (unsigned)(badt[t] >> 32);
#pragma unroll
            for (int off = 32; off; off >>= 1) { lo |= __shfl_xor(lo, off, 64); hi |= __shfl_xor(hi, off, 64); }
            const unsigned long long m = ((unsigned long long)hi << 32) | lo;
            if (!m) continue;
            if (tiles_i > 64) {
                strip_repair(role, lane, gb, S, gx, cw, C, H, W, j0 + 4 * t, CELLS, 1, 0, H);
            } else {
                for (int u = 0; u < tiles_i; ++u)
                    if ((m >> u) & 1)
                        strip_repair(role, lane, gb, S, gx, cw, C, H, W, j0 + 4 * t, CELLS, 1, 4 * u, 4 * u + 4 < H ? 4 * u + 4 : H);
            }
        }
    }
}
#undef D2T_PIN

__global__ void __launch_bounds__(W8_T)
k_corr_bwd_strip8w(const float* __restrict__ gout, const float* __restrict__ fm0, const float* __restrict__ fm1,
                   float* __restrict__ g0, float* __restrict__ g1,
                   int B, int C, int H, int W, int tiles_i, int tiles_j8)
{
    extern __shared__ __attribute__((aligned(16))) float ring8w[];   // W8_LDS bytes: two ring buffers
    // Logical order (strip, channel block, role, batch item) on the XCD-aware map of the LINEAR block id, as
    // k_corr_bwd_strip8: an XCD holds the strips of one (batch item, role, channel block) next to each other (they share
    // rows of S) and both roles of a batch item (they share gradOut[b]).
    const int nb = gridDim.y, lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * nb);
    const int tj8 = lid % tiles_j8, yb = (lid / tiles_j8) % nb, role = (lid / (tiles_j8 * nb)) & 1, b = lid / (2 * tiles_j8 * nb);
    if ((W8_ABL & 32) && role) return;                               // lab: role 0 alone on the chip
    if ((W8_ABL & 64) && !role) return;                              // lab: role 1 alone
    if (role) strip8w_body<1>(ring8w, gout, fm0, fm1, g0, g1, b, tj8, yb, C, H, W, tiles_i);
    else strip8w_body<0>(ring8w, gout, fm0, fm1, g0, g1, b, tj8, yb, C, H, W, tiles_i);
}

}  // namespace

bool corr_bwd8w_supported(int B, int C, int H, int W, int ps, int cs)
{
    // same envelope as the 4-pixel kernel, but no minimum width (the window is not clamped) beyond one tile column
    if (ps != CELLS || cs != 1 || B < 1 || C < 1 || W < TP) return false;
    const int tiles_i = (H + TP - 1) / TP;
    if (tiles_i < 5 || tiles_i > 250) return false;
    const bool fits = (C + 64LL) * H * W * 4 + 64LL * W < 0x7ffffff0LL && (1LL * H * W + 8LL * W) * CELLS * 4 < 0x7ffffff0LL;
    return fits && 2LL * B * ((W + 2 * TP - 1) / (2 * TP)) * ((C + W8_CH - 1) / W8_CH) <= 0x7fffffffLL;
}

// workgroups the launch would have: the dispatcher prefers this kernel when they fill the chip about once or more
long long corr_bwd8w_workgroups(int B, int C, int W)
{
    return 2LL * B * ((W + 2 * TP - 1) / (2 * TP)) * ((C + W8_CH - 1) / W8_CH);
}

int corr_bwd8w_f32(const float* gout, const float* fm0, const float* fm1, float* g0, float* g1,
                   int B, int C, int H, int W, hipStream_t st)
{
    const int tiles_i = (H + TP - 1) / TP, tiles_j8 = (W + 2 * TP - 1) / (2 * TP);
    D2T_ENSURE_DYNAMIC_LDS(k_corr_bwd_strip8w, W8_LDS);
    hipLaunchKernelGGL(k_corr_bwd_strip8w, dim3(2 * B * tiles_j8, (C + W8_CH - 1) / W8_CH), dim3(W8_T), W8_LDS, st,
                       gout, fm0, fm1, g0, g1, B, C, H, W, tiles_i, tiles_j8);
    return launch_status();
}

}}  // namespace d2t::tuned
