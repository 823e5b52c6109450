// d2t_corr_bwd8bf.hip -- gfx950 PointwiseCorrelation backward on the bf16 matrix pipe, f32 operands split in three
// (d_max = 8, stride 1, reference layout).  Same gather form and strip walk as d2t_corr_bwd8.hip:
//     gX[c][t] = sum over window slots w of  G[t][w] * S[c][w]
// but every f32 operand x is carried as three bf16 pieces, x = hi + mid + lo exactly (8 + 8 + 8 mantissa bits), and a
// product keeps six of the nine piece products (hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi; the dropped ones are below
// 2^-24 of the product), accumulated in f32 by v_mfma_f32_16x16x32_bf16.  Measured in lab/csrc/bf16x3_lab: 24 such MFMAs
// do the K = 128 of 32 v_mfma_f32_16x16x4_f32 in 0.46x the time (wall clock), and the result is as close to the exact sum as the f32
// chain is (<= 1.1e-6 of sum|terms| at K = 2048 against 1.7e-6).  The backward's contract is 1e-5 (the reference adds with
// atomics in no fixed order); this kernel is deterministic like the f32 one.
//
// What changes against d2t_corr_bwd8.hip:
//   * K = 32 per instruction: a lane (channel or pixel n, lane group g) holds EIGHT window slots = two 16-byte pieces of a
//     map row.  Block 0 = rows 0 and 1 of the super-step, block 1 = rows 2 and 3: piece h of lane group g is columns
//     4g .. 4g+3 of row 2q + h, so that ONE load instruction reads 64 contiguous bytes per channel (16-32 cache lines, as
//     the f32 kernel's row-wise k-blocks; with a lane group per row -- the first version of this file -- every load
//     touched 64 lines and the kernel ran 80.6 us against the f32 kernel's 72).  Block 2 collects the fifth column group
//     of the four rows (lane group g = row g) in piece 0; its piece 1 is zero on both sides.  3 blocks x 6 MFMAs per
//     (tile, c-tile) instead of 5 k-blocks x 4.
//   * the G ring holds bf16 pieces: [buffer][piece][block][live tile][lane] 16 bytes = 8 slots; a fragment is three
//     ds_read_b128.  G is split once, when it is written to the ring; S (32 contiguous bytes per lane and block, two
//     16-byte loads a super-step ahead, as before) is split in registers once per block and used by the five live tiles.
//   * three ring buffers and ONE barrier in the middle of a super-step (behind the ring writes of G(ss+1)): a buffer is
//     rewritten two barriers after its last read, so no wave ever waits with an empty matrix queue at the END of a
//     super-step (the f32 kernel keeps two buffers and preloads the last k-block's fragments to hide its barrier).
// Non-finite inputs: a piece of Inf / NaN is NaN, the tile's accumulators become non-finite and the wave repairs its
// region in the reference's form (strip_repair), as in the f32 kernels.  Finite values above the bf16 range (> 3.39e38)
// take that path too.
#include "../../detect-to-track_amd/csrc/d2t_corr_common.hpp"
#include <type_traits>

namespace d2t { namespace tuned {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BF_NB = 3;                            // K = 32 blocks per super-step: 2 x (2 rows x 16 columns) + the 4 rows' last 4 columns
constexpr int BF_CT = 2;                            // c-tiles (16 channels) per wave
constexpr int BF_WAVES = 8, BF_T = BF_WAVES * 64;
constexpr int BF_CH = BF_WAVES * BF_CT * 16;        // 256 channels per workgroup
constexpr int BF_PLANE = BF_NB * NACT * 64 * 16;    // bytes of one piece plane of a ring buffer: [block][tile][lane] x 16
constexpr int BF_BUF = 3 * BF_PLANE;                // 46,080 bytes: hi, mid, lo
constexpr int BF_LDS = 3 * BF_BUF;                  // 138,240 bytes: G(ss-1) (being retired), G(ss), G(ss+1)
constexpr int BF_PROD = 2 * NACT * 4 * 4 * 2 + NACT * 4 * 4;   // 400 producer threads, 4 ring quads each
constexpr int BF_OOR = 0x7ffffff0;                  // byte offset that every buffer range check rejects

struct Piece4 { bf16x4 hi, mid, lo; };
struct Piece8 { bf16x8 hi, mid, lo; };

__device__ __forceinline__ Piece4 split4(const f32x4& x)
{
    Piece4 s;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 h = (__bf16)x[i];                               // round to nearest even
        const float r1 = x[i] - (float)h;                            // exact
        const __bf16 m = (__bf16)r1;
        s.hi[i] = h; s.mid[i] = m; s.lo[i] = (__bf16)(r1 - (float)m);
    }
    return s;
}
__device__ __forceinline__ Piece8 join8(const Piece4& a, const Piece4& b)
{
    Piece8 s;
    s.hi = __builtin_shufflevector(a.hi, b.hi, 0, 1, 2, 3, 4, 5, 6, 7);
    s.mid = __builtin_shufflevector(a.mid, b.mid, 0, 1, 2, 3, 4, 5, 6, 7);
    s.lo = __builtin_shufflevector(a.lo, b.lo, 0, 1, 2, 3, 4, 5, 6, 7);
    return s;
}

#define D2T_BFMFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define D2T_BFMFMA_A(a, b, c) ((ABL & 16) ? bf_nomfma((a), (b), (c)) : D2T_BFMFMA((a), (b), (c)))

__device__ __forceinline__ f32x4 bf_nomfma(const bf16x8& a, const bf16x8& b, f32x4 c) { asm volatile("" : "+v"(c) : "v"(a), "v"(b)); return c; }

struct QuadBf { int off, info; };                                   // as Quad8 of d2t_corr_bwd8.hip: lo | hi << 8 | mask << 16

// Producer thread e (< BF_PROD) fills ONE 8-byte piece position -- (block q, live tile a, lane group g, tile pixel row
// tpi, piece h) -- of the four lanes (tile columns 0..3) that share it:
//   e < 320:  e = ((q * 5 + a) * 4 + g) * 4 + tpi) * 2 + h,  q = 0, 1:  map row xr = 2q + h, window columns 4g .. 4g+3
//   e >= 320: e - 320 = (a * 4 + g) * 4 + tpi,               q = 2, h = 0: map row xr = g, window columns 16 .. 19
__device__ __forceinline__ void prod_decode(int e, int& q, int& a, int& g, int& tpi, int& h)
{
    if (e < 320) { h = e & 1; tpi = (e >> 1) & 3; g = (e >> 3) & 3; const int qa = e >> 5; q = qa / NACT; a = qa - q * NACT; }
    else { const int f = e - 320; tpi = f & 3; g = (f >> 2) & 3; a = (f >> 4) < NACT ? (f >> 4) : 0; q = 2; h = 0; }
}
// Its ring quad `k` (0..3) -- one 16-byte run of gradOut:
//   role 0: 4 consecutive window columns of tile pixel (tpi, column k): component c = window column c of the piece;
//   role 1: 4 consecutive tile pixels (columns 0..3) for window column k of the piece (= one centre pixel): component c =
//           tile column c (transposed into the ring by g_put).
__device__ __forceinline__ QuadBf quadbf_desc(int role, int e, int k, int H, int W, int tiles_i, int j0, int col0)
{
    int q, a, g, tpi, h;
    prod_decode(e < BF_PROD ? e : 0, q, a, g, tpi, h);
    const int xr = q < 2 ? 2 * q + h : g, c0 = q < 2 ? 4 * g : 16;        // map row of the super-step, first window column
    const int ci = role ? 4 * a + tpi - xr : xr - 4 * a - tpi + 2 * DT;    // displaced - centre + d (constant over super-steps)
    const int tj = j0 + k, sj = col0 + c0 + (role ? k : 0);
    const int cj = role ? j0 - sj + DT : sj - tj + DT;                     // component c reads cell cj + c
    int mask = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bool col_ok = role ? j0 + c < W : tj < W;
        mask |= (cj + c >= 0 && cj + c < 2 * DT && col_ok) ? (1 << c) : 0;
    }
    if (ci < 0 || ci >= 2 * DT || e >= BF_PROD) mask = 0;
    const int pix0 = role ? xr * W + sj : (4 * (a - 2) + tpi) * W + tj;    // centre pixel at ss = 0
    int lo = 2 - a, hi = tiles_i + 2 - a;
    const int hi_t = (H - tpi + 3) / 4 + 2 - a;                            // 4(ss-2+a)+tpi < H
    const int hi_r = (H - xr + 3) / 4;                                     // 4ss+xr < H
    hi = hi < hi_t ? hi : hi_t;
    hi = hi < hi_r ? hi : hi_r;
    lo = lo < 0 ? 0 : lo;
    if (!mask || hi < lo) { lo = 0; hi = 0; }
    QuadBf d;
    d.off = (pix0 * CELLS + ci * CW + cj) * 4;
    d.info = lo | (hi << 8) | (mask << 16);
    return d;
}

__device__ __forceinline__ f32x4 quadbf_fix(const f32x4& v, int info)     // unvisited cells -> exact zeros (bit arithmetic: see quad8_fix)
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

template <int role, int ABL = 0>   // ABL: ablation mask for timing experiments (-DD2T_LAB): 1 no S reloads, 2 no tile stores, 4 no G requests, 8 no ring writes, 16 no MFMAs
__device__ __forceinline__ void stripbf_body(unsigned char* __restrict__ ring, const float* __restrict__ gout,
                                             const float* __restrict__ fm0, const float* __restrict__ fm1,
                                             float* __restrict__ g0, float* __restrict__ g1,
                                             int b, int tj, int yb, int C, int H, int W, int tiles_i)
{
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

    const int cw = yb * BF_CH + wave * (BF_CT * 16);         // first channel of this wave's first c-tile
    // S piece (block q, half h) at super-step 0, c-tile 0 (bytes): lane (channel n, lane group g).  Channels >= C lie
    // behind the buffer: zeros.  Rows >= H (last super-step) read the next plane or the range check's zeros: their G is 0.
    int sv[BF_NB][2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) sv[q][hf] = ((cw + n) * HW + (2 * q + hf) * W + col0 + 4 * g) * 4;
    sv[2][0] = ((cw + n) * HW + g * W + col0 + 16) * 4;
    sv[2][1] = BF_OOR;                                               // block 2 has no second piece: zeros
    const int s_step = 4 * W * 4, ct_step = 16 * HW * 4;
    auto s_load = [&](int ss, int q, int half, int ct) -> f32x4 {
        if ((q == 2 && half == 1) || ((ABL & 1) && ss > 0)) return (q == 2 && half == 1) ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{0.5f, 0.25f, 0.125f, 1.f};
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, sv[q][half] + ss * s_step + ct * ct_step, 0, 0);
        return __builtin_bit_cast(f32x4, v);
    };

    // ---- G production: threads 0..399, four quads each
    QuadBf qd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) qd[k] = quadbf_desc(role, tid, k, H, W, tiles_i, j0, col0);
    const int g_step = 4 * W * CELLS * 4;                            // gradOut bytes per 4 map rows
    f32x4 gn[4];
    auto g_load_all = [&](int ss) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lo = qd[k].info & 255, hi = (qd[k].info >> 8) & 255;
            const int v = ss >= lo && ss < hi ? qd[k].off + ss * g_step : BF_OOR;
            gn[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, v, 0, 0));
        }
    };
    // where this thread's piece goes: fragment (block, tile), lanes (g, tpi, tile column 0..3), half
    int q_pr, a_pr, g_pr, tpi_pr, h_pr;
    prod_decode(tid < BF_PROD ? tid : 0, q_pr, a_pr, g_pr, tpi_pr, h_pr);
    const int at_pr = (((q_pr * NACT + a_pr) * 64 + g_pr * 16 + tpi_pr * 4) * 2 + h_pr) * 8;   // bytes into a piece plane; + 16 per tile column
    auto put8 = [&](unsigned char* buf, int col, const Piece4& s) {
        unsigned char* at = buf + at_pr + col * 16;
        *reinterpret_cast<bf16x4*>(at) = s.hi;
        *reinterpret_cast<bf16x4*>(at + BF_PLANE) = s.mid;
        *reinterpret_cast<bf16x4*>(at + 2 * BF_PLANE) = s.lo;
    };
    auto g_put_part = [&](unsigned char* buf, int part) {            // part 0 / 1: tile columns 0, 1 / 2, 3
        if (tid >= BF_PROD) return;
        if (!role) {
#pragma unroll
            for (int k = 0; k < 2; ++k) put8(buf, 2 * part + k, split4(quadbf_fix(gn[2 * part + k], qd[2 * part + k].info)));   // quad k = tile column k
        } else {
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = quadbf_fix(gn[k], qd[k].info);
#pragma unroll
            for (int c = 0; c < 2; ++c) {                            // quad k = window column k
                const int cc = 2 * part + c;
                put8(buf, cc, split4(f32x4{v[0][cc], v[1][cc], v[2][cc], v[3][cc]}));
            }
        }
    };
    auto g_put_all = [&](unsigned char* buf) { g_put_part(buf, 0); g_put_part(buf, 1); };

    // ---- tile stores: lane (pixel n, channels 4g..4g+3 of each c-tile)
    unsigned long long badt = 0;                                     // tiles this lane stored a non-finite value for (bit u mod 64)
    const int x_lane = ((cw + 4 * g) * HW + (n >> 2) * W + j0 + (n & 3)) * 4;
    const bool col_ok = j0 + (n & 3) < W;
    auto store_tile = [&](const f32x4 (&d)[BF_CT], int u) {
        if (u < 0 || u >= tiles_i) return;                           // wave-uniform
        const int i = 4 * u + (n >> 2);
        const int base = col_ok && i < H ? x_lane + 4 * u * W * 4 : BF_OOR;
        bool bad = false;
#pragma unroll
        for (int ct = 0; ct < BF_CT; ++ct) bad = bad || nonfinite4(d[ct]);
        badt |= bad && base != BF_OOR ? 1ull << (u & 63) : 0ull;
#pragma unroll
        for (int ct = 0; ct < BF_CT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = d[ct][r];                            // (a bit_cast of the element lvalue d[ct][r] reads element 0)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rx,
                                                      base == BF_OOR ? BF_OOR : base + ct * ct_step + r * HW * 4, 0, 0);
            }
    };

    f32x4 acc[BF_CT][NACT], a4[BF_NB][2][BF_CT], done[BF_CT];
#pragma unroll
    for (int ct = 0; ct < BF_CT; ++ct) {
        done[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < NACT; ++a) acc[ct][a] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- prologue: ring[0] <- G(0), S pieces of super-step 0, registers <- G(1)
    g_load_all(0);
    for (int e = tid; e < 3 * 3 * NACT * 64; e += BF_T) {            // block 2 has no second piece: zero it once in every buffer and plane
        const int pl = e / (NACT * 64), r = e - pl * (NACT * 64);    // (LDS starts undefined, and 0 x NaN bits would poison the tile)
        *reinterpret_cast<unsigned long long*>(ring + pl * BF_PLANE + ((2 * NACT * 64 + r) * 2 + 1) * 8) = 0ull;
    }
#pragma unroll
    for (int q = 0; q < BF_NB; ++q)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int ct = 0; ct < BF_CT; ++ct) a4[q][hf][ct] = s_load(0, q, hf, ct);
    g_put_all(ring);
    g_load_all(1);
    lds_barrier();

    const unsigned char* lane_ring = ring + lane * 16;
    auto b_fetch = [&](Piece8& bv, int buf, int q, int a) {
        const unsigned char* at = lane_ring + buf * BF_BUF + (q * NACT + a) * 64 * 16;
        bv.hi = *reinterpret_cast<const bf16x8*>(at);
        bv.mid = *reinterpret_cast<const bf16x8*>(at + BF_PLANE);
        bv.lo = *reinterpret_cast<const bf16x8*>(at + 2 * BF_PLANE);
    };

    Piece8 sa[BF_CT], sn[BF_CT];                                     // S pieces of the current block / of the next one
#pragma unroll
    for (int ct = 0; ct < BF_CT; ++ct) {
        sa[ct] = join8(split4(a4[0][0][ct]), split4(a4[0][1][ct]));
        a4[0][0][ct] = s_load(1, 0, 0, ct); a4[0][1][ct] = s_load(1, 0, 1, ct);
    }

    // One super-step: G(ss) from ring buffer `cur`, G(ss+1) written to buffer `nxt` in front of its one barrier.
    // LO/HI: live accumulators [LO, HI) -- the first two super-steps carry tiles -2/-1 in acc[0..1], the last two carry
    // tiles past the map in acc[3..4].
    auto super_step = [&](int ss, int cur, int nxt, auto lo_c, auto hi_c) {
        constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
        auto block = [&](auto q_c) {
            constexpr int q = decltype(q_c)::value;
            constexpr int qn = q == BF_NB - 1 ? 0 : q + 1;           // the block whose S pieces are split under this block's MFMAs
            const int ssn = q == BF_NB - 1 ? ss + 1 : ss;
            Piece8 bvA, bvB;
            b_fetch(bvA, cur, q, LO);
            D2T_PIN();
#pragma unroll
            for (int a = LO; a < HI; ++a) {
                Piece8& bv = ((a - LO) & 1) ? bvB : bvA;
                Piece8& bn = ((a - LO) & 1) ? bvA : bvB;
                if (a + 1 < HI) b_fetch(bn, cur, q, a + 1);
                D2T_PIN();
                // six piece products, smallest first, the two c-tiles interleaved (two independent accumulator chains)
#pragma unroll
                for (int ct = 0; ct < BF_CT; ++ct) acc[ct][a] = D2T_BFMFMA_A(sa[ct].lo, bv.hi, acc[ct][a]);
#pragma unroll
                for (int ct = 0; ct < BF_CT; ++ct) acc[ct][a] = D2T_BFMFMA_A(sa[ct].hi, bv.lo, acc[ct][a]);
#pragma unroll
                for (int ct = 0; ct < BF_CT; ++ct) acc[ct][a] = D2T_BFMFMA_A(sa[ct].mid, bv.mid, acc[ct][a]);
#pragma unroll
                for (int ct = 0; ct < BF_CT; ++ct) acc[ct][a] = D2T_BFMFMA_A(sa[ct].mid, bv.hi, acc[ct][a]);
#pragma unroll
                for (int ct = 0; ct < BF_CT; ++ct) acc[ct][a] = D2T_BFMFMA_A(sa[ct].hi, bv.mid, acc[ct][a]);
#pragma unroll
                for (int ct = 0; ct < BF_CT; ++ct) acc[ct][a] = D2T_BFMFMA_A(sa[ct].hi, bv.hi, acc[ct][a]);
                D2T_PIN();
                // Everything that is not an MFMA goes BEHIND a tile's twelve queued MFMAs, a piece per tile: the tile store /
                // the ring writes of G(ss+1) and the barrier / the requests for G(ss+2) behind the first two tiles, the
                // split of the next block's S pieces (and their reload for the super-step after) behind the second and third.
                if (a == LO) {
                    if (q == 0) { if (!(ABL & 2)) store_tile(done, ss - 3); else asm volatile("" ::"v"(done[0]), "v"(done[1])); }   // complete since the end of the previous super-step
                    if (q == 1 && !(ABL & 8)) g_put_part(ring + nxt * BF_BUF, 0);  // G(ss+1), requested a super-step ago
                    if (q == 2 && !(ABL & 4)) g_load_all(ss + 2);    // past the map: out of range, zeros
                }
                if (a == LO + 1) {
                    if (q == 1) {
                        if (!(ABL & 8)) g_put_part(ring + nxt * BF_BUF, 1);
                        lds_barrier();                               // publish; `nxt` was last read two barriers ago
                    }
                    sn[0] = join8(split4(a4[qn][0][0]), split4(a4[qn][1][0]));
                    a4[qn][0][0] = s_load(ssn + 1, qn, 0, 0); a4[qn][1][0] = s_load(ssn + 1, qn, 1, 0);   // a whole super-step ahead
                }
                if (a == LO + 2) {
                    sn[1] = join8(split4(a4[qn][0][1]), split4(a4[qn][1][1]));
                    a4[qn][0][1] = s_load(ssn + 1, qn, 0, 1); a4[qn][1][1] = s_load(ssn + 1, qn, 1, 1);
                }
                if (a <= LO + 2) D2T_PIN();
            }
#pragma unroll
            for (int ct = 0; ct < BF_CT; ++ct) sa[ct] = sn[ct];
        };
        block(I0{});
        block(I1{});
        block(I2{});
        // tile ss-2 is complete: keep it for the store in the next super-step, rotate
#pragma unroll
        for (int ct = 0; ct < BF_CT; ++ct) {
            done[ct] = acc[ct][0];
#pragma unroll
            for (int a = 0; a + 1 < NACT; ++a) acc[ct][a] = acc[ct][a + 1];
            acc[ct][NACT - 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        D2T_PIN();
    };

    // tiles_i >= 5 (host-checked): two leading, tiles_i - 4 full, two trailing super-steps; ring buffers rotate 0, 1, 2
    int cur = 0;
    auto next = [&](int c) { return c == 2 ? 0 : c + 1; };
    super_step(0, cur, next(cur), I2{}, I5{}); cur = next(cur);
    super_step(1, cur, next(cur), I1{}, I5{}); cur = next(cur);
    int ss = 2;
    for (; ss <= tiles_i - 3; ++ss) { super_step(ss, cur, next(cur), I0{}, I5{}); cur = next(cur); }
    super_step(ss, cur, next(cur), I0{}, I4{}); cur = next(cur);
    super_step(ss + 1, cur, next(cur), I0{}, I3{});
    store_tile(done, tiles_i - 3);
    {
        f32x4 t0[BF_CT], t1[BF_CT];
#pragma unroll
        for (int ct = 0; ct < BF_CT; ++ct) { t0[ct] = acc[ct][0]; t1[ct] = acc[ct][1]; }
        store_tile(t0, tiles_i - 2);                                 // their remaining super-steps lie below the map
        store_tile(t1, tiles_i - 1);
    }
    if (__builtin_expect(__any(badt != 0), 0)) {                     // cold: non-finite (or > bf16 range) inputs only
        unsigned lo = (unsigned)badt, hi = (unsigned)(badt >> 32);
#pragma unroll
        for (int off = 32; off; off >>= 1) { lo |= __shfl_xor(lo, off, 64); hi |= __shfl_xor(hi, off, 64); }
        const unsigned long long m = ((unsigned long long)hi << 32) | lo;
        for (int ct = 0; ct < BF_CT; ++ct) {
            if (tiles_i > 64) {
                strip_repair(role, lane, gb, S, gx, cw + 16 * ct, C, H, W, j0, CELLS, 1, 0, H);
            } else {
                for (int u = 0; u < tiles_i; ++u)
                    if ((m >> u) & 1)
                        strip_repair(role, lane, gb, S, gx, cw + 16 * ct, C, H, W, j0, CELLS, 1, 4 * u, 4 * u + 4 < H ? 4 * u + 4 : H);
            }
        }
    }
}
#undef D2T_PIN

template <int ABL>
__global__ void __launch_bounds__(BF_T)
k_corr_bwd_strip8bf(const float* __restrict__ gout, const float* __restrict__ fm0, const float* __restrict__ fm1,
                    float* __restrict__ g0, float* __restrict__ g1,
                    int B, int C, int H, int W, int tiles_i, int tiles_j)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];   // BF_LDS bytes
    // The hardware deals workgroups to the XCDs by their LINEAR id (x fastest, then y): the map is applied to that id and
    // (strip, channel block, role, batch item) decoded from the result, as k_corr_bwd_strip8 does -- both roles of a batch
    // item (they share gradOut[b]) and the strips of a channel block (they share rows of S) stay on one XCD.
    const int nb = gridDim.y, lid = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * nb);
    const int tj = lid % tiles_j, yb = (lid / tiles_j) % nb, role = (lid / (tiles_j * nb)) & 1, b = lid / (2 * tiles_j * nb);
    if (role) stripbf_body<1, ABL>(ring, gout, fm0, fm1, g0, g1, b, tj, yb, C, H, W, tiles_i);
    else stripbf_body<0, ABL>(ring, gout, fm0, fm1, g0, g1, b, tj, yb, C, H, W, tiles_i);
}

}  // namespace

bool corr_bwd8bf_supported(int B, int C, int H, int W, int ps, int cs)
{
    return corr_bwd8_supported(B, C, H, W, ps, cs);
}

int corr_bwd8bf_f32(const float* gout, const float* fm0, const float* fm1, float* g0, float* g1,
                    int B, int C, int H, int W, hipStream_t st)
{
    const int tiles_i = (H + TP - 1) / TP, tiles_j = (W + TP - 1) / TP;
#define D2T_BF_LAUNCH(ABLV)                                                                                        \
    {                                                                                                              \
        D2T_ENSURE_DYNAMIC_LDS(k_corr_bwd_strip8bf<ABLV>, BF_LDS);                                                 \
        hipLaunchKernelGGL(k_corr_bwd_strip8bf<ABLV>, dim3(2 * B * tiles_j, (C + BF_CH - 1) / BF_CH), dim3(BF_T), BF_LDS, st, \
                           gout, fm0, fm1, g0, g1, B, C, H, W, tiles_i, tiles_j);                                  \
    }
#ifdef D2T_LAB
    static const int abl = lab_env_int("D2T_BF_ABL", 0);             // timing experiments only: results are wrong
    if (abl == 1) D2T_BF_LAUNCH(1) else if (abl == 2) D2T_BF_LAUNCH(2) else if (abl == 4) D2T_BF_LAUNCH(4) else if (abl == 12) D2T_BF_LAUNCH(12)
    else if (abl == 16) D2T_BF_LAUNCH(16) else if (abl == 15) D2T_BF_LAUNCH(15) else if (abl == 7) D2T_BF_LAUNCH(7) else
#endif
    D2T_BF_LAUNCH(0)
#undef D2T_BF_LAUNCH
    return launch_status();
}

}}  // namespace d2t::tuned
