// d2t_corr_tuned.hip -- gfx950-tuned f32 PointwiseCorrelation (d_max = 8, stride 1).
//
// The correlation is a banded GEMM: out[p][q] = sum_c FM0[c][p] * FM1[c][q] for q in the
// (2d x 2d) window of pixel p.  At the reference's shapes it is f32-FMA-bound, not HBM-bound
// (SURVEY.md F10), so the arithmetic runs on the exact-f32 matrix pipe:
// v_mfma_f32_16x16x4_f32 is bit-for-bit a k-ordered chain of fmaf (MI355X guide, "FP32-input
// MFMA"), i.e. the same ascending-channel FMA chain nvcc builds for the reference's inner loop
// (pointwise_correlation_cuda.cu:105-107).  Forward results are therefore BIT-IDENTICAL to the
// type-generic kernel and to the reference.
//
// Tiling.  M = 16 pixels p arranged 4x4 (a "p-tile"); their windows' union is 19 rows x 19
// columns of FM1.  That union is enumerated as groups of 4 consecutive columns (5 groups = 20
// columns per row): one 16-byte piece is one lane's B operand of FOUR MFMAs (four N-tiles).
// 16 groups x 4 columns = one "tile-group" = 4 N-tiles of 16 columns = one wave task; a p-tile has
// at most 6 tile-groups (95 groups), rows of the window that fall outside the map are not
// enumerated at all.  K = channels, 4 per MFMA, ascending.
//
// Forward kernels (all bit-identical): workgroups own SEGMENTS of vertically stacked p-tiles and
// copy the union of their windows into LDS by LDS-DMA, 16 channels at a time --
//   k_corr_fwd_seg        5 tiles / 15 waves, ring of 3 chunks    (grids >= 192 segments: B = 8)
//   k_corr_fwd_segx<2,4>  2 tiles /  6 waves, ring of 4 chunks    (medium grids)
//   k_corr_fwd_segx<1,4>  1 tile  /  3 waves, ring of 4 chunks    (the real model's B = 1 pairs)
// The epilogue stages the segment's outputs (including the structural zeros the reference gets
// from at::zeros, :192) in LDS and stores contiguous 16-byte runs.  The XCD-aware
// block map gives every XCD a contiguous run of segments so that neighbours share L2 lines.
// The first generation (one p-tile per workgroup, FM1 gathered straight from L2) measured
// texture-address-bound and is gone; DESIGN.md section 4.2 keeps its numbers.
#include "d2t_corr_common.hpp"
#include <type_traits>

namespace d2t { namespace tuned {

// In-kernel stamps for the developer harness lab/csrc/fwd_lab.hip (a separate diagnostic build, see
// the MI355X guide "In-kernel stamps").  The product library is built without D2T_LAB: no stamp
// executes there.
#ifdef D2T_LAB
__device__ unsigned long long* lab_stamps;                          // [workgroup][32]
#define D2T_STAMP(i)                                                                                  \
    do {                                                                                              \
        if (threadIdx.x == 0) {                                                                       \
            unsigned long long t_;                                                                    \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            lab_stamps[blockIdx.x * 32 + (i)] = t_;                                                   \
        }                                                                                             \
    } while (0)
#define D2T_STAMP_RT(i)                                                                               \
    do {                                                                                              \
        if (threadIdx.x == 0) {                                                                       \
            unsigned long long t_;                                                                    \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
            lab_stamps[blockIdx.x * 32 + (i)] = t_;                                                   \
        }                                                                                             \
    } while (0)
// per-wave clock reads for the backward strip kernel (lab/csrc/bwd_stamp_lab.hip); D2T_ABL: ablation mask of that lab
#ifndef D2T_ABL
#define D2T_ABL 0
#endif
__device__ unsigned long long* lab_wave_stamps;                     // [workgroup][wave][8]
#define D2T_WCLK(var) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#define D2T_WCLK_DECL(...) unsigned long long __VA_ARGS__
#define D2T_LAB_ONLY(...) __VA_ARGS__
#else
#define D2T_ABL 0
#define D2T_STAMP(i)
#define D2T_STAMP_RT(i)
#define D2T_WCLK(var)
#define D2T_WCLK_DECL(...)
#define D2T_LAB_ONLY(...)
#endif

// Where cell (ci, cj) of pixel (i, j) of batch item b lives in the correlation output / gradOut:
//     b * bs + (i*W + j) * ps + (ci*17 + cj) * cs          (floats)
// reference layout (B,H,W,17,17): ps = 289, cs = 1, bs = H*W*289; channel-major (289,H,W) per item --
// what correlation_tracker.py:64-70 makes of it with view + permute, and what ROIPool consumes --
// ps = 1, cs = H*W, bs = the caller's distance between items (a slice of a wider concat buffer).
// (struct CellLayout: d2t_tuned.hpp)

// Up to MAXLV correlation problems of one spatial shape in a single launch (the tracker's three pyramid
// levels, correlation_tracker.py:68-70: B = 1 grids fill a fraction of the chip each).  Workgroups
// [wg_end[l-1], wg_end[l]) belong to level l; levels are ordered heaviest (most channels) first by the
// host so that the long workgroups start first.
struct FwdLevels { const float* fm0[MAXLV]; const float* fm1[MAXLV]; float* out[MAXLV]; int C[MAXLV]; int wg_end[MAXLV]; int n; };
struct BwdLevels { const float* gout[MAXLV]; const float* fm0[MAXLV]; const float* fm1[MAXLV]; float* g0[MAXLV]; float* g1[MAXLV];
                   int C[MAXLV]; int wg_end[MAXLV]; int n; };

// ------------------------------------------------------------------------------------
// Forward, LDS-staged segment kernel (the default when the grid fills the chip).
// A tile-per-workgroup kernel gathers every p-tile's 19 x 20 window straight from L2 in 80-byte row
// pieces; rocprofv3 showed its texture-address unit 85 % busy (TA_TA_BUSY) at ~25 useful bytes per
// 64-byte sector, i.e. it is bound by the gather, not by MFMA or HBM.  Here one workgroup owns a
// segment of 5 vertically stacked p-tiles: the union of their windows (<= 35 rows x 20 columns) is
// copied ONCE per 16 channels into LDS (2.7x fewer bytes through the TA), and every wave reads its
// MFMA fragments from there with conflict-free ds_read_b128 (a tile's window rows are contiguous
// in the image, so a tile-group is 16 consecutive 16-byte slots).  A wave owns two (tile,
// tile-group) tasks.  At B=8, 38x63 the grid is 8 x 16 x 2 = 256 workgroups of exactly 5 tiles:
// one per CU.  Arithmetic is unchanged (ascending-channel MFMA chain): bit-identical results.
// ------------------------------------------------------------------------------------
#ifndef D2T_EXP_EPI
#define D2T_EXP_EPI 0      // tile-by-tile epilogue (round 4): measured 50.1 us against 47.5 us for the one-barrier form -- off
#endif
#ifndef D2T_FWD_LOADER
#define D2T_FWD_LOADER 0   // round 5 experiment (make EXTRA=-DD2T_FWD_LOADER=1): ONE loader wave issues every LDS-DMA instruction -- 80.1 us against 47.2
                           // (profiles/r05_a_ab_fwd_loader_wave.txt): a wave issues one such instruction per ~290 cycles, 41 per chunk take 5 us
#endif
#ifndef D2T_FWD_STAGGER
#define D2T_FWD_STAGGER 1  // round 5: the waves of a SIMD issue their LDS-DMA instructions at different k-steps of a chunk (A/B: -DD2T_FWD_STAGGER=0)
#endif
constexpr int SG_NU = 5;                            // p-tiles per segment
constexpr int SG_WAVES = 15;                        // computing waves: <= 30 (tile, tile-group) tasks, at most two per wave
constexpr int SG_THREADS = (SG_WAVES + D2T_FWD_LOADER) * 64;
constexpr int SG_KC = 16;                           // channels per staged chunk (4 k-steps)
constexpr int SG_ROWS = 4 * SG_NU + 2 * DT - 1;     // 35 window rows of a segment
constexpr int SG_SLOTS = SG_ROWS * NCG + 1;         // 176 slots of 16 bytes per channel (175 pieces + 1 pad)
constexpr int SG_BPL = SG_SLOTS * 4;                // 704 floats: plane stride = 0 mod 64 dwords -> the four
                                                    // k-slot lane groups of a ds_read_b128 hit disjoint banks
constexpr int SG_APL = SG_NU * 16;                  // 80 floats: FM0 pixels of one channel
constexpr int SG_BUF = SG_KC * (SG_BPL + SG_APL);   // floats per buffer (49 KB)
constexpr int SG_RING = 3;                          // staged chunks: one being read, one landed, one in flight
constexpr int SG_DUMMY = 256;                       // floats: where parked DMA instructions land
constexpr int SG_STAGE = SG_NU * 16 * CELLS;        // out staging (90.3 KB), aliases the ring
constexpr int SG_LDS = SG_RING * SG_BUF + SG_DUMMY > SG_STAGE ? SG_RING * SG_BUF + SG_DUMMY : SG_STAGE;
static_assert(SG_LDS * 4 <= 160 * 1024, "LDS budget");

constexpr int SG_AI = SG_KC * 4 * SG_NU / 64;       // 5 DMA wave-instructions move the FM0 pixels of a chunk
constexpr int SG_MAXDMA = (SG_KC * SG_SLOTS / 64 + SG_AI + SG_WAVES - 1) / SG_WAVES;   // 4: most a wave issues per chunk
static_assert(SG_KC * 4 * SG_NU % 64 == 0 && SG_MAXDMA <= SG_KC / 4, "DMA plan");

// The kernel.  Per chunk of 16 channels a wave issues its 2-3 LDS-DMA instructions of chunk ch+2, 4
// fragment fetches and 4 x NT x 4 MFMAs; the fragments of k-step ks+1 are fetched under the MFMAs of
// k-step ks -- across the chunk boundary too, which is what the third ring slot buys: chunk ch+1 has
// landed and is published by the barrier that sits BEHIND k-step 2 of chunk ch, so the matrix pipe
// never waits for an LDS round trip and a chunk's DMA has more than a chunk time to land.  LDS-DMA
// loads retire in order and a wave issues the same number (nd) for every chunk, so `vmcnt(nd)` at that
// barrier means "my part of chunk ch+1 has landed" (chunk ch+2's may be in flight).
//
// DMA instructions are the expensive part of the loop (measured with lab/csrc/fwd_lab: removing them
// takes 12 k cycles off a 67 k-cycle loop, ~50 cycles of SIMD issue each), so there are as few as the
// bytes allow: the image pitch is the segment's own row count (27 rows at 38-row maps, not the
// 35-row maximum), instructions are dealt round-robin to the waves, none is a parked dummy.
//
// Work distribution.  The segment's ACTIVE tile-groups (a tile at the top or bottom of the map has
// fewer window rows, hence fewer than 6) are numbered consecutively and dealt round-robin: task q
// goes to wave q mod 15.  A workgroup's waves are placed on the CU's four SIMDs cyclically (checked
// with HW_REG_HW_ID), so this also balances the matrix pipes: at 38 rows every segment has 27 tasks
// -> 7 / 7 / 7 / 6 per SIMD.  (The first version gave wave w the fixed slots w and w+15 of a 5 x 6
// table and issued the MFMAs of empty slots as well: 8 per SIMD.)
// fm0b / fm1b: the C channel planes this workgroup contracts over (one batch item; all of its channels, or one
// channel range of a split -- the buffer descriptors end behind them, so the last chunk and the two staged past the
// end arrive as zeros); outb: where the segment's cells go (the output of that batch item, or a partial-sum plane).
__device__ __forceinline__ void
seg_body(const float* __restrict__ fm0b, const float* __restrict__ fm1b, float* __restrict__ outb,
         int C, int H, int W, int tiles_i, int seg, int tj, CellLayout lay)
{
    __shared__ __attribute__((aligned(16))) float smem[SG_LDS];

    D2T_STAMP(0); D2T_STAMP_RT(8);
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef D2T_LAB
    if (lane == 0) { unsigned hw_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); lab_stamps[blockIdx.x * 32 + 16 + wave] = hw_; }
#endif
    const int u0 = seg * SG_NU, nu = tiles_i - u0 < SG_NU ? tiles_i - u0 : SG_NU;
    const int j0 = tj * TP, HW = H * W;
    const unsigned plane_bytes = (unsigned)C * HW * 4u;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fm1b), 0, plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fm0b), 0, plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro =
        __builtin_amdgcn_make_buffer_rsrc(outb, 0, (unsigned)HW * CELLS * 4u, 0x00020000);

    const int R0 = 4 * u0 - DT > 0 ? 4 * u0 - DT : 0;                // region rows [R0, R1) inside the map
    const int R1 = 4 * (u0 + nu) + DT - 1 < H ? 4 * (u0 + nu) + DT - 1 : H;
    const int nrows = R1 - R0;
    const int colL = j0 - DT;                                        // region columns [colL, colL+20): may leave the map

    // ---- staging by LDS-DMA (buffer_load_dwordx4 ... lds): a wave-instruction copies 64 pieces of
    // 16 bytes from per-lane global addresses into 64 CONSECUTIVE 16-byte LDS slots, no VGPR round
    // trip.  The LDS image of a chunk is [channel][row][column group] with a pitch of P slots per
    // channel (the segment's rows x 5, rounded up to 16 slots so that the plane stride stays 0 mod 64
    // dwords: the four k-slot lane groups of a ds_read_b128 hit disjoint banks), followed by the FM0
    // pixels [channel][20 pixel rows]; piece e of either part lands in slot e.  The buffer descriptor
    // covers exactly this batch item's C planes: a piece of a channel >= C (last chunk, and the two
    // chunks staged past the end) is out of range and arrives as exact zeros without touching memory.
    // Columns outside the map read whatever neighbours them in memory: MFMA columns are independent
    // and those cells are masked in the epilogue.  Pad slots and pixel rows past the map are parked
    // out of range.  Instruction y of a chunk (FM1 instructions first, then the 5 FM0 ones) belongs to
    // wave y mod 15.
    constexpr unsigned OOR = 0x7ffffff0u;                            // parked byte offset: always out of range
    const int P = (nrows * NCG + 15) & ~15;                          // slots per channel
    const int BPL = 4 * P;                                           // floats per channel plane
    const int a_base = SG_KC * BPL;                                  // float offset of the FM0 part in a buffer
    const int nBI = P >> 2;                                          // FM1 instructions per chunk (16 * P / 64)
    const int ndma = nBI + SG_AI;                                    // LDS-DMA instructions per chunk
    (void)ndma;
    const float rP = 1.0f / (float)P;
    const int chunk_bytes = SG_KC * HW * 4;
    using std::integral_constant;
    typedef integral_constant<int, 0> K0; typedef integral_constant<int, 1> K1;
    typedef integral_constant<int, 2> K2; typedef integral_constant<int, 3> K3;
#if D2T_FWD_LOADER
    // ---- the LOADER wave (wave 15): it issues all ndma LDS-DMA instructions of every chunk.  (Rounds 1-4 dealt them to the 15
    // computing waves; a wave's stream is in order, the CU's address unit takes ~55 cycles per 1 KB instruction of 16-byte pieces
    // -- csrc/d2t_corr_fwd_band.hip measured it with in-kernel clocks -- and every DMA a computing wave waits to issue is a slot
    // in which it issues no MFMA: removing the DMA took 12 k cycles off a 67 k-cycle loop in round 2.)  One barrier per chunk for
    // both roles; the loader arrives with `vmcnt(ndma)`: its instructions of chunk ch+1 have landed, chunk ch+2's are in flight.
    if (wave == SG_WAVES) {
        // addresses are recomputed per instruction (a dozen VALU operations against ~55 cycles of address-unit time each): no
        // per-instruction register arrays, a loop of a few dozen instructions
        auto stage = [&](int slot, int chunk) {
            const int cb = chunk * chunk_bytes;
            float* buf = smem + slot * SG_BUF;
#pragma unroll 2
            for (int k = 0; k < nBI; ++k) {                          // FM1: piece e of the [channel][row][column group] image lands in slot e
                const int e = k * 64 + lane;
                const int ch = (int)(((float)e + 0.5f) * rP), rem = e - ch * P;   // e / P, exact for e < 2^15
                const int row = rem / NCG, cg = rem - row * NCG;
                const int vo = row < nrows ? (ch * HW + (R0 + row) * W + colL + 4 * cg) * 4 + cb : (int)OOR;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr)(buf + k * 256), 16, vo, 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < SG_AI; ++k) {                        // FM0 pieces: (channel, pixel row of the segment)
                const int e = k * 64 + lane;
                const int ch = e / (4 * SG_NU), prow = e - ch * (4 * SG_NU);
                const int i = 4 * u0 + prow;
                const int vo = i < H ? (ch * HW + i * W + j0) * 4 + cb : (int)OOR;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (lds_ptr)(buf + a_base + k * 256), 16, vo, 0, 0, 0);
            }
        };
        auto wait_prev_and_barrier = [&]() {                         // all but the youngest ndma instructions have landed
            switch (nBI >> 2) {                                      // ndma = 4 (nBI / 4) + 5: nBI is a multiple of 4
                case 1: dma_wait_barrier<9>(); break;   case 2: dma_wait_barrier<13>(); break;  case 3: dma_wait_barrier<17>(); break;
                case 4: dma_wait_barrier<21>(); break;  case 5: dma_wait_barrier<25>(); break;  case 6: dma_wait_barrier<29>(); break;
                case 7: dma_wait_barrier<33>(); break;  case 8: dma_wait_barrier<37>(); break;  case 9: dma_wait_barrier<41>(); break;
                case 10: dma_wait_barrier<45>(); break; default: dma_wait_barrier<49>(); break;
            }
        };
        const int nchunks_l = (C + SG_KC - 1) / SG_KC;
        stage(0, 0);
        stage(1, 1);
        wait_prev_and_barrier();                                     // chunk 0 has landed
        int s_fre = 2;
        for (int ch = 0; ch < nchunks_l; ++ch) {
            stage(s_fre, ch + 2);                                    // into the slot of chunk ch-1 (every wave is past its last read of it)
            s_fre = s_fre == SG_RING - 1 ? 0 : s_fre + 1;
            wait_prev_and_barrier();                                 // chunk ch+1 has landed; publish
        }
    }
    auto dma = [&](int, int, auto) {};
    auto wait_prev_chunk_and_barrier = [&]() { lds_barrier(); };
#else
    const int nd = (ndma - wave + SG_WAVES - 1) / SG_WAVES;          // this wave's instructions per chunk (wave-uniform)
    // dv is UNSIGNED: a parked piece starts at OOR and is advanced by a chunk like every other piece; corr_fwd_supported() bounds
    // (C + 8 * SG_KC) * H * W * 4 below 0x7ffffff0, so OOR + (nchunks + 2) * chunk_bytes stays below 2^32 (no wrap back into the
    // planes) and above plane_bytes (still rejected by the range check) for every shape admitted -- C = 2048 has 130 chunks.
    unsigned dv[SG_MAXDMA];                                          // byte offset in the planes
    int dl[SG_MAXDMA];                                               // float offset in the buffer
    bool dA[SG_MAXDMA];
#pragma unroll
    for (int k = 0; k < SG_MAXDMA; ++k) {
        const int y = wave + SG_WAVES * k;
        dA[k] = y >= nBI;                                            // wave-uniform
        if (!dA[k]) {
            const int e = y * 64 + lane;
            const int ch = (int)(((float)e + 0.5f) * rP), rem = e - ch * P;   // e / P, exact for e < 2^15
            const int row = rem / NCG, cg = rem - row * NCG;
            dv[k] = row < nrows ? (unsigned)((ch * HW + (R0 + row) * W + colL + 4 * cg) * 4) : OOR;
            dl[k] = y * 256;
        } else {
            const int e = (y - nBI) * 64 + lane;                     // FM0 piece: (channel, pixel row of the segment)
            const int ch = e / (4 * SG_NU), prow = e - ch * (4 * SG_NU);
            const int i = 4 * u0 + prow;
            dv[k] = i < H ? (unsigned)((ch * HW + i * W + j0) * 4) : OOR;
            dl[k] = a_base + (y - nBI) * 256;
        }
    }
    auto dma = [&](int slot, int chunk, auto k_c) {                  // this wave's k-th DMA instruction of `chunk` into ring slot `slot`
        constexpr int k = decltype(k_c)::value;
        if (k >= nd) return;                                         // wave-uniform
        // Chunks are staged in ascending order, each of this wave's instructions once per chunk: its lane offset is ADVANCED by a chunk after
        // use (parked pieces stay out of range, see dv above) -- one v_add per DMA instruction, no select.  The f32 MFMA and
        // the vector ALU do not overlap on a SIMD (lab/csrc/mfma_valu_lab: their times add), so a vector instruction in this loop is matrix time.
        // (The chunk base in a per-chunk DESCRIPTOR would need none, but hipcc then rebuilds the descriptor behind a waterfall loop.)
        (void)chunk;
        const int v = (int)dv[k];                                    // the instruction's voffset is unsigned in hardware
        dv[k] += (unsigned)chunk_bytes;
        float* dst = smem + slot * SG_BUF + dl[k];
        if (dA[k]) __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (lds_ptr)dst, 16, v, 0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr)dst, 16, v, 0, 0, 0);
    };
    // "my instructions of the previous chunk have landed" (nd younger ones may be in flight) + barrier
    auto wait_prev_chunk_and_barrier = [&]() {
        if (nd >= 4) dma_wait_barrier<4>();
        else if (nd == 3) dma_wait_barrier<3>();
        else if (nd == 2) dma_wait_barrier<2>();
        else if (nd == 1) dma_wait_barrier<1>();
        else dma_wait_barrier<0>();
    };
#endif
    // chunk 0 goes out before anything else is computed; chunk 1 behind the task tables (every workgroup
    // of the chip starts at the same moment: chunk 0 alone lands sooner than both together)
    dma(0, 0, K0{}); dma(0, 0, K1{}); dma(0, 0, K2{}); dma(0, 0, K3{});
    D2T_STAMP(10);

    // ---- this wave's tasks: active tile-groups in (tile, tile-group) order, task q -> wave q mod 15
    int tgs[SG_NU];                                                  // tile-groups of tile t (0 past the segment)
#pragma unroll
    for (int t = 0; t < SG_NU; ++t) {
        const int u = u0 + t;
        const int wa = 4 * u - DT > 0 ? 4 * u - DT : 0;
        const int wb = 4 * u + TP + DT - 1 < H ? 4 * u + TP + DT - 1 : H;
        tgs[t] = t < nu && wb > wa ? ((wb - wa) * NCG + 15) >> 4 : 0;
    }
    int t_tile[2], t_T[2], t_ng[2];
    bool t_on[2];
    int l_off[2], a_off[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        int q = wave + SG_WAVES * k, t = 0;
#pragma unroll
        for (int tt = 0; tt < SG_NU; ++tt)
            if (t == tt && q >= tgs[tt]) { q -= tgs[tt]; ++t; }
        t_on[k] = t < SG_NU && wave < SG_WAVES;                       // q < tgs[t]; the loader wave has no task
        t = t_on[k] ? t : 0;
        const int T = t_on[k] ? q : 0;
        const int u = u0 + t;
        const int wa = 4 * u - DT > 0 ? 4 * u - DT : 0;              // tile's window rows inside the map
        const int wb = 4 * u + TP + DT - 1 < H ? 4 * u + TP + DT - 1 : H;
        const int ng = (wb - wa) * NCG;
        t_tile[k] = t; t_T[k] = T; t_ng[k] = ng;
        // lane's slot inside the tile-group (clamped to the tile's last group; masked in the epilogue)
        int gi = 16 * T + n;
        gi = gi < ng ? gi : (ng > 0 ? ng - 1 : 0);
        l_off[k] = ((wa - R0) * NCG + gi) * 4 + g * BPL;
        a_off[k] = a_base + g * SG_APL + t * 16 + n;
    }

    dma(1, 1, K0{}); dma(1, 1, K1{}); dma(1, 1, K2{}); dma(1, 1, K3{});

    f32x4 acc[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[k][s] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = (C + SG_KC - 1) / SG_KC;
#ifdef D2T_LAB
    unsigned long long lab_dma_wait = 0, lab_bar_wait = 0;
    const bool LAB_WAVE0 = wave == 0;
#endif
    struct Frag { f32x4 q[2]; float a[2]; };
    const int dma_pos = (wave >> 2) & 3;                             // D2T_FWD_STAGGER: where in a chunk this wave issues its DMA instructions
    (void)dma_pos;
    // BPLC: the FM1 image's plane pitch as a compile-time constant (0: the run-time value).  With the pitch in the template the k-step
    // stride of the fragment reads is an immediate of the ds_read: one address per chunk and fragment instead of one per k-step.  (A vector
    // instruction is matrix time on this chip -- lab/csrc/mfma_valu_lab.  Unrolling the loop over the three ring slots as well, so that no
    // address is computed in it at all, ran out of registers: 128 VGPRs + 45 spilled SGPRs.)
    auto run = [&](auto nt_c, auto bpl_c) {
        constexpr int NT = decltype(nt_c)::value;                    // tasks of this wave: 0, 1 or 2
        constexpr int BPLC = decltype(bpl_c)::value;
        auto fetch = [&](Frag& f, const float* buf, int ks) {
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                f.q[k] = *reinterpret_cast<const f32x4*>(buf + l_off[k] + ks * 4 * (BPLC ? BPLC : BPL));
                f.a[k] = buf[a_off[k] + ks * 4 * SG_APL];
            }
        };
        auto mfma = [&](const Frag& f, int s_lo, int s_hi) {
#pragma unroll
            for (int s = s_lo; s < s_hi; ++s)
#pragma unroll
                for (int k = 0; k < NT; ++k) acc[k][s] = D2T_MFMA(f.a[k], f.q[k][s], acc[k][s]);
        };
        Frag f0, f1;
        D2T_STAMP(11);
#ifdef D2T_LAB
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        D2T_STAMP(12);
#endif
        wait_prev_chunk_and_barrier();                               // chunk 0 has landed (chunk 1 may be in flight)
        D2T_STAMP(1);
        fetch(f0, smem, 0);
        int s_cur = 0;
        for (int ch = 0; ch < nchunks; ++ch) {
            const int s_nxt = s_cur == SG_RING - 1 ? 0 : s_cur + 1;  // chunk ch+1
            const int s_fre = s_nxt == SG_RING - 1 ? 0 : s_nxt + 1;  // chunk ch-1: every wave has passed the barrier
            const float* cur = smem + s_cur * SG_BUF;                // behind its last read of that slot
            // sched_barrier: the fetch of the NEXT k-step is issued before this k-step's MFMAs (hipcc
            // otherwise sinks it behind them and waits for the LDS round trip in front of every k-step);
            // a DMA instruction goes behind the MFMAs of a k-step.  Behind the barrier every wave of a
            // SIMD is released at once: each first issues MFMAs it already holds the operands of.
#define D2T_PIN() __builtin_amdgcn_sched_barrier(0)
#if D2T_FWD_STAGGER
            // A wave issues ALL its DMA instructions of the chunk at ONE of four points, chosen by which of its SIMD's waves it is
            // (waves go to the SIMDs cyclically: wave >> 2).  An LDS-DMA instruction of 64 scattered pieces holds its wave ~290
            // cycles (profiles/r05_a_ab_fwd_loader_wave.txt); with every wave issuing at the same points the four waves of a
            // SIMD sat in them together and the matrix pipe had nobody to take an MFMA from.
#define D2T_DMA_AT(p) if (dma_pos == (p)) { dma(s_fre, ch + 2, K0{}); dma(s_fre, ch + 2, K1{}); dma(s_fre, ch + 2, K2{}); dma(s_fre, ch + 2, K3{}); }
            D2T_DMA_AT(0) D2T_PIN();
            fetch(f1, cur, 1); D2T_PIN(); mfma(f0, 0, 4); D2T_DMA_AT(1) D2T_PIN();
            fetch(f0, cur, 2); D2T_PIN(); mfma(f1, 0, 4); D2T_DMA_AT(2) D2T_PIN();
            fetch(f1, cur, 3); D2T_PIN(); mfma(f0, 0, 4); D2T_DMA_AT(3) D2T_PIN();
#undef D2T_DMA_AT
#else
            fetch(f1, cur, 1); D2T_PIN(); mfma(f0, 0, 4); dma(s_fre, ch + 2, K0{}); D2T_PIN();
            fetch(f0, cur, 2); D2T_PIN(); mfma(f1, 0, 4); dma(s_fre, ch + 2, K1{}); dma(s_fre, ch + 2, K2{}); D2T_PIN();
            fetch(f1, cur, 3); D2T_PIN(); mfma(f0, 0, 4); dma(s_fre, ch + 2, K3{}); D2T_PIN();
#endif
#ifdef D2T_LAB
            unsigned long long tb0_, tb1_, tb2_;
            if (LAB_WAVE0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb0_)::"memory");
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (LAB_WAVE0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb1_)::"memory");
#endif
            wait_prev_chunk_and_barrier();                           // my part of chunk ch+1 has landed; publish
#ifdef D2T_LAB
            if (LAB_WAVE0) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tb2_)::"memory"); lab_dma_wait += tb1_ - tb0_; lab_bar_wait += tb2_ - tb1_; }
#endif
            mfma(f1, 0, 2); D2T_PIN();
            fetch(f0, smem + s_nxt * SG_BUF, 0); D2T_PIN(); mfma(f1, 2, 4); D2T_PIN();
#undef D2T_PIN
            s_cur = s_nxt;
        }
    };
    if (wave < SG_WAVES) {                                           // (the loader wave has met the same barriers in its own loop)
        // 576 floats: the pitch of every segment of a 38-row map (28 / 26 window rows x 5 column groups, rounded to 16 slots)
        auto run_nt = [&](auto nt_c) {
            if (BPL == 576) run(nt_c, integral_constant<int, 576>{});
            else run(nt_c, integral_constant<int, 0>{});
        };
        if (t_on[1]) run_nt(integral_constant<int, 2>{});            // wave-uniform
        else if (t_on[0]) run_nt(integral_constant<int, 1>{});
        else run_nt(integral_constant<int, 0>{});
    }
    D2T_STAMP(2);
#ifdef D2T_LAB
    if (threadIdx.x == 0) { lab_stamps[blockIdx.x * 32 + 13] = lab_dma_wait; lab_stamps[blockIdx.x * 32 + 14] = lab_bar_wait; }
#endif
    __syncthreads();                                                 // vmcnt(0): the zero chunks staged past the end
    D2T_STAMP(3);

    // ---- epilogue: [nu tiles][16 pixels][17][17] through LDS, then 4*nu contiguous runs ----
    // A lane's 4 x 4 block covers, for pixel (g, r) and displaced row rho, the cells cj = 4cg + s - r;
    // over the five column groups that is every cj in [0, 16], so the scatter itself writes the
    // structural zeros (cj = 16, ci = 16, displaced column outside the map) of the rows it covers.
    // Rows of the 17 x 17 map whose displaced row lies outside the tile's window (ci = 16 of the
    // tile's last pixel row; rows cut off by the map edge) are zero-filled here.
    // D2T_EXP_EPI (lab switch, off): tile by tile -- the stores of tile t issued as soon as its 16 x 17 x 17 block is staged,
    // so that the 22 MB all workgroups of the chip write at the same moment start to drain while the later tiles are still
    // scattered (nu barriers instead of one).  Measured in the alternating step (tools/ab.sh, three rounds,
    // profiles/r04_d_ab_fwd_epilogue_by_tile.txt): 50.1 / 50.1 / 50.2 us against 47.8 / 47.2 / 47.5 -- the extra barriers and
    // the five small store bursts cost more than the earlier start saves.
    const int nj = W - j0 < TP ? W - j0 : TP;
    const int run_ = nj * CELLS, run4 = run_ >> 2;                   // floats / whole float4s per pixel row
    const int prs = (H - 4 * u0 < 4 * nu ? H - 4 * u0 : 4 * nu);     // pixel rows that exist
    auto zero_rows = [&](int t_lo, int t_hi) {
        for (int e = t_lo * 16 * CW + tid; e < t_hi * 16 * CW; e += SG_THREADS) {
            const int p = e / CW, ci = e - p * CW, t = p >> 4, gg = (p >> 2) & 3;
            const int u = u0 + t;
            const int wa = 4 * u - DT > 0 ? 4 * u - DT : 0;
            const int wb = 4 * u + TP + DT - 1 < H ? 4 * u + TP + DT - 1 : H;
            const int rho = 4 * u + gg - DT + ci;
            if (rho < wa || rho >= wb) {
                float* row = smem + p * CELLS + ci * CW;
#pragma unroll
                for (int cj = 0; cj < CW; ++cj) row[cj] = 0.f;
            }
        }
    };
    auto scatter = [&](int k) {
        const int gi = 16 * t_T[k] + n;
        if (t_on[k] && gi < t_ng[k]) {
            const int u = u0 + t_tile[k];
            const int wa = 4 * u - DT > 0 ? 4 * u - DT : 0;
            const int rho = wa + gi / NCG, cg = gi - (gi / NCG) * NCG;   // displaced row, column group
            const int ci = rho - (4 * u + g) + DT;                   // di - i + d, pixel row i = 4u + g
            if (ci >= 0 && ci <= 2 * DT) {
                float* row = smem + (t_tile[k] * 16 + 4 * g) * CELLS + ci * CW;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int dj = colL + 4 * cg + s;
                    const bool live = ci < 2 * DT && dj >= 0 && dj < W;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cj = dj - (j0 + r) + DT;
                        if (cj >= 0 && cj <= 2 * DT) row[r * CELLS + cj] = live && cj < 2 * DT ? acc[k][s][r] : 0.f;
                    }
                }
            }
        }
    };
    auto store_rows = [&](int pr_lo, int pr_hi) {                    // pixel rows [pr_lo, pr_hi) of the segment, reference layout
        for (int e = pr_lo * run4 + tid; e < pr_hi * run4; e += SG_THREADS) {
            const int pr = e / run4, q = e - pr * run4;
            const int off = (((4 * u0 + pr) * W + j0) * CELLS + 4 * q) * 4;
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(smem + (size_t)pr * 4 * CELLS + 4 * q),
                                                   ro, off, 0, 0);   // plain write-back stores.  Round 1 used sc1 (write-through: 63 -> 51 us
                                                                     // then); with the ring-of-3 schedule plain stores are 0.9 us faster (A/B, round 2)
        }
        const int tail = run_ - 4 * run4;                            // 0..3 floats per pixel row (nj < 4)
        for (int e = pr_lo * tail + tid; e < pr_hi * tail; e += SG_THREADS) {
            const int pr = e / tail, q = 4 * run4 + (e - pr * tail);
            outb[((size_t)(4 * u0 + pr) * W + j0) * CELLS + q] = smem[(size_t)pr * 4 * CELLS + q];
        }
    };
#if D2T_EXP_EPI
    if (lay.cs == 1) {
        for (int t = 0; t < nu; ++t) {                               // wave-uniform trip count
            zero_rows(t, t + 1);
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (t_tile[k] == t) scatter(k);                      // wave-uniform
            __syncthreads();
            const int lo = 4 * t < prs ? 4 * t : prs, hi = 4 * t + 4 < prs ? 4 * t + 4 : prs;
            store_rows(lo, hi);
        }
        D2T_STAMP(4);
        D2T_STAMP(5);
        return;
    }
#endif
    zero_rows(0, nu);
#pragma unroll
    for (int k = 0; k < 2; ++k) scatter(k);
    __syncthreads();
    D2T_STAMP(4);
    // 16-byte stores (dword-aligned addresses): a pixel row of the strip is nj*289 contiguous floats in out and starts
    // 16-byte aligned in the LDS image
    if (lay.cs != 1) {
        // channel-major: cell c of pixel row pr is the 16-byte piece out[c][4*u0+pr][j0..j0+3]; consecutive
        // threads take consecutive cells (consecutive LDS words, pieces H*W*4 bytes apart in memory)
        for (int e = tid; e < prs * CELLS; e += SG_THREADS) {
            const int pr = e / CELLS, c = e - pr * CELLS;
            const float* src = smem + (size_t)pr * 4 * CELLS + c;
            float* dst = outb + (size_t)c * lay.cs + ((size_t)(4 * u0 + pr) * W + j0) * lay.ps;
            for (int r = 0; r < nj; ++r) dst[(size_t)r * lay.ps] = src[r * CELLS];
        }
        D2T_STAMP(5);
        return;
    }
    store_rows(0, prs);
    D2T_STAMP(5);
#ifdef D2T_LAB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the stores have been acknowledged
    D2T_STAMP(6); D2T_STAMP_RT(9);
#endif
}

__global__ void __launch_bounds__(SG_THREADS)
k_corr_fwd_seg(const float* __restrict__ fm0, const float* __restrict__ fm1, float* __restrict__ out,
               int C, int H, int W, int tiles_i, int tiles_j, int nseg, CellLayout lay)
{
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int seg = bid % nseg, tj = (bid / nseg) % tiles_j, b = bid / (nseg * tiles_j);
    const size_t item = (size_t)b * C * H * W;
    seg_body(fm0 + item, fm1 + item, out + (size_t)b * lay.bs, C, H, W, tiles_i, seg, tj, lay);
}

// ------------------------------------------------------------------------------------
// Forward, channel-split ("split-K").  The real model correlates B = 1 pairs with 512 / 1024 / 2048 channels
// (correlation_tracker.py:68-70): 38 five-tile segments for 256 CUs, and one-tile workgroups (190 of them) are bound
// by the texture-address unit -- a 19 x 20 window per tile is 2.9x the cache lines per tile of a five-tile segment
// (lab/csrc/ta_lab: ~2 cycles per 64-byte line touched).  So the channels of a level are split over S workgroups per
// segment: each runs the five-tile kernel body over its channel range and writes a partial-sum plane (reference
// layout) into the workspace; k_corr_fwd_combine adds the S planes of every cell in ascending split order --
// deterministic, no atomics -- and writes the caller's layout.  The sum is associated differently from the
// reference's single ascending-channel chain: results agree to f32 rounding (|delta| <= 1e-5 relative, tested), not
// bit for bit; D2T_IMPL_GENERIC stays the bit-exact anchor and grids that fill the chip by themselves never split.
// ------------------------------------------------------------------------------------
struct SplitLevels {
    const float* fm0[MAXLV]; const float* fm1[MAXLV];
    float* part[MAXLV];                                                     // S[l] planes of B*H*W*289 floats -- or, S[l] = 1, the output itself
    int C[MAXLV], S[MAXLV], cps[MAXLV], wg_end[MAXLV]; int n;               // cps: chunks (16 channels) per split
};

__global__ void __launch_bounds__(SG_THREADS)
k_corr_fwd_seg_split(SplitLevels lv, int B, int H, int W, int tiles_i, int tiles_j, int nseg, CellLayout lay)
{
    // logical id: the hardware deals workgroups to the 8 XCDs round-robin by blockIdx.x; give every XCD a contiguous run of
    // the logical order (level, split, batch item, strip, segment), so that neighbouring segments -- whose windows overlap --
    // share an L2.  (The map must be applied to the GLOBAL id: applied to the id inside a level or split, as at first, it
    // only matches the hardware's dealing when that range starts at a multiple of 8.)
    const int gl = xcd_remap(blockIdx.x, gridDim.x);
    int L = 0, wg0 = 0;
#pragma unroll
    for (int l = 1; l < MAXLV; ++l)
        if (l < lv.n && gl >= lv.wg_end[l - 1]) { L = l; wg0 = lv.wg_end[l - 1]; }
    const float* fm0 = lv.fm0[0]; const float* fm1 = lv.fm1[0]; float* part = lv.part[0];
    int C = lv.C[0], S = lv.S[0], cps = lv.cps[0];
#pragma unroll
    for (int l = 1; l < MAXLV; ++l)
        if (L == l) { fm0 = lv.fm0[l]; fm1 = lv.fm1[l]; part = lv.part[l]; C = lv.C[l]; S = lv.S[l]; cps = lv.cps[l]; }
    const int nsegs = B * tiles_j * nseg;
    const int lid = gl - wg0, s = lid / nsegs, bid = lid - s * nsegs;
    const int seg = bid % nseg, tj = (bid / nseg) % tiles_j, b = bid / (nseg * tiles_j);
    const int HW = H * W, c0 = s * cps * SG_KC;
    const int cn = C - c0 < cps * SG_KC ? C - c0 : cps * SG_KC;      // channels of this split (>= 1 by construction)
    const size_t item = ((size_t)b * C + c0) * HW;
    const CellLayout ref{CELLS, 1, 1LL * HW * CELLS};
    if (S == 1) seg_body(fm0 + item, fm1 + item, part + (size_t)b * lay.bs, cn, H, W, tiles_i, seg, tj, lay);   // a level that is not split
    else seg_body(fm0 + item, fm1 + item, part + ((size_t)s * B + b) * HW * CELLS, cn, H, W, tiles_i, seg, tj, ref);
}

// out[cell of pixel] = sum over the splits, ascending.  One workgroup = (level, batch item, 16 consecutive pixels):
// the 16 x 289 partial sums of a split are one contiguous run (all S loads of an element are issued together); the
// channel-major layout goes through LDS so that a cell's 16 pixels leave as one 64-byte run.
constexpr int CB_PIX = 16;
constexpr int CB_MAXS = 8;                                           // most splits of a level (fwd_split_plan)
struct CombineLevels { const float* part[MAXLV]; float* out[MAXLV]; int S[MAXLV]; int n; };

__global__ void __launch_bounds__(256)
k_corr_fwd_combine(CombineLevels lv, int B, int HW, CellLayout lay)
{
    __shared__ float tile[CB_PIX * (CELLS + 1)];
    const int L = blockIdx.z, b = blockIdx.y, p0 = blockIdx.x * CB_PIX;
    const float* part = lv.part[0]; float* out = lv.out[0]; int S = lv.S[0];
#pragma unroll
    for (int l = 1; l < MAXLV; ++l)
        if (L == l) { part = lv.part[l]; out = lv.out[l]; S = lv.S[l]; }
    if (S == 1) return;                                              // written in place by the segment kernel
    const int np = HW - p0 < CB_PIX ? HW - p0 : CB_PIX, nel4 = np * CELLS / 4;   // CB_PIX * 289 floats = 1156 whole float4
    const size_t plane = (size_t)B * HW * CELLS;
    const float* src = part + ((size_t)b * HW + p0) * CELLS;          // dword-aligned 16-byte accesses (H*W*289 need not be a multiple of 4)
    float* dst = out + (size_t)b * lay.bs;
    auto sum_at = [&](int e) {
        f32x4 v[CB_MAXS];
#pragma unroll
        for (int s = 0; s < CB_MAXS; ++s)
            v[s] = s < S ? f32x4(*reinterpret_cast<const f32x4u*>(src + s * plane + 4 * e)) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 a = v[0];
#pragma unroll
        for (int s = 1; s < CB_MAXS; ++s) a = s < S ? a + v[s] : a;  // ascending split order
        return a;
    };
    const int tail0 = nel4 * 4, ntail = np * CELLS - tail0;          // np < 16 (last block of the map): up to 3 floats left
    auto sum_tail = [&](int e) {
        const float* sp = part + ((size_t)b * HW + p0) * CELLS + e;
        float a = sp[0];
        for (int s = 1; s < S; ++s) a += sp[s * plane];
        return a;
    };
    if (lay.cs == 1) {                                               // reference layout: element-wise
        float* d = dst + (size_t)p0 * CELLS;
        for (int e = threadIdx.x; e < nel4; e += 256) *reinterpret_cast<f32x4u*>(d + 4 * e) = sum_at(e);
        if ((int)threadIdx.x < ntail) dst[(size_t)p0 * CELLS + tail0 + threadIdx.x] = sum_tail(tail0 + threadIdx.x);
        return;
    }
    for (int e = threadIdx.x; e < nel4; e += 256) {
        const f32x4 a = sum_at(e);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int f = 4 * e + k, p = f / CELLS, c = f - p * CELLS;
            tile[p * (CELLS + 1) + c] = a[k];
        }
    }
    if ((int)threadIdx.x < ntail) {
        const int f = tail0 + threadIdx.x, p = f / CELLS, c = f - p * CELLS;
        tile[p * (CELLS + 1) + c] = sum_tail(f);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < CELLS * CB_PIX; e += 256) {
        const int c = e / CB_PIX, p = e - c * CB_PIX;
        if (p < np) dst[(size_t)c * lay.cs + (size_t)(p0 + p) * lay.ps] = tile[p * (CELLS + 1) + c];
    }
}

#ifdef D2T_ENV_KNOBS
#include "../../lab/csrc/d2t_corr_fwd_segx.inc"       // scan builds only: the one- / two-tile forward the band-split kernel replaced
#endif

bool corr_fwd_supported(int B, int C, int H, int W, int d, int s)
{
    if (d != DT || s != 1 || B < 1 || C < 1 || H < 1 || W < WC) return false;
    const long long blocks = 1LL * B * ((H + TP - 1) / TP) * ((W + TP - 1) / TP);
    // 32-bit byte offsets inside one batch item (feature planes incl. the chunks staged past C; output)
    const bool offsets_fit = (C + 8LL * SG_KC) * H * W * 4 < 0x7ffffff0LL && 1LL * H * W * CELLS * 4 < 0x7ffffff0LL;
    return blocks <= 0x7fffffffLL && offsets_fit;
}

// Channel-split plan of a small-grid call (fewer than 192 five-tile segments): S[l] splits of cps[l] 16-channel chunks
// per level, S = 0 everywhere when the call does not split.  Splitting pays when a workgroup keeps enough chunks to
// amortise the five-tile kernel's prologue and epilogue (~12 us): measured crossover at B = 1, 38 x 75 between 512 and
// 768 channels (one-tile kernel 27 / 48 / 117 us at C = 512 / 1024 / 2048).
struct SplitPlan { int S[MAXLV], cps[MAXLV]; bool on; size_t ws_bytes; };

static SplitPlan fwd_split_plan(int nl, const int* C, int B, int H, int W)
{
    SplitPlan p;
    p.on = false; p.ws_bytes = 0;
    for (int l = 0; l < MAXLV; ++l) { p.S[l] = 0; p.cps[l] = 0; }
    const int tiles_i = (H + TP - 1) / TP, tiles_j = (W + TP - 1) / TP;
    const int nseg = (tiles_i + SG_NU - 1) / SG_NU;
    const long long segs = 1LL * B * tiles_j * nseg;
    if (segs >= 192 || nl < 1 || nl > MAXLV) return p;
    int budget = (int)(256 / segs);                                  // splits in all: one round of workgroups, one per CU
    budget = budget > nl - 1 + CB_MAXS ? nl - 1 + CB_MAXS : budget;  // (k_corr_fwd_combine adds at most CB_MAXS planes)
    int chunks[MAXLV], total = 0, most = 0;
    for (int l = 0; l < nl; ++l) { chunks[l] = (C[l] + SG_KC - 1) / SG_KC; total += chunks[l]; most = chunks[l] > most ? chunks[l] : most; }
    if (budget < nl + 1 || total < 40 || most < 40) return p;        // nothing to gain below ~640 channels
    int S[MAXLV];
    for (int l = 0; l < nl; ++l) S[l] = 1;
    for (int used = nl; used < budget; ++used) {                     // next split to the level with the longest workgroups
        int best = 0;
        for (int l = 1; l < nl; ++l)
            if (1LL * chunks[l] * S[best] > 1LL * chunks[best] * S[l]) best = l;
        if ((chunks[best] + S[best]) / (S[best] + 1) < 8 || S[best] >= CB_MAXS) break;   // keep >= 8 chunks per workgroup
        ++S[best];
    }
    for (int l = 0; l < nl; ++l) {
        p.cps[l] = (chunks[l] + S[l] - 1) / S[l];
        p.S[l] = (chunks[l] + p.cps[l] - 1) / p.cps[l];              // no empty split
        p.ws_bytes += (size_t)p.S[l] * B * H * W * CELLS * sizeof(float);
    }
    p.on = true;
    return p;
}

size_t corr_fwd_levels_ws_bytes(int nl, const int* C, int B, int H, int W) { return fwd_split_plan(nl, C, B, H, W).ws_bytes; }
size_t corr_fwd_ws_bytes(int B, int C, int H, int W, int d, int s)
{
    return corr_fwd_supported(B, C, H, W, d, s) ? corr_fwd_levels_ws_bytes(1, &C, B, H, W) : 0;
}

// nl problems of one spatial shape (B, H, W), level l with C[l] channels.  Small grids (the model's
// B = 1 pairs) go out as ONE launch, heaviest level first; grids that fill the chip by themselves are
// launched one after the other.
int corr_fwd_levels_f32(int nl, const float* const* fm0, const float* const* fm1, float* const* out, const int* C,
                        int B, int H, int W, CellLayout lay, hipStream_t st, void* ws, size_t ws_bytes)
{
    const int tiles_i = (H + TP - 1) / TP, tiles_j = (W + TP - 1) / TP;
    const int nseg = (tiles_i + SG_NU - 1) / SG_NU;
    const long long strip_blocks = 1LL * B * tiles_j * nseg;
#ifdef D2T_ENV_KNOBS
    if (lab_env_int("D2T_BAND_CFG", -1) > 0) {       // scan builds: the band-split kernel on any grid (lab/tools/band_scan.py)
        if (lab_env_int("D2T_BAND_LEVEL_LAUNCHES", 0)) {                 // A/B: one launch per level
            for (int l = 0; l < nl; ++l) {
                const int rc = corr_fwd_band_f32(lab_env_int("D2T_BAND_CFG", -1), 1, fm0 + l, fm1 + l, out + l, C + l, B, H, W, lay, st);
                if (rc != D2T_OK) return rc;
            }
            return D2T_OK;
        }
        return corr_fwd_band_f32(lab_env_int("D2T_BAND_CFG", -1), nl, fm0, fm1, out, C, B, H, W, lay, st);
    }
#endif
    if (strip_blocks >= 192) {                       // enough segments to give (nearly) every CU one
        for (int l = 0; l < nl; ++l)
            hipLaunchKernelGGL(k_corr_fwd_seg, dim3((int)strip_blocks), dim3(SG_THREADS), 0, st,
                               fm0[l], fm1[l], out[l], C[l], H, W, tiles_i, tiles_j, nseg, lay);
        return launch_status();
    }
    const SplitPlan plan = fwd_split_plan(nl, C, B, H, W);
    if (plan.on && ws && ws_bytes >= plan.ws_bytes) {                // channel split (a caller without workspace gets the kernels below)
        SplitLevels sl;
        CombineLevels cl;
        sl.n = cl.n = nl;
        float* w = static_cast<float*>(ws);
        int end = 0;
        for (int l = 0; l < MAXLV; ++l) {
            const int src = l < nl ? l : nl - 1;
            sl.fm0[l] = fm0[src]; sl.fm1[l] = fm1[src]; sl.C[l] = C[src]; sl.S[l] = plan.S[src]; sl.cps[l] = plan.cps[src];
            if (l < nl) {
                sl.part[l] = plan.S[l] == 1 ? out[l] : w;            // an unsplit level writes its output directly
                if (plan.S[l] > 1) w += (size_t)plan.S[l] * B * H * W * CELLS;
                end += (int)strip_blocks * plan.S[l];
            } else {
                sl.part[l] = sl.part[nl - 1];
            }
            sl.wg_end[l] = end;
            cl.part[l] = sl.part[l]; cl.out[l] = out[src]; cl.S[l] = plan.S[src];
        }
        hipLaunchKernelGGL(k_corr_fwd_seg_split, dim3(end), dim3(SG_THREADS), 0, st, sl, B, H, W, tiles_i, tiles_j, nseg, lay);
        int rc = launch_status();
        if (rc != D2T_OK) return rc;
        hipLaunchKernelGGL(k_corr_fwd_combine, dim3((H * W + CB_PIX - 1) / CB_PIX, B, nl), dim3(256), 0, st, cl, B, H * W, lay);
        return launch_status();
    }
    // small grids: the tile's WINDOW split over workgroups (d2t_corr_fwd_band.hip), one launch per level
    if (const int cfg = corr_fwd_band_config(B, H, W)) {
        if (lab_env_int("D2T_BAND_LEVEL_LAUNCHES", 0)) {                 // scan builds, A/B: one launch per level
            for (int l = 0; l < nl; ++l) {
                const int rc = corr_fwd_band_f32(cfg, 1, fm0 + l, fm1 + l, out + l, C + l, B, H, W, lay, st);
                if (rc != D2T_OK) return rc;
            }
            return D2T_OK;
        }
        return corr_fwd_band_f32(cfg, nl, fm0, fm1, out, C, B, H, W, lay, st);
    }
#ifndef D2T_ENV_KNOBS
    return D2T_EINVAL;                               // not reached: corr_fwd_band_config names a shape for every grid of the envelope
#else                                                // scan builds, D2T_BAND_CFG=0: the one- / two-tile kernels of rounds 1-4
    int order[MAXLV];
    for (int l = 0; l < nl; ++l) order[l] = l;
    for (int a = 0; a < nl; ++a)                     // heaviest first (nl <= 4)
        for (int b2 = a + 1; b2 < nl; ++b2)
            if (C[order[b2]] > C[order[a]]) { const int t = order[a]; order[a] = order[b2]; order[b2] = t; }
    const bool two = 1LL * B * tiles_j * ((tiles_i + 1) / 2) >= 160;   // medium grids: segments of 2 p-tiles (B = 2, 38 x 63 / 75, C = 2048:
                                                                        // 160 us against 181 us with one-tile workgroups, lab/tools/levels_cost.py)
    const int ns = two ? (tiles_i + 1) / 2 : tiles_i;
    const int per_level = B * tiles_j * ns;
    // One launch per level (D2T_EXP_LEVEL_LAUNCHES 0: all levels in one launch).  The tracker's three B = 1 levels are 190 one-tile
    // workgroups each, one per CU by LDS: three launches 179 us, one 570-workgroup launch 273 us (lab/tools/levels_cost.py) -- the
    // heavy level's workgroups queue behind the light ones'.
#ifndef D2T_EXP_LEVEL_LAUNCHES
#define D2T_EXP_LEVEL_LAUNCHES 1
#endif
    const int launches = D2T_EXP_LEVEL_LAUNCHES ? nl : 1, per_launch = D2T_EXP_LEVEL_LAUNCHES ? 1 : nl;
    for (int q = 0; q < launches; ++q) {
        FwdLevels lv;
        lv.n = per_launch;
        for (int l = 0; l < MAXLV; ++l) {
            const int src = order[q + (l < per_launch ? l : per_launch - 1)];
            lv.fm0[l] = fm0[src]; lv.fm1[l] = fm1[src]; lv.out[l] = out[src]; lv.C[l] = C[src];
            lv.wg_end[l] = per_level * ((l < per_launch ? l : per_launch - 1) + 1);
        }
        if (two)
            hipLaunchKernelGGL((k_corr_fwd_segx<2, 4>), dim3(per_level * per_launch), dim3(SegX<2, 4>::THREADS), 0, st,
                               lv, H, W, tiles_i, tiles_j, ns, lay);
        else                                         // small batches (B = 1 pairs): one p-tile per workgroup
            hipLaunchKernelGGL((k_corr_fwd_segx<1, 4, true>), dim3(per_level * per_launch), dim3(SegX<1, 4, true>::THREADS), 0, st,
                               lv, H, W, tiles_i, tiles_j, ns, lay);
    }
    return launch_status();
#endif
}

int corr_fwd_f32(const float* fm0, const float* fm1, float* out, int B, int C, int H, int W, int, int,
                 void* ws, size_t ws_bytes, hipStream_t st)
{
    const CellLayout lay{CELLS, 1, 1LL * H * W * CELLS};
    return corr_fwd_levels_f32(1, &fm0, &fm1, &out, &C, B, H, W, lay, st, ws, ws_bytes);
}

// ====================================================================================
// Backward.  The product's tuned backward is the 8-wave column-strip kernel of d2t_corr_bwd8.hip (reference layout; the
// channel-major gradient of the tracker's concat buffer is re-laid into the caller's workspace first).  The round-2 16-wave
// strip kernels live in lab/d2t_corr_bwd16.inc and are compiled into the lab build only.
// ====================================================================================
#ifdef D2T_LAB_KERNELS
#include "../../lab/csrc/d2t_corr_bwd16.inc"
#endif

// The tuned backward = the 8-wave strip kernel: d_max 8, stride 1, maps at least 17 rows high (five 4-row tiles alive at a time).
// Everything else (and f64) is the second tier's business (d2t_corr_blocked.hip), bit-identical to the anchor kernels.
bool corr_bwd_supported(int B, int C, int H, int W, int d, int s)
{
    return d == DT && s == 1 && H >= 1 && corr_bwd8_supported(B, C, H, W, CELLS, 1);
}

size_t corr_bwd_ws_bytes(int, int, int, int, int, int) { return 0; }

// The channel-major gradient of the tracker's concat buffer, one level: cell c of pixel p of item b at gout[b*bs + c*HW + p].  The
// 8-wave strip kernels read gradOut in the reference's layout ((2d+1)^2 contiguous cells per pixel); for this layout the levels call
// first re-lays every level's gradient into the caller's workspace -- 32 x 32 tiles through LDS, ~3 MB per level and item -- and then
// runs the same kernels as the per-level calls (round 4: the 16-wave kernel that reads the channel-major layout directly took
// 304 us for the tracker's three B = 1 levels against 195 us for three separate reference-layout calls).
struct RelayLevels { const float* src[MAXLV]; float* dst[MAXLV]; };

__global__ void __launch_bounds__(256)
k_corr_relay_cells(RelayLevels lv, int HW, long long bs)
{
    __shared__ float tile[32][33];
    const int L = blockIdx.z, b = blockIdx.y;
    const float* src = lv.src[0]; float* dst = lv.dst[0];
#pragma unroll
    for (int l = 1; l < MAXLV; ++l)
        if (L == l) { src = lv.src[l]; dst = lv.dst[l]; }
    src += (size_t)b * bs;
    dst += (size_t)b * HW * CELLS;
    constexpr int ctiles = (CELLS + 31) / 32;
    const int pt = blockIdx.x / ctiles, ct = blockIdx.x - pt * ctiles;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                     // read: cells down, pixels across (contiguous)
        const int c = ct * 32 + ty + 8 * r, p = pt * 32 + tx;
        tile[ty + 8 * r][tx] = c < CELLS && p < HW ? src[(size_t)c * HW + p] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {                                     // write: pixels down, cells across (contiguous)
        const int p = pt * 32 + ty + 8 * r, c = ct * 32 + tx;
        if (c < CELLS && p < HW) dst[(size_t)p * CELLS + c] = tile[tx][ty + 8 * r];
    }
}

size_t corr_bwd_levels_ws_bytes(int nl, const int* C, int B, int H, int W, CellLayout lay)
{
    if (lay.cs == 1) return 0;                                        // reference layout: read in place
    for (int l = 0; l < nl; ++l)
        if (!corr_bwd8_supported(B, C[l], H, W, CELLS, 1)) return 0;
    return (size_t)nl * B * H * W * CELLS * sizeof(float);
}

// nl problems of one spatial shape, every level launched by itself with the 8-wave kernel (channel blocks sized to the grid).
// Channel-major layout: needs the workspace of corr_bwd_levels_ws_bytes (the C ABI falls back to the layout-aware anchor kernel
// without it).  bwd_variant: 0 = the product's choice; the lab build (-DD2T_LAB_KERNELS) adds 1 = 16-wave strip kernel,
// 3 = bf16x3, 4 = strips 8 pixels wide, 5 = strips 4 pixels wide demanded.
int corr_bwd_levels_f32(int nl, const float* const* gout, const float* const* fm0, const float* const* fm1,
                        float* const* g0, float* const* g1, const int* C, int B, int H, int W, CellLayout lay, hipStream_t st,
                        int bwd_variant, void* ws, size_t ws_bytes)
{
#ifndef D2T_LAB_KERNELS
    (void)bwd_variant;
#endif
    const size_t relay =
#ifdef D2T_LAB_KERNELS
        bwd_variant == 1 ? 0 :
#endif
        corr_bwd_levels_ws_bytes(nl, C, B, H, W, lay);
    if (relay && ws && ws_bytes >= relay) {                           // channel-major gradient: re-lay, then the reference-layout kernels
        RelayLevels rl;
        const float* gref[MAXLV];
        const size_t per = (size_t)B * H * W * CELLS;
        for (int l = 0; l < MAXLV; ++l) {
            const int src = l < nl ? l : nl - 1;
            rl.src[l] = gout[src];
            rl.dst[l] = static_cast<float*>(ws) + src * per;
            gref[l] = rl.dst[l];
        }
        hipLaunchKernelGGL(k_corr_relay_cells, dim3(((H * W + 31) / 32) * ((CELLS + 31) / 32), B, nl), dim3(256), 0, st, rl, H * W, lay.bs);
        const int rc = launch_status();
        if (rc != D2T_OK) return rc;
        const CellLayout ref{CELLS, 1, 1LL * H * W * CELLS};
        return corr_bwd_levels_f32(nl, gref, fm0, fm1, g0, g1, C, B, H, W, ref, st, bwd_variant, nullptr, 0);
    }
#ifdef D2T_LAB_KERNELS
    const int tiles_i = (H + TP - 1) / TP, tiles_j = (W + TP - 1) / TP;
    int small[MAXLV], ns = 0;
    for (int l = 0; l < nl; ++l) {
        const long long wide = 2LL * B * tiles_j * ((C[l] + ST_CH - 1) / ST_CH);   // workgroups of 16 waves x 16 channels
        if (bwd_variant == 3 && corr_bwd8bf_supported(B, C[l], H, W, lay.ps, lay.cs)) {  // bf16 matrix pipe, operands split in three
            const int rc = corr_bwd8bf_f32(gout[l], fm0[l], fm1[l], g0[l], g1[l], B, C[l], H, W, st);
            if (rc != D2T_OK) return rc;
        } else if (bwd_variant == 4 && corr_bwd8w_supported(B, C[l], H, W, lay.ps, lay.cs)) {   // strips 8 pixels wide x 128 channels
            const int rc = corr_bwd8w_f32(gout[l], fm0[l], fm1[l], g0[l], g1[l], B, C[l], H, W, st);
            if (rc != D2T_OK) return rc;
        } else if (bwd_variant != 1 && corr_bwd8_supported(B, C[l], H, W, lay.ps, lay.cs)) {
            const int rc = corr_bwd8_f32(gout[l], fm0[l], fm1[l], g0[l], g1[l], B, C[l], H, W, st, 0);
            if (rc != D2T_OK) return rc;
        } else if (wide >= 100 && lay.cs == 1)
            hipLaunchKernelGGL(k_corr_bwd_strip<true>, dim3(2 * B * tiles_j, (C[l] + ST_CH - 1) / ST_CH), dim3(ST_THREADS), 0, st,
                               gout[l], fm0[l], fm1[l], g0[l], g1[l], B, C[l], H, W, tiles_i, tiles_j, lay);
        else if (wide >= 100)
            hipLaunchKernelGGL(k_corr_bwd_strip<false>, dim3(2 * B * tiles_j, (C[l] + ST_CH - 1) / ST_CH), dim3(ST_THREADS), 0, st,
                               gout[l], fm0[l], fm1[l], g0[l], g1[l], B, C[l], H, W, tiles_i, tiles_j, lay);
        else
            small[ns++] = l;
    }
    if (ns == 0) return launch_status();
    for (int a = 0; a < ns; ++a)                     // heaviest first (ns <= 4)
        for (int b2 = a + 1; b2 < ns; ++b2)
            if (C[small[b2]] > C[small[a]]) { const int t = small[a]; small[a] = small[b2]; small[b2] = t; }
    BwdLevels lv;
    lv.n = ns;
    int end = 0;
    for (int l = 0; l < MAXLV; ++l) {
        const int src = small[l < ns ? l : ns - 1];
        lv.gout[l] = gout[src]; lv.fm0[l] = fm0[src]; lv.fm1[l] = fm1[src]; lv.g0[l] = g0[src]; lv.g1[l] = g1[src];
        lv.C[l] = C[src];
        if (l < ns) end += 2 * B * tiles_j * ((C[src] + 63) / 64);
        lv.wg_end[l] = end;
    }
    if (end <= 256)                                                  // at most one per CU: 4 MFMA waves (64 channels)
        hipLaunchKernelGGL((k_corr_bwd_strip_n<4, 4>), dim3(end), dim3(512), 0, st, lv, B, H, W, tiles_i, tiles_j, lay);   // + 4 waves that only produce G
    else                                                             // small grids: 4 waves (64 channels) per workgroup
        hipLaunchKernelGGL((k_corr_bwd_strip_n<4, 0>), dim3(end), dim3(256), 0, st, lv, B, H, W, tiles_i, tiles_j, lay);
    return launch_status();
#else
    if (lay.cs != 1) return D2T_EWS;                                  // channel-major without its workspace: the C ABI does not get here
    for (int l = 0; l < nl; ++l) {
        if (!corr_bwd8_supported(B, C[l], H, W, lay.ps, lay.cs)) return D2T_EINVAL;
        const int rc = corr_bwd8_f32(gout[l], fm0[l], fm1[l], g0[l], g1[l], B, C[l], H, W, st, 0);
        if (rc != D2T_OK) return rc;
    }
    return D2T_OK;
#endif
}

int corr_bwd_f32(const float* gout, const float* fm0, const float* fm1, float* g0, float* g1,
                 int B, int C, int H, int W, int, int, void*, hipStream_t st, int bwd_variant)
{
    const CellLayout lay{CELLS, 1, 1LL * H * W * CELLS};
    return corr_bwd_levels_f32(1, &gout, &fm0, &fm1, &g0, &g1, &C, B, H, W, lay, st, bwd_variant, nullptr, 0);
}

}}  // namespace d2t::tuned
