// d2t_corr_fwd_band.hip -- gfx950-tuned f32 PointwiseCorrelation forward for SMALL and MEDIUM grids (d_max = 8, stride 1): the
// model's B = 1 pairs (correlation_tracker.py:57-70), BASELINE config 2, B <= 5 at the model's maps.  Bit-identical to the reference
// (pointwise_correlation_cuda.cu:84-107): every output cell is ONE ascending-channel chain of v_mfma_f32_16x16x4_f32,
// exactly as in d2t_corr_tuned.hip -- parallelism comes from splitting a tile's WINDOW over workgroups, never its channels.
//
// Why a second forward.  A 4 x 4 p-tile has 6 tile-groups (16 window slot groups = 4 N-tiles each); 190 tiles at 38 x 75
// are 1,140 (tile, tile-group) TASKS of 16 MFMAs per 16 channels.  The one-tile-per-workgroup kernel
// (k_corr_fwd_segx<1,4,true>, rounds 1-4) gave a CU all 6 tasks of a tile and left 66 CUs idle; its time was the L1-miss
// path of ONE CU streaming a 19 x 20 window per 16 channels -- 0.92 us per chunk where the matrix work is 0.32 us.
// Here a workgroup is
//     (tile row u, block of TW tiles side by side, band-set q of NB consecutive tile-groups)
// with one task per wave (or two waves per task, HT = 2, each multiplying two of its four N-tiles): the block's window
// is 4 TW + 16 columns wide, so a window row is ONE run of 128 / 144 bytes for 4 / 5 tiles instead of 80 bytes per tile,
// a band-set needs only its own 4 / 7 window rows, and TW is chosen so that the grid is ONE workgroup per CU (38 x 63:
// TW 4 -> 240 workgroups; 38 x 75: TW 5 -> 240).
//
// What the in-kernel clocks said on the way (lab/tools/band_stamps.py, profiles/r05_*_band_stamps.txt):
//   * the CU's address unit takes ~55 cycles per LDS-DMA instruction of 64 x 16 bytes whatever wave issues it, and a
//     wave that waits to issue one issues no MFMA: with the DMA dealt to the computing waves a chunk took 0.61 us
//     where the matrix work is 0.43 us.  So WL dedicated LOADER waves issue every DMA instruction; the computing
//     waves' stream is LDS reads and MFMAs only.  (4-byte-per-lane DMA -- 256 contiguous bytes per instruction -- is
//     no cheaper per instruction: 2.4x slower in all.)
//   * code that runs once per wave is not free: the first straight-from-the-registers epilogue (16 exec-masked dword stores, 4 KB of
//     straight-line code fetched cold by every wave of the chip at the same moment) took 4.3 k cycles of a 30 k-cycle kernel; a scatter
//     into a per-task LDS patch + a store loop took 11.8 k with one wave per task.  The epilogue is 16-byte stores straight from the
//     accumulators where a wave owns a whole task in the reference layout, and the LDS patch -- consecutive lanes on consecutive
//     cells / pixels -- for two-wave tasks and for the channel-major layout.
//   * at the METRIC shape (B = 8) this kernel loses to the 5-tile segments of d2t_corr_tuned.hip: 63.7-96.8 against 47 us, its matrix
//     work alone 35 us (profiles/r05_d_band_ablation_headline_shape.txt); it serves every grid below that kernel's threshold.
// Staging: ring of RING chunk images filled by LDS-DMA, counted vmcnt waits on the loader side, ONE barrier per chunk
// for both roles, placed in the middle of a chunk's MFMAs so that the next chunk's fragments are fetched under its
// second half.  A band's cells of a pixel are a contiguous run of its 17 x 17 block; the structural zeros (cj = 16,
// ci = 16, displaced columns outside the map) are written by the lanes that own those slots, window rows outside the
// map ("orphan rows") by the first / last band-set of the tile row.
#include "d2t_corr_common.hpp"

namespace d2t { namespace tuned {

#ifdef D2T_BAND_STAMPS
// in-kernel stamps of the stamp build (-DD2T_ENV_KNOBS -DD2T_BAND_STAMPS) (lab/tools/band_scan.py --stamps): [workgroup][16] clock reads of compute wave 0 (0-4) and of the first
// loader wave (8-13), s_memrealtime at entry / exit (14, 15).  The product library is built without D2T_ENV_KNOBS: no stamp executes there.
__device__ unsigned long long* band_stamps;
__device__ int band_dbg;                          // ablation bits: 1 no LDS-DMA, 2 no MFMA, 4 no fragment reads, 8 no stores
#define BAND_DBG (band_dbg)
#define BAND_STAMP(cond, i)                                                                           \
    do {                                                                                              \
        if (band_stamps && (cond) && (threadIdx.x & 63) == 0) {                                       \
            unsigned long long t_;                                                                    \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            band_stamps[(size_t)blockIdx.x * 16 + (i)] = t_;                                          \
        }                                                                                             \
    } while (0)
#define BAND_STAMP_RT(cond, i)                                                                        \
    do {                                                                                              \
        if (band_stamps && (cond) && (threadIdx.x & 63) == 0) {                                       \
            unsigned long long t_;                                                                    \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");           \
            band_stamps[(size_t)blockIdx.x * 16 + (i)] = t_;                                          \
        }                                                                                             \
    } while (0)
#define BAND_CLK(var) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#else
#define BAND_STAMP(cond, i)
#define BAND_STAMP_RT(cond, i)
#define BAND_CLK(var)
#define BAND_DBG 0
#endif

namespace {

// the levels of one launch: level l = workgroups [l pad, l pad + per)
struct BandLevels { const float* fm0[MAXLV]; const float* fm1[MAXLV]; float* out[MAXLV]; int C[MAXLV]; int per, pad; };

constexpr int imin(int a, int b) { return a < b ? a : b; }
constexpr int imax(int a, int b) { return a > b ? a : b; }
constexpr int MAXG = WR * NCG;                                       // 95 slot groups of an unclipped tile window
constexpr int PATCH = 4 * CW + 1;                                    // floats per pixel in a task's output patch: its 16 groups span <= 4 window rows x 17 cells

// TW tiles side by side, NB tile-groups per workgroup, HT waves per (tile, tile-group) task (2: each wave multiplies two of the
// task's four N-tiles), KC channels per staged chunk, RING chunk images, WL loader waves.
template <int TW, int NB, int HT, int KC, int RING, int WL>
struct Band {
    static constexpr int NG = TW + 4;                                // 16-byte column groups of the block's window
    static constexpr int WC = TW * NB * HT;                          // compute waves
    static constexpr int WAVES = WC + WL, THREADS = 64 * WAVES;      // + WL loader waves: they issue every LDS-DMA instruction
    static constexpr int NS = 4 / HT;                                // N-tiles (accumulators) per compute wave
    static constexpr int NQ = (6 + NB - 1) / NB;                     // band-sets per tile row and block
    static constexpr int rows_of(int q) { return (imin(16 * (q + 1) * NB, MAXG) - 1) / NCG - (16 * q * NB) / NCG + 1; }
    static constexpr int max_rows() { int m = 0; for (int q = 0; q < NQ; ++q) m = imax(m, rows_of(q)); return m; }
    static constexpr int NROWS = max_rows();                         // window rows a band-set stages at most
    static constexpr int P = (NROWS * NG + 15) / 16 * 16;            // slots per channel plane: plane stride = 0 mod 64 dwords
    static constexpr int BPL = 4 * P, APL = 16 * TW;                 // floats per channel: FM1 window part / FM0 pixels
    static constexpr int BUF = KC * (BPL + APL);                     // floats per staged chunk
    static constexpr int BI = KC * P / 64, AI = (KC * 4 * TW + 63) / 64;   // DMA wave-instructions (64 pieces of 16 bytes) per chunk: FM1 / FM0
    static constexpr int NDMA = (BI + AI + WL - 1) / WL;             // per loader wave and chunk (surplus ones are parked)
    static constexpr int DUMMY = 256;                                // floats: where parked DMA instructions land
    static constexpr int RINGF = RING * BUF + DUMMY;
    static constexpr int EPI = WC * 16 * PATCH;                      // epilogue: a patch of 16 pixels per compute wave (aliases the ring)
    static constexpr int LDS = RINGF > EPI ? RINGF : EPI;
    static constexpr int INFLIGHT = (RING - 2) * NDMA;               // a loader's DMA instructions that may be outstanding at a barrier
    static_assert(WAVES <= 16 && KC % 8 == 0 && (KC * P) % 64 == 0 && RING >= 3 && (HT == 1 || HT == 2), "shape");
    static_assert(INFLIGHT <= 63, "vmcnt is a 6-bit counter");
    static_assert(LDS * 4 <= 160 * 1024, "LDS budget");
};

// One workgroup: batch item b, tile row u, tile columns [TW vb, TW vb + TW), tile-groups [NB q, NB q + NB).
template <int TW, int NB, int HT, int KC, int RING, int WL>
__global__ void __launch_bounds__((TW * NB * HT + WL) * 64)
k_corr_fwd_band(BandLevels lv, int H, int W, int tiles_i, int tiles_j, int blocks_j, CellLayout lay)
{
    // level of this workgroup: the levels' workgroups are dealt in blocks of lv.pad (a multiple of 8: a level's workgroups keep the XCD
    // pattern of a launch of their own), heaviest level first
    const int level = blockIdx.x / lv.pad, local = blockIdx.x - level * lv.pad;
    if (local >= lv.per) return;                                     // padding (whole workgroup; nothing before this line syncs)
    const float* __restrict__ fm0 = lv.fm0[level];
    const float* __restrict__ fm1 = lv.fm1[level];
    float* __restrict__ out = lv.out[level];
    const int C = lv.C[level];
    using S = Band<TW, NB, HT, KC, RING, WL>;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    BAND_STAMP(wave == 0, 0); BAND_STAMP_RT(wave == 0, 14); BAND_STAMP(wave == S::WC, 8);
    // logical id (band-set innermost: the band-sets of a block share its FM0 pixels and overlap in window rows; give every
    // XCD a contiguous run of the logical order)
    int id = xcd_remap(local, lv.per);
    const int q = id % S::NQ; id /= S::NQ;
    const int vb = id % blocks_j; id /= blocks_j;
    const int u = id % tiles_i, b = id / tiles_i;

    const int HW = H * W;
    const int wa = 4 * u - DT > 0 ? 4 * u - DT : 0;                  // the tile row's window rows inside the map
    const int wb = 4 * u + TP + DT - 1 < H ? 4 * u + TP + DT - 1 : H;
    const int ng = (wb - wa) * NCG;                                  // slot groups of a tile of this row (> 0)
    const int q_last = ((ng + 15) / 16 - 1) / NB;
    if (q > q_last) return;                                          // the whole workgroup: nothing before this line syncs
    const int g_lo = 16 * NB * q, g_hi = g_lo + 16 * NB < ng ? g_lo + 16 * NB : ng;   // this band-set's groups [g_lo, g_hi)
    const int R0 = wa + g_lo / NCG, nrows = wa + (g_hi - 1) / NCG - R0 + 1;           // rows it stages
    const int v0 = TW * vb, j0 = 4 * v0, colL = j0 - DT;

    const unsigned plane_bytes = (unsigned)C * HW * 4u;
    const size_t item = (size_t)b * C * HW;
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fm1 + item), 0, plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fm0 + item), 0, plane_bytes, 0x00020000);
    float* outb = out + (size_t)b * lay.bs;
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(outb, 0, (unsigned)HW * CELLS * 4u, 0x00020000);

    // ---- this wave's role.  Waves [0, WC) compute: task = (tile v of the block, tile-group T), HT waves per task (wave half hs takes
    // N-tiles [NS hs, NS hs + NS)).  Waves [WC, WC + WL) are LOADERS: they issue every LDS-DMA instruction of the workgroup.  (With
    // the DMA instructions dealt to the computing waves -- the first version -- every wave issued its share right behind the chunk's
    // barrier, all of them queued at the CU's address unit together and no MFMA was issued meanwhile; a wave's stream is in order.)
    const bool loader = wave >= S::WC;                               // wave-uniform
    const int task = wave / HT, hs = wave - task * HT;
    const int v = task % TW, T = NB * q + (task / TW) % NB;
    const bool t_on = !loader && 16 * T < ng && v0 + v < tiles_j;    // wave-uniform
    int gi = 16 * T + n;
    const bool lane_on = t_on && gi < ng;
    gi = gi < ng ? gi : ng - 1;
    gi = gi < g_lo ? g_lo : gi;                                      // (a dead wave still reads a staged slot)
    const int rho = wa + gi / NCG, cg = gi - (gi / NCG) * NCG;       // displaced row, column group inside the tile's window
    const int l_off = ((rho - R0) * S::NG + v + cg) * 4 + S::NS * hs + g * S::BPL;
    const int a_off = KC * S::BPL + g * S::APL + (n >> 2) * (4 * TW) + 4 * v + (n & 3);

    typedef float frag_t __attribute__((ext_vector_type(S::NS)));
    f32x4 acc[S::NS];
#pragma unroll
    for (int s = 0; s < S::NS; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nchunks = (C + KC - 1) / KC;

    // Ring protocol (both roles meet at ONE barrier per chunk).  Barrier #k publishes chunk k: every loader has waited until at
    // most INFLIGHT = (RING - 2) NDMA of its DMA instructions are outstanding (LDS-DMA retires in order: chunks k+1 .. k+RING-2
    // may be in flight), every computing wave has its fragments of chunk k-1 in registers (lgkmcnt(0)), so behind the barrier
    // the loaders overwrite the slot of chunk k-1 with chunk k+RING-1.  The barrier sits in the MIDDLE of a chunk's MFMAs: the
    // fragments of chunk k+1 are fetched under the second half of chunk k's.  Chunk 0 alone is staged in front of barrier #0.
    if (loader) {
        // DMA plan: position k of a loader's sequence is instruction x = (wave - WC) + WL k of the chunk -- x < BI: 64 pieces of the FM1
        // image [channel][row][column group] (plane pitch P slots), BI <= x < BI + AI: 64 pieces (channel, pixel row, tile) of the
        // FM0 image, else parked (out of range: zeros into the dummy slot).  Rows past the band-set's own, pixel rows past the map
        // and tiles past the map's last tile column are parked too.
        constexpr int OOR = 0x7ffffff0;
        const int lw = wave - S::WC;
        int dv[S::NDMA];                                             // per-lane byte offset in the planes (chunk 0); the LDS side is recomputed
#pragma unroll
        for (int k = 0; k < S::NDMA; ++k) {
            const int x = lw + WL * k;
            if (x < S::BI) {                                         // wave-uniform
                const int e = x * 64 + lane;
                const int ch = e / S::P, rem = e - ch * S::P;
                const int row = rem / S::NG, cgb = rem - row * S::NG;   // pad slots: row >= NROWS >= nrows
                dv[k] = row < nrows ? (ch * HW + (R0 + row) * W + colL + 4 * cgb) * 4 : OOR;
            } else {
                const int e = (x - S::BI) * 64 + lane;
                const int ch = e / (4 * TW), rem = e - ch * (4 * TW);
                const int prow = rem / TW, pv = rem - prow * TW;
                const int pi = 4 * u + prow;
                dv[k] = x < S::BI + S::AI && ch < KC && pi < H && v0 + pv < tiles_j ? (ch * HW + pi * W + j0 + 4 * pv) * 4 : OOR;
            }
        }
        const int chunk_bytes = KC * HW * 4;
        const int dbg = BAND_DBG;
        auto stage = [&](int slot, int chunk) {                      // chunks past the end of C arrive as zeros
            if (dbg & 1) return;
            const int cb = chunk * chunk_bytes;
            // chunks nobody multiplies (behind chunk nchunks, which is fetched but unused) are parked whole: the ring is quiet when the
            // last barrier has been passed, and the epilogue may use its memory
            float* buf = chunk <= nchunks ? smem + slot * S::BUF : nullptr;
            float* dummy = smem + RING * S::BUF;
#pragma unroll
            for (int k = 0; k < S::NDMA; ++k) {
                const int x = lw + WL * k;                           // wave-uniform
                const int vo = dv[k] == OOR || !buf ? OOR : dv[k] + cb;
                float* dst = x < S::BI ? (buf ? buf + x * 256 : dummy)
                           : x < S::BI + S::AI ? (buf ? buf + KC * S::BPL + (x - S::BI) * 256 : dummy) : dummy;   // else parked (vo = OOR): keeps the count equal
                if (x < S::BI) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr)dst, 16, vo, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (lds_ptr)dst, 16, vo, 0, 0, 0);
            }
        };
        stage(0, 0);
        BAND_STAMP(lw == 0, 9);
        dma_wait_barrier<0>();                                       // barrier #0: chunk 0 has landed
        BAND_STAMP(lw == 0, 10);
#pragma unroll
        for (int p = 1; p < RING - 1; ++p) stage(p, p);              // chunks 1 .. RING-2 go out under the first MFMAs
        int slot = RING - 1;                                         // where chunk ch + RING - 1 goes
#ifdef D2T_BAND_STAMPS
        unsigned long long t_issue = 0, t_wait = 0, ta_, tb_, tc_;
#endif
        for (int ch = 0; ch < nchunks; ++ch) {
            BAND_CLK(ta_);
            stage(slot, ch + RING - 1);
            BAND_CLK(tb_);
            slot = slot + 1 == RING ? 0 : slot + 1;
            dma_wait_barrier<S::INFLIGHT>();                         // barrier #(ch+1): chunk ch+1 has landed
#ifdef D2T_BAND_STAMPS
            BAND_CLK(tc_); t_issue += tb_ - ta_; t_wait += tc_ - tb_;
#endif
        }
        BAND_STAMP(lw == 0, 11);
#ifdef D2T_BAND_STAMPS
        if (band_stamps && lw == 0 && lane == 0) { band_stamps[(size_t)blockIdx.x * 16 + 12] = t_issue; band_stamps[(size_t)blockIdx.x * 16 + 13] = t_wait; }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the parked instructions staged past the end
    } else {
        struct Frag { frag_t q[KC / 4]; float a[KC / 4]; };
        const int dbg = BAND_DBG;
        auto fetch = [&](Frag& f, const float* cur) {
            if (dbg & 4) return;
#pragma unroll
            for (int ks = 0; ks < KC / 4; ++ks) {
                f.q[ks] = *reinterpret_cast<const frag_t*>(cur + l_off + ks * 4 * S::BPL);
                f.a[ks] = cur[a_off + ks * 4 * S::APL];
            }
        };
        auto mfma = [&](const Frag& f, int ks_lo, int ks_hi) {
            if (!t_on || (dbg & 2)) return;                          // wave-uniform
#pragma unroll
            for (int ks = ks_lo; ks < ks_hi; ++ks)
#pragma unroll
                for (int s = 0; s < S::NS; ++s) acc[s] = D2T_MFMA(f.a[ks], f.q[ks][s], acc[s]);
        };
        Frag cur_f, nxt_f;
        lds_barrier();                                               // barrier #0
        BAND_STAMP(wave == 0, 1);
        fetch(cur_f, smem);
        int slot = 0;                                                // ring slot of chunk ch
        for (int ch = 0; ch < nchunks; ++ch) {
            const int next_slot = slot + 1 == RING ? 0 : slot + 1;
            mfma(cur_f, 0, KC / 8);
            lds_barrier();                                           // barrier #(ch+1)
            fetch(nxt_f, smem + next_slot * S::BUF);                 // lands under the second half
            __builtin_amdgcn_sched_barrier(0);
            mfma(cur_f, KC / 8, KC / 4);
            cur_f = nxt_f;
            slot = next_slot;
        }
        BAND_STAMP(wave == 0, 2);
    }

    // ---- epilogue.  acc[s][r] belongs to pixel (row g, column r) of the tile and window slot (rho, column 4 cg + NS hs + s): cell
    // ci = rho - i + d, cj = 4 cg + NS hs + s - r of that pixel.  A task's cells of a pixel are ONE contiguous run of its 17 x 17 block
    // (the task's groups in row-major order), so a wave scatters its accumulators into a private LDS patch [16 pixels][PATCH] -- structural
    // zeros (cj = 16, ci = 16, displaced columns outside the map) included -- and the task's waves store the runs with consecutive
    // lanes on consecutive cells.  (Straight from the registers, as at first: 12-16 exec-masked store instructions per wave, each
    // touching a dozen lines -- and 4 KB of straight-line code that every wave of the chip fetched cold at the same moment.)
    if (BAND_DBG & 8) return;                                        // (stamp builds: no stores; kernel-wide, nothing below is skipped by a part of a workgroup)
    const int gf = 16 * T, gl = (gf + 15 < ng ? gf + 15 : ng - 1);   // the task's first / last group (wave-uniform)
    if (HT == 1 && lay.cs == 1) {
        // Reference layout, one wave per task: straight from the accumulators.  The four s of a lane are adjacent cells, so a lane
        // stores ONE run per pixel -- 16 bytes for the inner column groups, the part inside [0, 16] for the first and the last one.
        // (A vector-memory instruction costs its wave ~15 cycles per line it touches: the 16 dword stores of the first version and
        // these 12 take the same 4 k cycles; an LDS patch + store loop with consecutive lanes on consecutive cells touches a
        // quarter of the lines but took 11.8 k cycles in loop overhead -- profiles/r05_b_band_stamps_*.txt.)
        const int i = 4 * u + g, ci = rho - i + DT;
        if (lane_on && i < H && ci >= 0 && ci <= 2 * DT) {
            const int jt = j0 + 4 * v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = jt + r;
                if (j < W) {
                    unsigned val[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int cj = 4 * cg + s - r, dj = j + cj - DT;
                        val[s] = __builtin_bit_cast(unsigned, ci < 2 * DT && cj < 2 * DT && dj >= 0 && dj < W ? acc[s % S::NS][r] : 0.f);
                    }
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
                    const int base = ((i * W + j) * CELLS + ci * CW + 4 * cg - r) * 4;   // byte offset of cell cj = 4 cg - r (s = 0)
                    if (cg >= 1 && cg <= 3) {
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{val[0], val[1], val[2], val[3]}, ro, base, 0, 0);
                    } else if (cg == 0) {                            // cells cj = s - r >= 0: s = r .. 3
                        if (r == 0) __builtin_amdgcn_raw_buffer_store_b128(u32x4{val[0], val[1], val[2], val[3]}, ro, base, 0, 0);
                        else if (r == 1) __builtin_amdgcn_raw_buffer_store_b96(u32x3{val[1], val[2], val[3]}, ro, base + 4, 0, 0);
                        else if (r == 2) __builtin_amdgcn_raw_buffer_store_b64(u32x2{val[2], val[3]}, ro, base + 8, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b32(val[3], ro, base + 12, 0, 0);
                    } else {                                         // cg = 4: cells cj = 16 - r + s <= 16: s = 0 .. r
                        if (r == 0) __builtin_amdgcn_raw_buffer_store_b32(val[0], ro, base, 0, 0);
                        else if (r == 1) __builtin_amdgcn_raw_buffer_store_b64(u32x2{val[0], val[1]}, ro, base, 0, 0);
                        else if (r == 2) __builtin_amdgcn_raw_buffer_store_b96(u32x3{val[0], val[1], val[2]}, ro, base, 0, 0);
                        else __builtin_amdgcn_raw_buffer_store_b128(u32x4{val[0], val[1], val[2], val[3]}, ro, base, 0, 0);
                    }
                }
            }
        }
    } else {
    // Two waves per task, or the channel-major layout: through LDS.  A task's cells of a pixel are ONE contiguous run of its 17 x 17
    // block, so its waves scatter their accumulators into the task's patch [16 pixels][PATCH] -- structural zeros included -- and
    // the runs are stored from there.
    __syncthreads();                                                 // every wave is past its last read of the ring; the ring is quiet
    BAND_STAMP(wave == 0, 5);
    const int rho_f = wa + gf / NCG;                                 // first window row of the task: patch index k = (rho - rho_f) * 17 + cj
    float* patch = smem + task * 16 * PATCH;
    if (lane_on) {
        const int i = 4 * u + g, ci = rho - i + DT;
        const int krow = (rho - rho_f) * CW;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = j0 + 4 * v + r;
#pragma unroll
            for (int s = 0; s < S::NS; ++s) {
                const int cj = 4 * cg + S::NS * hs + s - r, dj = j + cj - DT;
                if (cj >= 0 && cj <= 2 * DT)
                    patch[(4 * g + r) * PATCH + krow + cj] = ci < 2 * DT && cj < 2 * DT && dj >= 0 && dj < W ? acc[s][r] : 0.f;
            }
        }
    }
    // pixel p = (row pg, column pr) of a task's tile: its run is patch[p][kA .. kB]; cell index in the pixel's 17 x 17 block = fb + k
    auto run_of = [&](int tgf, int tgl, int pg, int pr, int& kA, int& kB, int& fb) -> bool {
        const int trf = wa + tgf / NCG, tcf = tgf - (tgf / NCG) * NCG, trl = wa + tgl / NCG, tcl = tgl - (tgl / NCG) * NCG;
        const int pi = 4 * u + pg;
        const int rlo = trf > pi - DT ? trf : pi - DT, rhi = trl < pi + DT ? trl : pi + DT;
        int cjA = 4 * tcf - pr; cjA = rlo == trf && cjA > 0 ? cjA : 0;
        int cjB = 4 * tcl + 3 - pr; cjB = rhi == trl && cjB < 2 * DT ? cjB : 2 * DT;
        kA = (rlo - trf) * CW + cjA; kB = (rhi - trf) * CW + cjB;
        fb = (trf - pi + DT) * CW;
        return pi < H && rlo <= rhi;
    };
    if (lay.cs == 1) {
        // reference layout: a pixel's cells are contiguous -- the task's own waves store its patch, consecutive lanes on consecutive cells
        if (HT > 1) __syncthreads(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // HT = 1: the patch is the wave's own
        if (t_on) {
            for (int e = hs * 64 + lane; e < 16 * PATCH; e += 64 * HT) {
                const int p = e / PATCH, k = e - p * PATCH;
                const int pj = j0 + 4 * v + (p & 3);
                int kA, kB, fb;
                if (run_of(gf, gl, p >> 2, p & 3, kA, kB, fb) && pj < W && k >= kA && k <= kB)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, patch[e]), ro,
                                                          (((4 * u + (p >> 2)) * W + pj) * lay.ps + (fb + k) * lay.cs) * 4, 0, 0);
            }
        }
    } else {
        // channel-major (the tracker's concat buffer): cell c of pixel (i, j) at c * cs + (i * W + j) -- what is contiguous is a ROW OF
        // PIXELS of one cell, so the whole workgroup walks (band, cell, pixel row, the block's 4 TW pixel columns) with consecutive
        // threads on consecutive pixels: 16 TW contiguous bytes per (cell, pixel row) instead of one line per dword
        __syncthreads();
        for (int tb = 0; tb < NB; ++tb) {
            const int tT = NB * q + tb;
            if (16 * tT >= ng) break;                                // wave-uniform
            const int tgf = 16 * tT, tgl = tgf + 15 < ng ? tgf + 15 : ng - 1;
            for (int e = tid; e < PATCH * 4 * (4 * TW); e += S::THREADS) {
                const int px = e % (4 * TW), rest = e / (4 * TW), pg = rest & 3, k = rest >> 2;
                const int pv = px >> 2, pr = px & 3, pj = j0 + px;
                int kA, kB, fb;
                if (run_of(tgf, tgl, pg, pr, kA, kB, fb) && pj < W && k >= kA && k <= kB)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, smem[((tb * TW + pv) * 16 + 4 * pg + pr) * PATCH + k]), ro,
                                                          (((4 * u + pg) * W + pj) * lay.ps + (fb + k) * lay.cs) * 4, 0, 0);
            }
        }
    }
    }
    BAND_STAMP(wave == 0, 3);
    // orphan rows: cells whose displaced row lies outside the tile row's window -- above the map (first band-set: ci < d - i), below
    // it or ci = 16 of the tile's last pixel row (last band-set: ci >= wb - i + d): structural zeros nobody computes.  Per pixel they
    // are one or two contiguous runs of whole 17-cell rows; consecutive threads take consecutive cells.
    if (q == 0 || q == q_last) {
        for (int og = 0; og < 4; ++og) {
            const int oi = 4 * u + og;
            if (oi >= H) break;
            int n_top = q == 0 ? DT - oi : 0;
            n_top = n_top < 0 ? 0 : n_top;                           // (<= 8)
            int b_lo = q == q_last ? wb - oi + DT : CW;
            b_lo = b_lo > CW ? CW : b_lo;                            // (>= 9: wb > oi)
            const int L = (n_top + CW - b_lo) * CW;                  // orphan cells per pixel of this pixel row
            for (int e = tid; e < 4 * TW * L; e += S::THREADS) {
                const int pj = e / L, k = e - pj * L;
                const int cell = k < n_top * CW ? k : b_lo * CW + (k - n_top * CW);
                const int oj = j0 + pj;
                if (oj < W) __builtin_amdgcn_raw_buffer_store_b32(0u, ro, ((oi * W + oj) * lay.ps + cell * lay.cs) * 4, 0, 0);
            }
        }
    }
#ifdef D2T_BAND_STAMPS
    if (band_stamps) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores have been acknowledged
    BAND_STAMP(wave == 0, 4); BAND_STAMP_RT(wave == 0, 15);
#endif
}

template <int TW, int NB, int HT, int KC, int RING, int WL>
int launch_band(int nl, const float* const* fm0, const float* const* fm1, float* const* out, const int* C, int B, int H, int W, CellLayout lay,
                hipStream_t st)
{
    using S = Band<TW, NB, HT, KC, RING, WL>;
    const int tiles_i = (H + TP - 1) / TP, tiles_j = (W + TP - 1) / TP, blocks_j = (tiles_j + TW - 1) / TW;
    const long long per = 1LL * B * tiles_i * blocks_j * S::NQ, pad = (per + 7) / 8 * 8;
    if (nl < 1 || nl > MAXLV || pad * nl > 0x7fffffffLL) return D2T_ETOOBIG;
    BandLevels lv;
    lv.per = (int)per; lv.pad = (int)pad;
    int order[MAXLV];
    for (int l = 0; l < nl; ++l) order[l] = l;
    for (int a = 0; a < nl; ++a)                                     // heaviest first (nl <= 4): the long workgroups start first
        for (int b2 = a + 1; b2 < nl; ++b2)
            if (C[order[b2]] > C[order[a]]) { const int t = order[a]; order[a] = order[b2]; order[b2] = t; }
    for (int l = 0; l < MAXLV; ++l) {
        const int src = order[l < nl ? l : nl - 1];
        lv.fm0[l] = fm0[src]; lv.fm1[l] = fm1[src]; lv.out[l] = out[src]; lv.C[l] = C[src];
    }
    auto kfn = k_corr_fwd_band<TW, NB, HT, KC, RING, WL>;
    D2T_ENSURE_DYNAMIC_LDS(kfn, S::LDS * 4);
    hipLaunchKernelGGL(kfn, dim3((unsigned)(pad * nl)), dim3(S::THREADS), S::LDS * 4, st, lv, H, W, tiles_i, tiles_j, blocks_j, lay);
    return launch_status();
}

}  // namespace

// Which shape of workgroup for a grid of B items of H x W (0 = the band kernels do not take it).  id = 100 HT + 10 TW + NB.
int corr_fwd_band_config(int B, int H, int W)
{
    const int forced = lab_env_int("D2T_BAND_CFG", -1);
    if (forced >= 0) return forced;
    const int tiles_i = (H + TP - 1) / TP, tiles_j = (W + TP - 1) / TP;
    const long long tasks = 6LL * B * tiles_i * tiles_j;
    if (tasks > 2600) {
        // Medium grids (B = 3 .. 5 at the model's maps: too few 5-tile segments for k_corr_fwd_seg, several rounds of band workgroups).
        // lab/tools/band_scan.py, us, two-tile segments -> band: B 4 C 256 38 x 63 51 -> 33 (TW 4 x NB 2: 480 workgroups of 8 + 4 waves, two
        // per CU), B 3 C 1024 38 x 75 161 -> 98 (NB 2), B 5 C 256 38 x 63 52 -> 44 and B 5 C 1024 38 x 75 171 -> 151 (NB 1; NB 2 loses there).
        const long long n42 = 3LL * B * tiles_i * ((tiles_j + 3) / 4);
        return n42 > 400 && n42 <= 512 ? 142 : 141;
    }
    // One workgroup per CU is what pays (profiles/r05_b_band_scan.txt, us): 38 x 63 -- 16 tile columns, TW 4: 240 workgroups -- B = 1 C = 256
    // 11.8 (one-tile kernel 17.7); 38 x 75 -- 19 tile columns -- TW 4: 300 workgroups, 44 CUs carry two: C = 2048 83.6; TW 5 with two waves
    // per task (10 computing + 6 loader waves): 240 workgroups, 69.0 (one-tile kernel 115).  Grids of several rounds (B = 2): TW 4.
    const long long n4 = 6LL * B * tiles_i * ((tiles_j + 3) / 4), n5 = 6LL * B * tiles_i * ((tiles_j + 4) / 5);
    return n4 > 256 && n5 <= 256 ? 1251 : 141;
}

// nl levels of one spatial shape in ONE launch (the tracker's three B = 1 calls, correlation_tracker.py:68-70): the workgroups of the next
// level start as CUs come free instead of behind a launch boundary.
int corr_fwd_band_f32(int cfg, int nl, const float* const* fm0, const float* const* fm1, float* const* out, const int* C, int B, int H, int W,
                      CellLayout lay, hipStream_t st)
{
    switch (cfg) {
#define D2T_BAND_CASE(id, TW, NB, HT, KC, RING, WL) case id: return launch_band<TW, NB, HT, KC, RING, WL>(nl, fm0, fm1, out, C, B, H, W, lay, st);
        D2T_BAND_CASE(141, 4, 1, 1, 16, 3, 4)
        D2T_BAND_CASE(1251, 5, 1, 2, 16, 3, 6)
        D2T_BAND_CASE(142, 4, 2, 1, 16, 3, 4)
#ifdef D2T_ENV_KNOBS                                                  /* scan builds (lab/tools/band_scan.py) */
        D2T_BAND_CASE(1141, 4, 1, 1, 16, 3, 8)
        D2T_BAND_CASE(151, 5, 1, 1, 16, 3, 4)
        D2T_BAND_CASE(251, 5, 1, 2, 16, 3, 4)
        D2T_BAND_CASE(241, 4, 1, 2, 16, 3, 4)
        D2T_BAND_CASE(131, 3, 1, 1, 16, 3, 4)
        D2T_BAND_CASE(123, 2, 3, 1, 16, 3, 4)
        D2T_BAND_CASE(143, 4, 3, 1, 16, 3, 4)
        D2T_BAND_CASE(126, 2, 6, 1, 16, 3, 4)
        D2T_BAND_CASE(133, 3, 3, 1, 16, 3, 4)
#endif
#undef D2T_BAND_CASE
        default: return D2T_EINVAL;
    }
}

#ifdef D2T_BAND_STAMPS
extern "C" int d2t_lab_band_stamps(void* p)                          // scan builds: where the stamps go (NULL: off)
{
    unsigned long long* q = static_cast<unsigned long long*>(p);
    return static_cast<int>(hipMemcpyToSymbol(HIP_SYMBOL(band_stamps), &q, sizeof(q)));
}
extern "C" int d2t_lab_band_dbg(int bits) { return static_cast<int>(hipMemcpyToSymbol(HIP_SYMBOL(band_dbg), &bits, sizeof(bits))); }
#endif

}}  // namespace d2t::tuned
