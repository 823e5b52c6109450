// d2t_corr_fwd_band.hip -- gfx950-tuned f32 PointwiseCorrelation forward for SMALL grids (d_max = 8, stride 1): the
// model's B = 1 pairs (correlation_tracker.py:57-70) and BASELINE config 2.  Bit-identical to the reference
// (pointwise_correlation_cuda.cu:84-107): every output cell is ONE ascending-channel chain of v_mfma_f32_16x16x4_f32,
// exactly as in d2t_corr_tuned.hip -- parallelism comes from splitting a tile's WINDOW over workgroups, never its channels.
//
// Why a second forward.  A 4 x 4 p-tile has 6 tile-groups (16 window slot groups = 4 N-tiles each); 190 tiles at 38 x 75
// are 1,140 (tile, tile-group) TASKS of 16 MFMAs per 16 channels.  The one-tile-per-workgroup kernel
// (k_corr_fwd_segx<1,4,true>) gives a CU all 6 tasks of a tile and leaves 66 CUs idle; its time is the L1-miss path of ONE
// CU streaming a 19 x 20 window per 16 channels (34.6 cache lines per channel: notebook 4.2-6) -- 0.92 us per chunk where
// the matrix work is 0.32 us.  Here a workgroup is
//     (tile row u, block of TW tiles side by side, band-set q of NB consecutive tile-groups)
// with ONE task per wave (TW * NB waves): the block's window is 4 TW + 16 columns wide, so a window row is one 96 / 128
// byte run for 2 / 4 tiles instead of 80 bytes per tile, a band-set needs only its own 4 / 7 / 10 window rows, and the
// FM0 pixels of the block are staged once for its NB bands.  Lines per channel and task: 5.8 (one tile, all bands) ->
// 3.5 (TW 4, NB 1), 2.5 (TW 4, NB 2); tasks per SIMD: 1 or 2 instead of 1.5 on three quarters of the chip.
//
// Staging and loop are those of k_corr_fwd_segx (ring of chunk images filled by LDS-DMA, counted vmcnt waits, one
// barrier per chunk, the next chunk's fragments fetched under this chunk's MFMAs).  The epilogue stores straight from
// the accumulators: a band's cells of a pixel are a contiguous run of its 17 x 17 block, so no LDS staging, no barrier;
// the structural zeros (cj = 16, ci = 16, displaced columns outside the map) are written by the lanes that own those
// slots, window rows outside the map ("orphan rows") by the first / last band-set of the tile row.
#include "d2t_corr_common.hpp"

namespace d2t { namespace tuned {

namespace {

constexpr int imin(int a, int b) { return a < b ? a : b; }
constexpr int imax(int a, int b) { return a > b ? a : b; }
constexpr int MAXG = WR * NCG;                                       // 95 slot groups of an unclipped tile window

template <int TW, int NB, int KC, int RING, int WL>
struct Band {
    static constexpr int NG = TW + 4;                                // 16-byte column groups of the block's window
    static constexpr int WC = TW * NB;                               // compute waves: one (tile, tile-group) task each
    static constexpr int WAVES = WC + WL, THREADS = 64 * WAVES;      // + WL loader waves: they issue every LDS-DMA instruction
    static constexpr int NQ = (6 + NB - 1) / NB;                     // band-sets per tile row and block
    static constexpr int rows_of(int q) { return (imin(16 * (q + 1) * NB, MAXG) - 1) / NCG - (16 * q * NB) / NCG + 1; }
    static constexpr int max_rows() { int m = 0; for (int q = 0; q < NQ; ++q) m = imax(m, rows_of(q)); return m; }
    static constexpr int NROWS = max_rows();                         // window rows a band-set stages at most
    static constexpr int P = (NROWS * NG + 15) / 16 * 16;            // slots per channel plane: plane stride = 0 mod 64 dwords
    static constexpr int BPL = 4 * P, APL = 16 * TW;                 // floats per channel: FM1 window part / FM0 pixels
    static constexpr int BUF = KC * (BPL + APL);                     // floats per staged chunk
    static constexpr int BI = KC * P / 64, AI = KC * 4 * TW / 64;    // DMA wave-instructions per chunk: FM1 / FM0
    static constexpr int NDMA = (BI + AI + WL - 1) / WL;             // per loader wave and chunk (surplus ones are parked)
    static constexpr int DUMMY = 256;                                // floats: where parked DMA instructions land
    static constexpr int LDS = RING * BUF + DUMMY;
    static constexpr int INFLIGHT = (RING - 2) * NDMA;               // a loader's DMA instructions that may be outstanding at a barrier
    static_assert(WAVES <= 16 && KC % 8 == 0 && (KC * 4 * TW) % 64 == 0 && (KC * P) % 64 == 0 && RING >= 3, "shape");
    static_assert(INFLIGHT <= 63, "vmcnt is a 6-bit counter");
    static_assert(LDS * 4 <= 160 * 1024, "LDS budget");
};

// One workgroup: batch item b, tile row u, tile columns [TW vb, TW vb + TW), tile-groups [NB q, NB q + NB).
template <int TW, int NB, int KC, int RING, int WL>
__global__ void __launch_bounds__((TW * NB + WL) * 64)
k_corr_fwd_band(const float* __restrict__ fm0, const float* __restrict__ fm1, float* __restrict__ out,
                int C, int H, int W, int tiles_i, int tiles_j, int blocks_j, CellLayout lay)
{
    using S = Band<TW, NB, KC, RING, WL>;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // logical id (band-set innermost: the band-sets of a block share its FM0 pixels and overlap in window rows; give every
    // XCD a contiguous run of the logical order)
    int id = xcd_remap(blockIdx.x, gridDim.x);
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

    // ---- this wave's role.  Waves [0, WC) compute: tile v of the block, tile-group T.  Waves [WC, WC + WL) are LOADERS: they issue
    // every LDS-DMA instruction of the workgroup.  (With the DMA instructions dealt to the computing waves -- the first version --
    // every wave issued its share right behind the chunk's barrier, all of them queued at the CU's address unit together and no
    // MFMA was issued meanwhile: 0.61 us per chunk where the matrix work is 0.43 us; a wave's instruction stream is in order.)
    const bool loader = wave >= S::WC;                               // wave-uniform
    const int v = wave % TW, T = NB * q + (wave / TW) % NB;
    const bool t_on = !loader && 16 * T < ng && v0 + v < tiles_j;    // wave-uniform
    int gi = 16 * T + n;
    const bool lane_on = t_on && gi < ng;
    gi = gi < ng ? gi : ng - 1;
    gi = gi < g_lo ? g_lo : gi;                                      // (a dead wave still reads a staged slot)
    const int rho = wa + gi / NCG, cg = gi - (gi / NCG) * NCG;       // displaced row, column group inside the tile's window
    const int l_off = ((rho - R0) * S::NG + v + cg) * 4 + g * S::BPL;
    const int a_off = KC * S::BPL + g * S::APL + (n >> 2) * (4 * TW) + 4 * v + (n & 3);

    f32x4 acc[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nchunks = (C + KC - 1) / KC;

    // Ring protocol (both roles meet at ONE barrier per chunk).  Barrier #k publishes chunk k: every loader has waited until at
    // most INFLIGHT = (RING - 2) NDMA of its DMA instructions are outstanding (LDS-DMA retires in order: chunks k+1 .. k+RING-2
    // may be in flight), every computing wave has its fragments of chunk k-1 in registers (lgkmcnt(0)), so behind the barrier
    // the loaders overwrite the slot of chunk k-1 with chunk k+RING-1.  The barrier sits in the MIDDLE of a chunk's MFMAs: the
    // fragments of chunk k+1 are fetched under the second half of chunk k's.
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
                const int xa = x - S::BI;
                const int e = xa * 64 + lane;
                const int ch = e / (4 * TW), rem = e - ch * (4 * TW);
                const int prow = rem / TW, pv = rem - prow * TW;
                const int pi = 4 * u + prow;
                dv[k] = xa < S::AI && pi < H && v0 + pv < tiles_j ? (ch * HW + pi * W + j0 + 4 * pv) * 4 : OOR;
            }
        }
        const int chunk_bytes = KC * HW * 4;
        auto stage = [&](int slot, int chunk) {                      // chunks past the end of C arrive as zeros
            const int cb = chunk * chunk_bytes;
            float* buf = smem + slot * S::BUF;
#pragma unroll
            for (int k = 0; k < S::NDMA; ++k) {
                const int x = lw + WL * k;                           // wave-uniform
                const int vo = dv[k] == OOR ? OOR : dv[k] + cb;
                if (x < S::BI) __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr)(buf + x * 256), 16, vo, 0, 0, 0);
                else if (x < S::BI + S::AI) __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (lds_ptr)(buf + KC * S::BPL + (x - S::BI) * 256), 16, vo, 0, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (lds_ptr)(smem + RING * S::BUF), 16, vo, 0, 0, 0);   // parked (vo = OOR): keeps the count equal
            }
        };
#pragma unroll
        for (int p = 0; p < RING - 1; ++p) stage(p, p);
        dma_wait_barrier<S::INFLIGHT>();                             // barrier #0: chunk 0 has landed
        int slot = RING - 1;                                         // where chunk ch + RING - 1 goes
        for (int ch = 0; ch < nchunks; ++ch) {
            stage(slot, ch + RING - 1);
            slot = slot + 1 == RING ? 0 : slot + 1;
            dma_wait_barrier<S::INFLIGHT>();                         // barrier #(ch+1): chunk ch+1 has landed
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the zero chunks staged past the end
    } else {
        struct Frag { f32x4 q[KC / 4]; float a[KC / 4]; };
        auto fetch = [&](Frag& f, const float* cur) {
#pragma unroll
            for (int ks = 0; ks < KC / 4; ++ks) {
                f.q[ks] = *reinterpret_cast<const f32x4*>(cur + l_off + ks * 4 * S::BPL);
                f.a[ks] = cur[a_off + ks * 4 * S::APL];
            }
        };
        auto mfma = [&](const Frag& f, int ks_lo, int ks_hi) {
            if (!t_on) return;                                       // wave-uniform
#pragma unroll
            for (int ks = ks_lo; ks < ks_hi; ++ks)
#pragma unroll
                for (int s = 0; s < 4; ++s) acc[s] = D2T_MFMA(f.a[ks], f.q[ks][s], acc[s]);
        };
        Frag cur_f, nxt_f;
        lds_barrier();                                               // barrier #0
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
    }

    // ---- epilogue: straight from the accumulators.  acc[s][r] belongs to pixel (row g, column r) of the tile and window
    // slot (rho, column 4 cg + s): cell ci = rho - i + d, cj = 4 cg + s - r of that pixel.
    const int i = 4 * u + g, ci = rho - i + DT;
    if (lane_on && i < H && ci >= 0 && ci <= 2 * DT) {
        const int jt = j0 + 4 * v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = jt + r;
            if (j < W) {
                const int pix_off = (i * W + j) * lay.ps + ci * CW * lay.cs;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int cj = 4 * cg + s - r, dj = j + cj - DT;
                    if (cj >= 0 && cj <= 2 * DT) {
                        const float val = ci < 2 * DT && cj < 2 * DT && dj >= 0 && dj < W ? acc[s][r] : 0.f;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), ro, (pix_off + cj * lay.cs) * 4, 0, 0);
                    }
                }
            }
        }
    }
    // orphan rows: cells whose displaced row lies outside the tile row's window -- above the map (first band-set), below
    // it or ci = 16 of the tile's last pixel row (last band-set): structural zeros nobody computes
    if (q == 0 || q == q_last) {
        for (int e = tid; e < 16 * TW * CW; e += S::THREADS) {
            const int p = e / CW, oci = e - p * CW;
            const int ov = p >> 4, prow = (p >> 2) & 3, r = p & 3;
            const int oi = 4 * u + prow, oj = j0 + 4 * ov + r;
            const int orho = oi + oci - DT;
            const bool mine = (orho < wa && q == 0) || (orho >= wb && q == q_last);
            if (mine && oi < H && oj < W) {
                const int off = (oi * W + oj) * lay.ps + oci * CW * lay.cs;
#pragma unroll
                for (int cj = 0; cj < CW; ++cj) __builtin_amdgcn_raw_buffer_store_b32(0u, ro, (off + cj * lay.cs) * 4, 0, 0);
            }
        }
    }
}

template <int TW, int NB, int KC, int RING, int WL>
int launch_band(const float* fm0, const float* fm1, float* out, int B, int C, int H, int W, CellLayout lay, hipStream_t st)
{
    using S = Band<TW, NB, KC, RING, WL>;
    const int tiles_i = (H + TP - 1) / TP, tiles_j = (W + TP - 1) / TP, blocks_j = (tiles_j + TW - 1) / TW;
    const long long nwg = 1LL * B * tiles_i * blocks_j * S::NQ;
    if (nwg > 0x7fffffffLL) return D2T_ETOOBIG;
    auto kfn = k_corr_fwd_band<TW, NB, KC, RING, WL>;
    D2T_ENSURE_DYNAMIC_LDS(kfn, S::LDS * 4);
    hipLaunchKernelGGL(kfn, dim3((unsigned)nwg), dim3(S::THREADS), S::LDS * 4, st, fm0, fm1, out, C, H, W, tiles_i, tiles_j, blocks_j, lay);
    return launch_status();
}

}  // namespace

// Which shape of workgroup for a grid of B items of H x W (0 = the band kernels do not take it).  Cost model per 16-channel
// chunk on the busiest CU, in cycles: matrix pipe 512 per task on its busiest SIMD, L1-miss path 2.7 per 128-byte line
// (ta_roof.json) -- see the table in the file header; workgroups are dealt evenly over the 256 CUs.
int corr_fwd_band_config(int B, int H, int W)
{
    const int forced = lab_env_int("D2T_BAND_CFG", -1);
    if (forced >= 0) return forced;
    const int tiles_i = (H + TP - 1) / TP, tiles_j = (W + TP - 1) / TP;
    const long long tasks = 6LL * B * tiles_i * tiles_j;
    if (tasks > 2600) return 0;                                      // larger grids: the segment kernels of d2t_corr_tuned.hip
    struct Cand { int id, tw, nb; double lines; };
    static const Cand cands[] = {{41, 4, 1, 14.0}, {42, 4, 2, 20.0}, {43, 4, 3, 26.0}, {23, 2, 3, 22.5}};
    int best = 0; double best_t = 1e30;
    for (const Cand& c : cands) {
        const long long nwg = 1LL * B * tiles_i * ((tiles_j + c.tw - 1) / c.tw) * ((6 + c.nb - 1) / c.nb);
        const double per_cu = (double)((nwg + 255) / 256);
        const double mfma = per_cu * 512.0 * ((c.tw * c.nb + 3) / 4);
        const double ta = per_cu * c.lines * 16 * 2.7;
        const double t = (mfma > ta ? mfma : ta) + 120.0;           // + barrier / fetch per chunk
        if (t < best_t) { best_t = t; best = c.id; }
    }
    return best;
}

int corr_fwd_band_f32(int cfg, const float* fm0, const float* fm1, float* out, int B, int C, int H, int W, CellLayout lay, hipStream_t st)
{
    switch (cfg) {
#define D2T_BAND_CASE(id, TW, NB, KC, RING, WL) case id: return launch_band<TW, NB, KC, RING, WL>(fm0, fm1, out, B, C, H, W, lay, st);
        D2T_BAND_CASE(41, 4, 1, 16, 4, 2)
        D2T_BAND_CASE(42, 4, 2, 16, 4, 2)
        D2T_BAND_CASE(43, 4, 3, 16, 4, 2)
        D2T_BAND_CASE(23, 2, 3, 16, 4, 2)
        D2T_BAND_CASE(22, 2, 2, 16, 4, 2)
#ifdef D2T_ENV_KNOBS                                                  /* scan builds (tools/band_scan.py) */
        D2T_BAND_CASE(141, 4, 1, 32, 3, 2)
        D2T_BAND_CASE(142, 4, 2, 32, 3, 2)
        D2T_BAND_CASE(143, 4, 3, 32, 3, 2)
        D2T_BAND_CASE(123, 2, 3, 32, 3, 2)
        D2T_BAND_CASE(241, 4, 1, 16, 4, 1)
        D2T_BAND_CASE(242, 4, 2, 16, 4, 1)
        D2T_BAND_CASE(341, 4, 1, 16, 4, 4)
        D2T_BAND_CASE(342, 4, 2, 16, 4, 4)
        D2T_BAND_CASE(441, 4, 1, 16, 3, 2)
        D2T_BAND_CASE(442, 4, 2, 16, 3, 2)
        D2T_BAND_CASE(541, 4, 1, 32, 4, 2)
        D2T_BAND_CASE(542, 4, 2, 32, 3, 4)
#endif
#undef D2T_BAND_CASE
        default: return D2T_EINVAL;
    }
}

}}  // namespace d2t::tuned
