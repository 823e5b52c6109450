// seg_lab.hip -- tuning harness (developer tool): ablations of the LDS-staged forward segment kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte load, dword aligned

#define D2T_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int TP = 4;                      // p-tile edge: 4x4 pixels = MFMA M = 16
constexpr int DT = 8;                      // d_max the tuned kernels are built for
constexpr int WR = TP + 2 * DT - 1;        // 19 window rows (and needed columns)
constexpr int NCG = (WR + 3) / 4;          // 5 column groups per window row
constexpr int WC = NCG * 4;                // 20 loaded columns
constexpr int CW = 2 * DT + 1;             // 17
constexpr int CELLS = CW * CW;             // 289
constexpr int FWD_WAVES = 6;               // >= max tile-groups = ceil(19*5/16)
constexpr int FWD_THREADS = FWD_WAVES * 64;

// Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of logical tiles
// (bijective for any grid size).  Placement only affects L2 reuse, never results.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

constexpr int ACH = 64;                    // FM0 channels per staged chunk (16 k-steps)

constexpr int SG_NU = 5;                            // p-tiles per segment
constexpr int SG_WAVES = 15;                        // 30 (tile, tile-group) tasks, two per wave
constexpr int SG_THREADS = SG_WAVES * 64;
constexpr int SG_KC = 16;                           // channels per staged chunk (4 k-steps)
constexpr int SG_ROWS = 4 * SG_NU + 2 * DT - 1;     // 35 window rows of a segment
constexpr int SG_BPL = SG_ROWS * WC;                // 700 floats: FM1 region of one channel
constexpr int SG_APL = SG_NU * 16;                  // 80 floats: FM0 pixels of one channel
constexpr int SG_BUF = SG_KC * (SG_BPL + SG_APL);   // floats per buffer (48.75 KB)
constexpr int SG_STAGE = SG_NU * 16 * CELLS;        // out staging (90.3 KB), aliases the buffers
constexpr int SG_LDS = 2 * SG_BUF > SG_STAGE ? 2 * SG_BUF : SG_STAGE;

constexpr int SG_NPB = SG_KC * SG_ROWS * NCG;       // 2800 FM1 pieces (16 bytes) per chunk
constexpr int SG_NPA = SG_KC * 4 * SG_NU;           // 320 FM0 pieces per chunk
constexpr int SG_BI = (SG_NPB + 63) / 64;           // 44 wave-instructions move the FM1 region of a chunk
constexpr int SG_BIW = (SG_BI + SG_WAVES - 1) / SG_WAVES;   // 3 per wave
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int EPI>
__global__ void __launch_bounds__(SG_THREADS)
k_seg(const float* __restrict__ fm0, const float* __restrict__ fm1, float* __restrict__ out,
               int C, int H, int W, int tiles_i, int tiles_j, int nseg)
{
    __shared__ __attribute__((aligned(16))) float smem[SG_LDS];

    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int seg = bid % nseg, tj = (bid / nseg) % tiles_j, b = bid / (nseg * tiles_j);
    const int u0 = seg * SG_NU, nu = tiles_i - u0 < SG_NU ? tiles_i - u0 : SG_NU;
    const int j0 = tj * TP, HW = H * W;
    const int R0 = 4 * u0 - DT > 0 ? 4 * u0 - DT : 0;                // region rows [R0, R1) inside the map
    const int R1 = 4 * (u0 + nu) + DT - 1 < H ? 4 * (u0 + nu) + DT - 1 : H;
    const int nrows = R1 - R0;
    const int colL = j0 - DT;                                        // region columns [colL, colL+20): may leave the map

    // ---- staging by LDS-DMA (buffer_load_dwordx4 ... lds): a wave-instruction copies 64 pieces of
    // 16 bytes from per-lane global addresses into 64 CONSECUTIVE 16-byte LDS slots, no VGPR round
    // trip.  The LDS image is [channel][row][column group] with a fixed 35-row pitch, so piece e
    // lands in slot e.  The buffer descriptor covers exactly this batch item's C planes: a piece of a
    // channel >= C (last chunk) is out of range and arrives as exact zeros.  Columns outside the
    // map read whatever neighbours them in memory: MFMA columns are independent and those cells are
    // masked in the epilogue.  Pieces of rows the segment does not have are parked out of range.
    const unsigned plane_bytes = (unsigned)C * HW * 4u;
    const __amdgpu_buffer_rsrc_t r1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fm1 + (size_t)b * C * HW), 0, plane_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r0 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fm0 + (size_t)b * C * HW), 0, plane_bytes, 0x00020000);
    constexpr int OOR = 0x7ffffff0;                                  // parked byte offset: always out of range
    int pb_voff[SG_BIW];
    bool pb_on[SG_BIW];
#pragma unroll
    for (int k = 0; k < SG_BIW; ++k) {
        const int e = (wave + SG_WAVES * k) * 64 + lane;
        const int ch = e / (SG_ROWS * NCG), rem = e - ch * (SG_ROWS * NCG);
        const int row = rem / NCG, cg = rem - row * NCG;
        pb_on[k] = e < SG_NPB;
        pb_voff[k] = row < nrows ? (ch * HW + (R0 + row) * W + colL + 4 * cg) * 4 : OOR;
    }
    int pa_voff = OOR;
    const bool pa_on = wave < (SG_NPA + 63) / 64;
    {
        const int e = wave * 64 + lane;                              // FM0 piece: (channel, pixel row of the segment)
        const int ch = e / (4 * SG_NU), prow = e - ch * (4 * SG_NU);
        const int i = 4 * u0 + prow;
        if (pa_on && e < SG_NPA && i < H) pa_voff = (ch * HW + i * W + j0) * 4;
    }
    const int chunk_bytes = SG_KC * HW * 4;
    auto stage = [&](float* buf, int chunk) {
        const int cb = chunk * chunk_bytes;
#pragma unroll
        for (int k = 0; k < SG_BIW; ++k) {
            if (pb_on[k]) {
                const int v = pb_voff[k] == OOR ? OOR : pb_voff[k] + cb;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr)(buf + (wave + SG_WAVES * k) * 256), 16, v, 0, 0, 0);
            }
        }
        if (pa_on) {
            const int v = pa_voff == OOR ? OOR : pa_voff + cb;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r0, (lds_ptr)(buf + SG_KC * SG_BPL + wave * 256), 16, v, 0, 0, 0);
        }
    };

    // ---- this wave's two tasks: id = tile*6 + tile-group
    int t_tile[2], t_off[2], t_ng[2];
    bool t_on[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int id = wave + SG_WAVES * k, t = id / 6, T = id - t * 6;
        const int u = u0 + t;
        const int wa = 4 * u - DT > 0 ? 4 * u - DT : 0;              // tile's window rows inside the map
        const int wb = 4 * u + TP + DT - 1 < H ? 4 * u + TP + DT - 1 : H;
        const int ng = (wb - wa) * NCG;
        t_tile[k] = t;
        t_ng[k] = ng;
        t_on[k] = t < nu && 16 * T < ng;
        t_off[k] = ((wa - R0) * NCG + 16 * T) * 4;                   // float offset of the tile-group's first slot
    }
    // lane's slot inside the tile-group (clamped to the tile's last group; masked in the epilogue);
    // a wave without a second task recomputes a valid slot into a dead accumulator: no branch
    int l_off[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int T = (wave + SG_WAVES * k) % 6;
        int gi = 16 * T + n;
        gi = gi < t_ng[k] ? gi : (t_ng[k] > 0 ? t_ng[k] - 1 : 0);
        l_off[k] = t_on[k] ? t_off[k] - 16 * T * 4 + gi * 4 + g * SG_BPL : g * SG_BPL;
    }
    const int a_off0 = SG_KC * SG_BPL + g * SG_APL + (t_on[0] ? t_tile[0] : 0) * 16 + n;
    const int a_off1 = SG_KC * SG_BPL + g * SG_APL + (t_on[1] ? t_tile[1] : 0) * 16 + n;

    f32x4 acc[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[k][s] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = (C + SG_KC - 1) / SG_KC;
    stage(smem, 0);
    __syncthreads();                                                 // vmcnt(0) + barrier: chunk 0 has landed
    for (int ch = 0; ch < nchunks; ++ch) {
        const float* cur = smem + (ch & 1) * SG_BUF;
        stage(smem + ((ch + 1) & 1) * SG_BUF, ch + 1);               // lands during this chunk's MFMAs (past the end: zeros)
#pragma unroll
        for (int ks = 0; ks < SG_KC / 4; ++ks) {
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(cur + l_off[0] + ks * 4 * SG_BPL);
            const f32x4 q1 = *reinterpret_cast<const f32x4*>(cur + l_off[1] + ks * 4 * SG_BPL);
            const float a0 = cur[a_off0 + ks * 4 * SG_APL], a1 = cur[a_off1 + ks * 4 * SG_APL];
            acc[0][0] = D2T_MFMA(a0, q0.x, acc[0][0]);
            acc[1][0] = D2T_MFMA(a1, q1.x, acc[1][0]);
            acc[0][1] = D2T_MFMA(a0, q0.y, acc[0][1]);
            acc[1][1] = D2T_MFMA(a1, q1.y, acc[1][1]);
            acc[0][2] = D2T_MFMA(a0, q0.z, acc[0][2]);
            acc[1][2] = D2T_MFMA(a1, q1.z, acc[1][2]);
            acc[0][3] = D2T_MFMA(a0, q0.w, acc[0][3]);
            acc[1][3] = D2T_MFMA(a1, q1.w, acc[1][3]);
        }
        __syncthreads();
    }

    if (EPI == 0) { out[(size_t)blockIdx.x * SG_THREADS + tid] = acc[0][0].x + acc[0][1].x + acc[0][2].x + acc[0][3].x + acc[1][0].x + acc[1][1].x + acc[1][2].x + acc[1][3].x; return; }
    if (EPI != 3) for (int e = tid; e < nu * 16 * CELLS; e += SG_THREADS) smem[e] = 0.f;
    __syncthreads();
    if (EPI == 4) { out[(size_t)blockIdx.x * SG_THREADS + tid] = smem[tid] + acc[0][0].x + acc[1][1].y; return; }
    if (EPI != 3)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int T = (wave + SG_WAVES * k) % 6;
        const int gi = 16 * T + n;
        if (t_on[k] && gi < t_ng[k]) {
            const int u = u0 + t_tile[k];
            const int wa = 4 * u - DT > 0 ? 4 * u - DT : 0;
            const int rho = wa + gi / NCG, cg = gi - (gi / NCG) * NCG;   // displaced row, column group
            const int ci = rho - (4 * u + g) + DT;                   // di - i + d, pixel row i = 4u + g
            if (ci >= 0 && ci < 2 * DT) {
                float* row = smem + (t_tile[k] * 16 + 4 * g) * CELLS + ci * CW;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int dj = colL + 4 * cg + s;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cj = dj - (j0 + r) + DT;
                        if (cj >= 0 && cj < 2 * DT && dj >= 0 && dj < W) row[r * CELLS + cj] = acc[k][s][r];
                    }
                }
            }
        }
    }
    __syncthreads();
    if (EPI == 2) { out[(size_t)blockIdx.x * SG_THREADS + tid] = smem[tid * 7]; return; }
    // 16-byte stores: a pixel row of the strip is nj*289 contiguous floats in
    // out and starts 16-byte aligned in the LDS image; 4x fewer store instructions than dwords
    const int nj = W - j0 < TP ? W - j0 : TP;
    const int run = nj * CELLS, run4 = run >> 2;                     // floats / whole float4s per pixel row
    const int prs = (H - 4 * u0 < 4 * nu ? H - 4 * u0 : 4 * nu);     // pixel rows that exist
    for (int e = tid; e < prs * run4; e += SG_THREADS) {
        const int pr = e / run4, q = e - pr * run4;
        float* dst = out + (((size_t)b * H + 4 * u0 + pr) * W + j0) * CELLS + 4 * q;
        *reinterpret_cast<f32x4u*>(dst) = *reinterpret_cast<const f32x4*>(smem + (size_t)pr * 4 * CELLS + 4 * q);
    }
    const int tail = run - 4 * run4;                                 // 0..3 floats per pixel row (nj < 4)
    for (int e = tid; e < prs * tail; e += SG_THREADS) {
        const int pr = e / tail, q = 4 * run4 + (e - pr * tail);
        out[(((size_t)b * H + 4 * u0 + pr) * W + j0) * CELLS + q] = smem[(size_t)pr * 4 * CELLS + q];
    }
}


template <typename F> float time_it(F f, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f(i);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) f(i);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / iters;
}
int main() {
    const int B = 8, C = 256, H = 38, W = 63;
    const size_t in_n = (size_t)B * C * H * W, out_n = (size_t)B * H * W * CELLS;
    const int NS = 6;
    std::vector<float*> f0(NS), f1(NS), o(NS);
    std::vector<float> h(in_n);
    for (size_t i = 0; i < in_n; ++i) h[i] = (float)rand() / RAND_MAX;
    for (int s = 0; s < NS; ++s) {
        hipMalloc(&f0[s], in_n * 4); hipMalloc(&f1[s], in_n * 4); hipMalloc(&o[s], out_n * 4);
        hipMemcpy(f0[s], h.data(), in_n * 4, hipMemcpyHostToDevice); hipMemcpy(f1[s], h.data(), in_n * 4, hipMemcpyHostToDevice);
    }
    const int ti = (H + 3) / 4, tj = (W + 3) / 4, nseg = (ti + SG_NU - 1) / SG_NU, blocks = B * tj * nseg;
#define RUN(A, NAME) { float us = time_it([&](int i) { hipLaunchKernelGGL((k_seg<A>), dim3(blocks), dim3(SG_THREADS), 0, 0, f0[i % NS], f1[i % NS], o[i % NS], C, H, W, ti, tj, nseg); }, 50); printf("%-50s %8.1f us\n", NAME, us); }
    RUN(1, "full");
    RUN(0, "no epilogue");
    RUN(4, "zero-fill only");
    RUN(2, "zero-fill + scatter (no global stores)");
    RUN(3, "global stores only (no zero/scatter)");
    return 0;
}
