// bwd_lab.hip -- tuning harness (developer tool): ablations of the column-strip backward.
// Generated from ../d2t_corr_tuned.hip by the snippet in the commit message; timing only.
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
constexpr int CA = 256;                    // FM0 channels staged in LDS per pass

// Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of logical tiles
// (bijective for any grid size).  Placement only affects L2 reuse, never results.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

constexpr int ST_WAVES = 16;
constexpr int ST_THREADS = ST_WAVES * 64;
constexpr int ST_CH = ST_WAVES * 16;                // channels per workgroup pass
constexpr int NACT = 5;                             // tiles alive during one super-step
constexpr int KB_SS = 5;                            // k-blocks per super-step
constexpr int RING = NACT * 256;                    // floats of G per k-block: [tile][lane][4]
// (1600 of the 2048 quad slots of a super-step are real: 5 k-blocks x 5 tiles x 64 lanes)
constexpr int Q_PER_THREAD = 2;                     // every thread produces 2 quads: 2048 slots, 448 unused
constexpr int RING_SS = Q_PER_THREAD * ST_THREADS * 4;   // 8192 floats = 32 KB per buffer

// Four consecutive elements (columns s = 0..3) of the G ring for k-block 5*ss + q, live tile a,
// lane l.  The cell inside a gradOut row depends only on (q, a, l); the row advances by 4 map
// rows per super-step.  role 0: the four cells are adjacent in one gradOut row; role 1: they sit
// in four adjacent centre pixels, one cell to the left each time.
__device__ __forceinline__ f32x4 strip_quad(const float* __restrict__ gb, int role, int ss, int q, int a, int l,
                                            int H, int W, int tiles_i, int j0, int col0)
{
    const int t = l & 15, gg = l >> 4;
    const int u = ss - 2 + a;                                       // tile row
    const int x = 4 * q + gg, xr = (x * 13) >> 6, cg = x - xr * NCG; // x / 5, x % 5 for x in 0..19
    const int rho = 4 * ss + xr;                                    // slot row
    const int ti = 4 * u + (t >> 2), tj = j0 + (t & 3);             // tile pixel
    const int sj = col0 + 4 * cg;                                   // first slot column (all 4 in the map)
    const int ci = role ? ti - rho + DT : rho - ti + DT;            // displaced - centre + d
    const int cj = role ? tj - sj + DT : sj - tj + DT;              // for s = 0; role 0: +s, role 1: -s
    const bool ok = u >= 0 && u < tiles_i && ti < H && tj < W && rho < H && ci >= 0 && ci < 2 * DT;
    const int pix = role ? rho * W + sj : ti * W + tj;              // centre pixel for s = 0
    const int off = pix * CELLS + ci * CW + cj;                     // fits int32 (checked by the C ABI)
    const int dp = role ? CELLS - 1 : 1;                            // next s: next centre & cell-1, or cell+1
    const int dc = role ? -1 : 1;
    f32x4 v;
    // branch-free: an element that is not needed loads gb[0] and is replaced by 0
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = cj + dc * s;
        const bool use = ok && c >= 0 && c < 2 * DT;
        const float x_ = gb[use ? off + dp * s : 0];
        v[s] = use ? x_ : 0.f;
    }
    return v;
}

template <int NOG, int NOBAR, int NOS, int NOMFMA>
__global__ void __launch_bounds__(ST_THREADS)
k_bwd(const float* __restrict__ gout, const float* __restrict__ fm0, const float* __restrict__ fm1,
                 float* __restrict__ g0, float* __restrict__ g1,
                 int B, int C, int H, int W, int tiles_i, int tiles_j)
{
    __shared__ __attribute__((aligned(16))) float ring[2][RING_SS];  // 64 KB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);                // (role, b) pairs stay on one XCD
    const int tj = bid % tiles_j, b = (bid / tiles_j) % B, role = bid / (tiles_j * B);
    const int j0 = tj * TP, HW = H * W;
    const int wleft = j0 - DT + role;                                // role 1 window is shifted by one
    const int col0 = wleft < 0 ? 0 : (wleft > W - WC ? W - WC : wleft);
    const float* S = role ? fm0 : fm1;
    float* gx = role ? g1 : g0;
    const float* gb = gout + (size_t)b * HW * CELLS;

    const int cw = blockIdx.y * ST_CH + wave * 16;                   // first channel of this wave's c-tile
    const int cl = cw + n < C ? cw + n : C - 1;                      // lane's channel (clamped; never stored)
    const float* sp = S + ((size_t)b * C + cl) * HW + col0;

    f32x4 acc[NACT];
#pragma unroll
    for (int a = 0; a < NACT; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};

    // S fragment of k-block 5*ss + q: group 20*ss + 4q + g.  Rows past the map carry G = 0.
    auto s_load = [&](int ss, int q) -> f32x4 {
        const int x = 4 * q + g, xr = (x * 13) >> 6, cg = x - xr * NCG;
        int rho = 4 * ss + xr;
        rho = rho < H ? rho : H - 1;
        return *reinterpret_cast<const f32x4u*>(sp + rho * W + 4 * cg);
    };
    auto store_tile = [&](const f32x4& d, int u) {
        const int i = 4 * u + (n >> 2), j = j0 + (n & 3);
        if (u < 0 || u >= tiles_i || i >= H || j >= W) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = cw + 4 * g + r;
            if (c < C) gx[((size_t)b * C + c) * HW + i * W + j] = d[r];
        }
    };
    // quad k of this thread in a super-step's ring: slot tid + k*1024 = (q, a, lane); slots >= 1600
    // are padding (they decode to q >= 5, rows that belong to the next super-step: never read)
    auto g_quad = [&](int ss, int k) -> f32x4 {
        const int e = tid + k * ST_THREADS;
        const int q = e / (NACT * 64), r = e - q * (NACT * 64);
        return strip_quad(gb, role, ss, q, r >> 6, r & 63, H, W, tiles_i, j0, col0);
    };

    // prologue: ring[0] <- super-step 0
#pragma unroll
    for (int k = 0; k < Q_PER_THREAD; ++k)
        reinterpret_cast<f32x4*>(ring[0])[tid + k * ST_THREADS] = g_quad(0, k);
    f32x4 av = s_load(0, 0);
    __syncthreads();

    // The loop body is straight-line code (no branches): every load is unconditional, so the
    // compiler can retire them with counted s_waitcnt instead of draining at block boundaries.
    for (int ss = 0; ss < tiles_i; ++ss) {
        const int cur = ss & 1;
        const f32x4* rb = reinterpret_cast<const f32x4*>(ring[cur]);
        f32x4 gn[Q_PER_THREAD];
#pragma unroll
        for (int k = 0; k < Q_PER_THREAD; ++k) gn[k] = NOG ? f32x4{0.f,0.f,0.f,(float)ss} : g_quad(ss + 1, k);
#pragma unroll
        for (int q = 0; q < KB_SS; ++q) {
            const f32x4 a4 = av;
            if (!NOS) av = q + 1 < KB_SS ? s_load(ss, q + 1) : s_load(ss + 1, 0); else av.x += 1.f;
            f32x4 bv[NACT];
#pragma unroll
            for (int a = 0; a < NACT; ++a) bv[a] = rb[(q * NACT + a) * 64 + lane];
#pragma unroll
            for (int s = 0; s < 4; ++s) {                            // s outer: 5 independent accumulators
#pragma unroll
                for (int a = 0; a < NACT; ++a) { if (!NOMFMA) acc[a] = D2T_MFMA(a4[s], bv[a][s], acc[a]); else acc[a][s] += a4[s] * bv[a][s]; }
            }
        }
#pragma unroll
        for (int k = 0; k < Q_PER_THREAD; ++k)
            reinterpret_cast<f32x4*>(ring[cur ^ 1])[tid + k * ST_THREADS] = gn[k];
        if (!NOBAR) __syncthreads();
        store_tile(acc[0], ss - 2);                                  // complete after its 5th super-step
#pragma unroll
        for (int a = 0; a + 1 < NACT; ++a) acc[a] = acc[a + 1];
        acc[NACT - 1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    store_tile(acc[0], tiles_i - 2);                                 // their remaining super-steps lie below the map
    store_tile(acc[1], tiles_i - 1);
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
    std::vector<float*> f0(NS), f1(NS), go(NS), g0(NS), g1(NS);
    std::vector<float> h(out_n);
    for (size_t i = 0; i < out_n; ++i) h[i] = (float)rand() / RAND_MAX;
    for (int s = 0; s < NS; ++s) {
        hipMalloc(&f0[s], in_n * 4); hipMalloc(&f1[s], in_n * 4); hipMalloc(&g0[s], in_n * 4); hipMalloc(&g1[s], in_n * 4); hipMalloc(&go[s], out_n * 4);
        hipMemcpy(f0[s], h.data(), in_n * 4, hipMemcpyHostToDevice); hipMemcpy(f1[s], h.data(), in_n * 4, hipMemcpyHostToDevice);
        hipMemcpy(go[s], h.data(), out_n * 4, hipMemcpyHostToDevice);
    }
    const int ti = (H + 3) / 4, tj = (W + 3) / 4;
#define RUN(A, Bb, Cc, Dd, NAME) { float us = time_it([&](int i) { hipLaunchKernelGGL((k_bwd<A, Bb, Cc, Dd>), dim3(2 * B * tj, 1), dim3(ST_THREADS), 0, 0, go[i % NS], f0[i % NS], f1[i % NS], g0[i % NS], g1[i % NS], B, C, H, W, ti, tj); }, 50); printf("%-44s %8.1f us\n", NAME, us); }
    RUN(0, 0, 0, 0, "full");
    RUN(1, 0, 0, 0, "no G production (ring const)");
    RUN(1, 1, 0, 0, "no G, no barrier");
    RUN(1, 1, 1, 0, "no G, no barrier, no S loads (pure MFMA+LDS)");
    RUN(0, 0, 0, 1, "no MFMA (VALU fma instead)");
    RUN(0, 1, 0, 0, "G production but no barrier (racy)");
    return 0;
}
