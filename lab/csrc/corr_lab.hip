// corr_lab.hip -- tuning harness (developer tool, not part of libd2t_ops.so): times ablations of
// the correlation forward main loop on the B=8 C=256 38x63 shape to find which side bounds it.
//   hipcc -O3 --offload-arch=gfx950 -o corr_lab corr_lab.hip && ./corr_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int TP = 4, DT = 8, WR = 19, NCG = 5, WC = 20, CW = 17, CELLS = 289;
constexpr int WAVES = 6, THREADS = WAVES * 64, CA = 256;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// MODE 0: full; 1: aligned loads (wrong data, timing only); 2: loads only (one MFMA per step to
// keep data live); 3: MFMA only (no streamed loads); 4: no XCD remap
template <int MODE, int PF_WAVES, int EPI = 1, int AST = 1>
__global__ void __launch_bounds__(THREADS, PF_WAVES)
k_fwd(const float* __restrict__ fm0, const float* __restrict__ fm1, float* __restrict__ out,
      int C, int H, int W, int tiles_i, int tiles_j)
{
    __shared__ float smem[16 * CELLS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    const int bid = MODE == 4 ? blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
    const int tj = bid % tiles_j, ti = (bid / tiles_j) % tiles_i, b = bid / (tiles_j * tiles_i);
    const int i0 = ti * TP, j0 = tj * TP, HW = H * W;
    const int wr_lo = DT - i0 > 0 ? DT - i0 : 0;
    const int wr_hi = H + DT - i0 < WR ? H + DT - i0 : WR;
    const int NG = (wr_hi - wr_lo) * NCG, ntg = (NG + 15) >> 4;
    int col0 = j0 - DT;
    col0 = col0 < 0 ? 0 : (col0 > W - WC ? W - WC : col0);
    const int gsel = 16 * wave + n, gi = gsel < NG ? gsel : NG - 1;
    const int wr = wr_lo + gi / NCG, di = i0 - DT + wr;
    int djs = col0 + 4 * (gi % NCG);
    int boff = g * HW + di * W + djs;
    if (MODE == 1) boff &= ~3;
    const int kstride = 4 * HW;
    const int am = tid & 15;
    const int ai = i0 + (am >> 2) < H ? i0 + (am >> 2) : H - 1;
    const int aj = j0 + (am & 3) < W ? j0 + (am & 3) : W - 1;
    const float* ap = fm0 + (size_t)b * C * HW + ai * W + aj;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const bool active = wave < ntg;

    for (int c0 = 0; c0 < C; c0 += CA) {
        const int cc = C - c0 < CA ? C - c0 : CA;
        if (AST) {
        __syncthreads();
        for (int e = tid; e < CA * 16; e += THREADS) { const int c = e >> 4; smem[e] = c < cc ? ap[(size_t)(c0 + c) * HW] : 0.f; }
        __syncthreads();
        }
        if (active) {
            const float* bq = fm1 + ((size_t)b * C + c0) * HW;
            const int nfull = cc >> 2;
            for (int ks = 0; ks < nfull; ++ks) {
                f32x4 q;
                if (MODE == 3) { q = acc0; q.x = (float)ks; }
                else q = *reinterpret_cast<const f32x4u*>(bq + ks * kstride + boff);
                const float a_ = smem[(ks * 4 + g) * 16 + n];
                acc0 = MFMA(a_, q.x, acc0);
                if (MODE != 2) { acc1 = MFMA(a_, q.y, acc1); acc2 = MFMA(a_, q.z, acc2); acc3 = MFMA(a_, q.w, acc3); }
                else { acc1.x += q.y; acc2.x += q.z; acc3.x += q.w; }
            }
        }
    }
    if (!EPI) { if (active) out[(size_t)blockIdx.x * THREADS + tid] = acc0.x + acc1.x + acc2.x + acc3.x; return; }
    __syncthreads();
    for (int e = tid; e < 16 * CELLS; e += THREADS) smem[e] = 0.f;
    __syncthreads();
    if (active && gsel < NG) {
        const int ci = wr - g;
        if (ci >= 0 && ci < 2 * DT) {
            float* row = smem + (4 * g) * CELLS + ci * CW;
            const f32x4 a4[4] = {acc0, acc1, acc2, acc3};
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int cj = djs + t - (j0 + r) + DT; if (cj >= 0 && cj < 2 * DT) row[r * CELLS + cj] = a4[t][r]; }
        }
    }
    __syncthreads();
    const int nj = W - j0 < TP ? W - j0 : TP;
    for (int pi = 0; pi < TP; ++pi) {
        const int i = i0 + pi;
        if (i >= H) break;
        float* dst = out + (((size_t)b * H + i) * W + j0) * CELLS;
        const float* src = smem + pi * 4 * CELLS;
        for (int e = tid; e < nj * CELLS; e += THREADS) dst[e] = src[e];
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

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, C = argc > 2 ? atoi(argv[2]) : 256, H = 38, W = argc > 3 ? atoi(argv[3]) : 63;
    const size_t in_n = (size_t)B * C * H * W, out_n = (size_t)B * H * W * CELLS;
    const int NSETS = 6;
    std::vector<float*> f0(NSETS), f1(NSETS), o(NSETS);
    std::vector<float> h(in_n);
    for (size_t i = 0; i < in_n; ++i) h[i] = (float)rand() / RAND_MAX;
    for (int s = 0; s < NSETS; ++s) {
        hipMalloc(&f0[s], in_n * 4); hipMalloc(&f1[s], in_n * 4); hipMalloc(&o[s], out_n * 4);
        hipMemcpy(f0[s], h.data(), in_n * 4, hipMemcpyHostToDevice); hipMemcpy(f1[s], h.data(), in_n * 4, hipMemcpyHostToDevice);
    }
    const int ti = (H + 3) / 4, tj = (W + 3) / 4, blocks = B * ti * tj;
    printf("B=%d C=%d H=%d W=%d blocks=%d\n", B, C, H, W, blocks);
#define RUN(MODE, PW, NAME) { float us = time_it([&](int i) { hipLaunchKernelGGL((k_fwd<MODE, PW>), dim3(blocks), dim3(THREADS), 0, 0, f0[i % NSETS], f1[i % NSETS], o[i % NSETS], C, H, W, ti, tj); }, 50); printf("%-40s %8.1f us\n", NAME, us); }
    RUN(0, 8, "full, simple loop, 8 waves/SIMD cap");
    RUN(0, 1, "full, simple loop, no cap");
    RUN(1, 8, "aligned loads (wrong data)");
    RUN(2, 8, "loads only (1 MFMA/step)");
    RUN(3, 8, "MFMA only (no streamed loads)");
    RUN(4, 8, "full, no XCD remap");
#define RUN2(MODE, PW, EPI, AST, NAME) { float us = time_it([&](int i) { hipLaunchKernelGGL((k_fwd<MODE, PW, EPI, AST>), dim3(blocks), dim3(THREADS), 0, 0, f0[i % NSETS], f1[i % NSETS], o[i % NSETS], C, H, W, ti, tj); }, 50); printf("%-40s %8.1f us\n", NAME, us); }
    RUN2(0, 8, 0, 1, "full, no epilogue");
    RUN2(0, 8, 1, 0, "full, no A staging");
    RUN2(0, 8, 0, 0, "loads+MFMA only");
    RUN2(3, 8, 0, 0, "pure MFMA loop (+ds_read)");
    RUN2(3, 8, 0, 1, "MFMA + A staging");
    RUN2(3, 8, 1, 0, "MFMA + epilogue");
    RUN2(2, 8, 0, 0, "pure loads loop");
    return 0;
}
