// bf16x3_lab.hip -- developer harness (not product): VERDICT round 2, item 10 -- what would the correlation's channel
// contraction cost, and how accurate would it be, on the bf16 matrix pipe with every f32 operand split into three
// bf16 pieces (x = hi + mid + lo, 6 of the 9 piece products kept: hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi)?
//
// Timing (every wave of the chip, operands in registers, 4 independent accumulators as in the product kernels;
// s_memtime ticks per K = 128 channels of 4 tiles AND wall-clock time per MFMA per SIMD from HIP events -- trust the
// latter: under full matrix load the s_memtime counter slows down with the clock):
//   f32      32 x v_mfma_f32_16x16x4_f32                               (the product's instruction)
//   presplit 24 x v_mfma_f32_16x16x32_bf16 on pieces split beforehand  (the matrix time alone)
//   inline   the same with BOTH operands split in registers per use    (no reuse of a split: worst case)
//   inlineB  A pieces split beforehand, B split per use                (the forward's shape: FM0 pixels are reused by
//                                                                       every tile-group, FM1 windows are not)
// Numerics: one 16 x 16 tile, K channels, U[0,1) and N(0,1) data: max |error| / sum |terms| against a double
// reference for the f32 MFMA chain and for the split form (the contract is 1e-5).
//
//   hipcc -O3 --offload-arch=gfx950 -o bf16x3_lab bf16x3_lab.hip && ./bf16x3_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct Split { bf16x8 hi, mid, lo; };

__device__ __forceinline__ Split split3(const f32x8 x)
{
    Split s;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];                               // round to nearest even
        const float r1 = x[i] - (float)h;                            // exact
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;                              // exact
        s.hi[i] = h; s.mid[i] = m; s.lo[i] = (__bf16)r2;
    }
    return s;
}

// acc += a . b over 32 channels, six piece products, smallest first
__device__ __forceinline__ f32x4 mac6(const Split& a, const Split& b, f32x4 acc)
{
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.mid, b.mid, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.mid, b.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.mid, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, acc, 0, 0, 0);
    return acc;
}

template <int MODE>   // 0 f32, 1 presplit, 2 inline (both), 3 inline B only, 4 presplit with the K = 16 instruction (48 MFMAs)
__global__ void __launch_bounds__(1024) k_time(const float* __restrict__ src, float* __restrict__ sink, unsigned long long* __restrict__ cyc, int iters)
{
    const int lane = threadIdx.x & 63;
    f32x8 xa, xb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { xa[i] = src[(lane * 8 + i) & 1023]; xb[i] = src[(lane * 8 + i + 512) & 1023]; }
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    Split sa = split3(xa), sb = split3(xb);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {                             // one iteration = K 128 for 4 tiles (4 accumulators)
        if (MODE == 4) {
            const s16x4 p0 = __builtin_bit_cast(s16x4, __builtin_shufflevector(sa.hi, sa.hi, 0, 1, 2, 3)), p1 = __builtin_bit_cast(s16x4, __builtin_shufflevector(sb.hi, sb.hi, 0, 1, 2, 3));
            const s16x4 p2 = __builtin_bit_cast(s16x4, __builtin_shufflevector(sa.mid, sa.mid, 0, 1, 2, 3)), p3 = __builtin_bit_cast(s16x4, __builtin_shufflevector(sb.lo, sb.lo, 0, 1, 2, 3));
#pragma unroll
            for (int k = 0; k < 8; ++k)                              // 8 blocks of 16 channels, 6 products each
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(p2, p3, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(p0, p3, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(p2, p1, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(p0, p1, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(p2, p2, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(p0, p0, acc[t], 0, 0, 0);
                }
        } else if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 32; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[k & 7], xb[(k + t) & 7], acc[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                            // 4 blocks of 32 channels
                if (MODE == 2) {
                    xa[k] += 1.0f;                                   // (new values every block: the split cannot be hoisted)
                    sa = split3(xa);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (MODE >= 2) {
                        xb[(k + t) & 7] += 1.0f;
                        sb = split3(xb);
                    }
                    acc[t] = mac6(sa, sb, acc[t]);
                }
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

// one tile: D[m][n] = sum_k A[m][k] B[k][n], A (16, K), B (K, 16) row-major; one wave
__global__ void __launch_bounds__(64) k_tile(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ Df32, float* __restrict__ Dbf, int K)
{
    const int lane = threadIdx.x, mn = lane & 15, g = lane >> 4;
    f32x4 a32 = {0.f, 0.f, 0.f, 0.f}, a16 = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4)                                // ascending channels, 4 per instruction: the product's chain
        a32 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[mn * K + k0 + g], B[(k0 + g) * 16 + mn], a32, 0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += 32) {
        f32x8 xa, xb;
#pragma unroll
        for (int i = 0; i < 8; ++i) { xa[i] = A[mn * K + k0 + 8 * g + i]; xb[i] = B[(k0 + 8 * g + i) * 16 + mn]; }
        a16 = mac6(split3(xa), split3(xb), a16);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { Df32[(4 * g + r) * 16 + mn] = a32[r]; Dbf[(4 * g + r) * 16 + mn] = a16[r]; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main()
{
    // ---- timing
    const int wgs = 256, iters = 2000;
    float *src, *sink; unsigned long long* cyc;
    CK(hipMalloc(&src, 1024 * 4)); CK(hipMalloc(&sink, wgs * 1024 * 4)); CK(hipMalloc(&cyc, wgs * 16 * 8));
    std::vector<float> h(1024);
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    for (auto& v : h) v = u(rng);
    CK(hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice));
    const char* names[5] = {"f32 16x16x4 (32 MFMAs)", "bf16x3 presplit (24 MFMAs)", "bf16x3, A and B split per use", "bf16x3, B split per use", "bf16x3 presplit, K=16 instr (48)"};
    for (int wpb = 256; wpb <= 1024; wpb *= 2)                       // 1, 2 and 4 waves per SIMD
        for (int mode = 0; mode < 5; ++mode) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL(k_time<0>, dim3(wgs), dim3(wpb), 0, 0, src, sink, cyc, iters);
                if (mode == 1) hipLaunchKernelGGL(k_time<1>, dim3(wgs), dim3(wpb), 0, 0, src, sink, cyc, iters);
                if (mode == 2) hipLaunchKernelGGL(k_time<2>, dim3(wgs), dim3(wpb), 0, 0, src, sink, cyc, iters);
                if (mode == 3) hipLaunchKernelGGL(k_time<3>, dim3(wgs), dim3(wpb), 0, 0, src, sink, cyc, iters);
                if (mode == 4) hipLaunchKernelGGL(k_time<4>, dim3(wgs), dim3(wpb), 0, 0, src, sink, cyc, iters);
                CK(hipEventRecord(e1, 0));
                CK(hipDeviceSynchronize());
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            std::vector<unsigned long long> c(wgs * (wpb / 64));
            CK(hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost));
            double mean = 0;
            for (auto v : c) mean += (double)v;
            mean /= c.size();
            const double mfmas = (mode == 0 ? 128.0 : mode == 4 ? 192.0 : 96.0) * iters * (wpb / 256);   // per SIMD
            printf("%d wave(s)/SIMD  %-34s %9.1f s_memtime cycles per (K=128 x 4 tiles) per wave; kernel %.3f ms = %.2f ns per MFMA per SIMD\n",
                   wpb / 256, names[mode], mean / iters, ms, ms * 1e6 / mfmas);
        }
    // ---- numerics
    for (int dist = 0; dist < 2; ++dist)
        for (int K : {256, 512, 1024, 2048}) {
            std::vector<float> A(16 * K), B(K * 16);
            std::normal_distribution<float> nd(0.f, 1.f);
            for (auto& v : A) v = dist ? nd(rng) : u(rng);
            for (auto& v : B) v = dist ? nd(rng) : u(rng);
            float *dA, *dB, *d32, *d16;
            CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&d32, 1024)); CK(hipMalloc(&d16, 1024));
            CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(k_tile, dim3(1), dim3(64), 0, 0, dA, dB, d32, d16, K);
            CK(hipDeviceSynchronize());
            std::vector<float> r32(256), r16(256);
            CK(hipMemcpy(r32.data(), d32, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(r16.data(), d16, 1024, hipMemcpyDeviceToHost));
            double e32 = 0, e16 = 0;
            for (int m = 0; m < 16; ++m)
                for (int n = 0; n < 16; ++n) {
                    double ref = 0, mag = 0;
                    for (int k = 0; k < K; ++k) { const double p = (double)A[m * K + k] * (double)B[k * 16 + n]; ref += p; mag += std::fabs(p); }
                    e32 = std::fmax(e32, std::fabs(r32[m * 16 + n] - ref) / mag);
                    e16 = std::fmax(e16, std::fabs(r16[m * 16 + n] - ref) / mag);
                }
            printf("%s K=%4d  max |err| / sum|terms|:  f32 MFMA chain %.2e   bf16x3 (6 products) %.2e\n", dist ? "N(0,1)" : "U[0,1)", K, e32, e16);
            (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(d32); (void)hipFree(d16);
        }
    return 0;
}
