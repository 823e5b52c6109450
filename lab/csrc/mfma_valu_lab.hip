// mfma_valu_lab.hip -- developer harness (not product): do v_mfma_f32_16x16x4_f32 and f32 vector instructions overlap on a SIMD, or do they
// share its f32 FMA lanes?  Every wave of the chip runs ITERS rounds of (M independent MFMAs, V independent v_fma_f32); four variants
// per occupancy: MFMA only, VALU only, both in ONE wave, and -- two waves per SIMD -- even waves MFMA only / odd waves VALU only.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o mfma_valu_lab mfma_valu_lab.hip && ./mfma_valu_lab
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int M, int V, bool SPLIT, int OP = 0>
__global__ void __launch_bounds__(512) k_mix(float* out, int iters, float a, float b)
{
    f32x4 acc[8];
    float v[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    const int wave = threadIdx.x >> 6;
    const bool do_m = !SPLIT || (wave & 4) == 0, do_v = !SPLIT || (wave & 4) != 0;   // SPLIT: waves 0-3 (one per SIMD) MFMA, waves 4-7 VALU
    for (int it = 0; it < iters; ++it) {
        if (M > 0 && do_m) {
#pragma unroll
            for (int i = 0; i < M; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 7], 0, 0, 0);
        }
        if (V > 0 && do_v) {
#pragma unroll
            for (int i = 0; i < V; ++i) {
                if (OP == 0) v[i & 15] = __builtin_fmaf(v[i & 15], a, b);
                else {                                               // integer / logic vector instruction
                    unsigned u = __builtin_bit_cast(unsigned, v[i & 15]);
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(u) : "v"(u), "v"(it));
                    v[i & 15] = __builtin_bit_cast(float, u);
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int M, int V, bool SPLIT, int OP = 0>
static float run(int threads, int iters, float* out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto k = k_mix<M, V, SPLIT, OP>;
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main()
{
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 2000;
    // per round: 8 MFMAs (8 x 32 = 256 matrix cycles) and 32 v_fma (32 x 4 = 128 vector cycles) per wave
    printf("one wave per SIMD (256 threads per CU), us: MFMA only %.1f | VALU only %.1f | both in one wave %.1f\n",
           run<8, 0, false>(256, iters, out), run<0, 32, false>(256, iters, out), run<8, 32, false>(256, iters, out));
    printf("two waves per SIMD (512 threads per CU), us: MFMA only (both waves) %.1f | VALU only (both) %.1f | both in every wave %.1f | "
           "wave A MFMA only + wave B VALU only %.1f (alone: MFMA %.1f, VALU %.1f with one wave per SIMD)\n",
           run<8, 0, false>(512, iters, out), run<0, 32, false>(512, iters, out), run<8, 32, false>(512, iters, out),
           run<8, 32, true>(512, iters, out), run<8, 0, false>(256, iters, out), run<0, 32, false>(256, iters, out));
    printf("the same with 64 v_fma per round: both in one wave %.1f | split over two waves %.1f\n",
           run<8, 64, false>(256, iters, out), run<8, 64, true>(512, iters, out));
    printf("v_xor_b32 instead of v_fma_f32 (32 per round), one wave per SIMD: VALU only %.1f | both in one wave %.1f; two waves per SIMD, split: %.1f\n",
           run<0, 32, false, 1>(256, iters, out), run<8, 32, false, 1>(256, iters, out), run<8, 32, true, 1>(512, iters, out));
    return 0;
}
