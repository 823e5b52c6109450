// div_lab.hip -- is q1 = fma(fma(-n, a * r, a), r, a * r) with r = 1.0f / n (IEEE) the correctly rounded a / n for the divisors ROIPool
// meets (n = bin pixel counts, 1 .. 65,025, and their negatives)?  Exhaustive over n, pseudo-random a across the whole f32 range (normal,
// denormal, huge, +-0, Inf, NaN) plus the neighbourhood of every representable quotient's rounding boundary is not feasible; this lab
// samples 2^20 values of a per n (6.8e10 pairs) and counts bit mismatches against the compiler's IEEE division.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off div_lab.hip -o div_lab && ./div_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void k(unsigned long long* bad, unsigned long long* bad_safe, int nmax, int samples, unsigned* first)
{
    const int n0 = blockIdx.x + 1;
    for (int sgn = 0; sgn < 2; ++sgn) {
        const float nf = sgn ? -(float)n0 : (float)n0;
        const float r = 1.0f / nf;
        unsigned long long b = 0, bs = 0;
        for (int s = threadIdx.x; s < samples; s += blockDim.x) {
            uint32_t u = mix((uint32_t)s * 2654435761u + (uint32_t)n0 * 40503u + sgn);
            if ((s & 7) == 0) u = (u & 0x807fffffu) | ((100u + (u >> 23) % 60u) << 23);       // typical magnitudes 2^-27 .. 2^32
            const float a = __uint_as_float(u);
            const float want = a / nf;
            const float q0 = a * r;
            const float e = __builtin_fmaf(-nf, q0, a);
            const float q1 = __builtin_fmaf(e, r, q0);
            const bool safe = __builtin_fabsf(q0) < 1e30f && (__builtin_fabsf(q0) > 1e-30f || q0 == 0.0f);
            const bool same = __float_as_uint(q1) == __float_as_uint(want) || (want != want && q1 != q1);
            if (!same) { ++b; if (safe && u != 0x80000000u) { ++bs; atomicCAS(first, 0u, u); } }   // a = -0 cannot be a running sum that started at +0
        }
        atomicAdd(bad, b); atomicAdd(bad_safe, bs);
    }
}
int main()
{
    unsigned long long *bad, *bad_safe; unsigned* first;
    hipMalloc(&bad, 8); hipMalloc(&bad_safe, 8); hipMalloc(&first, 4);
    hipMemset(bad, 0, 8); hipMemset(bad_safe, 0, 8); hipMemset(first, 0, 4);
    const int nmax = 65025, samples = 1 << 20;
    hipLaunchKernelGGL(k, dim3(nmax), dim3(256), 0, 0, bad, bad_safe, nmax, samples, first);
    unsigned long long hb = 0, hs = 0; unsigned hf = 0;
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hs, bad_safe, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
    printf("pairs %.3e  mismatches (all a) %llu  mismatches with (|q0| in [1e-30, 1e30] or q0 == 0) and a != -0: %llu  first bad a bits 0x%08x\n",
           2.0 * nmax * samples, hb, hs, hf);
    return 0;
}
