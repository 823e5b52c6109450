// occ_lab.hip -- developer harness (not product): how many workgroups of a given shape does a CU of the MI355X hold AT A TIME?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o occ_lab occ_lab.hip && ./occ_lab
// Every workgroup reads the 100 MHz clock at entry and exit and idles ~20 us in between; the host counts how many lifetimes overlap.
// Asked because k_roipool_fwd_sat2<7> (1,024 threads, 60 VGPRs, 55.7 KB of dynamic LDS) runs ONE workgroup per CU at a time although the
// runtime's occupancy query says two.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int THREADS, int REGS>
__global__ void __launch_bounds__(THREADS) k_idle(unsigned long long* st, int spin, float* sink)
{
    extern __shared__ float lds[];
    unsigned long long t0, t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    float keep[REGS];                                               // register pressure on request
#pragma unroll
    for (int i = 0; i < REGS; ++i) keep[i] = threadIdx.x * 0.5f + i;
    lds[threadIdx.x] = keep[0];
    __syncthreads();
    for (int it = 0; it < spin; ++it) {
#pragma unroll
        for (int i = 0; i < REGS; ++i) keep[i] = keep[i] * 1.0001f + lds[(threadIdx.x + i) % THREADS];
        __builtin_amdgcn_s_sleep(8);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) s += keep[i];
    if (s == 12345.678f) sink[0] = s;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) { st[2 * blockIdx.x] = t0; st[2 * blockIdx.x + 1] = t1; }
}

template <int THREADS, int REGS>
static void run(int nwg, int lds_bytes, int spin)
{
    unsigned long long* st; float* sink;
    hipMalloc(&st, nwg * 16); hipMalloc(&sink, 4);
    auto k = k_idle<THREADS, REGS>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, THREADS, lds_bytes);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k));
    for (int rep = 0; rep < 2; ++rep) { hipMemset(st, 0, nwg * 16); hipLaunchKernelGGL(k, dim3(nwg), dim3(THREADS), lds_bytes, 0, st, spin, sink); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(2 * nwg);
    hipMemcpy(h.data(), st, nwg * 16, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0;
    for (int i = 0; i < nwg; ++i) { lo = std::min(lo, h[2 * i]); hi = std::max(hi, h[2 * i + 1]); }
    int maxc = 0;
    for (int s = 1; s < 40; ++s) {
        const unsigned long long m = lo + (hi - lo) * s / 40;
        int c = 0;
        for (int i = 0; i < nwg; ++i) c += h[2 * i] <= m && h[2 * i + 1] >= m;
        maxc = std::max(maxc, c);
    }
    printf("threads %4d  VGPRs %3d  LDS %6d B  %4d workgroups: occupancy query %d per CU, resident at a time (max of 39 samples) %4d = %.2f per CU, span %.1f us\n",
           THREADS, fa.numRegs, lds_bytes, nwg, occ, maxc, maxc / 256.0, (hi - lo) / 100.0);
    hipFree(st); hipFree(sink);
}

int main()
{
    const int spin = 300;
    run<1024, 8>(512, 55712, spin);
    run<1024, 20>(512, 55712, spin);
    run<1024, 24>(512, 55712, spin);
    run<1024, 26>(512, 55712, spin);
    run<1024, 28>(512, 55712, spin);
    run<1024, 30>(512, 55712, spin);
    run<1024, 32>(512, 55712, spin);
    run<1024, 40>(512, 55712, spin);
    run<768, 40>(512, 55712, spin);
    run<512, 40>(512, 55712, spin);
    run<512, 8>(1024, 30000, spin);
    return 0;
}
