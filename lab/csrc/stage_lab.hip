// stage_lab.hip -- tuning harness (developer tool): how fast can one CU pull 16-byte window pieces
// of a (C, H, W) map into LDS?  mode 0: LDS-DMA (buffer_load_dwordx4 ... lds), mode 1: global
// load to registers + ds_write_b128, both with NW waves per workgroup and one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
constexpr int KC = 16, NCG = 5;

template <int MODE, int NW, int ROWS>
__device__ __forceinline__ void stage_body(const float* __restrict__ fm, float* __restrict__ sink, int C, int H, int W, int nchunks)
{
    constexpr int SLOTS = (ROWS * NCG + 15) / 16 * 16;
    constexpr int NI = KC * SLOTS / 64;                  // wave-instructions per chunk
    constexpr int PER = (NI + NW - 1) / NW;
    __shared__ __attribute__((aligned(16))) float smem[MODE == 3 ? 3 : 2][KC * SLOTS * 4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int HW = H * W;
    const int tile = blockIdx.x;                          // window origin varies per block
    const int r0 = (tile * 4) % (H - ROWS > 0 ? H - ROWS : 1), c0 = (tile * 4) % (W - 20);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fm), 0, (unsigned)C * HW * 4u, 0x00020000);
    int voff[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int x = wave + NW * k, e = x * 64 + lane;
        const int ch = e / SLOTS, rem = e - ch * SLOTS, row = rem / NCG, cg = rem - row * NCG;
        voff[k] = (x < NI && row < ROWS) ? (ch * HW + (r0 + row) * W + c0 + 4 * cg) * 4 : 0x7ffffff0;
    }
    float acc = 0.f;
    for (int chn = 0; chn < nchunks; ++chn) {
        float* buf = smem[MODE == 3 ? chn % 3 : (chn & 1)];
        const int cb = (chn % (C / KC)) * KC * HW * 4;
        if (MODE == 2) {
            constexpr int NI4 = KC * SLOTS * 4 / 64;             // dword instructions per chunk
            constexpr int PER4 = (NI4 + NW - 1) / NW;
#pragma unroll
            for (int k = 0; k < PER4; ++k) {
                const int x = wave + NW * k, f = x * 64 + lane;
                const int ch = f / (SLOTS * 4), rem = f - ch * (SLOTS * 4), row = rem / 20, col = rem - row * 20;
                const int vo = (x < NI4 && row < ROWS) ? (ch * HW + (r0 + row) * W + c0 + col) * 4 + cb : 0x7ffffff0;
                if (x < NI4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(buf + x * 64), 4, vo, 0, 0, 0);
            }
        } else if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int x = wave + NW * k;
                if (x < NI) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(buf + x * 256), 16, voff[k] == 0x7ffffff0 ? voff[k] : voff[k] + cb, 0, 0, 0);
            }
        } else {
            f32x4 v[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int x = wave + NW * k;
                v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (x < NI && voff[k] != 0x7ffffff0)
                    v[k] = *reinterpret_cast<const f32x4u*>(reinterpret_cast<const char*>(fm) + voff[k] + cb);
            }
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int x = wave + NW * k;
                if (x < NI) *reinterpret_cast<f32x4*>(buf + x * 256 + lane * 4) = v[k];
            }
        }
        if (MODE == 3) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory"); }
        else { __syncthreads(); acc += buf[(lane * 37 + chn) & (KC * SLOTS * 4 - 1)]; }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

__global__ void __launch_bounds__(3 * 64) k_stage_0_3_19(const float* fm, float* sink, int C, int H, int W, int n) { stage_body<0, 3, 19>(fm, sink, C, H, W, n); }
__global__ void __launch_bounds__(3 * 64) k_stage_1_3_19(const float* fm, float* sink, int C, int H, int W, int n) { stage_body<1, 3, 19>(fm, sink, C, H, W, n); }
__global__ void __launch_bounds__(6 * 64) k_stage_0_6_19(const float* fm, float* sink, int C, int H, int W, int n) { stage_body<0, 6, 19>(fm, sink, C, H, W, n); }
__global__ void __launch_bounds__(6 * 64) k_stage_1_6_19(const float* fm, float* sink, int C, int H, int W, int n) { stage_body<1, 6, 19>(fm, sink, C, H, W, n); }
__global__ void __launch_bounds__(15 * 64) k_stage_0_15_35(const float* fm, float* sink, int C, int H, int W, int n) { stage_body<0, 15, 35>(fm, sink, C, H, W, n); }
__global__ void __launch_bounds__(15 * 64) k_stage_1_15_35(const float* fm, float* sink, int C, int H, int W, int n) { stage_body<1, 15, 35>(fm, sink, C, H, W, n); }
__global__ void __launch_bounds__(12 * 64) k_stage_0_12_19(const float* fm, float* sink, int C, int H, int W, int n) { stage_body<0, 12, 19>(fm, sink, C, H, W, n); }
__global__ void __launch_bounds__(12 * 64) k_stage_1_12_19(const float* fm, float* sink, int C, int H, int W, int n) { stage_body<1, 12, 19>(fm, sink, C, H, W, n); }

__global__ void __launch_bounds__(3 * 64) k_stage_2_3_19(const float* fm, float* sink, int C, int H, int W, int n) { stage_body<2, 3, 19>(fm, sink, C, H, W, n); }
__global__ void __launch_bounds__(15 * 64) k_stage_2_15_35(const float* fm, float* sink, int C, int H, int W, int n) { stage_body<2, 15, 35>(fm, sink, C, H, W, n); }
__global__ void __launch_bounds__(3 * 64) k_stage_3_3_19(const float* fm, float* sink, int C, int H, int W, int n) { stage_body<3, 3, 19>(fm, sink, C, H, W, n); }
__global__ void __launch_bounds__(6 * 64) k_stage_3_6_19(const float* fm, float* sink, int C, int H, int W, int n) { stage_body<3, 6, 19>(fm, sink, C, H, W, n); }
__global__ void __launch_bounds__(15 * 64) k_stage_3_15_35(const float* fm, float* sink, int C, int H, int W, int n) { stage_body<3, 15, 35>(fm, sink, C, H, W, n); }
static void report(const char* name, float ms, int nchunks, int rows)
{
    const double us = ms * 1000.0 / 10, kb = KC * rows * 80.0 / 1024.0;
    printf("%-28s %8.1f us  %6.3f us/chunk  %6.1f KB/chunk  %6.1f KB/us/CU\n", name, us, us / nchunks, kb, kb * nchunks / us);
}
#define RUN(MODE, NW, ROWS, NAME)                                                                              \
    {                                                                                                          \
        hipEvent_t e0, e1;                                                                                     \
        hipEventCreate(&e0); hipEventCreate(&e1);                                                              \
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_stage_##MODE##_##NW##_##ROWS, dim3(256), dim3(NW * 64), 0, 0, fm, sink, C, H, W, nchunks); \
        hipEventRecord(e0);                                                                                    \
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k_stage_##MODE##_##NW##_##ROWS, dim3(256), dim3(NW * 64), 0, 0, fm, sink, C, H, W, nchunks); \
        hipEventRecord(e1); hipEventSynchronize(e1);                                                           \
        float ms; hipEventElapsedTime(&ms, e0, e1);                                                            \
        report(NAME, ms, nchunks, ROWS);                                                                       \
    }

int main()
{
    const int C = 2048, H = 38, nchunks = 128;
    float *fmw, *sink;
    hipMalloc(&fmw, (size_t)C * H * 80 * 4); hipMalloc(&sink, 64);
    hipMemset(fmw, 0, (size_t)C * H * 80 * 4);
    const float* fm = fmw;
    for (int W : {75, 63, 64, 76}) {                       // 64, 76: every row 16-byte aligned
        printf("W = %d\n", W);
        RUN(0, 3, 19, "dma  3 waves 19 rows")
        RUN(1, 3, 19, "regs 3 waves 19 rows")
        RUN(3, 3, 19, "dma  3 waves 19 rows, 3 in flight")
        RUN(3, 6, 19, "dma  6 waves 19 rows, 3 in flight")
        RUN(0, 15, 35, "dma  15 waves 35 rows")
        RUN(1, 15, 35, "regs 15 waves 35 rows")
        RUN(3, 15, 35, "dma 15 waves 35 rows, 3 in flight")
    }
    return 0;
}
