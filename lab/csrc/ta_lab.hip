// ta_lab.hip -- developer harness (not product): what does one S-fragment load of the backward strip kernel cost the
// texture-address unit, by lane -> address mapping?  Every wave of the chip streams 16-byte pieces of a (C, H, W) map,
// 16 channels x 64 bytes per wave-instruction, with DEPTH loads in flight; results are xor-ed so nothing is dead.
//   map 0: lane l -> (channel l & 15, piece l >> 4)      (the MFMA A-operand layout: neighbouring lanes = neighbouring channels)
//   map 1: lane l -> (channel l >> 2, piece l & 3)       (a quad of lanes = 64 contiguous bytes of one channel)
//   map 2: as map 1 through LDS-DMA (buffer_load ... lds), no VGPR destination
//   map 3: 1 KB contiguous per instruction (upper bound)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;

template <int MAP>
__global__ void __launch_bounds__(512) k_ta(const float* __restrict__ fm, unsigned* __restrict__ sink, int C, int H, int W, int iters)
{
    __shared__ __attribute__((aligned(16))) float smem[8 * 8 * 256];   // 8 waves x 8 slots x 1 KB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int HW = H * W;
    const int strip = blockIdx.x % 16, b = blockIdx.x / 16;
    const unsigned long long a = reinterpret_cast<unsigned long long>(fm + (size_t)(b % 8) * C * HW);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                                                         (unsigned)C * HW * 4u, 0x00020000);
    const int ch = MAP == 0 ? (lane & 15) : (lane >> 2), pc = MAP == 0 ? (lane >> 4) : (lane & 3);
    const int col0 = strip * 4 > W - 20 ? W - 20 : strip * 4;
    int voff = MAP == 3 ? wave * 32 * HW * 4 + lane * 16 : ((wave * 32 + ch) * HW + col0 + 4 * pc) * 4;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const int row = it % (H - 1);
#pragma unroll
        for (int k = 0; k < 8; ++k) {                     // 8 loads: 2 c-tiles x 4 rows (row-k-block mapping)
            const int v = voff + (k & 1) * 16 * HW * 4 + (MAP == 3 ? ((row + (k >> 1)) * 1024) % (16 * HW * 4 - 1024) : (((row + (k >> 1)) % H) * W) * 4);
            if (MAP == 2) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + (wave * 8 + k) * 256), 16, v, 0, 0, 0);
            } else {
                const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, v, 0, 0);
                acc ^= x;
            }
        }
    }
    if (MAP == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc[0] = __float_as_uint(smem[threadIdx.x]); }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[threadIdx.x] = acc[0];
}

int main()
{
    const int B = 8, C = 256, H = 38, W = 63, iters = 400;
    const size_t n = (size_t)B * C * H * W;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f;
    float* fm; unsigned* sink;
    hipMalloc(&fm, n * 4); hipMalloc(&sink, 4096); hipMemcpy(fm, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto t = [&](auto kern, const char* name) {
        for (int r = 0; r < 3; ++r) {
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, fm, sink, C, H, W, 10);
            hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, fm, sink, C, H, W, iters);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double ninstr = 8.0 * iters * 8;          // per CU
            printf("%-40s %8.1f us  = %6.1f ns per wave-instruction per CU (%.1f B/clk/CU at 2.4 GHz)\n", name, ms * 1e3, ms * 1e6 / ninstr,
                   1024.0 / (ms * 1e6 / ninstr * 2.4));
        }
    };
    t(k_ta<0>, "map 0: lane = (channel, piece)");
    t(k_ta<1>, "map 1: quad of lanes = 64 B of a channel");
    t(k_ta<2>, "map 2: map 1 via LDS-DMA");
    t(k_ta<3>, "map 3: 1 KB contiguous");
    return 0;
}
