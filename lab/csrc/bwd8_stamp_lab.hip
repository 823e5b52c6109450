// bwd8_stamp_lab.hip -- developer harness (not product): builds the PRODUCT 8-wave backward strip kernel with per-wave
// clock reads and prints where the waves of a workgroup of the metric shape spend their cycles, and the shader clock
// the chip holds under this load (s_memtime / s_memrealtime).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DD2T_LAB lab/bwd8_stamp_lab.hip -o lab/bwd8_stamp_lab
#define D2T_LAB 1
#include "../../detect-to-track_amd/csrc/d2t_corr_bwd8.hip"
#include <algorithm>
#include <cstdio>
#include <vector>

using namespace d2t::tuned;

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 8, C = argc > 2 ? atoi(argv[2]) : 256, H = 38, W = argc > 3 ? atoi(argv[3]) : 63;
    const int NS = 4, iters = 200;
    const size_t nin = (size_t)B * C * H * W, nout = (size_t)B * H * W * 289;
    std::vector<float> h(nin), hg(nout);
    for (size_t i = 0; i < nin; ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f;
    for (size_t i = 0; i < nout; ++i) hg[i] = (float)((i * 40503u >> 4) & 0xffff) / 65536.f;
    float *f0[NS], *f1[NS], *go[NS], *g0[NS], *g1[NS];
    for (int s = 0; s < NS; ++s) {
        hipMalloc(&f0[s], nin * 4); hipMalloc(&f1[s], nin * 4); hipMalloc(&go[s], nout * 4);
        hipMalloc(&g0[s], nin * 4); hipMalloc(&g1[s], nin * 4);
        hipMemcpy(f0[s], h.data(), nin * 4, hipMemcpyHostToDevice);
        hipMemcpy(f1[s], h.data(), nin * 4, hipMemcpyHostToDevice);
        hipMemcpy(go[s], hg.data(), nout * 4, hipMemcpyHostToDevice);
    }
    const int tiles_j = (W + 3) / 4;
    const int blocks = 2 * B * tiles_j * ((C + 255) / 256), NW = 8;
    unsigned long long* st;
    hipMalloc(&st, (size_t)blocks * NW * 16 * 8);
    hipMemset(st, 0, (size_t)blocks * NW * 16 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(lab8_stamps), &st, sizeof(st));
    // ablations (timing only; results are wrong): 1 no S reloads, 2 no tile stores, 4 no G loads, 8 no ring writes
    typedef void (*launch_t)(const float*, const float*, const float*, float*, float*, int, int, int, int);
    const launch_t abl[] = {lab8_launch<0>, lab8_launch<1>, lab8_launch<2>, lab8_launch<4>, lab8_launch<12>, lab8_launch<6>, lab8_launch<7>, lab8_launch<15>};
    const char* abl_name[] = {"all in", "no S reloads", "no tile stores", "no G loads", "no G loads, no ring writes", "no stores, no G loads", "no S, no stores, no G loads", "MFMA + LDS reads + barriers only"};
    {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rnd = 0; rnd < 3; ++rnd)
            for (int v = 0; v < 8; ++v) {
                for (int i = 0; i < NS; ++i) abl[v](go[i], f0[i], f1[i], g0[i], g1[i], B, C, H, W);
                hipEventRecord(a);
                for (int i = 0; i < 100; ++i) abl[v](go[i % NS], f0[i % NS], f1[i % NS], g0[i % NS], g1[i % NS], B, C, H, W);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                printf("round %d  %-36s %.1f us\n", rnd, abl_name[v], ms * 10.f);
            }
    }
    auto run = [&](int i) { abl[0](go[i % NS], f0[i % NS], f1[i % NS], g0[i % NS], g1[i % NS], B, C, H, W); };
    for (int i = 0; i < 2 * NS; ++i) run(i);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) run(i);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("B=%d C=%d %dx%d: %d workgroups, %.1f us per launch (stamped build)\n", B, C, H, W, blocks, ms * 1000.f / iters);
    std::vector<unsigned long long> s((size_t)blocks * NW * 16);
    hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"prologue (descriptors, S + G(0), barrier)", "loop", "final stores + drain", "  slice: tile store", "  slice: ring put",
                           "  slice: G load issue", "  barrier wait", "  rotate", "shader clock in the loop (MHz)"};
    for (int role = 0; role < 2; ++role) {
        std::vector<double> v[9], tot;
        for (int w = 0; w < blocks; ++w) {
            const int xcd = w & 7, qq = blocks >> 3, rr = blocks & 7;    // xcd_remap on the host, then the kernel's decode
            const int bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (w >> 3);
            if (((bid / tiles_j) & 1) != role) continue;
            for (int k = 0; k < NW; ++k) {
                const unsigned long long* p = &s[((size_t)w * NW + k) * 16];
                v[0].push_back((double)(p[1] - p[0])); v[1].push_back((double)(p[2] - p[1])); v[2].push_back((double)(p[3] - p[2]));
                for (int j = 0; j < 5; ++j) v[3 + j].push_back((double)p[6 + j]);
                v[8].push_back((double)(p[2] - p[1]) / (double)(p[5] - p[4]) * 100.0);
                tot.push_back((double)(p[3] - p[0]));
            }
        }
        printf("role %d (median over waves, shader clock cycles)\n", role);
        for (int k = 0; k < 9; ++k) printf("  %-44s %9.0f\n", names[k], med(v[k]));
        printf("  %-44s %9.0f\n", "wave total", med(tot));
    }
    // per-wave view of one interior workgroup: barrier wait by wave
    for (int w = 40; w < 42; ++w) {
        printf("workgroup %d: barrier wait by wave:", w);
        for (int k = 0; k < NW; ++k) printf(" %llu", s[((size_t)w * NW + k) * 16 + 9]);
        printf("\n");
    }
    return 0;
}
