// fwd_lab.hip -- developer harness (not product): builds the PRODUCT forward kernel source with
// in-kernel stamps and prints where a workgroup of the metric shape spends its cycles.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DD2T_LAB lab/fwd_lab.hip -o lab/fwd_lab
#define D2T_LAB 1
#include "../../detect-to-track_amd/csrc/d2t_corr_tuned.hip"
#include "../../detect-to-track_amd/csrc/d2t_corr_fwd_band.hip"     // the rest of the correlation units the tuned one links against
#include "../../detect-to-track_amd/csrc/d2t_corr_bwd8.hip"
#include <algorithm>
#include <cstdio>
#include <vector>

using namespace d2t::tuned;

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 8, C = argc > 2 ? atoi(argv[2]) : 256, H = 38, W = argc > 3 ? atoi(argv[3]) : 63;
    const int NS = 6, iters = 60;
    const size_t nin = (size_t)B * C * H * W, nout = (size_t)B * H * W * 289;
    std::vector<float> h(nin);
    for (size_t i = 0; i < nin; ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f;
    float *f0[NS], *f1[NS], *o[NS];
    for (int s = 0; s < NS; ++s) {
        hipMalloc(&f0[s], nin * 4); hipMalloc(&f1[s], nin * 4); hipMalloc(&o[s], nout * 4);
        hipMemcpy(f0[s], h.data(), nin * 4, hipMemcpyHostToDevice);
        hipMemcpy(f1[s], h.data(), nin * 4, hipMemcpyHostToDevice);
    }
    const int tiles_i = (H + 3) / 4, tiles_j = (W + 3) / 4, nseg = (tiles_i + SG_NU - 1) / SG_NU;
    const int blocks = B * tiles_j * nseg;
    unsigned long long* st;
    hipMalloc(&st, (size_t)blocks * 32 * 8);
    hipMemset(st, 0, (size_t)blocks * 32 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(lab_stamps), &st, sizeof(st));
    for (int i = 0; i < 2 * NS; ++i) corr_fwd_f32(f0[i % NS], f1[i % NS], o[i % NS], B, C, H, W, 8, 1, nullptr, 0, nullptr);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) corr_fwd_f32(f0[i % NS], f1[i % NS], o[i % NS], B, C, H, W, 8, 1, nullptr, 0, nullptr);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("B=%d C=%d %dx%d: %d workgroups, %.1f us per launch (stamped build)\n", B, C, H, W, blocks, ms * 1000.f / iters);
    std::vector<unsigned long long> s((size_t)blocks * 32);
    hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"start -> chunk 0 landed (prologue)", "main loop", "loop end -> drained+barrier",
                           "zero-fill + scatter", "copy-out issue", "stores acknowledged"};
    std::vector<double> seg[6], tot, clk, t0s;
    unsigned long long tmin = ~0ull;
    for (int w = 0; w < blocks; ++w) tmin = std::min(tmin, s[w * 32 + 0]);
    for (int w = 0; w < blocks; ++w) {
        const unsigned long long* p = &s[(size_t)w * 32];
        for (int k = 0; k < 6; ++k) seg[k].push_back((double)(p[k + 1] - p[k]));
        tot.push_back((double)(p[6] - p[0]));
        clk.push_back((double)(p[6] - p[0]) / (double)(p[9] - p[8]) * 100e6 / 1e9);
        t0s.push_back((double)(p[0] - tmin));
    }
    const double ghz = med(clk);
    printf("in-kernel clock %.2f GHz (median over workgroups)\n", ghz);
    for (int k = 0; k < 6; ++k) printf("  %-38s %8.0f cycles  %6.2f us\n", names[k], med(seg[k]), med(seg[k]) / ghz / 1e3);
    printf("  %-38s %8.0f cycles  %6.2f us\n", "workgroup total", med(tot), med(tot) / ghz / 1e3);
    {
        std::vector<double> a, b2, c, dw, bw;
        for (int w = 0; w < blocks; ++w) {
            const unsigned long long* p = &s[(size_t)w * 32];
            a.push_back((double)(p[10] - p[0])); b2.push_back((double)(p[11] - p[10])); c.push_back((double)(p[12] - p[11]));
            dw.push_back((double)p[13]); bw.push_back((double)p[14]);
        }
        printf("  prologue: start->DMA issued %.0f, ->task tables done %.0f, ->own chunk-0 parts landed %.0f, ->barrier passed %.0f\n",
               med(a), med(b2), med(c), med(seg[0]) - med(a) - med(b2) - med(c));
        printf("  main loop, wave 0: waiting for its DMA %.0f cycles, at the barrier %.0f cycles (sum over chunks)\n", med(dw), med(bw));
    }
    for (int w = 0; w < 3; ++w) {
        printf("  wg %d wave->simd:", w * 97 % blocks);
        for (int k = 0; k < SG_WAVES; ++k) printf(" %llu", (s[(size_t)(w * 97 % blocks) * 32 + 16 + k] >> 4) & 3);
        printf("  cu %llu\n", (s[(size_t)(w * 97 % blocks) * 32 + 16] >> 8) & 15);
    }
    std::sort(t0s.begin(), t0s.end());
    printf("  start skew across workgroups: median %.0f, max %.0f cycles\n", t0s[t0s.size() / 2], t0s.back());
    return 0;
}
