// bwd_stamp_lab.hip -- developer harness (not product): builds the round-2 16-wave backward strip kernel (lab/d2t_corr_bwd16.inc) with
// per-wave clock reads and prints where the waves of a workgroup of the metric shape spend their cycles.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DD2T_LAB -DD2T_LAB_KERNELS lab/bwd_stamp_lab.hip -o lab/bwd_stamp_lab
#define D2T_LAB 1
#include "../../detect-to-track_amd/csrc/d2t_corr_tuned.hip"
#include "../../detect-to-track_amd/csrc/d2t_corr_fwd_band.hip"     // the rest of the correlation units the tuned one links against
#include "../../detect-to-track_amd/csrc/d2t_corr_bwd8.hip"
#include "d2t_corr_bwd8w.hip"
#include "d2t_corr_bwd8bf.hip"
#include <algorithm>
#include <cstdio>
#include <vector>

using namespace d2t::tuned;

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 8, C = argc > 2 ? atoi(argv[2]) : 256, H = 38, W = argc > 3 ? atoi(argv[3]) : 63;
    const int NS = 4, iters = 40;
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
    const int blocks = 2 * B * tiles_j * ((C + ST_CH - 1) / ST_CH);
    unsigned long long* st;
    hipMalloc(&st, (size_t)blocks * ST_WAVES * 8 * 8);
    hipMemset(st, 0, (size_t)blocks * ST_WAVES * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(lab_wave_stamps), &st, sizeof(st));
    auto run = [&](int i) { corr_bwd_f32(go[i % NS], f0[i % NS], f1[i % NS], g0[i % NS], g1[i % NS], B, C, H, W, 8, 1, nullptr, 0); };
    for (int i = 0; i < 2 * NS; ++i) run(i);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) run(i);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("B=%d C=%d %dx%d: %d workgroups, %.1f us per launch (stamped build)\n", B, C, H, W, blocks, ms * 1000.f / iters);
    std::vector<unsigned long long> s((size_t)blocks * ST_WAVES * 8);
    hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
    const char* names[] = {"prologue (descriptors, first G, barrier)", "loop", "final stores", "  loop: k-blocks (MFMA + fetch)",
                           "  loop: ring put + G load issue", "  loop: barrier wait", "  loop: tile store + rotate"};
    for (int role = 0; role < 2; ++role) {
        std::vector<double> v[7], tot;
        for (int w = 0; w < blocks; ++w) {
            const int xcd = w & 7, qq = blocks >> 3, rr = blocks & 7;    // xcd_remap on the host, then the kernel's decode
            const int bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (w >> 3);
            if (((bid / tiles_j) & 1) != role) continue;
            for (int k = 0; k < ST_WAVES; ++k) {
                const unsigned long long* p = &s[((size_t)w * ST_WAVES + k) * 8];
                v[0].push_back((double)(p[1] - p[0])); v[1].push_back((double)(p[2] - p[1])); v[2].push_back((double)(p[3] - p[2]));
                v[3].push_back((double)p[4]); v[4].push_back((double)p[5]); v[5].push_back((double)p[6]); v[6].push_back((double)p[7]);
                tot.push_back((double)(p[3] - p[0]));
            }
        }
        printf("role %d (median over waves, shader clock cycles)\n", role);
        for (int k = 0; k < 7; ++k) printf("  %-44s %9.0f\n", names[k], med(v[k]));
        printf("  %-44s %9.0f\n", "wave total", med(tot));
    }
    // split by actual role: recompute role from xcd_remap on the host
    return 0;
}
