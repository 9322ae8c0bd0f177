// Strip timeline of the BcResNet front kernel (trunk_b.hip compiled with -DNWW_TRACE): s_memtime stamps of workgroup 0's eight waves at the
// phase boundaries of its first 16 (clip, strip) items - item top | next strip's rows requested | conv + pool tasks done | behind barrier 1 |
// next rows split into the planes | depthwise done | behind barrier 2 - at full load (8192 clips, two workgroups per CU).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DNWW_TRACE -I nanowakeword_amd/csrc -I include tools/ubench/front_trace.hip -o tools/ubench/front_trace
// run:   tools/ubench/front_trace [B=8192]
#include "../../nanowakeword_amd/csrc/trunk_b.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8192, H = 101, W = 64;
    const int Ho = (H / 2 - 1) / 2 + 1, Wo = (W / 2 - 1) / 2 + 1;
    std::vector<float> x((size_t)B * H * W), w1(32 * 9), al(32), be(32), dw(9 * 32);
    uint32_t st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : x) v = 80.0f * rnd() - 40.0f;
    for (auto& v : w1) v = rnd();
    for (auto& v : al) v = 1.0f + rnd();
    for (auto& v : be) v = rnd();
    for (auto& v : dw) v = rnd();
    float *dx, *dw1, *dal, *dbe, *ddw, *d1, *x1;
    const size_t no = (size_t)B * Ho * Wo * 32;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dw1, w1.size() * 4); hipMalloc(&dal, 128); hipMalloc(&dbe, 128); hipMalloc(&ddw, dw.size() * 4);
    hipMalloc(&d1, no * 4); hipMalloc(&x1, no * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dal, al.data(), 128, hipMemcpyHostToDevice); hipMemcpy(dbe, be.data(), 128, hipMemcpyHostToDevice);
    hipMemcpy(ddw, dw.data(), dw.size() * 4, hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreate(&s);
    unsigned char* pack; hipMalloc(&pack, bc_front_b_packed_bytes());
    const float fws = 32768.0f, fin = 4.0f;
    launch_bc_front_b_pack_f16(dw1, pack, fws, s);
    Conv1DwArgs a{dx, dw1, nullptr, dal, dbe, ddw, d1, x1, B, H, W, ACT_RELU, 2, 2};
    a.wpack = pack; a.f16_in = fin; a.f16_clamp = 8192.0f; a.f16_unscale = 1.0f / (fin * fws); a.bn_pos = 1;
    for (int i = 0; i < 3; ++i) launch_bc_front_b(a, 3, 256, s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    launch_bc_front_b(a, 3, 256, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("bc_front_b B=%d: %.4f ms per launch (%s)\n", B, ms, hipGetErrorString(hipGetLastError()));
    std::vector<unsigned long long> tr(8 * 16 * 8);
    hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_front_trace), tr.size() * 8);
    printf("clocks (s_memtime, 100 MHz ticks x the shader / reference ratio - compare rows, not absolute) since the item's top, items 4..11 averaged:\n");
    printf("          rows requested | tasks done | barrier 1 | rows split | depthwise done | barrier 2 | next item's top\n");
    for (int wv = 0; wv < 8; ++wv) {
        double acc[7] = {0, 0, 0, 0, 0, 0, 0}; int n = 0;
        for (int it = 4; it < 12; ++it, ++n) {
            const unsigned long long* r = &tr[(wv * 16 + it) * 8];
            for (int k = 1; k < 7; ++k) acc[k - 1] += (double)(r[k] - r[0]);
            acc[6] += (double)(tr[(wv * 16 + it + 1) * 8] - r[0]);
        }
        printf("  wave %d:", wv);
        for (int k = 0; k < 7; ++k) printf(" %9.1f", acc[k] / n);
        printf("\n");
    }
    printf("per item (wave 0): top-to-top, tasks, barrier-1 wait, depthwise, barrier-2 wait\n");
    for (int it = 0; it < 15; ++it) {
        const unsigned long long* r = &tr[(0 * 16 + it) * 8];
        printf("  item %2d: %7llu %7llu %7llu %7llu %7llu\n", it, tr[(it + 1) * 8] - r[0], r[2] - r[1], r[3] - r[2], r[5] - r[4], r[6] - r[5]);
    }
    return 0;
}
