// Clip timeline of the fused convolution module (convmod_x3.hip compiled with -DNWW_TRACE): s_memtime of workgroup 0's waves over its second clip -
// LayerNorm + split | 5 x conv1 block | windows loaded | depthwise done | rows re-split | 5 x conv2 block.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DNWW_TRACE -I tools/ubench -I nanowakeword_amd/csrc -I include tools/ubench/convmod_trace.hip nanowakeword_amd/csrc/lin_x3.hip -o tools/ubench/convmod_trace
// run:   tools/ubench/convmod_trace [B=2048] [T=101]
#include "convmod_x3.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 2048, T = argc > 2 ? atoi(argv[2]) : 101, D = 144;
    std::vector<float> h((size_t)B * T * D), w1((size_t)2 * D * D), b1(2 * D), w2((size_t)D * D), b2(D), dw((size_t)D * 31), pc(4 * D), ln(2 * D);
    uint32_t st = 1;
    auto rnd = [&](float a) { st = st * 1664525u + 1013904223u; return (((st >> 8) & 0xffff) / 65536.0f - 0.5f) * a; };
    for (auto& v : h) v = rnd(4.0f);
    for (auto& v : w1) v = rnd(0.2f);
    for (auto& v : w2) v = rnd(0.2f);
    for (auto& v : b1) v = rnd(0.1f);
    for (auto& v : b2) v = rnd(0.1f);
    for (auto& v : dw) v = rnd(0.3f);
    for (auto& v : pc) v = 1.0f + rnd(0.1f);
    for (auto& v : ln) v = 1.0f + rnd(0.1f);
    auto up = [&](const std::vector<float>& v) { float* d; hipMalloc(&d, v.size() * 4); hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice); return d; };
    float *dh = up(h), *dw1 = up(w1), *db1 = up(b1), *dw2 = up(w2), *db2 = up(b2), *ddw = up(dw), *dpc = up(pc), *dln = up(ln), *dtaps;
    void *pk1a, *pk1b, *pk2;
    hipMalloc(&pk1a, lin_x3_packed_bytes(D, D, 1, 2)); hipMalloc(&pk1b, lin_x3_packed_bytes(D, D, 1, 2)); hipMalloc(&pk2, lin_x3_packed_bytes(D, D, 1, 2)); hipMalloc(&dtaps, 31 * D * 4);
    hipStream_t s; hipStreamCreate(&s);
    const float ws = 65536.0f;
    launch_lin_x3_pack(dw1, db1, pk1a, D, D, 1, D, s, 2, ws);
    launch_lin_x3_pack(dw1 + D * D, db1 + D, pk1b, D, D, 1, D, s, 2, ws);
    launch_lin_x3_pack(dw2, db2, pk2, D, D, 1, D, s, 2, ws);
    launch_convmod_taps(ddw, dtaps, D, 31, s);
    ConvModArgs a{dh, dh, dln, dln + D, (const unsigned char*)pk1a, (const unsigned char*)pk1b, (const unsigned char*)pk2, dtaps, dpc, dpc + D, dpc + 2 * D, B, T, 1.0f / ws, 1.0f / ws};
    for (int i = 0; i < 3; ++i) launch_convmod_x3(a, D, s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < 10; ++i) launch_convmod_x3(a, D, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("convmod_x3 B=%d T=%d: %.4f ms per launch (%s)\n", B, T, ms / 10, hipGetErrorString(hipGetLastError()));
    std::vector<unsigned long long> tr(4 * 32);
    hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_cm_trace), tr.size() * 8);
    printf("clocks of each phase (second clip of workgroup 0): ln+split | 5 x conv1 block | windows | depthwise | re-split | 5 x conv2 block | total\n");
    for (int wv = 0; wv < 4; ++wv) {
        const unsigned long long* r = &tr[wv * 32];
        printf("  wave %d:", wv);
        for (int k = 1; k <= 14; ++k) printf(" %6llu", r[k] - r[k - 1]);
        printf(" | %7llu\n", r[14] - r[0]);
    }
    return 0;
}
