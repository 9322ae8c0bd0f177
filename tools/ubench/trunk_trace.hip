// Phase timeline of the fused trunk (trunk_b.hip compiled with -DNWW_TRACE): s_memtime stamps per wave at the phase
// boundaries of the first items of the first workgroups, plus plain launch timing of the shapes.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNWW_TRACE -I nanowakeword_amd/csrc -I include tools/ubench/trunk_trace.hip -o tools/ubench/trunk_trace
// run:   tools/ubench/trunk_trace [B=4096] [grid=256] [products=6]      (NWW_TRUNK_STRIPS as in the library)
#include "../../nanowakeword_amd/csrc/trunk_b.hip"
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, H = 101, W = 64, G = argc > 2 ? atoi(argv[2]) : 256;
    const int NWv = 8;
    const int P = argc > 3 ? atoi(argv[3]) : 6;                  // products: 6 / 9 (bf16 terms) or 3 (two binary16 terms)
    std::vector<float> x((size_t)B * H * W), w1(16 * 9), b1(16), w2(32 * 16 * 9), b2(32);
    uint32_t st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : x) v = 40.0f * rnd();
    for (auto& v : w1) v = rnd(); for (auto& v : b1) v = 0.1f * rnd();
    for (auto& v : w2) v = 0.2f * rnd(); for (auto& v : b2) v = 0.1f * rnd();
    float *dx, *dw1, *db1, *dw2, *db2, *dout;
    unsigned long long* dtr;
    const size_t ntr = 16 * 6 * 8 * 8 + 64;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dw1, w1.size() * 4); hipMalloc(&db1, 64); hipMalloc(&dw2, w2.size() * 4); hipMalloc(&db2, 128);
    hipMalloc(&dout, (size_t)B * 32 * 25 * 16 * 4); hipMalloc(&dtr, ntr * 8);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db1, b1.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dw2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db2, b2.data(), 128, hipMemcpyHostToDevice);
    hipMemset(dtr, 0, ntr * 8);
    TrunkArgs a{dx, dw1, db1, nullptr, nullptr, dw2, db2, nullptr, nullptr, dout, B, H, W, ACT_RELU};
    hipStream_t s; hipStreamCreate(&s);
    unsigned char* dpack; hipMalloc(&dpack, trunk_b_packed_bytes());
    if (P == 3) {
        launch_trunk_b_pack_f16(dw1, dw2, dpack, 32768.0f, 65536.0f, s);
        a.f16_in = 64.0f; a.f16_k1 = 64.0f * 32768.0f; a.f16_s1 = 256.0f; a.f16_k2 = 256.0f * 65536.0f; a.f16_so = 1.0f;
    } else {
        launch_trunk_b_pack(dw1, dw2, dpack, s);
    }
    a.wpack = dpack;
    for (int i = 0; i < 20; ++i) launch_cnn_trunk_b(a, P, G, s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < 50; ++i) launch_cnn_trunk_b(a, P, G, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("trunk_b B=%d grid<=%d: %.4f ms per launch\n", B, G, ms / 50);
    a.trace = dtr;
    launch_cnn_trunk_b(a, P, G, s);
    hipStreamSynchronize(s);
    std::vector<unsigned long long> tr(ntr);
    hipMemcpy(tr.data(), dtr, ntr * 8, hipMemcpyDeviceToHost);
    for (int blk = 0; blk < 16; blk += 5) {
        const unsigned long long* g = &tr[16 * 6 * 8 * 8 + blk * 4];
        const unsigned long long* q = &tr[(((size_t)blk * 6 + 1) * NWv + 0) * 8];
        const unsigned long long* q2 = &tr[(((size_t)blk * 6 + 3) * NWv + 0) * 8];
        printf("block slot %d: entry -> first item %.2f us, item loop %.2f us; items 1-2: %llu clocks in %.2f us -> %.0f MHz\n", blk, (g[1] - g[0]) / 100.0,
               (g[2] - g[1]) / 100.0, q2[0] - q[0], (q2[6] - q[6]) / 100.0, 100.0 * (q2[0] - q[0]) / (double)(q2[6] - q[6]));
    }
    for (int blk : {0, 15}) {      // slot 0 = first workgroup (strip 0), slot 15 = last workgroup (last strip)
        for (int it = 1; it < 3; ++it) {
            const unsigned long long t0 = tr[(((size_t)blk * 6 + it) * NWv + 0) * 8 + 0];
            printf("block %d item %d (clocks from wave 0's item start: start, conv1 of the next item done (A waves), conv2 tile 1 / 2 / 3 / 4 done, -, after the barrier)\n", blk, it);
            for (int wv = 0; wv < NWv; ++wv) {
                printf("  wave %d:", wv);
                for (int k = 0; k < 8; ++k) {
                    if (k == 6) continue;
                    const unsigned long long v = tr[(((size_t)blk * 6 + it) * NWv + wv) * 8 + k];
                    if (v) printf(" %7lld", (long long)(v - t0)); else printf("       -");
                }
                printf("\n");
            }
        }
    }
    return 0;
}
