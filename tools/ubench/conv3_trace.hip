// Item timeline of conv3_x3 (compiled with -DNWW_TRACE): s_memtime of workgroup 0's eight waves per item - item top | rows staged (split + LDS
// stores) | behind the barrier | next item's loads requested | tiles done (MFMA loop + epilogue stores) | behind the closing barrier.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNWW_TRACE -I nanowakeword_amd/csrc -I include tools/ubench/conv3_trace.hip -o tools/ubench/conv3_trace
// run:   tools/ubench/conv3_trace [B=2048] [H=25] [W=16] [Cout=32] [pool=1] [avg_ow=0] [f16=1]
#include "../../nanowakeword_amd/csrc/conv3_x3.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 2048, H = argc > 2 ? atoi(argv[2]) : 25, W = argc > 3 ? atoi(argv[3]) : 16, Cout = argc > 4 ? atoi(argv[4]) : 32;
    const int pool = argc > 5 ? atoi(argv[5]) : 1, avg = argc > 6 ? atoi(argv[6]) : 0, f16 = argc > 7 ? atoi(argv[7]) : 1;
    std::vector<float> x((size_t)B * 32 * H * W), w((size_t)Cout * 32 * 9), bias(Cout), al(Cout, 1.0f), be(Cout, 0.0f);
    uint32_t st = 1;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : x) v = rnd() * 4.0f;
    for (auto& v : w) v = rnd() * 0.2f;
    for (auto& v : bias) v = rnd() * 0.1f;
    float *dx, *dw, *db, *dal, *dbe, *dout;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dw, w.size() * 4); hipMalloc(&db, Cout * 4); hipMalloc(&dal, Cout * 4); hipMalloc(&dbe, Cout * 4);
    hipMalloc(&dout, (size_t)B * Cout * H * W * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, bias.data(), Cout * 4, hipMemcpyHostToDevice); hipMemcpy(dal, al.data(), Cout * 4, hipMemcpyHostToDevice); hipMemcpy(dbe, be.data(), Cout * 4, hipMemcpyHostToDevice);
    ConvMfmaArgs a{dx, dw, db, dal, dbe, dout, B, H, W, Cout, ACT_RELU, pool};
    if (avg) { a.avg_kw = 7; a.avg_sw = 6; a.avg_ow = 4; a.avg_y = 1; }
    if (pool && !avg) a.seq_out = 1;
    if (f16) { a.h2_in = 1024.0f; a.h2_w = 32768.0f; }
    hipStream_t s; hipStreamCreate(&s);
    const size_t lds = conv3_x3_lds_bytes(H, W, a.avg_ow);
    const int grid = 256 * (lds * 2 <= 160 * 1024 ? 2 : 1);
    for (int i = 0; i < 5; ++i) launch_conv3_x3(a, grid, s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < 20; ++i) launch_conv3_x3(a, grid, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int items = B * (Cout / 32);
    printf("conv3_x3 B=%d %dx%d Cout=%d pool=%d avg=%d f16=%d grid=%d: %.4f ms per launch, %.0f ns per item and workgroup (%s)\n", B, H, W, Cout, pool, avg, f16, grid, ms / 20,
           ms / 20 * 1e6 / ((items + grid - 1) / grid), hipGetErrorString(hipGetLastError()));
    std::vector<unsigned long long> tr(8 * 16 * 8);
    hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_c3_trace), tr.size() * 8);
    printf("clocks since the item's top (items 1..: averaged): staged | behind barrier | next loads requested | tiles done | behind barrier | next item's top\n");
    const int nit = (items + grid - 1) / grid < 16 ? (items + grid - 1) / grid : 16;
    for (int wv = 0; wv < 8; ++wv) {
        double acc[6] = {0, 0, 0, 0, 0, 0}; int n = 0;
        for (int it = 1; it + 1 < nit; ++it, ++n) {
            const unsigned long long* r = &tr[(wv * 16 + it) * 8];
            for (int k = 1; k < 6; ++k) acc[k - 1] += (double)(r[k] - r[0]);
            acc[5] += (double)(tr[(wv * 16 + it + 1) * 8] - r[0]);
        }
        if (n == 0) break;
        printf("  wave %d:", wv);
        for (int k = 0; k < 6; ++k) printf(" %7.1f", acc[k] / n);
        printf("\n");
    }
    return 0;
}
