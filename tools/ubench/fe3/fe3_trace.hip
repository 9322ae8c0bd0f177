// Phase timeline of the matrix-pipe frontend (frontend3.hip compiled with -DNWW_TRACE): s_memtime stamps per wave at the phase
// boundaries of the first six items of the first eight workgroups, plus plain launch timing.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -DNWW_TRACE -I tools/ubench/fe3 -I nanowakeword_amd/csrc -I include tools/ubench/fe3/fe3_trace.hip tools/ubench/fe3/fe3_tables.cpp nanowakeword_amd/csrc/fe_tables.cpp -o tools/ubench/fe3/fe3_trace
// run:   tools/ubench/fe3/fe3_trace [B=4096] [grid=512]
#include "frontend3.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096, G = argc > 2 ? atoi(argv[2]) : 768, N = 16000;
    FeParams p;
    std::vector<float> w, fb;
    fe_default_window(p.win_length, w);
    fe_default_melfb(p, fb);
    FeTables tb;
    std::vector<Fe3Plan> pl(1);
    if (!fe_build_tables(p, w.data(), fb.data(), &tb).empty() || !fe3_build_plan(p, w.data(), pl.data()).empty()) return 1;
    int max_taps = 0;
    for (int j = 0; j < p.n_mels; ++j) max_taps = tb.mel_cnt[j] > max_taps ? tb.mel_cnt[j] : max_taps;
    const int T = fe_num_frames(p, N);
    std::vector<int16_t> x((size_t)B * N);
    uint32_t st = 12345;
    for (auto& v : x) { st = st * 1664525u + 1013904223u; v = (int16_t)((st >> 10) % 16384) - 8192; }
    int16_t* dx; FeTables* dtb; Fe3Plan* dpl; float* dout; unsigned long long* dtr;
    const size_t ntr = 8 * 6 * 16 * 8;
    hipMalloc(&dx, x.size() * 2); hipMalloc(&dtb, sizeof(tb)); hipMalloc(&dpl, sizeof(Fe3Plan)); hipMalloc(&dout, (size_t)B * T * 64 * 4); hipMalloc(&dtr, ntr * 8);
    hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dtb, &tb, sizeof(tb), hipMemcpyHostToDevice);
    hipMemcpy(dpl, pl.data(), sizeof(Fe3Plan), hipMemcpyHostToDevice); hipMemset(dtr, 0, ntr * 8);
    hipStream_t s; hipStreamCreate(&s);
    for (int i = 0; i < 10; ++i) fe3_launch(dx, N, B, N, T, p, dtb, dpl, dout, nullptr, 1, max_taps, G, s, nullptr);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < 50; ++i) fe3_launch(dx, N, B, N, T, p, dtb, dpl, dout, nullptr, 1, max_taps, G, s, nullptr);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("fe3 B=%d grid<=%d max_taps=%d: %.4f ms per launch (%s)\n", B, G, max_taps, ms / 50, hipGetErrorString(hipGetLastError()));
    g_fe3_trace = dtr;
    fe3_launch(dx, N, B, N, T, p, dtb, dpl, dout, nullptr, 1, max_taps, G, s, nullptr);
    hipStreamSynchronize(s);
    std::vector<unsigned long long> tr(ntr);
    hipMemcpy(tr.data(), dtr, ntr * 8, hipMemcpyDeviceToHost);
    printf("clocks from wave 0's item start: item start | staged | behind barrier 1 | stage 1 done | behind barrier 2 | stage 2 done | behind barrier 3 | mel + copy-out done\n");
    for (int blk : {0, 5}) {
        for (int it = 2; it < 4; ++it) {
            const unsigned long long t0 = tr[((blk * 6 + it) * 16 + 0) * 8];
            printf("block %d item %d\n", blk, it);
            for (int wv = 0; wv < FE3_NW; wv += FE3_NW / 4) {
                printf("  wave %d:", wv);
                for (int k = 0; k < 8; ++k) printf(" %7lld", (long long)(tr[((blk * 6 + it) * 16 + wv) * 8 + k] - t0));
                printf("\n");
            }
        }
    }
    return 0;
}
