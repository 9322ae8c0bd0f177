// Step timeline of the register-resident recurrence (rnn_x3.hip compiled with -DNWW_TRACE): s_memtime stamps of workgroup 0's waves at
// the phase boundaries of every step - step top | (input product issued) h planes addressed | recurrent product issued | gates done, h written |
// before the barrier | behind it - plus plain launch timing.  GRU, H = 128, fused input projection (the GRU head's instance) or precomputed xg.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNWW_TRACE -I nanowakeword_amd/csrc -I include tools/ubench/rnn_trace.hip -o tools/ubench/rnn_trace
// run:   tools/ubench/rnn_trace [B=2048] [T=101] [fin=64|0] [H=128]
#include "../../nanowakeword_amd/csrc/rnn_x3.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 2048, T = argc > 2 ? atoi(argv[2]) : 101, fin = argc > 3 ? atoi(argv[3]) : 64, H = argc > 4 ? atoi(argv[4]) : 128;
    const int G = 3;
    std::vector<float> whh((size_t)G * H * H), wih((size_t)G * H * 64), b(G * H), x((size_t)B * T * 64), xg((size_t)B * (T + 1) * G * H);
    uint32_t st = 1;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : whh) v = rnd() * 0.2f;
    for (auto& v : wih) v = rnd() * 0.2f;
    for (auto& v : b) v = rnd() * 0.1f;
    for (auto& v : x) v = rnd() * 40.0f;
    for (auto& v : xg) v = rnd();
    float *dwhh, *dwih, *db, *dx, *dxg, *dlast;
    hipMalloc(&dwhh, whh.size() * 4); hipMalloc(&dwih, wih.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dx, x.size() * 4); hipMalloc(&dxg, xg.size() * 4);
    hipMalloc(&dlast, (size_t)B * 2 * H * 4);
    hipMemcpy(dwhh, whh.data(), whh.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dwih, wih.data(), wih.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dxg, xg.data(), xg.size() * 4, hipMemcpyHostToDevice);
    GruArgs a;
    a.xg = dxg; a.w_hh = dwhh; a.b_hh = db; a.seq_out = nullptr; a.ld_seq = 2 * H; a.last_out = dlast; a.ld_last = 2 * H; a.col_off = 0;
    a.B = B; a.T = T; a.H = H; a.reverse = 0; a.steps = T; a.products = 3; a.w_scale = 32768.0f;
    if (fin) { a.x_in = dx; a.w_ih = dwih; a.b_ih = db; a.fin = fin; a.x_scale = 4.0f; a.x_clamp = 8192.0f; a.wi_scale = 32768.0f; }
    hipStream_t s; hipStreamCreate(&s);
    for (int i = 0; i < 5; ++i) launch_rnn_x3(a, 3, s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < 20; ++i) launch_rnn_x3(a, 3, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("gru H=%d B=%d T=%d fin=%d: %.4f ms per launch = %.0f ns per step (%s)\n", H, B, T, fin, ms / 20, ms / 20 / T * 1e6, hipGetErrorString(hipGetLastError()));
    std::vector<unsigned long long> tr(8 * 128 * 8);
    hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_rnn_trace), tr.size() * 8);
    const int nw = H == 128 ? 8 : H / 16;
    printf("s_memtime ticks (100 MHz: 10 ns) since the step's top, averaged over steps 20..%d: hplanes | products issued | gates + h written | before barrier | behind barrier | next top\n", T - 2);
    for (int wv = 0; wv < nw; ++wv) {
        double acc[6] = {0, 0, 0, 0, 0, 0};
        int n = 0;
        for (int stp = 20; stp + 1 < T && stp + 1 < 128; ++stp, ++n) {
            const unsigned long long* r = &tr[(wv * 128 + stp) * 8];
            for (int k = 1; k < 6; ++k) acc[k - 1] += (double)(r[k] - r[0]);
            acc[5] += (double)(tr[(wv * 128 + stp + 1) * 8] - r[0]);
        }
        printf("  wave %d:", wv);
        for (int k = 0; k < 6; ++k) printf(" %7.1f", acc[k] / n);
        printf("\n");
    }
    return 0;
}
