// Output-block timeline of the input-stationary Linear (lin_x3.hip compiled with -DNWW_TRACE): s_memtime of workgroup 0's waves per 32-output block -
// block top | next block's fetch issued | products issued | loads landed (vmcnt 0) | epilogue + stores issued | behind the barrier.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNWW_TRACE -I nanowakeword_amd/csrc -I include tools/ubench/lin_trace.hip -o tools/ubench/lin_trace
// run:   tools/ubench/lin_trace [M=206848] [epi=0|1|2] (K = 144; N = 432 head-major qkv | 144 + residual | 144 GLU with LayerNorm)
#include "../../nanowakeword_amd/csrc/lin_x3.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 206848, epi = argc > 2 ? atoi(argv[2]) : 0, K = 144, T = 101;
    const int N = epi == 0 ? 432 : 144, parts = epi == 2 ? 2 : 1;
    std::vector<float> x((size_t)M * K), w((size_t)N * parts * K), b(N * parts), lw(K, 1.0f), lb(K, 0.0f);
    uint32_t st = 1;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : x) v = rnd() * 4.0f;
    for (auto& v : w) v = rnd() * 0.2f;
    for (auto& v : b) v = rnd() * 0.1f;
    float *dx, *dw, *db, *dout, *dres, *dlw, *dlb; void* pk;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dw, w.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dout, (size_t)M * N * 4); hipMalloc(&dres, (size_t)M * N * 4);
    hipMalloc(&dlw, K * 4); hipMalloc(&dlb, K * 4); hipMalloc(&pk, lin_x3_packed_bytes(K, N, parts, 2));
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dlw, lw.data(), K * 4, hipMemcpyHostToDevice); hipMemcpy(dlb, lb.data(), K * 4, hipMemcpyHostToDevice);
    hipMemset(dres, 0, (size_t)M * N * 4);
    hipStream_t s; hipStreamCreate(&s);
    launch_lin_x3_pack(dw, db, pk, K, N, parts, N, s, 2, 32768.0f);
    LinArgs a{dx, K, dout, N, epi == 1 ? dres : nullptr, N, 1.0f, epi == 2 ? dlw : nullptr, epi == 2 ? dlb : nullptr, static_cast<const unsigned char*>(pk), M, N};
    a.h2 = 1; a.w_un = 1.0f / 32768.0f;
    if (epi == 0) { a.qkv_T = T; a.qkv_dh = 36; }
    for (int i = 0; i < 3; ++i) launch_lin_x3(a, K, epi, epi == 2, s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < 10; ++i) launch_lin_x3(a, K, epi, epi == 2, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("lin_x3 K=%d N=%d epi=%d M=%d: %.4f ms per launch (%s)\n", K, N, epi, M, ms / 10, hipGetErrorString(hipGetLastError()));
    std::vector<unsigned long long> tr(8 * 16 * 8);
    hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_lin_trace), tr.size() * 8);
    const int nblk = (N + 31) / 32 < 16 ? (N + 31) / 32 : 16;
    printf("clocks since the block's top (blocks 1..%d averaged): block entered (fetch issued) | products issued | vmcnt(0) | stores issued | behind barrier | next top\n", nblk - 2);
    for (int wv = 0; wv < 8; ++wv) {
        if (tr[(wv * 16 + 1) * 8] == 0) break;
        double acc[6] = {0, 0, 0, 0, 0, 0}; int n = 0;
        for (int blk = 1; blk + 1 < nblk; ++blk, ++n) {
            const unsigned long long* r = &tr[(wv * 16 + blk) * 8];
            for (int k = 1; k < 6; ++k) acc[k - 1] += (double)(r[k] - r[0]);
            acc[5] += (double)(tr[(wv * 16 + blk + 1) * 8] - r[0]);
        }
        printf("  wave %d:", wv);
        for (int k = 0; k < 6; ++k) printf(" %7.1f", acc[k] / n);
        printf("\n");
    }
    return 0;
}
