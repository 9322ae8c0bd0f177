// Phase timeline of the bulk split-operand GEMM at fc1's shape (M = 4096, N = 128, K = 12800, split-K 16): s_memtime stamps of
// workgroup 0's waves at (0) arrival at the first barrier of a k-tile, (1) its release, (2) after split + store + second barrier,
// (3) after the tile's MFMAs; plus the launch time.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DX3_TRACE -I nanowakeword_amd/csrc -I include tools/ubench/gemm_x3_trace.hip -o tools/ubench/gemm_x3_trace
#include "../../nanowakeword_amd/csrc/gemm_x3.hip"
#include <stdio.h>
#include <vector>
int main() {
    const int M = 4096, N = 128, K = 12800, SK = 16;
    std::vector<float> A((size_t)M * K), W((size_t)N * K);
    uint32_t st = 1;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : A) v = rnd();
    for (auto& v : W) v = 0.1f * rnd();
    float *dA, *dW, *dC, *dws; void* dx3;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dC, (size_t)M * N * 4); hipMalloc(&dws, (size_t)SK * M * N * 4);
    hipMalloc(&dx3, gemm_x3_weight_bytes(N, K));
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreate(&s);
    launch_split_weights_x3(dW, dx3, N, K, s);
    GemmArgs g;
    g.A = dA; g.lda = K; g.W = dW; g.C = dC; g.ldc = N; g.M = M; g.N = N; g.K = K; g.bias = nullptr; g.alpha = nullptr; g.beta = nullptr; g.act = ACT_NONE;
    g.res = nullptr; g.ldres = 0; g.rscale = 1.f; g.Wx3 = dx3; g.splitk = SK; g.splitk_ws = dws;
    for (int i = 0; i < 10; ++i) launch_gemm_x3(g, s);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < 50; ++i) launch_gemm_x3(g, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("gemm_x3 bulk (row-major A): %.4f ms per launch: %s\n", ms / 50, hipGetErrorString(hipGetLastError()));
    unsigned long long tr[4 * 16 * 4];
    hipMemcpyFromSymbol(tr, HIP_SYMBOL(x3_trace_buf), sizeof(tr));
    for (int w = 0; w < 4; ++w) {
        printf("wave %d (clocks since its first stamp): tile: arrive  release  staged  multiplied\n", w);
        const unsigned long long t0 = tr[(w * 16) * 4];
        for (int k = 0; k < 12; ++k)
            printf("   tile %2d: %7llu %7llu %7llu %7llu\n", k, tr[(w * 16 + k) * 4] - t0, tr[(w * 16 + k) * 4 + 1] - t0, tr[(w * 16 + k) * 4 + 2] - t0, tr[(w * 16 + k) * 4 + 3] - t0);
    }
    return 0;
}
