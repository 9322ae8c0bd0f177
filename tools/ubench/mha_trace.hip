// Unit timeline of the two-term attention core (mha_h2.hip compiled with -DNWW_TRACE): s_memtime of workgroup 0's waves per (clip, head) unit -
// unit top | maxima exchanged (two barriers) | K, V scaled, split, stored | behind the barrier | scores done | softmax + P V done | next rows' maxima taken | next top.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNWW_TRACE -I nanowakeword_amd/csrc -I include tools/ubench/mha_trace.hip -o tools/ubench/mha_trace
// run:   tools/ubench/mha_trace [B=2048] [T=101] [D=144] [heads=4]
#include "../../nanowakeword_amd/csrc/mha_h2.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 2048, T = argc > 2 ? atoi(argv[2]) : 101, D = argc > 3 ? atoi(argv[3]) : 144, NH = argc > 4 ? atoi(argv[4]) : 4;
    std::vector<float> qkv((size_t)3 * B * T * D);
    uint32_t st = 1;
    for (auto& v : qkv) { st = st * 1664525u + 1013904223u; v = (((st >> 8) & 0xffff) / 65536.0f - 0.5f) * 4.0f; }
    float *dq, *dout;
    hipMalloc(&dq, qkv.size() * 4); hipMalloc(&dout, (size_t)B * T * D * 4);
    hipMemcpy(dq, qkv.data(), qkv.size() * 4, hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreate(&s);
    for (int i = 0; i < 3; ++i) launch_mha_h2(dq, dout, B, T, D, NH, s, 1);
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int i = 0; i < 10; ++i) launch_mha_h2(dq, dout, B, T, D, NH, s, 1);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mha_h2 B=%d T=%d D=%d heads=%d: %.4f ms per launch (%s)\n", B, T, D, NH, ms / 10, hipGetErrorString(hipGetLastError()));
    std::vector<unsigned long long> tr(4 * 16 * 8);
    hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_mha_trace), tr.size() * 8);
    printf("clocks since the unit's top (units 1..13 averaged): maxima exchanged | K, V stored | behind barrier | scores done | softmax + P V done | maxima of the next rows | next top\n");
    for (int wv = 0; wv < 4; ++wv) {
        double acc[7] = {0, 0, 0, 0, 0, 0, 0}; int n = 0;
        for (int u = 1; u + 1 < 15; ++u, ++n) {
            const unsigned long long* r = &tr[(wv * 16 + u) * 8];
            for (int k = 1; k < 7; ++k) acc[k - 1] += (double)(r[k] - r[0]);
            acc[6] += (double)(tr[(wv * 16 + u + 1) * 8] - r[0]);
        }
        printf("  wave %d:", wv);
        for (int k = 0; k < 7; ++k) printf(" %7.1f", acc[k] / n);
        printf("\n");
    }
    return 0;
}
