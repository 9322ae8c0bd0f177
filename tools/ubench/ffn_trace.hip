// Hidden-block timeline of the fused Conformer feed-forward (ffn_x3.hip compiled with -DNWW_TRACE): s_memtime stamps of workgroup 0's four
// waves around every second hidden block of the main loop - block top | fetches issued | first product + epilogue done | second product issued |
// fetches landed (vmcnt 0) | behind the barrier - plus plain launch timing.  D = 144 (the default Conformer width), two-term form.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNWW_TRACE -I nanowakeword_amd/csrc -I include tools/ubench/ffn_trace.hip -o tools/ubench/ffn_trace
// run:   tools/ubench/ffn_trace [M=206848] [mode: 0 plain, 1 + input_proj (K = 64) in front, 2 + conv2 (K = 144) + residual in front, 3 = 2 + LayerNorm + time sums behind]
#include "../../nanowakeword_amd/csrc/ffn_x3.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 206848, D = 144, mode = argc > 2 ? atoi(argv[2]) : 0;
    std::vector<float> h((size_t)M * D), w1((size_t)4 * D * D), w2((size_t)4 * D * D), b1(4 * D), b2(D), lw(D, 1.0f), lb(D, 0.0f);
    uint32_t st = 1;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : h) v = rnd() * 4.0f;
    for (auto& v : w1) v = rnd() * 0.2f;
    for (auto& v : w2) v = rnd() * 0.2f;
    for (auto& v : b1) v = rnd() * 0.1f;
    for (auto& v : b2) v = rnd() * 0.1f;
    float *dh, *dw1, *dw2, *db1, *db2, *dlw, *dlb; void* pk;
    hipMalloc(&dh, h.size() * 4); hipMalloc(&dw1, w1.size() * 4); hipMalloc(&dw2, w2.size() * 4); hipMalloc(&db1, b1.size() * 4); hipMalloc(&db2, b2.size() * 4);
    hipMalloc(&dlw, D * 4); hipMalloc(&dlb, D * 4); hipMalloc(&pk, ffn_x3_packed_bytes(D));
    hipMemcpy(dh, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dw2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db1, b1.data(), b1.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db2, b2.data(), b2.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dlw, lw.data(), D * 4, hipMemcpyHostToDevice); hipMemcpy(dlb, lb.data(), D * 4, hipMemcpyHostToDevice);
    hipStream_t s; hipStreamCreate(&s);
    launch_ffn_x3_pack(dw1, db1, dw2, pk, D, s, 32768.0f, 32768.0f, mode ? 1 : 0);
    FfnArgs a{dh, dlw, dlb, static_cast<const unsigned char*>(pk), db2, M, 0.5f};
    a.h2_x = 512.0f; a.h2_w1 = 32768.0f; a.h2_h = 64.0f; a.h2_w2 = 32768.0f;
    if (mode) {
        const int KP = mode == 1 ? 64 : 144;
        std::vector<float> px((size_t)M * KP), pw((size_t)D * KP);
        for (auto& v : px) v = rnd() * 4.0f;
        for (auto& v : pw) v = rnd() * 0.2f;
        float *dpx, *dpw, *dms; void* ppk;
        hipMalloc(&dpx, px.size() * 4); hipMalloc(&dpw, pw.size() * 4); hipMalloc(&ppk, ffn_x3_pro_tile_bytes(KP) * 5); hipMalloc(&dms, ffn_x3_msum_bytes(M, D));
        hipMemcpy(dpx, px.data(), px.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dpw, pw.data(), pw.size() * 4, hipMemcpyHostToDevice);
        launch_ffn_x3_pro_pack(dpw, ppk, D, KP, 32768.0f, s);
        a.px = dpx; a.ppacked = (const unsigned char*)ppk; a.pb = db2; a.pro_k = KP; a.pro_res = mode >= 2; a.p_un = 1.0f / 32768.0f;
        if (mode == 3) { a.ln2_w = dlw; a.ln2_b = dlb; a.msum = dms; a.T = 101; a.m_scale = 68719476736.0f / 64.0f; }
    }
    for (int i = 0; i < 3; ++i) { hipMemcpyAsync(dh, h.data(), h.size() * 4, hipMemcpyHostToDevice, s); launch_ffn_x3(a, D, s); }
    hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemcpy(dh, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEventRecord(e0, s);
    launch_ffn_x3(a, D, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("ffn_x3 D=%d M=%d: %.4f ms per launch (%s)\n", D, M, ms, hipGetErrorString(hipGetLastError()));
#ifndef NWW_TRACE      // built without -DNWW_TRACE: timing only (20 launches back to back)
    {
        hipEventRecord(e0, s);
        for (int i = 0; i < 20; ++i) launch_ffn_x3(a, D, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("ffn_x3 D=%d M=%d mode %d: %.4f ms per launch over 20 launches, no stamps\n", D, M, mode, ms / 20);
        return 0;
    }
#else
    std::vector<unsigned long long> tr(4 * 32 * 8);
    hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_ffn_trace), tr.size() * 8);
    printf("clocks since the block's top (even blocks 2..14 averaged): fetches issued | product 1 + epilogue | product 2 issued | vmcnt(0) | behind barrier | next stamped top (2 blocks later)\n");
    for (int wv = 0; wv < 4; ++wv) {
        double acc[6] = {0, 0, 0, 0, 0, 0}; int n = 0;
        for (int hb = 2; hb + 2 < 16; hb += 2, ++n) {
            const unsigned long long* r = &tr[(wv * 32 + hb) * 8];
            for (int k = 1; k < 6; ++k) acc[k - 1] += (double)(r[k] - r[0]);
            acc[5] += (double)(tr[(wv * 32 + hb + 2) * 8] - r[0]);
        }
        printf("  wave %d:", wv);
        for (int k = 0; k < 6; ++k) printf(" %7.1f", acc[k] / n);
        printf("\n");
    }
    printf("prologue of workgroup 0 (clocks since the kernel's first stamp): rows split, fetches issued | landed + barrier | products done | LayerNorm + fragments | ... | main loop done | update done | sums done\n");
    for (int wv = 0; wv < 4; ++wv) {
        const unsigned long long* p0 = &tr[(wv * 32 + 30) * 8];
        const unsigned long long* p1 = &tr[(wv * 32 + 31) * 8];
        printf("  wave %d: %7llu %7llu %7llu %7llu | %8llu %8llu %8llu", wv, p0[1] - p0[0], p0[2] - p0[0], p0[3] - p0[0], p0[4] - p0[0], p1[0] - p0[0], p1[1] - p0[0], p1[2] > p0[0] ? p1[2] - p0[0] : 0ULL);
        if (p1[2] > p0[0]) printf("   (behind the update: stats %llu | barrier %llu | staged %llu | summed %llu | stored %llu)", p1[3] - p1[1], p1[4] - p1[1], p1[5] - p1[1], p1[6] - p1[1], p1[2] - p1[1]);
        printf("\n");
    }
    return 0;
#endif
}
