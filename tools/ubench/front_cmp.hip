// BcResNet front kernel: the split-operand bf16 version (trunk_b.hip: bc_front_b_kernel) against the float32-MFMA one
// (trunk.hip: conv1_pool_dw_nhwc_kernel) on the same random input - element-wise comparison of d and xs, and timing.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I nanowakeword_amd/csrc -I include tools/ubench/front_cmp.hip -o tools/ubench/front_cmp
#include "../../nanowakeword_amd/csrc/trunk_b.hip"
#include "../../nanowakeword_amd/csrc/trunk.hip"
#include <stdio.h>
#include <math.h>
#include <vector>

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, H = argc > 2 ? atoi(argv[2]) : 101, W = argc > 3 ? atoi(argv[3]) : 64;
    const int Ho = (H / 2 - 1) / 2 + 1, Wo = (W / 2 - 1) / 2 + 1;
    std::vector<float> x((size_t)B * H * W), w1(32 * 9), al(32), be(32), dw(9 * 32);
    uint32_t st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : x) v = 40.0f * rnd();
    for (auto& v : w1) v = rnd();
    const bool neg = getenv("NEG_ALPHA") != nullptr; for (auto& v : al) v = neg ? 2.0f * rnd() : 1.0f + rnd(); for (auto& v : be) v = rnd(); for (auto& v : dw) v = rnd();
    float *dx, *dw1, *dal, *dbe, *ddw, *d0, *x0, *d1, *x1;
    const size_t no = (size_t)B * Ho * Wo * 32;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dw1, w1.size() * 4); hipMalloc(&dal, 128); hipMalloc(&dbe, 128); hipMalloc(&ddw, dw.size() * 4);
    hipMalloc(&d0, no * 4); hipMalloc(&x0, no * 4); hipMalloc(&d1, no * 4); hipMalloc(&x1, no * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dal, al.data(), 128, hipMemcpyHostToDevice); hipMemcpy(dbe, be.data(), 128, hipMemcpyHostToDevice);
    hipMemcpy(ddw, dw.data(), dw.size() * 4, hipMemcpyHostToDevice);
    hipMemset(d1, 0xff, no * 4); hipMemset(x1, 0xff, no * 4);
    hipStream_t s; hipStreamCreate(&s);
    unsigned char* pack; hipMalloc(&pack, bc_front_b_packed_bytes());
    launch_bc_front_b_pack(dw1, pack, s);
    Conv1DwArgs a{dx, dw1, nullptr, dal, dbe, ddw, d0, x0, B, H, W, ACT_RELU, 2, 2};
    hipError_t e = launch_conv1_pool_dw_nhwc(a, 256, s);
    printf("f32 launch: %s\n", hipGetErrorString(e));
    Conv1DwArgs b = a; b.d_out = d1; b.xs_out = x1; b.wpack = pack;
    e = launch_bc_front_b(b, 6, 256, s);
    printf("x3 launch: %s, rows %d\n", hipGetErrorString(e), bc_front_b_rows(H, W, 2));
    e = hipStreamSynchronize(s);
    printf("sync: %s\n", hipGetErrorString(e));
    std::vector<float> hd0(no), hx0(no), hd1(no), hx1(no);
    hipMemcpy(hd0.data(), d0, no * 4, hipMemcpyDeviceToHost); hipMemcpy(hx0.data(), x0, no * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hd1.data(), d1, no * 4, hipMemcpyDeviceToHost); hipMemcpy(hx1.data(), x1, no * 4, hipMemcpyDeviceToHost);
    for (int px = 0; px < 2; ++px) {
        printf("pixel %d f32:", px); for (int c = 0; c < 32; ++c) printf(" %7.3f", hx0[px * 32 + c]); printf("\n");
        printf("pixel %d x3 :", px); for (int c = 0; c < 32; ++c) printf(" %7.3f", hx1[px * 32 + c]); printf("\n");
    }
    int bad = 0; double worst = 0;
    for (size_t k = 0; k < no; ++k) {
        const double dd = fabs((double)hx0[k] - hx1[k]), sc = fmax(1.0, fabs((double)hx0[k]));
        if (!(dd <= 1e-3 * sc)) {
            if (bad < 4) {
                const int c = k % 32; size_t p = k / 32; const int ox = p % Wo; p /= Wo; const int oy = p % Ho; const int bb = (int)(p / Ho);
                printf("xs mismatch b %d oy %d ox %d c %d: f32 %.6g x3 %.6g\n", bb, oy, ox, c, hx0[k], hx1[k]);
            }
            ++bad;
        }
        if (dd / sc > worst) worst = dd / sc;
    }
    printf("xs: %d of %zu mismatched, worst rel %.3g\n", bad, no, worst);
    bad = 0; worst = 0;
    for (size_t k = 0; k < no; ++k) {
        const double dd = fabs((double)hd0[k] - hd1[k]), sc = fmax(1.0, fabs((double)hd0[k]));
        if (!(dd <= 1e-3 * sc)) ++bad;
        if (dd / sc > worst) worst = dd / sc;
    }
    printf("d : %d of %zu mismatched, worst rel %.3g\n", bad, no, worst);
    if (B >= 1024) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int v = 0; v < 2; ++v) {
            for (int i = 0; i < 5; ++i) { if (v) launch_bc_front_b(b, 6, 256, s); else launch_conv1_pool_dw_nhwc(a, 256, s); }
            hipEventRecord(e0, s);
            for (int i = 0; i < 20; ++i) { if (v) launch_bc_front_b(b, 6, 256, s); else launch_conv1_pool_dw_nhwc(a, 256, s); }
            hipEventRecord(e1, s); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.4f ms per launch (B = %d)\n", v ? "x3 " : "f32", ms / 20, B);
        }
    }
    return 0;
}
