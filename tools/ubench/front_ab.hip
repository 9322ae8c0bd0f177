// BcResNet front kernel (trunk_b.hip: bc_front_b_kernel, the two-term binary16 form the default arithmetic runs): launch time at the BASELINE
// config-3 shape and an FNV-1a hash of both outputs - the regression check for changes that must stay bit-identical (build the file before and
// after a change, the hashes of the two binaries must agree for every shape / option).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I nanowakeword_amd/csrc -I include tools/ubench/front_ab.hip -o tools/ubench/front_ab
//        (-DFRONT_AB_SRC='"path/to/another/trunk_b.hip"' builds against another copy of the kernel)
// run:   tools/ubench/front_ab [B=8192] [H=101] [W=64] [act16: 0 float32 out, 1 bf16, 2 binary16] [neg: 1 = some folded-BN factors negative]
#ifndef FRONT_AB_SRC
#define FRONT_AB_SRC "../../nanowakeword_amd/csrc/trunk_b.hip"
#endif
#include FRONT_AB_SRC
#include <stdio.h>
#include <vector>

static unsigned long long fnv(const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    unsigned long long h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ULL; }
    return h;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8192, H = argc > 2 ? atoi(argv[2]) : 101, W = argc > 3 ? atoi(argv[3]) : 64;
    const int act16 = argc > 4 ? atoi(argv[4]) : 0, neg = argc > 5 ? atoi(argv[5]) : 0;
    const int Ho = (H / 2 - 1) / 2 + 1, Wo = (W / 2 - 1) / 2 + 1;
    std::vector<float> x((size_t)B * H * W), w1(32 * 9), al(32), be(32), dw(9 * 32);
    uint32_t st = 12345;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : x) v = 80.0f * rnd() - 40.0f;
    for (auto& v : w1) v = rnd();
    for (auto& v : al) v = neg ? 2.0f * rnd() : 1.0f + rnd();
    for (auto& v : be) v = rnd();
    for (auto& v : dw) v = rnd();
    float *dx, *dw1, *dal, *dbe, *ddw, *d1, *x1;
    const size_t no = (size_t)B * Ho * Wo * 32, ob = no * (act16 ? 2 : 4);
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dw1, w1.size() * 4); hipMalloc(&dal, 128); hipMalloc(&dbe, 128); hipMalloc(&ddw, dw.size() * 4);
    hipMalloc(&d1, ob); hipMalloc(&x1, ob);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dal, al.data(), 128, hipMemcpyHostToDevice); hipMemcpy(dbe, be.data(), 128, hipMemcpyHostToDevice);
    hipMemcpy(ddw, dw.data(), dw.size() * 4, hipMemcpyHostToDevice);
    hipMemset(d1, 0xff, ob); hipMemset(x1, 0xff, ob);
    hipStream_t s; hipStreamCreate(&s);
    unsigned char* pack; hipMalloc(&pack, bc_front_b_packed_bytes());
    const float fws = 32768.0f, fin = 4.0f;
    launch_bc_front_b_pack_f16(dw1, pack, fws, s);
    Conv1DwArgs a{dx, dw1, nullptr, dal, dbe, ddw, d1, x1, B, H, W, ACT_RELU, 2, 2};
    a.wpack = pack; a.f16_in = fin; a.f16_clamp = 8192.0f; a.f16_unscale = 1.0f / (fin * fws); a.bn_pos = neg ? 0 : 1;
    a.bf16_out = act16; a.d_scale = 16.0f; a.xs_scale = 32.0f;
    hipError_t e = launch_bc_front_b(a, 3, 256, s);
    hipError_t e2 = hipStreamSynchronize(s);
    printf("B=%d H=%d W=%d act16=%d neg=%d rows/strip %d: launch %s, sync %s\n", B, H, W, act16, neg, bc_front_b_rows(H, W, 2), hipGetErrorString(e), hipGetErrorString(e2));
    std::vector<unsigned char> hd(ob), hx(ob);
    hipMemcpy(hd.data(), d1, ob, hipMemcpyDeviceToHost); hipMemcpy(hx.data(), x1, ob, hipMemcpyDeviceToHost);
    printf("hash d %016llx  xs %016llx\n", fnv(hd.data(), ob), fnv(hx.data(), ob));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) launch_bc_front_b(a, 3, 256, s);
    hipEventRecord(e0, s);
    const int K = 20;
    for (int i = 0; i < K; ++i) launch_bc_front_b(a, 3, 256, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%.4f ms per launch\n", ms / K);
    return 0;
}
