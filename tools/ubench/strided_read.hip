// Micro-benchmark: HBM read bandwidth of the split-operand GEMM's A-operand access pattern on gfx950.
// A [4096][12800] float32 (210 MB); 512 workgroups = 32 row blocks (128 rows) x 16 K chunks (800 floats); every
// workgroup walks its chunk in pieces of PIECE bytes per row: 128 rows x PIECE bytes per step, one 16-byte load per
// lane.  PIECE = 128 is what gemm_x3 does per 32-k tile; larger pieces = more contiguous bytes per row per step.
// build: hipcc --offload-arch=gfx950 -O3 strided_read.hip -o strided_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int PIECE>
__global__ void __launch_bounds__(256) k(const float* __restrict__ A, float* __restrict__ out, int M, int K, int kchunk) {
    constexpr int LPR = PIECE / 16;                 // lanes per row
    constexpr int RPP = 256 / LPR;                  // rows per pass
    const int bm = blockIdx.x * 128, k0 = blockIdx.y * kchunk;
    const int lr = threadIdx.x / LPR, lc = threadIdx.x % LPR;
    float4 acc = make_float4(0, 0, 0, 0);
    for (int k = 0; k < kchunk; k += PIECE / 4) {
#pragma unroll
        for (int p = 0; p < 128 / RPP; ++p) {
            if (k + 4 * lc >= kchunk) continue;
            const float4 v = *reinterpret_cast<const float4*>(A + (size_t)(bm + lr + RPP * p) * K + k0 + k + 4 * lc);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    out[(blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
template <typename F> static float time_ms(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 10; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}
int main() {
    const int M = 4096, K = 12800;
    float *A, *out;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&out, 512 * 256 * 4);
    hipMemset(A, 0, (size_t)M * K * 4);
    // a 1 GB buffer touched between runs would defeat the 256 MB infinity cache; A itself (210 MB) mostly fits, so also
    // run with a cache-flushing pass in between
    float* flush;
    if (hipMalloc(&flush, (size_t)1 << 30) != hipSuccess) { printf("flush alloc failed\n"); return 1; }
#define RUN(P, KC)                                                                                         \
    {                                                                                                      \
        float ms = time_ms([&] { hipMemsetAsync(flush, 1, (size_t)1 << 30, 0);                             \
                                 hipLaunchKernelGGL(k<P>, dim3(32, K / KC), dim3(256), 0, 0, A, out, M, K, KC); }); \
        float ms0 = time_ms([&] { hipMemsetAsync(flush, 1, (size_t)1 << 30, 0); });                        \
        printf("piece %4d B, k-chunk %5d: %.3f ms -> %.2f TB/s\n", P, KC, ms - ms0, 210e6 / ((ms - ms0) * 1e-3) / 1e12); \
    }
    RUN(128, 800) RUN(256, 800) RUN(512, 800) RUN(1024, 800) RUN(128, 1600) RUN(512, 1600) RUN(128, 400) RUN(512, 400)
    return 0;
}
