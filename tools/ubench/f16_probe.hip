// Probe of the gfx950 facts the matrix-pipe frontend (frontend3.hip) is built on.  Prints PASS / FAIL lines.
//   1  v_mfma_f32_16x16x32_f16 fragment layouts: A[i = lane & 15][k = 8 (lane >> 4) + e], B[k = 8 (lane >> 4) + e][j = lane & 15],
//      C[row = 4 (lane >> 4) + r][col = lane & 15] - checked with A = 0/1 selector rows and an ASYMMETRIC B
//   2  binary16 subnormal operands of that MFMA (A and B side): kept or flushed?
//   3  v_fma_mixlo_f16 / v_fma_mixhi_f16: hi = RN16(acc * 2^-12) and lo' = RN16(acc - hi * 2^12) in one instruction each
//   4  int16 -> binary16 (v_cvt_f16_i16, round to nearest even) and back, for the exact two-term split of a PCM sample
//   5  issue rate of back-to-back v_mfma_f32_16x16x32_f16 (one wave per SIMD, 2 / 4 accumulators) beside 32x32x16
// build: hipcc --offload-arch=gfx950 -O3 f16_probe.hip -o f16_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(64) k_layout(const _Float16* A, const _Float16* B, float* C) {   // A [16][32], B [32][16] row-major
    const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = A[i * 32 + 8 * g + e]; b[e] = B[(8 * g + e) * 16 + i]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[(4 * g + r) * 16 + i] = c[r];
}

__global__ void __launch_bounds__(64) k_mix(const float* acc, uint32_t* out, int n) {
    const int t = threadIdx.x;
    if (t >= n) return;
    const float a = acc[2 * t], b = acc[2 * t + 1];
    uint32_t hi = 0, lo = 0;
    const float dn = 0.000244140625f, up = 4096.0f;     // 2^-12, 2^12
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(hi) : "v"(a), "v"(dn));
    asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(hi) : "v"(b), "v"(dn));
    // lo' = acc - hi * 2^12: source 0 = -hi (binary16 half of the packed register), source 1 = 2^12 (f32), source 2 = acc (f32)
    asm volatile("v_fma_mixlo_f16 %0, -%1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(up), "v"(a));
    asm volatile("v_fma_mixhi_f16 %0, -%1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(up), "v"(b));
    out[2 * t] = hi;
    out[2 * t + 1] = lo;
}

__global__ void __launch_bounds__(64) k_cvt(const int16_t* x, uint16_t* hi, int16_t* lo, int n) {
    for (int t = threadIdx.x; t < n; t += 64) {
        const _Float16 h = (_Float16)x[t];                  // v_cvt_f16_i16, RNE
        const short back = (short)h;
        hi[t] = __builtin_bit_cast(uint16_t, h);
        lo[t] = (int16_t)(x[t] - back);
    }
}

template <int NACC, bool BIG>
__global__ void __launch_bounds__(256) k_rate(float* out, int iters, uint32_t seed) {
    uint4 av = make_uint4(0x3c003c00u + (seed + threadIdx.x) % 7, 0x38003c00u, 0x3c003800u, 0x34003c00u);
    f16x8 A = __builtin_bit_cast(f16x8, av), B = A;
    float s = 0.f;
    if (BIG) {
        f32x16 acc[NACC];
        for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rep = 0; rep < 8 / NACC; ++rep)
#pragma unroll
                for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc[a], 0, 0, 0);
        for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    } else {
        f32x4 acc[NACC];
        for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int rep = 0; rep < 8 / NACC; ++rep)
#pragma unroll
                for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, acc[a], 0, 0, 0);
        for (int a = 0; a < NACC; ++a) for (int r = 0; r < 4; ++r) s += acc[a][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float h2f(uint16_t h) { _Float16 v = __builtin_bit_cast(_Float16, h); return (float)v; }

int main() {
    _Float16 *A, *B; float* C;
    hipMallocManaged(&A, 16 * 32 * 2); hipMallocManaged(&B, 32 * 16 * 2); hipMallocManaged(&C, 256 * 4);
    // 1: layout.  A[i][k] = 1 if k == (i * 7 + 3) % 32: C[i][j] must be B[(7 i + 3) % 32][j] = 100 k + j
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (_Float16)(k == (i * 7 + 3) % 32 ? 1.0f : 0.0f);
    for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (_Float16)(float)(64 * k + j);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, A, B, C);
    hipDeviceSynchronize();
    int bad = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) bad += C[i * 16 + j] != (float)(64 * ((i * 7 + 3) % 32) + j);
    printf("%s 1 layout of v_mfma_f32_16x16x32_f16 (%d wrong of 256)\n", bad ? "FAIL" : "PASS", bad);
    // 2: subnormals.  A[0][0] = 2^-20 (subnormal), B[0][0] = 2^10 -> C[0][0] = 2^-10 if kept; and the B side
    for (int n = 0; n < 512; ++n) { A[n] = (_Float16)0.0f; B[n] = (_Float16)0.0f; }
    A[0] = __builtin_bit_cast(_Float16, (uint16_t)0x0010);   // 16 x 2^-24 = 2^-20
    B[0] = (_Float16)1024.0f;
    A[1 * 32 + 1] = (_Float16)1024.0f;
    B[1 * 16 + 1] = __builtin_bit_cast(_Float16, (uint16_t)0x0001);   // 2^-24
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, A, B, C);
    hipDeviceSynchronize();
    printf("%s 2 subnormal A operand kept (C = %g, expected %g)\n", C[0] == ldexpf(1.f, -10) ? "PASS" : "FAIL", C[0], ldexpf(1.f, -10));
    printf("%s 2 subnormal B operand kept (C = %g, expected %g)\n", C[1 * 16 + 1] == ldexpf(1.f, -14) ? "PASS" : "FAIL", C[17], ldexpf(1.f, -14));
    // 3: fma_mix split
    float* acc; uint32_t* out;
    const int n = 64;
    hipMallocManaged(&acc, 2 * n * 4); hipMallocManaged(&out, 2 * n * 4);
    uint32_t r = 12345u;
    for (int t = 0; t < 2 * n; ++t) {
        r = r * 1664525u + 1013904223u;
        const float m = 1.0f + (float)(r >> 9) / 8388608.0f;                 // [1, 2) with 23 random bits
        acc[t] = ((r & 1) ? -1.f : 1.f) * ldexpf(m, (int)((r >> 3) % 27));   // up to 2^27
    }
    acc[0] = 0.f; acc[1] = 1.0f; acc[2] = 134217727.0f * 0.99f;
    hipLaunchKernelGGL(k_mix, dim3(1), dim3(64), 0, 0, acc, out, n);
    hipDeviceSynchronize();
    bad = 0;
    double worst = 0;
    for (int t = 0; t < n; ++t)
        for (int q = 0; q < 2; ++q) {
            const float a = acc[2 * t + q];
            const uint16_t hb = (uint16_t)(out[2 * t] >> (16 * q)), lb = (uint16_t)(out[2 * t + 1] >> (16 * q));
            const _Float16 he = (_Float16)(a * 0.000244140625f);
            const _Float16 le = (_Float16)(a - (float)he * 4096.0f);
            if (hb != __builtin_bit_cast(uint16_t, he) || lb != __builtin_bit_cast(uint16_t, le)) ++bad;
            const double rec = ((double)h2f(hb) * 4096.0 + (double)h2f(lb));
            if (a != 0.f) worst = fmax(worst, fabs(rec - a) / fabs(a));
        }
    printf("%s 3 v_fma_mix{lo,hi}_f16 split (%d mismatches of %d; worst |hi 2^12 + lo' - acc| / |acc| = %.3g = 2^%.1f)\n", bad ? "FAIL" : "PASS", bad, 2 * n, worst, log2(worst));
    // 4: int16 round trip
    int16_t *x, *lo; uint16_t* hi;
    hipMallocManaged(&x, 65536 * 2); hipMallocManaged(&lo, 65536 * 2); hipMallocManaged(&hi, 65536 * 2);
    for (int v = 0; v < 65536; ++v) x[v] = (int16_t)(v - 32768);
    hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, x, hi, lo, 65536);
    hipDeviceSynchronize();
    bad = 0;
    int maxlo = 0;
    for (int v = 0; v < 65536; ++v) {
        const double rec = (double)h2f(hi[v]) + (double)lo[v];
        bad += rec != (double)x[v];
        if (abs(lo[v]) > maxlo) maxlo = abs(lo[v]);
    }
    printf("%s 4 int16 = RN16 + lo exactly for all 65536 values (%d wrong, max |lo| = %d)\n", bad ? "FAIL" : "PASS", bad, maxlo);
    // 5: issue rates
    float* o;
    hipMalloc(&o, 256 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](auto kern, const char* name, double flops_per_mfma) {
        const int iters = 20000;
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, o, 100, 1u);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, o, iters, 1u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n_mfma = (double)iters * 8;             // per wave
        printf("  5 %-34s %.3f ms: %.1f ns per MFMA per wave (one wave per SIMD) = %.0f TFLOP/s\n", name, ms, ms * 1e6 / n_mfma,
               n_mfma * 1024 * flops_per_mfma / (ms * 1e-3) / 1e12);
    };
    timeit(k_rate<1, false>, "16x16x32 f16, 1 accumulator", 2.0 * 16 * 16 * 32);
    timeit(k_rate<2, false>, "16x16x32 f16, 2 accumulators", 2.0 * 16 * 16 * 32);
    timeit(k_rate<4, false>, "16x16x32 f16, 4 accumulators", 2.0 * 16 * 16 * 32);
    timeit(k_rate<1, true>, "32x32x16 f16, 1 accumulator", 2.0 * 32 * 32 * 16);
    timeit(k_rate<2, true>, "32x32x16 f16, 2 accumulators", 2.0 * 32 * 32 * 16);
    return 0;
}
