// Can two DIFFERENT kernels, launched on two streams, share a CU on gfx950 - one matrix-pipe wave per SIMD (256
// registers, AGPR accumulators) next to VALU waves of another kernel (<= 128 registers)?  And if they do, do they overlap
// the way two such waves of ONE workgroup do (tools/ubench/mfma_valu_overlap.hip)?
//
//   kernel M: 256 threads (one wave per SIMD), v_mfma_f32_32x32x16_bf16 on AGPR accumulators, LDS_M bytes of LDS
//   kernel V: 256 threads, __launch_bounds__(256, 4) (<= 128 VGPRs), v_fma_f32 chains, LDS_V bytes of LDS
// Every workgroup records (XCC, SE, CU), its start and end on the constant 100 MHz clock.  Reported: time of M alone, V
// alone, both on one stream, both on two streams, and for the two-stream run on how many CUs an M and a V workgroup were
// resident at the same time (and for what share of M's residency).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/coresidency.hip -o tools/ubench/coresidency
// run:   GPU_MAX_HW_QUEUES=4 tools/ubench/coresidency [itersM] [itersV] [ldsM KB] [ldsV KB] [gridM] [gridV] [V: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_accvgpr_read, 3 v_accvgpr_write] [M accumulator chains 1/2]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Rec { unsigned long long t0, t1; uint32_t hw, xcc; };

__device__ __forceinline__ void stamp(Rec* r, bool begin) {
    if (threadIdx.x == 0) {
        if (begin) {
            r->t0 = wall_clock64();
            r->hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
            r->xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
        } else {
            r->t1 = wall_clock64();
        }
    }
}

template <int NACC>      // 1: every MFMA depends on the previous one (one accumulator chain), 2: two chains alternate, 3: one chain, B operand in AGPRs
__global__ void __launch_bounds__(256) kern_m(int iters, float* out, Rec* rec) {
    extern __shared__ float lds[];
    stamp(rec + blockIdx.x, true);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    asm volatile("" : "+a"(acc0), "+a"(acc1));
    bf16x8 a, b;
#pragma unroll
    for (int r = 0; r < 8; ++r) { a[r] = (__bf16)(float)(threadIdx.x + r); b[r] = (__bf16)(float)(r + 1); }
    lds[threadIdx.x] = 1.0f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (NACC == 3) {        // B operand read from AGPRs
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "a"(b));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "a"(b));
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                if (NACC == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            }
        }
    }
    asm volatile("" : "+a"(acc0), "+a"(acc1));
    if (blockIdx.x == 0 && threadIdx.x == 0) {      // calibration: s_memtime ticks and 100 MHz ticks over iters x 16 MFMAs
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
        printf("calibration: %d MFMAs: %llu s_memtime ticks (%.2f per MFMA), %llu ticks of the 100 MHz clock -> s_memtime runs at %.0f MHz\n",
               iters * 16, c1 - c0, (double)(c1 - c0) / (iters * 16), w1 - w0, 100.0 * (double)(c1 - c0) / (double)(w1 - w0));
    }
    float s = lds[threadIdx.x];
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 12345.678f) out[threadIdx.x] = s;
    stamp(rec + blockIdx.x, false);
}

template <int PK>
__global__ void __launch_bounds__(256, 4) kern_v(int iters, float* out, Rec* rec) {
    extern __shared__ float lds[];
    stamp(rec + blockIdx.x, true);
    float x[8], ag[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { x[r] = (float)(threadIdx.x + r); ag[r] = x[r] + 1.0f; asm volatile("" : "+a"(ag[r])); }
    lds[threadIdx.x] = 1.0f;
    const float m = 1.0000001f, c = 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (PK == 1) x[r] = fmaf(x[r], m, c);          // hipcc's SLP pass pairs these into v_pk_fma_f32
                else if (PK == 2) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x[r]) : "a"(ag[r]));   // AGPR -> VGPR moves
                else if (PK == 3) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(ag[r]) : "v"(x[r]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[r]) : "v"(m), "v"(c));
            }
        }
    }
    float s = lds[threadIdx.x];
#pragma unroll
    for (int r = 0; r < 8; ++r) { asm volatile("" : "+a"(ag[r])); s += x[r] + ag[r]; }
    if (s == 12345.678f) out[threadIdx.x] = s;
    stamp(rec + blockIdx.x, false);
}

static uint32_t cu_key(const Rec& r) {      // gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0]
    return ((r.xcc & 0xf) << 8) | ((r.hw >> 8) & 0xff);
}

int main(int argc, char** argv) {
    const int itM = argc > 1 ? atoi(argv[1]) : 4000, itV = argc > 2 ? atoi(argv[2]) : 4000;
    const int ldsM = (argc > 3 ? atoi(argv[3]) : 58) * 1024, ldsV = (argc > 4 ? atoi(argv[4]) : 50) * 1024;
    const int gM = argc > 5 ? atoi(argv[5]) : 256, gV = argc > 6 ? atoi(argv[6]) : 512, pk = argc > 7 ? atoi(argv[7]) : 0, nacc = argc > 8 ? atoi(argv[8]) : 2;
    auto km = nacc == 1 ? kern_m<1> : nacc == 3 ? kern_m<3> : kern_m<2>;
    auto kv = pk == 1 ? kern_v<1> : pk == 2 ? kern_v<2> : pk == 3 ? kern_v<3> : kern_v<0>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(km), hipFuncAttributeMaxDynamicSharedMemorySize, ldsM);
    hipFuncSetAttribute(reinterpret_cast<const void*>(kv), hipFuncAttributeMaxDynamicSharedMemorySize, ldsV);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(km));
    printf("kern_m: %d regs, kern_v: ", fa.numRegs);
    hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kv));
    printf("%d regs; LDS M %d KB, V %d KB; grids %d / %d; V = %s, M = %d accumulator chain(s)\n", fa.numRegs, ldsM / 1024, ldsV / 1024, gM, gV, pk == 1 ? "v_pk_fma_f32" : pk == 2 ? "v_accvgpr_read_b32" : pk == 3 ? "v_accvgpr_write_b32" : "v_fma_f32", nacc);
    float* out; Rec *rm, *rv;
    hipMalloc(&out, 4096); hipMalloc(&rm, gM * sizeof(Rec)); hipMalloc(&rv, gV * sizeof(Rec));
    hipStream_t s1, s2; hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    auto timeit = [&](int mode) {       // 0: M alone, 1: V alone, 2: both on s1, 3: M on s1 and V on s2
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(e0, s1);
            hipStreamWaitEvent(s2, e0, 0);
            if (mode != 1) hipLaunchKernelGGL(km, dim3(gM), dim3(256), ldsM, s1, itM, out, rm);
            if (mode != 0) hipLaunchKernelGGL(kv, dim3(gV), dim3(256), ldsV, mode == 3 ? s2 : s1, itV, out, rv);
            hipEventRecord(e2, s2);
            hipStreamWaitEvent(s1, e2, 0);
            hipEventRecord(e1, s1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, ms);
        }
        return best;
    };
    const float tm = timeit(0), tv = timeit(1), t1 = timeit(2), t2 = timeit(3);
    printf("M alone %.3f ms | V alone %.3f ms | one stream %.3f ms | two streams %.3f ms\n", tm, tv, t1, t2);
    std::vector<Rec> hm(gM), hv(gV);
    hipMemcpy(hm.data(), rm, gM * sizeof(Rec), hipMemcpyDeviceToHost);
    hipMemcpy(hv.data(), rv, gV * sizeof(Rec), hipMemcpyDeviceToHost);
    std::map<uint32_t, std::vector<Rec>> byv;
    for (auto& r : hv) byv[cu_key(r)].push_back(r);
    std::map<uint32_t, int> cus;
    double share = 0; int shared_cus = 0;
    for (auto& r : hm) {
        cus[cu_key(r)]++;
        unsigned long long ov = 0;
        for (auto& v : byv[cu_key(r)]) {
            const unsigned long long a = std::max(r.t0, v.t0), b = std::min(r.t1, v.t1);
            if (b > a) ov = std::max(ov, b - a);
        }
        if (ov) { ++shared_cus; share += (double)ov / (double)(r.t1 - r.t0); }
    }
    printf("two-stream run: %zu distinct CUs seen by M, %zu by V; M workgroups that had a V workgroup on their CU at the same time: %d of %d (mean overlap %.0f %% of M's residency)\n",
           cus.size(), byv.size(), shared_cus, gM, shared_cus ? 100.0 * share / shared_cus : 0.0);
    unsigned long long m0 = ~0ull, m1 = 0, v0 = ~0ull, v1 = 0;
    for (auto& r : hm) { m0 = std::min(m0, r.t0); m1 = std::max(m1, r.t1); }
    for (auto& r : hv) { v0 = std::min(v0, r.t0); v1 = std::max(v1, r.t1); }
    const unsigned long long z = std::min(m0, v0);
    printf("spans (us from first start, 100 MHz clock): M %.1f..%.1f, V %.1f..%.1f\n", (m0 - z) / 100.0, (m1 - z) / 100.0, (v0 - z) / 100.0, (v1 - z) / 100.0);
    return 0;
}
