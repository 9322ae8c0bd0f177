// lin_x3.h - input-stationary short-K Linear layers on the bf16 matrix cores (lin_x3.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct LinArgs {
    const float* x; int ldx;         // [M][K] input rows
    float* out; int ldc;             // [M][N]
    const float* res; int ldres;     // epilogue 1: out = res + rscale * (y + bias)   (res may be out)
    float rscale;
    const float* ln_w; const float* ln_b;      // LayerNorm over the K input features first (epilogue 2 instances only)
    const unsigned char* packed;     // launch_lin_x3_pack output
    int M, N;                        // N = output features written (for the GLU epilogue: of the gated product)
    int nblk = 0;                    // filled by the launcher
    // epilogue 0 only, qkv_T > 0: N = 3 D outputs are stored head-major, out[which = q|k|v][clip][head][t][dh] (clip = row / qkv_T,
    // t = row % qkv_T, head = (col % D) / qkv_dh) - every (clip, head) block of the attention kernel is then one contiguous run
    int qkv_T = 0, qkv_dh = 0;
    // h2 = 1 (NWW_ARITH_F16X3): two binary16 terms per operand, three products per float32 product.  The weights are packed with
    // terms = 2 as binary16 terms of W x ws (ws a power of two putting the largest weight in [2^14, 2^15)), w_un = 1 / ws; an input row
    // is scaled by ITS OWN power of two (largest element into [2^14, 2^15), found in registers after the LayerNorm) and the
    // accumulators are multiplied back per row - no bound on the tensor is needed (dual_x3.hip: DualArgs::h2).
    int h2 = 0;
    float w_un = 1.0f;
};

// bytes of one packed 32-output block: parts x K/16 x terms fragments of 1 KB + parts x 32 biases, padded to whole 4 KB copy steps
__host__ __device__ inline size_t lin_x3_block_bytes(int K, int parts, int terms = 3) {
    return ((size_t)parts * (K / 16) * terms * 1024 + (size_t)parts * 128 + 4095) & ~(size_t)4095;
}
bool lin_x3_supported(int K, int N, bool h2 = true);      // K = 192 / 256: two-term (h2) instances only
size_t lin_x3_packed_bytes(int K, int n_out, int parts, int terms = 3);
// W [parts * gate_off .. ][K] float32, bias or nullptr -> packed; parts = 2, gate_off = N for the GLU pairing (rows j and N + j)
// terms = 3: three bf16 terms per weight; terms = 2: two binary16 terms of W x ws (LinArgs::h2)
hipError_t launch_lin_x3_pack(const float* W, const float* bias, void* out, int K, int n_out, int parts, int gate_off, hipStream_t s,
                              int terms = 3, float ws = 1.0f);
// epi: 0 plain, 1 residual, 2 GLU; ln: LayerNorm prologue (epi 2 only)
hipError_t launch_lin_x3(const LinArgs& a, int K, int epi, bool ln, hipStream_t s);
