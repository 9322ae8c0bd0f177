// convmod_x3.h - the Conformer's convolution module in one clip-resident launch (convmod_x3.hip):
// h <- h + conv2( swish( BN( depthwise_k31( GLU( conv1( LayerNorm(h) ) ) ) ) ) )
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct ConvModArgs {
    const float* h;                  // [B][T][D] residual stream
    float* out;                      // [B][T][D]; may be h (a workgroup reads and writes only its own clip)
    const float* ln_w; const float* ln_b;      // conv_module.layer_norm [D]
    // weight tiles in lin_x3's packing, launch_lin_x3_pack(W, bias, K = D, n_out = D, parts = 1, .., terms = 2, ws): one 20 KB chunk per 32 outputs
    const unsigned char* packed1a;   // conv1 rows 0 .. D - 1 (the GLU's a half), scale ws1
    const unsigned char* packed1b;   // conv1 rows D .. 2 D - 1 (the gate), the SAME scale ws1
    const unsigned char* packed2;    // conv2, scale ws2
    const float* dw_t;               // depthwise weights tap-major [31][D] (launch_convmod_taps)
    const float* dw_b;               // depthwise bias [D]
    const float* bn_a; const float* bn_b;      // folded BatchNorm alpha / beta [D]
    int B, T;
    float w1_un, w2_un;              // 1 / ws1, 1 / ws2
};

bool convmod_x3_supported(int T, int D, int kernel_size);
hipError_t launch_convmod_taps(const float* w /*[D][KT]*/, float* wt /*[KT][D]*/, int D, int KT, hipStream_t s);
hipError_t launch_convmod_x3(const ConvModArgs& a, int D, hipStream_t s);
