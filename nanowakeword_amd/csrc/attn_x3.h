// attn_x3.h - the Conformer's attention module in ONE launch (attn_x3.hip): h <- h + out_proj(softmax(q k^T / sqrt(dh)) v) with
// q, k, v = in_proj(h) computed per head inside the kernel from the clip's own rows (architectures.py:471-493, 512-513).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct AttnArgs {
    const float* h;                  // [B][T][D] residual stream (read twice: operand rows, then the residual)
    float* out;                      // [B][T][D]; may be h (a workgroup reads and writes only its own clip)
    const unsigned char* packed;     // launch_attn_x3_pack output (weight chunks in kernel order)
    const float* bc;                 // [D] out_proj.bias + out_proj.weight . v-bias (the v bias commutes with the softmax average)
    int B, T;
    float w_un;                      // 1 / (scale of the packed in_proj weights)
    float cK, cV;                    // powers of two that bring the raw k / v accumulators into the binary16 range (plan-time bounds)
    float o_un;                      // 1 / (scale of the packed out_proj weights x in_proj weight scale x cV)
    float qscale;                    // 1 / sqrt(head dim)
    int stagger = 0;                 // set by the launcher: start-up delay step in shader clocks (workgroups start in eight phases)
};

bool attn_x3_supported(int T, int D, int n_head);
size_t attn_x3_packed_bytes(int D, int n_head);
// in_w [3D][D], in_b [3D], out_w [D][D], out_b [D] float32 -> packed chunks (two binary16 terms of W x ws) and bc [D]
hipError_t launch_attn_x3_pack(const float* in_w, const float* in_b, const float* out_w, const float* out_b, void* packed, float* bc,
                               int D, int n_head, float ws_in, float ws_out, hipStream_t s);
hipError_t launch_attn_x3(const AttnArgs& a, int D, int n_head, hipStream_t s);
