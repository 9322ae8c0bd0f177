// split_h2.h - the two-term binary16 split of the "f16x3" arithmetic (trunk_b.hip, gemm_x3.hip, ...).
//   v (already times its power-of-two scale) = hi + lo,  hi = RN16(v),  lo = RN16(v - hi)
// hi carries 11 significant bits; v - hi (exact in float32) has at most 12 left, of which lo keeps 11: v is represented to
// 2^-23 of itself (exactly, half of the time).  Four VALU instructions per PAIR of values: v_cvt_pk_f16_f32, two
// v_fma_mix_f32 (v - hi with the binary16 half read in place: no conversion back), v_cvt_pk_f16_f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 nww_f16x2 __attribute__((ext_vector_type(2)));
typedef float nww_f32x2 __attribute__((ext_vector_type(2)));

// (a, b) -> hi = (RN16(a), RN16(b)), lo = (RN16(a - hi.a), RN16(b - hi.b)), each as two binary16 packed in a dword (a low)
__device__ __forceinline__ void nww_split2h(float a, float b, uint32_t& hi, uint32_t& lo) {
    const nww_f32x2 v = {a, b};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, nww_f16x2));
    float ra, rb;
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(ra) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(rb) : "v"(hi), "v"(b));
    const nww_f32x2 r = {ra, rb};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, nww_f16x2));
}

// ---- 16-bit storage of activation tensors (nww_config.act_dtype): ACT16_BF16 = bf16 (round to nearest even, no scale),
// ACT16_F16 = binary16 of value x a plan-time power-of-two scale (round to nearest even, saturating at +-65504)
enum { ACT16_F32 = 0, ACT16_BF16 = 1, ACT16_F16 = 2 };
__device__ __forceinline__ uint32_t nww_pk_bf16(float a, float b) {
    union { __bf16 h[2]; uint32_t u; } c;
    c.h[0] = (__bf16)a; c.h[1] = (__bf16)b;
    return c.u;
}
__device__ __forceinline__ uint32_t nww_pk_f16_sat(float a, float b) {
    const nww_f32x2 v = {__builtin_amdgcn_fmed3f(a, -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(b, -65504.0f, 65504.0f)};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, nww_f16x2));
}
// kind is wave-uniform; f16: (a, b) times mul
__device__ __forceinline__ uint32_t nww_pk_act16(int kind, float a, float b, float mul) {
    return kind == ACT16_F16 ? nww_pk_f16_sat(a * mul, b * mul) : nww_pk_bf16(a, b);
}
// the two values of a packed dword back to float32 (f16: still times the tensor's scale)
__device__ __forceinline__ void nww_unpk_act16(int kind, uint32_t u, float& a, float& b) {
    if (kind == ACT16_F16) {
        const nww_f16x2 h = __builtin_bit_cast(nww_f16x2, u);
        a = (float)h[0]; b = (float)h[1];
    } else {
        a = __uint_as_float(u << 16); b = __uint_as_float(u & 0xffff0000u);
    }
}
