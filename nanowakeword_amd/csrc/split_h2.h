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
