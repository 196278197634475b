// Shared pieces of the bf16x6 kernels (mlp_x6_kernels.hip, wgrad_x6_kernels.hip): vector types and the exact three-way split
// x = x1 + x2 + x3 of a float32 into bf16 parts (round to nearest even: 8 + 8 + 8 significand bits, signed residuals).
#pragma once
#include "pxo_common.h"

namespace pxo {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// exact three-way split
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x;
  const float r = x - (float)a;
  b = (__bf16)r;
  c = (__bf16)(r - (float)b);
}
// the same for a pair, packed [lo half = first | hi half = second]: v_cvt_pk_bf16_f32, widened back with a shift / a mask
// (the residuals as 2-vectors: v_pk_add_f32, one instruction per pair)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(f32x2 x) {
  const bf16x2 h = {(__bf16)x[0], (__bf16)x[1]};
  uint32_t b;
  __builtin_memcpy(&b, &h, 4);
  asm volatile("" : "+v"(b));     // ONE v_cvt_pk_bf16_f32: without this the low half is converted a second time for `b << 16`
  return b;
}
__device__ __forceinline__ f32x2 widen_pk_bf16(uint32_t b) { return f32x2{__uint_as_float(b << 16), __uint_as_float(b & 0xffff0000u)}; }
__device__ __forceinline__ void split3_pair(f32x2 x, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  p1 = cvt_pk_bf16(x);
  const f32x2 r = x - widen_pk_bf16(p1);
  p2 = cvt_pk_bf16(r);
  p3 = cvt_pk_bf16(r - widen_pk_bf16(p2));
}
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  split3_pair(f32x2{a, b}, p1, p2, p3);
}


}  // namespace pxo
