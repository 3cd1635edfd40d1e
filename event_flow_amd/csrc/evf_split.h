// Exact 3-way bf16 split of fp32 values, two at a time:  a = hi + mid + lo  with every term a bf16 (round to nearest even
// at each step; the residuals are exact fp32 subtractions).  v_cvt_pk_bf16_f32 (gfx950) rounds and packs a pair in ONE
// instruction -- the integer form ((u + 0x7FFF + ((u >> 16) & 1)) >> 16 per element, then shifts / ors to pack) costs
// ~4x the VALU work and gives the same bits for every finite input (fp32 denormals are preserved in this build).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 evf_bf16x2 __attribute__((ext_vector_type(2)));
typedef float evf_f32x2 __attribute__((ext_vector_type(2)));

// low half = bf16(a), high half = bf16(b)
__device__ __forceinline__ uint32_t evf_pk_bf16(float a, float b) {
  const evf_f32x2 f = {a, b};
  const evf_bf16x2 r = __builtin_convertvector(f, evf_bf16x2);
  return *(const uint32_t*)&r;
}

__device__ __forceinline__ void evf_split3_pair(float a, float b, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
  hi = evf_pk_bf16(a, b);
  const float ra = a - __uint_as_float(hi << 16), rb = b - __uint_as_float(hi & 0xFFFF0000u);
  mid = evf_pk_bf16(ra, rb);
  const float sa = ra - __uint_as_float(mid << 16), sb = rb - __uint_as_float(mid & 0xFFFF0000u);
  lo = evf_pk_bf16(sa, sb);
}
