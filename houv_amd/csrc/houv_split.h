// houv_split.h -- an fp32 value as the sum of three bf16 parts, x = hi + mid + lo: round-to-nearest at each step, so the parts
// carry 8 + 8 + 8 significant bits and every residual is exact in fp32.  bf16 x bf16 is exact in fp32 and the bf16 MFMAs
// accumulate in fp32, so a product summed from the part products of weight >= 2^-16 (hi*hi, hi*mid, mid*hi, hi*lo, mid*mid,
// lo*hi) is fp32-grade -- on a matrix pipe that runs bf16 at 16x the fp32-input rate (gemm.hip, attention.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace houv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {   // v_cvt_pk_bf16_f32: a in the low half
  const f32x2 p = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(p, bf16x2));
}
__device__ __forceinline__ float bf16_lo_as_f32(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float bf16_hi_as_f32(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }

// four consecutive values -> 8 bytes per part
template <int NPART>
__device__ __forceinline__ void split4(const float4 v, uint2 (&out)[NPART]) {
  float r0 = v.x, r1 = v.y, r2 = v.z, r3 = v.w;
#pragma unroll
  for (int p = 0; p < NPART; ++p) {
    const unsigned a = pack_bf16(r0, r1), b = pack_bf16(r2, r3);
    out[p] = make_uint2(a, b);
    if (p + 1 < NPART) {
      r0 -= bf16_lo_as_f32(a); r1 -= bf16_hi_as_f32(a);
      r2 -= bf16_lo_as_f32(b); r3 -= bf16_hi_as_f32(b);
    }
  }
}

// eight values (element j of an MFMA operand fragment) -> one 16-byte fragment per part
template <int NPART>
__device__ __forceinline__ void split8(const float (&v)[8], uint4 (&out)[NPART]) {
  uint2 a[NPART], b[NPART];
  split4<NPART>(make_float4(v[0], v[1], v[2], v[3]), a);
  split4<NPART>(make_float4(v[4], v[5], v[6], v[7]), b);
#pragma unroll
  for (int p = 0; p < NPART; ++p) out[p] = make_uint4(a[p].x, a[p].y, b[p].x, b[p].y);
}

}  // namespace houv
