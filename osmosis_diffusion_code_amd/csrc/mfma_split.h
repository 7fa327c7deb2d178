// fp32 on the bf16 / fp16 matrix cores of gfx950: exact operand splitting and the 32x32x16 MFMA on 16-byte
// fragments.  Shared by the convolution kernels (igemm.hip) and the attention kernels (attention.hip).
//
// fp32 has no fast matrix path on gfx950 (no xf32; the f32 MFMA runs at the vector rate, 1/16 of the bf16 MFMA).
// An fp32 value splits EXACTLY into three bf16 terms  a = a0 + a1 + a2  (8+8+8 significand bits, RNE), so
//   a*b = a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0) + O(2^-24 |ab|):
//   NP = 3 planes, 6 bf16 MFMAs per product group -> fp32-class accuracy ("bf16x6");
//   NP = 2 planes, 3 bf16 MFMAs                   -> ~2^-16 relative      ("bf16x3");
//   NP = 1: ONE plane of IEEE half, fp16 x fp16 -> fp32 (the reference's use_fp16 arithmetic).
// Products are exact in the fp32 accumulator of the MFMA; only the dropped terms and the fp32 accumulation order
// differ from an exact-fp32 fmaf chain.
#pragma once
#include "osm_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float4 sel4(bool ok, float4 v) {
  return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {   // RNE, v_cvt_pk_bf16_f32
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// 4 fp32 -> NP planes of 4 16-bit elements (2 packed dwords per plane); 22 VALU ops for NP = 3.
template <int NP>
__device__ __forceinline__ void split_planes(float4 x, uint2 (&pl)[NP]) {
  if constexpr (NP == 1) {   // fp16 arithmetic: ONE half plane, RNE
    const osm::floatx4_t f = {x.x, x.y, x.z, x.w};
    pl[0] = __builtin_bit_cast(uint2, __builtin_convertvector(f, osm::half4_t));
    return;
  }
  unsigned a = cvt_pk_bf16(x.x, x.y), b = cvt_pk_bf16(x.z, x.w);
  pl[0] = make_uint2(a, b);
  float r0 = x.x - bf_lo(a), r1 = x.y - bf_hi(a), r2 = x.z - bf_lo(b), r3 = x.w - bf_hi(b);
  a = cvt_pk_bf16(r0, r1);
  b = cvt_pk_bf16(r2, r3);
  pl[1] = make_uint2(a, b);
  if (NP == 3) {
    r0 -= bf_lo(a); r1 -= bf_hi(a); r2 -= bf_lo(b); r3 -= bf_hi(b);
    pl[NP - 1] = make_uint2(cvt_pk_bf16(r0, r1), cvt_pk_bf16(r2, r3));
  }
}

// "f16x3" arithmetic (round 3): an fp32 value PRE-SCALED into the fp16 range splits into two IEEE-half terms
// a = h0 + h1 + O(2^-23 |a|) (11 + 11 significand bits, RNE), and a product is the three fp16 MFMAs h0g0 + h0g1 + h1g0:
// half the matrix work of bf16x6 for ~22-bit operands.  The residual a - h0 is one v_fma_mix_f32 (the half operand is
// read in place): 8 VALU for 4 values against 22 for three bf16 planes.
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {   // RNE, v_cvt_pk_f16_f32
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2));
}
__device__ __forceinline__ float sub_h_lo(unsigned h, float x) {       // x - (float)low half of h, exact
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
}
__device__ __forceinline__ float sub_h_hi(unsigned h, float x) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
}
// packed pair of second-plane halfs: (half)(x0 - lo(h)) | (half)(x1 - hi(h)) << 16.  The difference is exact in fp32, so the
// single rounding of v_fma_mixlo / mixhi_f16 gives the bits of cvt_pk_f16(sub_h_lo, sub_h_hi) in 2 instructions instead of 3
__device__ __forceinline__ unsigned resid_pk_f16(unsigned h, float x0, float x1) {
  unsigned r;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(h), "v"(x1));
  return r;
}
__device__ __forceinline__ void split_f16x2(float4 x, uint2 (&pl)[2]) {
  const unsigned a = cvt_pk_f16(x.x, x.y), b = cvt_pk_f16(x.z, x.w);
  pl[0] = make_uint2(a, b);
  pl[1] = make_uint2(resid_pk_f16(a, x.x, x.y), resid_pk_f16(b, x.z, x.w));
}
__device__ __forceinline__ f32x16 mma16h(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// 8 fp32 (two float4: elements 0-3, 4-7 of a fragment) -> NP 16-byte fragments
template <int NP>
__device__ __forceinline__ void split_frag8(float4 lo4, float4 hi4, uint4 (&pl)[NP]) {
  uint2 lo[NP], hi[NP];
  split_planes<NP>(lo4, lo);
  split_planes<NP>(hi4, hi);
#pragma unroll
  for (int q = 0; q < NP; ++q) pl[q] = make_uint4(lo[q].x, lo[q].y, hi[q].x, hi[q].y);
}

// one 32x32x16 MFMA on 16-byte fragments: bf16 planes (NP = 2, 3) or fp16 (NP = 1).
// A: lane l holds A[row = l & 31][k = 8 (l >> 5) + e], B: B[k = 8 (l >> 5) + e][col = l & 31], e = 0..7;
// C/D: col = l & 31, row = (r & 3) + 8 (r >> 2) + 4 (l >> 5), r = 0..15.
template <int NP>
__device__ __forceinline__ f32x16 mma16(uint4 a, uint4 b, f32x16 c) {
  if constexpr (NP == 1)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// sum over the NP-plane product group: c += (a0 + a1 + ..)(b0 + b1 + ..) without the O(2^-8 NP) terms
template <int NP>
__device__ __forceinline__ f32x16 mma_split(const uint4 (&a)[NP], const uint4 (&b)[NP], f32x16 c) {
#pragma unroll
  for (int pa = NP - 1; pa >= 0; --pa)
#pragma unroll
    for (int pb = NP - 1 - pa; pb >= 0; --pb) c = mma16<NP>(a[pa], b[pb], c);
  return c;
}

}  // namespace
