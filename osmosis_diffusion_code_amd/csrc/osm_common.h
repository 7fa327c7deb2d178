// Shared helpers for the gfx950 kernels of libosmosis_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/osmosis_hip.h"

// ---------------------------------------------------------------- activation storage type of this build family
// Every source that touches NHWC activations is compiled twice: as is (activations fp32, entry points `osm_*`) and
// with -DOSM_ACT_F16 (activations IEEE half in HBM, fp32 arithmetic / accumulation everywhere, entry points
// `osm_*_h`) -- the reference's `use_fp16` storage (unet.py:544,733; fp16_util.py:13-20) with GroupNorm32's fp32
// arithmetic (nn.py:17-19).  Both families live in libosmosis_hip.so.
#ifdef OSM_ACT_F16
typedef _Float16 act_t;
typedef osm_half_t abi_act_t;      // how the C ABI spells the element type (IEEE binary16 bits)
#define OSM_FN(name) name##_h
#define OSM_ACT_IS_F16 1
#else
typedef float act_t;
typedef float abi_act_t;
#define OSM_FN(name) name
#define OSM_ACT_IS_F16 0
#endif
#define OSM_ACT(p) reinterpret_cast<act_t*>(p)
#define OSM_CACT(p) reinterpret_cast<const act_t*>(p)

namespace osm {

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float floatx4_t __attribute__((ext_vector_type(4)));

// 4 consecutive activations <-> fp32 registers (16-byte access for fp32 storage, 8-byte for half storage)
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const _Float16* p) {
  const floatx4_t f = __builtin_convertvector(*reinterpret_cast<const half4_t*>(p), floatx4_t);
  return make_float4(f[0], f[1], f[2], f[3]);
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(_Float16* p, float4 v) {
  const floatx4_t f = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<half4_t*>(p) = __builtin_convertvector(f, half4_t);   // RNE
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const _Float16* p) { return (float)*p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(_Float16* p, float v) { *p = (_Float16)v; }
// VEC (1, 4 or 8) consecutive activations <-> fp32 registers; 8 halfs = one 16-byte access
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx8_t __attribute__((ext_vector_type(8)));
template <int VEC, typename T>
__device__ __forceinline__ void ldv(const T* p, float (&v)[VEC]) {
  if constexpr (VEC == 1) {
    v[0] = ld1(p);
  } else if constexpr (VEC == 8 && sizeof(T) == 2) {
    const floatx8_t f = __builtin_convertvector(*reinterpret_cast<const half8_t*>(p), floatx8_t);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = f[e];
  } else {
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) {
      const float4 t = ld4(p + 4 * q);
      v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
  }
}
template <int VEC, typename T>
__device__ __forceinline__ void stv(T* p, const float (&v)[VEC]) {
  if constexpr (VEC == 1) {
    st1(p, v[0]);
  } else if constexpr (VEC == 8 && sizeof(T) == 2) {
    floatx8_t f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = v[e];
    *reinterpret_cast<half8_t*>(p) = __builtin_convertvector(f, half8_t);
  } else {
#pragma unroll
    for (int q = 0; q < VEC / 4; ++q) st4(p + 4 * q, make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]));
  }
}
// alignment of a 4-element activation vector access
inline bool aligned_act4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & (4 * sizeof(act_t) - 1)) == 0; }

inline char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OSM_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return OSM_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// sigmoid via the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each); the accurate expf /
// IEEE divide cost ~10x the VALU work and turn the HBM-bound GroupNorm passes VALU-bound.
__device__ __forceinline__ float sigmoid_f(float z) { return __frcp_rn(1.0f + __expf(-z)); }
__device__ __forceinline__ float silu_f(float z) { return z * sigmoid_f(z); }
// d silu(z) / dz = s (1 + z (1 - s)),  s = sigmoid(z)
__device__ __forceinline__ float dsilu_f(float z) {
  const float s = sigmoid_f(z);
  return s * (1.0f + z * (1.0f - s));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace osm

#define OSM_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) return osm::fail(OSM_ERR_INVALID, __VA_ARGS__); \
  } while (0)
