// Shared helpers for the gfx950 kernels of libosmosis_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/osmosis_hip.h"

namespace osm {

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
