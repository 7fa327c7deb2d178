// Pure-MFMA micro-benchmark: cycles per v_mfma_f32_32x32x16_bf16 with 1 or 2 waves per SIMD, and with a
// VALU-heavy partner wave on the same SIMD.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: all waves MFMA; 1: waves 4..7 do VALU work instead; 2: waves 4..7 do LDS stores
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[8192];
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  uint4 ua = make_uint4(threadIdx.x, 1, 2, 3), ub = make_uint4(5, threadIdx.x, 7, 8);
  bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
  float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0 || wave < 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[3], 0, 0, 0);
      }
    }
  } else if (MODE == 1) {
    for (int it = 0; it < iters * 4; ++it) {
#pragma unroll
      for (int j = 0; j < 48; ++j) { v0 = v0 * v1 + v2; v1 = v1 * v2 + v3; v2 = v2 * v3 + v0; v3 = v3 * v0 + v1; }
    }
  } else {
    for (int it = 0; it < iters * 8; ++it) {
#pragma unroll
      for (int j = 0; j < 12; ++j) reinterpret_cast<float4*>(lds)[(threadIdx.x + 64 * j) & 2047] = make_float4(v0, v1, v2, v3 + j);
      __builtin_amdgcn_s_waitcnt(0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
  float s = v0 + v1 + v2 + v3 + lds[threadIdx.x];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int threads) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 64);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, threads>>>(out, cyc, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<256, threads>>>(out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  const int mwaves = (MODE == 0) ? threads / 64 : 4;
  double nm = (double)iters * 48;
  double tf = 256.0 * mwaves * nm * 32768.0 * 2 / 2 / (ms * 1e-3) / 1e12;   // 32x32x16x2 flop
  printf("%-34s threads=%d  %.3f ms  %.0f TFLOP/s  cycles/MFMA (wave0) = %.1f  clock ~ %.2f GHz\n", name, threads, ms, tf,
         (double)h[0] / nm, (double)h[0] / (ms * 1e6));
}

int main() {
  run<0>("MFMA only, 1 wave/SIMD", 256);
  run<0>("MFMA only, 2 waves/SIMD", 512);
  run<1>("MFMA + VALU partner", 512);
  run<2>("MFMA + LDS-store partner", 512);
  return 0;
}
