// How many VALU instructions fit beside v_mfma_f32_32x32x16_bf16 on one SIMD of gfx950?
//   mode A: every wave issues { 1 MFMA, K independent v_fma_f32 } repeatedly; 1 or 2 waves per SIMD
//   mode B: waves 0-3 issue MFMAs only, waves 4-7 VALU only (independent chains): both rates are reported
// hipcc --offload-arch=gfx950 -O3 valu_mfma.hip -o valu_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, bool SPLIT>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  uint4 ua = make_uint4(threadIdx.x, 1, 2, 3), ub = make_uint4(5, threadIdx.x, 7, 8);
  bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  const float c1 = out[0], c2 = out[1];
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  if (!SPLIT) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < K; ++q) v[(j * K + q) & 7] = fmaf(v[(j * K + q) & 7], c1, c2);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, K, 0);
      }
    }
  } else if (wave < 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16 * K; ++j) v[j & 7] = fmaf(v[j & 7], c1, c2);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[2 + blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, bool SPLIT>
void run(int threads) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, (2 + 256 * 512) * 4); hipMalloc(&cyc, 64);
  hipMemset(out, 0, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<K, SPLIT><<<256, threads>>>(out, cyc, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<K, SPLIT><<<256, threads>>>(out, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  const double nm = iters * 16.0;
  if (!SPLIT)
    printf("interleaved K=%2d VALU per MFMA, %d waves/SIMD: %.3f ms, wave0 %.1f clk per {MFMA + K VALU}, per SIMD %.1f clk per MFMA\n", K,
           threads / 256, ms, h[0] / nm, h[0] / nm / (threads / 256));
  else
    printf("split: MFMA waves %.1f clk/MFMA; VALU waves: %d VALU in %.0f clk = %.2f clk/VALU  (%.3f ms)\n", h[0] / nm, 16 * K * iters,
           (double)h[4], h[4] / (nm * K), ms);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, false>(256); run<2, false>(256); run<4, false>(256); run<5, false>(256); run<6, false>(256); run<7, false>(256); run<8, false>(256); run<10, false>(256);
  run<0, false>(512); run<4, false>(512); run<6, false>(512); run<7, false>(512); run<8, false>(512); run<10, false>(512); run<12, false>(512);
  run<8, true>(512); run<16, true>(512);
  return 0;
}
