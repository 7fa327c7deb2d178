// Marginal cost of one instruction of each kind inside an MFMA stream on gfx950 (two waves per SIMD, 512-thread workgroups, one per CU):
// every wave issues { 1 v_mfma_f32_32x32x16_f16, K ops of kind OP } repeatedly; reported: SIMD cycles per MFMA.
//   OP 0 v_fma_f32   1 v_pk_fma_f32   2 v_cvt_pk_f16_f32   3 v_fma_mixlo_f16   4 ds_read_b128 (conflict-free, waited 8 later)
//   5 buffer_load_dwordx4 (L1 / L2 hits, waited 8 later)   6 v_pk_add_f32   7 ds_write_b128
// hipcc --offload-arch=gfx950 -O3 op_cost.hip -o op_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int OP, int K>
__global__ __launch_bounds__(512, 2) void k(float* out, const f32x4* __restrict__ src, unsigned long long* cyc, int iters) {
  __shared__ f32x4 lds[4096];
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = f32x4{1.f, 2.f, 3.f, (float)i};
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  uint4 ua = make_uint4(threadIdx.x, 1, 2, 3), ub = make_uint4(5, threadIdx.x, 7, 8);
  f16x8 a = __builtin_bit_cast(f16x8, ua), b = __builtin_bit_cast(f16x8, ub);
  f32x2 v2[8];
  float v[16];
  f32x4 ld[8];
  unsigned h[8];
  for (int i = 0; i < 8; ++i) { v2[i] = f32x2{(float)threadIdx.x, (float)i}; h[i] = threadIdx.x + i; ld[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x + i;
  const float c1 = out[0], c2 = out[1];
  const f32x2 c12 = {c1, c2};
  const unsigned laddr = (unsigned)(size_t)(lds + threadIdx.x);
  const f32x4* gp = src + (blockIdx.x & 7) * 512 + threadIdx.x;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j & 3], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < K; ++q) {
        const int n = j * K + q;
        if (OP == 0) v[n & 15] = fmaf(v[n & 15], c1, c2);
        if (OP == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %2" : "+v"(v2[n & 7]) : "v"(v2[(n + 3) & 7]), "v"(c12));
        if (OP == 6) asm volatile("v_pk_add_f32 %0, %1, %2" : "+v"(v2[n & 7]) : "v"(v2[(n + 3) & 7]), "v"(c12));
        if (OP == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[n & 7]) : "v"(v[n & 15]), "v"(v[(n + 5) & 15]));
        if (OP == 3) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(h[n & 7]) : "v"(h[(n + 3) & 7]), "v"(v[n & 15]));
        if (OP == 4) {
          if ((n & 7) == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ld[0]), "+v"(ld[1]), "+v"(ld[2]), "+v"(ld[3]), "+v"(ld[4]), "+v"(ld[5]), "+v"(ld[6]), "+v"(ld[7]));
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[n & 7]) : "v"(laddr), "n"(((n * 8192) & 0xffff) & ~15));
        }
        if (OP == 5) {
          if ((n & 7) == 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ld[0]), "+v"(ld[1]), "+v"(ld[2]), "+v"(ld[3]), "+v"(ld[4]), "+v"(ld[5]), "+v"(ld[6]), "+v"(ld[7]));
          asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(ld[n & 7]) : "v"(gp), "n"((n & 3) * 1024));
        }
        if (OP == 7) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(laddr), "v"(ld[n & 7]), "n"(((n * 8192) & 0xffff) & ~15) : "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(ld[0]), "+v"(ld[1]), "+v"(ld[2]), "+v"(ld[3]), "+v"(ld[4]), "+v"(ld[5]), "+v"(ld[6]), "+v"(ld[7]));
  unsigned long long t1 = __builtin_readcyclecounter();
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v2[i][0] + v2[i][1] + (float)h[i] + ld[i][0] + ld[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[2 + blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x][0];
}

template <int OP, int K>
void run(const char* name) {
  float* out; unsigned long long* cyc; f32x4* src;
  hipMalloc(&out, (2 + 256 * 512) * 4); hipMalloc(&cyc, 64); hipMalloc(&src, 8 * 512 * 16 + 8192);
  hipMemset(out, 0, 8); hipMemset(src, 0, 8 * 512 * 16 + 8192);
  const int iters = 500;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP, K><<<256, 512>>>(out, src, cyc, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP, K><<<256, 512>>>(out, src, cyc, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  const double nm = iters * 16.0;
  // s_memtime / readcyclecounter runs at 100 MHz on gfx950: report time-based cycles too (at the clock the event time implies: unknown -> ns)
  printf("%-22s K=%2d : %.3f ms  -> %.1f ns per {MFMA + K ops} per SIMD pair-of-waves slot (= %.1f ns per MFMA issue of ONE wave; 2 waves share the SIMD)\n",
         name, K, ms, ms * 1e6 / nm / 2, ms * 1e6 / nm);
  hipFree(out); hipFree(cyc); hipFree(src);
}
#define RUNS(OP, NAME) run<OP, 0>(NAME); run<OP, 2>(NAME); run<OP, 4>(NAME); run<OP, 6>(NAME); run<OP, 8>(NAME);
int main() {
  RUNS(0, "v_fma_f32") RUNS(1, "v_pk_fma_f32") RUNS(6, "v_pk_add_f32") RUNS(2, "v_cvt_pk_f16_f32") RUNS(3, "v_fma_mixlo_f16")
  run<4, 1>("ds_read_b128"); run<4, 2>("ds_read_b128"); run<4, 4>("ds_read_b128");
  run<5, 1>("global_load_dwordx4"); run<5, 2>("global_load_dwordx4");
  run<7, 1>("ds_write_b128"); run<7, 2>("ds_write_b128");
  return 0;
}
