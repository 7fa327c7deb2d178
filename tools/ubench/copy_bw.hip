// HBM streaming calibration: what does a plain float4 copy / a read-only pass / (reads + SiLU + a write) reach on this part, and with
// which grid / loads in flight?  The yardstick for the GroupNorm passes (csrc/norm.hip), which are this shape of kernel.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/copy_bw.hip -o tools/ubench/copy_bw && tools/ubench/copy_bw
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE, int UNROLL>   // 0 copy, 1 copy + silu, 2 read only (sum), 3 two reads + silu + one write
__global__ __launch_bounds__(256) void k(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ o,
                                         long long n, float* sink) {
  const long long stride = (long long)gridDim.x * 256;
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * UNROLL) {
    float4 v[UNROLL], w[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long long j = i + u * stride;
      v[u] = j < n ? a[j] : make_float4(0, 0, 0, 0);
      if (MODE == 3) w[u] = j < n ? b[j] : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long long j = i + u * stride;
      float4 r = v[u];
      if (MODE == 1 || MODE == 3) {
        auto f = [](float z) { return z * __frcp_rn(1.0f + __expf(-z)); };
        r = make_float4(f(r.x * 1.1f + 0.1f), f(r.y * 1.1f + 0.1f), f(r.z * 1.1f + 0.1f), f(r.w * 1.1f + 0.1f));
        if (MODE == 3) { r.x *= w[u].x; r.y *= w[u].y; r.z *= w[u].z; r.w *= w[u].w; }
      }
      if (MODE == 2) acc += r.x + r.y + r.z + r.w;
      else if (j < n) o[j] = r;
    }
  }
  if (MODE == 2 && acc == 12345.f) *sink = acc;
}

constexpr int NSET = 6;     // rotate through buffer sets > the 256 MB Infinity Cache

template <int MODE, int UNROLL>
void run(const char* name, float4* a, float4* b, float4* o, long long n, int grid, float* sink, double bytes) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, UNROLL>), dim3(grid), dim3(256), 0, 0, a, b, o, n, sink);
  hipEventRecord(e0);
  const int reps = 24;
  for (int i = 0; i < reps; ++i) {
    const long long off = (long long)(i % NSET) * n;
    hipLaunchKernelGGL((k<MODE, UNROLL>), dim3(grid), dim3(256), 0, 0, a + off, b + off, o + off, n, sink);
  }
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-26s grid %6d x256, %d float4 in flight per thread and operand  %7.1f us  %6.2f TB/s\n", name, grid, UNROLL,
         ms / reps * 1e3, bytes / (ms / reps) / 1e9);
}

int main() {
  const long long nf4 = 65536LL * 256 / 4;      // one 256 x 256 x 256 fp32 tensor = 67 MB
  float4 *a, *b, *o;
  float* sink;
  hipMalloc(&a, nf4 * 16 * NSET);
  hipMalloc(&b, nf4 * 16 * NSET);
  hipMalloc(&o, nf4 * 16 * NSET);
  hipMalloc(&sink, 4);
  hipMemset(a, 0, nf4 * 16 * NSET);
  hipMemset(b, 0, nf4 * 16 * NSET);
  const double B = nf4 * 16.0;
  for (int grid : {1024, 2048, 4096, 8192, 16384}) {
    run<0, 4>("copy", a, b, o, nf4, grid, sink, 2 * B);
    run<0, 8>("copy", a, b, o, nf4, grid, sink, 2 * B);
    run<1, 4>("copy + silu", a, b, o, nf4, grid, sink, 2 * B);
    run<2, 8>("read only", a, b, o, nf4, grid, sink, B);
    run<3, 4>("2 reads + silu + write", a, b, o, nf4, grid, sink, 3 * B);
  }
  return 0;
}
