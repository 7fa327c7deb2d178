// HBM-bound helpers of the UNet: resampling, layout conversion, softmax rows, timestep embedding,
// small linear layers.  All operate on NHWC matrix views (row = pixel, ld = row stride).
#include "osm_common.h"

namespace {

// ---------------------------------------------------------------- 2x2 sum-pool (avg-pool / upsample-bwd)
// reference: Downsample(use_conv=False) = AvgPool2d(2,2)  (unet.py:213-215)
template <int VEC>
__global__ __launch_bounds__(256) void pool2x2_kernel(const act_t* __restrict__ x, long long ldx,
                                                       act_t* __restrict__ y, long long ldy, int B, int H,
                                                       int W, int C, float scale, const act_t* __restrict__ x2 = nullptr,
                                                       long long ldx2 = 0, act_t* __restrict__ y2 = nullptr, long long ldy2 = 0) {
  if (blockIdx.y == 1) { x = x2; ldx = ldx2; y = y2; ldy = ldy2; }     // the second tensor of a pair (osm_*_pair)
  const int Ho = H / 2, Wo = W / 2, vpr = C / VEC;
  const long long total = (long long)B * Ho * Wo * vpr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % vpr);
    long long r = i / vpr;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const long long in0 = ((long long)(b * H + 2 * ho) * W + 2 * wo);
    const long long out = ((long long)(b * Ho + ho) * Wo + wo);
    const int c = v * VEC;
    if (VEC == 4) {
      const float4 a0 = osm::ld4(x + in0 * ldx + c);
      const float4 a1 = osm::ld4(x + (in0 + 1) * ldx + c);
      const float4 a2 = osm::ld4(x + (in0 + W) * ldx + c);
      const float4 a3 = osm::ld4(x + (in0 + W + 1) * ldx + c);
      float4 o;
      o.x = ((a0.x + a1.x) + (a2.x + a3.x)) * scale;
      o.y = ((a0.y + a1.y) + (a2.y + a3.y)) * scale;
      o.z = ((a0.z + a1.z) + (a2.z + a3.z)) * scale;
      o.w = ((a0.w + a1.w) + (a2.w + a3.w)) * scale;
      osm::st4(y + out * ldy + c, o);
    } else {
      const float s = (osm::ld1(x + in0 * ldx + c) + osm::ld1(x + (in0 + 1) * ldx + c)) +
                      (osm::ld1(x + (in0 + W) * ldx + c) + osm::ld1(x + (in0 + W + 1) * ldx + c));
      osm::st1(y + out * ldy + c, s * scale);
    }
  }
}

// ---------------------------------------------------------------- nearest 2x upsample (/ avg-pool-bwd)
// reference: Upsample(use_conv=False) = F.interpolate(scale_factor=2, mode="nearest") (unet.py:186)
template <int VEC>
__global__ __launch_bounds__(256) void upsample2x_kernel(const act_t* __restrict__ x, long long ldx,
                                                          act_t* __restrict__ y, long long ldy, int B, int H,
                                                          int W, int C, float scale, const act_t* __restrict__ x2 = nullptr,
                                                          long long ldx2 = 0, act_t* __restrict__ y2 = nullptr, long long ldy2 = 0) {
  if (blockIdx.y == 1) { x = x2; ldx = ldx2; y = y2; ldy = ldy2; }     // the second tensor of a pair (osm_*_pair)
  const int Ho = 2 * H, Wo = 2 * W, vpr = C / VEC;
  const long long total = (long long)B * Ho * Wo * vpr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % vpr);
    long long r = i / vpr;
    const int wo = (int)(r % Wo);
    r /= Wo;
    const int ho = (int)(r % Ho);
    const int b = (int)(r / Ho);
    const long long in = ((long long)(b * H + ho / 2) * W + wo / 2);
    const long long out = ((long long)(b * Ho + ho) * Wo + wo);
    const int c = v * VEC;
    if (VEC == 4) {
      float4 a = osm::ld4(x + in * ldx + c);
      a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale;
      osm::st4(y + out * ldy + c, a);
    } else {
      osm::st1(y + out * ldy + c, osm::ld1(x + in * ldx + c) * scale);
    }
  }
}

// ---------------------------------------------------------------- stride-2 pick / place, per-image row vector (round 5)
// The UNet variants no shipped Osmosis config uses (unet.py:160-219 with use_conv, :329-332 without scale-shift norm):
//   pick:  y[b][i][j] = x[b][2 i][2 j]            a stride-2 3x3 convolution = the stride-1 convolution, every other pixel kept
//   place: y[b][2 i][2 j] = x[b][i][j], 0 elsewhere  its adjoint (the gradient of the kept pixels, before the stride-1 data gradient)
//   rowvec: y[b][p][c] += v[b][c]                  h + emb_out (additive conditioning), emb + label_emb[y]
template <int MODE>      // 0 pick (x: H x W -> y: H/2 x W/2), 1 place (x: H/2 x W/2 -> y: H x W)
__global__ __launch_bounds__(256) void stride2_kernel(const act_t* __restrict__ x, long long ldx, act_t* __restrict__ y, long long ldy,
                                                       int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2;
  const int oh = MODE == 0 ? Ho : H, ow = MODE == 0 ? Wo : W;
  const long long total = (long long)B * oh * ow * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int w = (int)(r % ow);
    r /= ow;
    const int h = (int)(r % oh);
    const int b = (int)(r / oh);
    if (MODE == 0) {
      osm::st1(y + ((long long)(b * Ho + h) * Wo + w) * ldy + c, osm::ld1(x + ((long long)(b * H + 2 * h) * W + 2 * w) * ldx + c));
    } else {
      const bool kept = !(h & 1) && !(w & 1);
      const float v = kept ? osm::ld1(x + ((long long)(b * Ho + h / 2) * Wo + w / 2) * ldx + c) : 0.f;
      osm::st1(y + ((long long)(b * H + h) * W + w) * ldy + c, v);
    }
  }
}
__global__ __launch_bounds__(256) void add_rowvec_kernel(act_t* __restrict__ y, long long ldy, const float* __restrict__ v, long long ldv,
                                                          int B, long long HW, int C) {
  const long long total = (long long)B * HW * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long row = i / C;
    act_t* d = y + row * ldy + c;
    osm::st1(d, osm::ld1(d) + v[(row / HW) * ldv + c]);
  }
}

// ---------------------------------------------------------------- layout
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, act_t* __restrict__ y,
                                                            long long ldy, int B, int C, int HW) {
  const long long total = (long long)B * HW * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long row = i / C;  // b*HW + p
    const int p = (int)(row % HW);
    const int b = (int)(row / HW);
    osm::st1(y + row * ldy + c, x[((long long)b * C + c) * HW + p]);
  }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const act_t* __restrict__ x, long long ldx,
                                                            float* __restrict__ y, int B, int C, int HW) {
  const long long total = (long long)B * HW * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const long long bc = i / HW;
    const int c = (int)(bc % C);
    const int b = (int)(bc / C);
    y[i] = osm::ld1(x + ((long long)b * HW + p) * ldx + c);
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void copy2d_kernel(const act_t* __restrict__ x, long long ldx,
                                                      act_t* __restrict__ y, long long ldy, long long M, int C,
                                                      int accumulate) {
  const int vpr = C / VEC;
  const long long total = M * vpr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / vpr;
    const int c = (int)(i - row * vpr) * VEC;
    if (VEC == 4) {
      float4 a = osm::ld4(x + row * ldx + c);
      act_t* d = y + row * ldy + c;
      if (accumulate) {
        const float4 o = osm::ld4(d);
        a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
      }
      osm::st4(d, a);
    } else {
      float a = osm::ld1(x + row * ldx + c);
      if (accumulate) a += osm::ld1(y + row * ldy + c);
      osm::st1(y + row * ldy + c, a);
    }
  }
}

// ---------------------------------------------------------------- softmax over rows (one wave per row)
// reference: th.softmax(weight.float(), dim=-1) (unet.py:431)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, float* __restrict__ P,
                                                            float* __restrict__ PT, long long nrows, int T) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const float* s = S + row * T;
  const long long mat = row / T;
  const int t = (int)(row - mat * T);
  if (T <= 1024) {   // the row lives in registers (<= 16 values per lane): one read, one write
    float v[16];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = lane + 64 * k;
      v[k] = i < T ? s[i] : -INFINITY;
      mx = fmaxf(mx, v[k]);
    }
    mx = osm::wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      v[k] = expf(v[k] - mx);          // exp(-inf) = 0 for the padding lanes
      sum += v[k];
    }
    sum = osm::wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = lane + 64 * k;
      if (i < T) {
        const float pv = v[k] * inv;
        P[row * T + i] = pv;
        if (PT) PT[(mat * T + i) * T + t] = pv;
      }
    }
    return;
  }
  float mx = -INFINITY;
  for (int i = lane; i < T; i += 64) mx = fmaxf(mx, s[i]);
  mx = osm::wave_max(mx);
  float sum = 0.f;
  for (int i = lane; i < T; i += 64) sum += expf(s[i] - mx);
  sum = osm::wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int i = lane; i < T; i += 64) {
    const float pv = expf(s[i] - mx) * inv;
    P[row * T + i] = pv;
    if (PT) PT[(mat * T + i) * T + t] = pv;
  }
}

// dS = P * (dP - sum_s dP*P)
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ P,
                                                                const float* __restrict__ dP,
                                                                float* __restrict__ dS, float* __restrict__ dST,
                                                                long long nrows, int T) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const float* pr = P + row * T;
  const float* dp = dP + row * T;
  const long long mat = row / T;
  const int t = (int)(row - mat * T);
  if (T <= 1024) {
    float pv[16], dv[16];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = lane + 64 * k;
      pv[k] = i < T ? pr[i] : 0.f;
      dv[k] = i < T ? dp[i] : 0.f;
      dot += pv[k] * dv[k];
    }
    dot = osm::wave_sum(dot);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = lane + 64 * k;
      if (i < T) {
        const float v = pv[k] * (dv[k] - dot);
        dS[row * T + i] = v;
        if (dST) dST[(mat * T + i) * T + t] = v;
      }
    }
    return;
  }
  float dot = 0.f;
  for (int i = lane; i < T; i += 64) dot += pr[i] * dp[i];
  dot = osm::wave_sum(dot);
  for (int i = lane; i < T; i += 64) {
    const float v = pr[i] * (dp[i] - dot);
    dS[row * T + i] = v;
    if (dST) dST[(mat * T + i) * T + t] = v;
  }
}

// Transposed copies for T % 64 == 0 (every attention resolution of the network): the strided 4-byte transposed
// stores of the row kernels ran at ~1.2 TB/s, so the row kernels then write only the row-major result and this
// kernel transposes 64 x 64 tiles through LDS (256-byte segments on both sides).
__global__ __launch_bounds__(256) void transpose64_kernel(const float* __restrict__ X, float* __restrict__ Y, int T) {
  __shared__ float tile[64][65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long base = (long long)blockIdx.z * T * T;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll 4
  for (int rr = wave; rr < 64; rr += 4) tile[rr][lane] = X[base + (long long)(r0 + rr) * T + c0 + lane];
  __syncthreads();
#pragma unroll 4
  for (int cc = wave; cc < 64; cc += 4) Y[base + (long long)(c0 + cc) * T + r0 + lane] = tile[lane][cc];
}

// ---------------------------------------------------------------- timestep embedding (nn.py:103-121)
__global__ void temb_kernel(const float* __restrict__ t, float* __restrict__ out, int B, int dim,
                            float max_period) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, k = i % half;
  const float freq = expf(-logf(max_period) * (float)k / (float)half);
  const float arg = t[b] * freq;
  out[(long long)b * dim + k] = cosf(arg);
  out[(long long)b * dim + half + k] = sinf(arg);
  if ((dim & 1) && k == 0) out[(long long)b * dim + dim - 1] = 0.f;
}

// ---------------------------------------------------------------- y[B][N] = act(x)[B][K] W[N][K]^T + b
// one wave per output feature n; W streamed once with 16-byte loads (weight-bandwidth bound).
template <int VEC>
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ Wt,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      int B, int K, int N, int silu_in, int silu_out) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const float* w = Wt + (long long)n * K;
  for (int b = 0; b < B; ++b) {
    const float* xb = x + (long long)b * K;
    float acc = 0.f;
    if (VEC == 4) {
      for (int k = 4 * lane; k < K; k += 256) {
        const float4 wv = *reinterpret_cast<const float4*>(w + k);
        float4 xv = *reinterpret_cast<const float4*>(xb + k);
        if (silu_in) {
          xv.x = osm::silu_f(xv.x); xv.y = osm::silu_f(xv.y);
          xv.z = osm::silu_f(xv.z); xv.w = osm::silu_f(xv.w);
        }
        acc += wv.x * xv.x + wv.y * xv.y + wv.z * xv.z + wv.w * xv.w;
      }
    } else {
      for (int k = lane; k < K; k += 64) {
        float xv = xb[k];
        if (silu_in) xv = osm::silu_f(xv);
        acc += w[k] * xv;
      }
    }
    acc = osm::wave_sum(acc);
    if (lane == 0) {
      float v = acc + (bias ? bias[n] : 0.f);
      if (silu_out) v = osm::silu_f(v);
      y[(long long)b * N + n] = v;
    }
  }
}

inline int grid_for(long long total) {
  long long b = (total + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// (float family only below this kernel) 2-D strided conversion between the two activation storage types
#ifdef OSM_ACT_F16
template <bool TO_HALF>
__global__ __launch_bounds__(256) void convert2d_kernel(const void* __restrict__ x, long long ldx, void* __restrict__ y,
                                                         long long ldy, long long M, int C) {
  const int vpr = C / 4;
  const long long total = M * vpr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / vpr;
    const int c = (int)(i - row * vpr) * 4;
    if (TO_HALF)
      osm::st4(reinterpret_cast<_Float16*>(y) + row * ldy + c, osm::ld4(reinterpret_cast<const float*>(x) + row * ldx + c));
    else
      osm::st4(reinterpret_cast<float*>(y) + row * ldy + c, osm::ld4(reinterpret_cast<const _Float16*>(x) + row * ldx + c));
  }
}
#endif

}  // namespace

extern "C" int OSM_FN(osm_pool2x2)(const abi_act_t* x_, long long ldx, abi_act_t* y_, long long ldy, int B, int H, int W,
                                   int C, float scale, void* stream) {
  const act_t* x = OSM_CACT(x_);
  act_t* y = OSM_ACT(y_);
  OSM_REQUIRE(x && y, "osm_pool2x2: null pointer");
  OSM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "osm_pool2x2: H, W must be even");
  const bool v4 = C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && osm::aligned_act4(x) && osm::aligned_act4(y);
  const long long total = (long long)B * (H / 2) * (W / 2) * (C / (v4 ? 4 : 1));
  if (v4)
    hipLaunchKernelGGL((pool2x2_kernel<4>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, B, H, W, C, scale);
  else
    hipLaunchKernelGGL((pool2x2_kernel<1>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, B, H, W, C, scale);
  return osm::check_launch("pool2x2_kernel");
}

extern "C" int OSM_FN(osm_upsample2x)(const abi_act_t* x_, long long ldx, abi_act_t* y_, long long ldy, int B, int H,
                                      int W, int C, float scale, void* stream) {
  const act_t* x = OSM_CACT(x_);
  act_t* y = OSM_ACT(y_);
  OSM_REQUIRE(x && y, "osm_upsample2x: null pointer");
  OSM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "osm_upsample2x: bad shape");
  const bool v4 = C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && osm::aligned_act4(x) && osm::aligned_act4(y);
  const long long total = (long long)B * (2 * H) * (2 * W) * (C / (v4 ? 4 : 1));
  if (v4)
    hipLaunchKernelGGL((upsample2x_kernel<4>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, B, H, W, C, scale);
  else
    hipLaunchKernelGGL((upsample2x_kernel<1>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, B, H, W, C, scale);
  return osm::check_launch("upsample2x_kernel");
}

extern "C" int OSM_FN(osm_stride2_pick)(const abi_act_t* x_, long long ldx, abi_act_t* y_, long long ldy, int B, int H, int W, int C,
                                        void* stream) {
  OSM_REQUIRE(x_ && y_, "osm_stride2_pick: null pointer");
  OSM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "osm_stride2_pick: H, W must be even");
  hipLaunchKernelGGL((stride2_kernel<0>), dim3(grid_for((long long)B * (H / 2) * (W / 2) * C)), dim3(256), 0, (hipStream_t)stream,
                     OSM_CACT(x_), ldx, OSM_ACT(y_), ldy, B, H, W, C);
  return osm::check_launch("stride2_kernel<0>");
}
extern "C" int OSM_FN(osm_stride2_place)(const abi_act_t* x_, long long ldx, abi_act_t* y_, long long ldy, int B, int H, int W, int C,
                                         void* stream) {
  OSM_REQUIRE(x_ && y_, "osm_stride2_place: null pointer");
  OSM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "osm_stride2_place: H, W (of the OUTPUT) must be even");
  hipLaunchKernelGGL((stride2_kernel<1>), dim3(grid_for((long long)B * H * W * C)), dim3(256), 0, (hipStream_t)stream,
                     OSM_CACT(x_), ldx, OSM_ACT(y_), ldy, B, H, W, C);
  return osm::check_launch("stride2_kernel<1>");
}
extern "C" int OSM_FN(osm_add_rowvec)(abi_act_t* y_, long long ldy, const float* v, long long ldv, int B, long long HW, int C,
                                      void* stream) {
  OSM_REQUIRE(y_ && v, "osm_add_rowvec: null pointer");
  OSM_REQUIRE(B > 0 && C > 0 && HW > 0 && ldv >= C, "osm_add_rowvec: bad shape");
  hipLaunchKernelGGL(add_rowvec_kernel, dim3(grid_for((long long)B * HW * C)), dim3(256), 0, (hipStream_t)stream, OSM_ACT(y_), ldy, v,
                     ldv, B, HW, C);
  return osm::check_launch("add_rowvec_kernel");
}

// Two tensors of the same shape resampled by ONE launch (an up / down ResBlock resamples its input and its normalised input,
// and in the backward its two gradients: unet.py:263-270): saves a dependent launch (~4.5 us) per pair.  4-element vectors only.
extern "C" int OSM_FN(osm_resample_pair)(int up, const abi_act_t* x1_, long long ldx1, abi_act_t* y1_, long long ldy1,
                                         const abi_act_t* x2_, long long ldx2, abi_act_t* y2_, long long ldy2, int B, int H,
                                         int W, int C, float scale, void* stream) {
  const act_t *x1 = OSM_CACT(x1_), *x2 = OSM_CACT(x2_);
  act_t *y1 = OSM_ACT(y1_), *y2 = OSM_ACT(y2_);
  OSM_REQUIRE(x1 && y1 && x2 && y2, "osm_resample_pair: null pointer");
  OSM_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "osm_resample_pair: bad shape");
  OSM_REQUIRE(C % 4 == 0 && ldx1 % 4 == 0 && ldy1 % 4 == 0 && ldx2 % 4 == 0 && ldy2 % 4 == 0 && osm::aligned_act4(x1) &&
              osm::aligned_act4(y1) && osm::aligned_act4(x2) && osm::aligned_act4(y2),
              "osm_resample_pair: C and the row strides must be multiples of 4, pointers aligned to 4 elements");
  OSM_REQUIRE(up || (H % 2 == 0 && W % 2 == 0), "osm_resample_pair: pooling needs even H, W");
  const long long total = up ? (long long)B * (2 * H) * (2 * W) * (C / 4) : (long long)B * (H / 2) * (W / 2) * (C / 4);
  const dim3 grid(grid_for(total), 2);
  if (up)
    hipLaunchKernelGGL((upsample2x_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, x1, ldx1, y1, ldy1, B, H, W, C, scale,
                       x2, ldx2, y2, ldy2);
  else
    hipLaunchKernelGGL((pool2x2_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, x1, ldx1, y1, ldy1, B, H, W, C, scale,
                       x2, ldx2, y2, ldy2);
  return osm::check_launch("resample pair kernel");
}

extern "C" int OSM_FN(osm_nchw_to_nhwc)(const float* x, abi_act_t* y_, long long ldy, int B, int C, int HW,
                                        void* stream) {
  act_t* y = OSM_ACT(y_);
  OSM_REQUIRE(x && y && B > 0 && C > 0 && HW > 0 && ldy >= C, "osm_nchw_to_nhwc: bad argument");
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long long)B * C * HW)), dim3(256), 0,
                     (hipStream_t)stream, x, y, ldy, B, C, HW);
  return osm::check_launch("nchw_to_nhwc_kernel");
}

extern "C" int OSM_FN(osm_nhwc_to_nchw)(const abi_act_t* x_, long long ldx, float* y, int B, int C, int HW,
                                        void* stream) {
  const act_t* x = OSM_CACT(x_);
  OSM_REQUIRE(x && y && B > 0 && C > 0 && HW > 0 && ldx >= C, "osm_nhwc_to_nchw: bad argument");
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((long long)B * C * HW)), dim3(256), 0,
                     (hipStream_t)stream, x, ldx, y, B, C, HW);
  return osm::check_launch("nhwc_to_nchw_kernel");
}

extern "C" int OSM_FN(osm_copy2d)(const abi_act_t* x_, long long ldx, abi_act_t* y_, long long ldy, long long M, int C,
                                  int accumulate, void* stream) {
  const act_t* x = OSM_CACT(x_);
  act_t* y = OSM_ACT(y_);
  OSM_REQUIRE(x && y && M > 0 && C > 0, "osm_copy2d: bad argument");
  const bool v4 = C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && osm::aligned_act4(x) && osm::aligned_act4(y);
  const long long total = M * (C / (v4 ? 4 : 1));
  if (v4)
    hipLaunchKernelGGL((copy2d_kernel<4>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, M, C, accumulate);
  else
    hipLaunchKernelGGL((copy2d_kernel<1>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, M, C, accumulate);
  return osm::check_launch("copy2d_kernel");
}

#ifndef OSM_ACT_F16
// per-image max |x| of an [B][HW][C] activation tensor (row stride ldx) as OSM_MAXABS_PARTS partial maxima per image: block
// (j, img) writes the bit pattern of its non-negative partial to out[img][j] (a NaN anywhere sets NaN bits, which compare
// above every finite pattern: it poisons the scale like it would poison an fp32 convolution).  No atomics, no clearing:
// every slot is written on every call; the consumer (osm_conv_desc::x_maxabs, f16x3 Winograd kernel) folds the partials.
__global__ __launch_bounds__(512) void maxabs_kernel(const float* __restrict__ x, long long ldx, long long rows_per_img, int C4,
                                                     unsigned* __restrict__ out) {
  const int img = blockIdx.y;
  const float* __restrict__ xb = x + (long long)img * rows_per_img * ldx;
  const long long total = rows_per_img * C4;          // float4 vectors of this image
  const bool dense = ldx == 4LL * C4;
  const long long stride = (long long)gridDim.x * blockDim.x;
  float m = 0.f;
  unsigned nanbits = 0;
  auto take = [&](float4 v) {
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));   // fmaxf drops NaNs ...
    if (v.x != v.x || v.y != v.y || v.z != v.z || v.w != v.w) nanbits = 0x7fc00000u;      // ... so they are tracked apart
  };
  auto at = [&](long long i) -> float4 {
    if (dense) return reinterpret_cast<const float4*>(xb)[i];
    const long long r = i / C4;
    return *reinterpret_cast<const float4*>(xb + r * ldx + (i - r * C4) * 4);
  };
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 7 * stride < total; i += 8 * stride) {   // eight independent 16-byte loads in flight per thread
    float4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = at(i + q * stride);
#pragma unroll
    for (int q = 0; q < 8; ++q) take(v[q]);
  }
  for (; i < total; i += stride) take(at(i));
  unsigned bits = __float_as_uint(m) | nanbits;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, o, 64));
  __shared__ unsigned wmax[8];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = bits;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int q = 1; q < 8; ++q) bits = max(bits, wmax[q]);
    out[(long long)img * gridDim.x + blockIdx.x] = bits;
  }
}
extern "C" int osm_maxabs_parts(void) { return OSM_MAXABS_PARTS; }
extern "C" int osm_maxabs(const float* x, long long ldx, int B, long long rows_per_img, int C, float* out, void* stream) {
  OSM_REQUIRE(x && out && B > 0 && rows_per_img > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldx >= C && osm::aligned16(x),
              "osm_maxabs: bad argument (C, ldx multiples of 4, 16-byte aligned x)");
  hipLaunchKernelGGL(maxabs_kernel, dim3(OSM_MAXABS_PARTS, B), dim3(512), 0, (hipStream_t)stream, x, ldx, rows_per_img, C / 4,
                     reinterpret_cast<unsigned*>(out));
  return osm::check_launch("maxabs_kernel");
}
#endif

#ifdef OSM_ACT_F16
extern "C" int osm_half_to_f32(const osm_half_t* x, long long ldx, float* y, long long ldy, long long M, int C,
                               void* stream) {
  OSM_REQUIRE(x && y && M > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && osm::aligned16(y) &&
              osm::aligned_act4(x), "osm_half_to_f32: bad argument (C, ld multiples of 4; aligned pointers)");
  hipLaunchKernelGGL((convert2d_kernel<false>), dim3(grid_for(M * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     (const void*)x, ldx, (void*)y, ldy, M, C);
  return osm::check_launch("convert2d_kernel");
}
extern "C" int osm_f32_to_half(const float* x, long long ldx, osm_half_t* y, long long ldy, long long M, int C,
                               void* stream) {
  OSM_REQUIRE(x && y && M > 0 && C > 0 && C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && osm::aligned16(x) &&
              osm::aligned_act4(y), "osm_f32_to_half: bad argument (C, ld multiples of 4; aligned pointers)");
  hipLaunchKernelGGL((convert2d_kernel<true>), dim3(grid_for(M * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                     (const void*)x, ldx, (void*)y, ldy, M, C);
  return osm::check_launch("convert2d_kernel");
}
#else

extern "C" int osm_softmax_rows(const float* S, float* P, float* PT, int nmat, int T, void* stream) {
  OSM_REQUIRE(S && P && nmat > 0 && T > 0, "osm_softmax_rows: bad argument");
  const long long nrows = (long long)nmat * T;
  if (PT && T % 64 == 0) {
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       S, P, (float*)nullptr, nrows, T);
    hipLaunchKernelGGL(transpose64_kernel, dim3(T / 64, T / 64, nmat), dim3(256), 0, (hipStream_t)stream, P, PT, T);
    return osm::check_launch("softmax_rows_kernel + transpose64_kernel");
  }
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     S, P, PT, nrows, T);
  return osm::check_launch("softmax_rows_kernel");
}

extern "C" int osm_softmax_rows_bwd(const float* P, const float* dP, float* dS, float* dST, int nmat, int T,
                                    void* stream) {
  OSM_REQUIRE(P && dP && dS && nmat > 0 && T > 0, "osm_softmax_rows_bwd: bad argument");
  const long long nrows = (long long)nmat * T;
  if (dST && T % 64 == 0) {
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, P, dP, dS, (float*)nullptr, nrows, T);
    hipLaunchKernelGGL(transpose64_kernel, dim3(T / 64, T / 64, nmat), dim3(256), 0, (hipStream_t)stream, dS, dST, T);
    return osm::check_launch("softmax_rows_bwd_kernel + transpose64_kernel");
  }
  hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, P, dP, dS, dST, nrows, T);
  return osm::check_launch("softmax_rows_bwd_kernel");
}

extern "C" int osm_timestep_embedding(const float* t, float* out, int B, int dim, float max_period,
                                      void* stream) {
  OSM_REQUIRE(t && out && B > 0 && dim >= 2, "osm_timestep_embedding: bad argument");
  const int n = B * (dim / 2);
  hipLaunchKernelGGL(temb_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, out, B, dim,
                     max_period);
  return osm::check_launch("temb_kernel");
}

extern "C" int osm_linear(const float* x, const float* W, const float* b, float* y, int B, int K, int N,
                          int silu_in, int silu_out, void* stream) {
  OSM_REQUIRE(x && W && y && B > 0 && K > 0 && N > 0, "osm_linear: bad argument");
  const bool v4 = K % 4 == 0 && osm::aligned16(x) && osm::aligned16(W);
  if (v4)
    hipLaunchKernelGGL((linear_kernel<4>), dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, W, b, y, B, K, N, silu_in, silu_out);
  else
    hipLaunchKernelGGL((linear_kernel<1>), dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, W, b, y, B, K, N, silu_in, silu_out);
  return osm::check_launch("linear_kernel");
}

extern "C" int osm_version(void) { return (0 << 16) | (2 << 8) | 0; }
extern "C" const char* osm_last_error(void) { return osm::err_buf(); }
#endif   // !OSM_ACT_F16
