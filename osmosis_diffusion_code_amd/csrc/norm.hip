// GroupNorm(32) (+FiLM scale/shift) (+SiLU), forward and data-gradient, on NHWC matrix views.
// Reference: nn.py:17-19,93-100 (GroupNorm32, fp32, eps 1e-5), unet.py:263,287 (SiLU),
// unet.py:327-331 (out_norm(h) * (1 + scale) + shift).
//
// HBM-bound.  NHWC makes a group's channels contiguous inside a pixel row, so one workgroup sweeps a
// slab of full pixel rows with 16-byte coalesced loads and keeps per-thread partial sums for the
// (fixed) channel vector(s) it owns; a structured (deterministic, atomics-free) LDS pass folds
// threads into the 32 groups; chunk partials are combined in fp64 by a finalize kernel.
//   pass 1 (stats / grad-stats):  read x (and dy)         -> [B][chunk][G][2] partials -> [B][G][2]
//   pass 2 (apply / grad-apply):  read x (and dy), write y / dx
#include "osm_common.h"
#include <cstdlib>
#include <string>

namespace {

// Pixel rows per chunk (one workgroup).  Low-resolution layers (8x8 ... 32x32 with 512-2048 channels) are
// latency-bound, not bandwidth-bound: a fixed 64-row chunk left them with 1-16 workgroups and 64 dependent
// row iterations per thread (measured 80 us for a 262 KB tensor); chunks shrink so that >= ~64 workgroups exist.
// (round 2: HW / 512 instead of HW / 256 -- 512 workgroups on a 128x128 tensor, 256 on 64x64 ... -- measured -8 % over the step's
// GroupNorm kernels; HW / 384, / 768, / 1024 and caps of 32 / 48 / 128 rows were all worse)
#ifndef GN_PPC_DIV
#define GN_PPC_DIV 512
#define GN_PPC_MAX 64
#endif
__host__ __device__ inline int gn_ppc(int HW) {
  int p = HW / GN_PPC_DIV;
  return p < 4 ? 4 : (p > GN_PPC_MAX ? GN_PPC_MAX : p);
}
constexpr int NJMAX = 4;  // channel vectors per thread (C <= 4096)

struct GNArgs {
  const act_t* x;
  const act_t* dy;
  const float* stats;   // [B][G][2] mean, rstd
  const float* gstats;  // [B][G][2] m1, m2 (backward)
  const float* gamma;
  const float* beta;
  const float* film;    // [B][2C] or null
  const act_t* addend;
  const act_t* addend2;   // second optional addend of the backward apply (out = dGN(dy) + addend + addend2)
  act_t* out;
  float* part;
  float* fin;           // finalized statistics written by the one-launch kernel / the fused apply prologue
  unsigned* maxabs;     // optional [B][OSM_MAXABS_PARTS] partial max |out| (bit patterns), the contract of osm_maxabs: the
                        // f16x3 convolution that reads `out` next needs its range and this pass holds every value
  unsigned* maxabs_in;  // the same for the INPUT x, from the forward statistics pass (a 1x1 f16x3 convolution reads x itself)
  double n;             // elements per group
  int fuse;             // apply kernels: combine the chunk partials in the prologue (no finalize launch)
  long long ldx, lddy, ldo, ldadd, ldadd2, ldf;
  int B, HW, C, G, gs, nchunk, ppc, silu;
  float eps;
};

__device__ __forceinline__ void gn_fwd_elem(float x, float mean, float rstd, float ga, float be,
                                            bool film, float sc, float sh, float& xh, float& z) {
  xh = (x - mean) * rstd;
  z = xh * ga + be;
  if (film) z = z * (1.0f + sc) + sh;
}

// workgroup `wg` of `nwg` (<= OSM_MAXABS_PARTS) publishes its partial max |out| of image b: slot wg, and zeros in the slots no
// workgroup owns (every slot is rewritten on every call: no clearing, no atomics).  Called by ALL threads of the workgroup.
__device__ __forceinline__ void gn_publish_max(unsigned* __restrict__ maxabs, int b, int wg, int nwg, float m, unsigned nanbits) {
  __shared__ unsigned wmax[16];
  unsigned bits = __float_as_uint(m) | nanbits;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, o, 64));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = bits;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) bits = max(bits, wmax[w]);
    unsigned* sl = maxabs + (long long)b * OSM_MAXABS_PARTS;
    for (int s2 = wg; s2 < OSM_MAXABS_PARTS; s2 += nwg) sl[s2] = s2 == wg ? bits : 0u;
  }
}

// MODE 0: sums of (x, x^2).   MODE 1: sums of (dxh, dxh*xh) for the backward.
template <int VEC, int MODE>
__global__ __launch_bounds__(256) void gn_reduce_kernel(GNArgs a) {
  __shared__ float red[256 * NJMAX * 2];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int vpr = a.C / VEC;
  const int colT = vpr < 256 ? vpr : 256;
  const int rowT = 256 / colT;
  const int nj = (vpr + 255) / 256;
  const int tc = tid % colT, tr = tid / colT;
  const int p0 = chunk * a.ppc;
  const int p1 = min(a.HW, p0 + a.ppc);

  float s1[NJMAX], s2[NJMAX];
#pragma unroll
  for (int j = 0; j < NJMAX; ++j) s1[j] = s2[j] = 0.f;
  float imax = 0.f;
  unsigned inan = 0u;

  if (tr < rowT) {
#pragma unroll
    for (int j = 0; j < NJMAX; ++j) {
      const int v = tc + colT * j;
      if (j >= nj || v >= vpr) break;
      const int c = v * VEC;
      const int g = c / a.gs;
      float mean = 0.f, rstd = 0.f;
      float ga[VEC], be[VEC], sc[VEC], sh[VEC];
      if (MODE == 1) {
        mean = a.stats[(b * a.G + g) * 2];
        rstd = a.stats[(b * a.G + g) * 2 + 1];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          ga[e] = a.gamma[c + e];
          be[e] = a.beta[c + e];
          sc[e] = a.film ? a.film[(long long)b * a.ldf + c + e] : 0.f;
          sh[e] = a.film ? a.film[(long long)b * a.ldf + a.C + c + e] : 0.f;
        }
      }
#pragma unroll 4
      for (int p = p0 + tr; p < p1; p += rowT) {
        const long long row = (long long)b * a.HW + p;
        float xv[VEC], dv[VEC];
        osm::ldv<VEC>(a.x + row * a.ldx + c, xv);
        if (MODE == 1) osm::ldv<VEC>(a.dy + row * a.lddy + c, dv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          if (MODE == 0) {
            s1[j] += xv[e];
            s2[j] += xv[e] * xv[e];
            const float fx = (float)xv[e];
            imax = fmaxf(imax, fabsf(fx));
            if (fx != fx) inan = 0x7fc00000u;
          } else {
            float xh, z;
            gn_fwd_elem(xv[e], mean, rstd, ga[e], be[e], a.film != nullptr, sc[e], sh[e], xh, z);
            float dz = dv[e] * (a.silu ? osm::dsilu_f(z) : 1.0f);
            if (a.film) dz *= (1.0f + sc[e]);
            const float dxh = dz * ga[e];
            s1[j] += dxh;
            s2[j] += dxh * xh;
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NJMAX; ++j) {
    red[(tid * NJMAX + j) * 2] = s1[j];
    red[(tid * NJMAX + j) * 2 + 1] = s2[j];
  }
  __syncthreads();
  if (tid < a.G) {
    const int g = tid;
    const int v0 = g * a.gs / VEC, v1 = (g + 1) * a.gs / VEC;
    float t1 = 0.f, t2 = 0.f;
    for (int v = v0; v < v1; ++v) {
      const int c_t = v % colT, j = v / colT;
      for (int r = 0; r < rowT; ++r) {
        const int t = r * colT + c_t;
        t1 += red[(t * NJMAX + j) * 2];
        t2 += red[(t * NJMAX + j) * 2 + 1];
      }
    }
    float* o = a.part + (((long long)b * a.nchunk + chunk) * a.G + g) * 2;
    o[0] = t1;
    o[1] = t2;
  }
  if (MODE == 0 && a.maxabs_in) gn_publish_max(a.maxabs_in, b, chunk, gridDim.x, imax, inan);
}

// one wave per (b, g): lanes stride over the chunk partials, fp64 combine, shuffle reduce.
template <int MODE>
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ part,
                                                           float* __restrict__ out, int B, int G, int nchunk,
                                                           double n, float eps) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B * G) return;
  const int b = i / G, g = i % G;
  double s1 = 0.0, s2 = 0.0;
  for (int k = lane; k < nchunk; k += 64) {
    const float2 pp = *reinterpret_cast<const float2*>(part + (((long long)b * nchunk + k) * G + g) * 2);
    s1 += (double)pp.x;
    s2 += (double)pp.y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if (lane != 0) return;
  if (MODE == 0) {
    const double mean = s1 / n;
    double var = s2 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    out[i * 2] = (float)mean;
    out[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  } else {
    out[i * 2] = (float)(s1 / n);
    out[i * 2 + 1] = (float)(s2 / n);
  }
}

// Finalize (as gn_finalize_kernel<0>) and expand to the per-channel table a convolution applies while staging:
// table[b][0..3][c] = mean | rstd | gamma (1 + scale) | beta (1 + scale) + shift.   One wave per (b, g).
__global__ __launch_bounds__(256) void gn_finalize_table_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                 float* __restrict__ table, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ film,
                                                                 long long ldf, int B, int G, int C, int nchunk, double n,
                                                                 float eps) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= B * G) return;
  const int b = i / G, g = i % G;
  double s1 = 0.0, s2 = 0.0;
  for (int k = lane; k < nchunk; k += 64) {
    const float2 pp = *reinterpret_cast<const float2*>(part + (((long long)b * nchunk + k) * G + g) * 2);
    s1 += (double)pp.x;
    s2 += (double)pp.y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  const double mu = s1 / n;
  double var = s2 / n - mu * mu;
  if (var < 0.0) var = 0.0;
  const float mean = (float)mu, rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (lane == 0) {
    out[i * 2] = mean;
    out[i * 2 + 1] = rstd;
  }
  const int gs = C / G;
  float* t = table + (long long)b * 4 * C;
  for (int e = lane; e < gs; e += 64) {
    const int c = g * gs + e;
    const float sc = film ? film[(long long)b * ldf + c] : 0.f;
    const float sh = film ? film[(long long)b * ldf + C + c] : 0.f;
    t[c] = mean;
    t[C + c] = rstd;
    t[2 * C + c] = gamma[c] * (1.0f + sc);
    t[3 * C + c] = beta[c] * (1.0f + sc) + sh;
  }
}

// Statistics from the column sums a convolution wrote next to its output ([B][nchunk][2][C], igemm.hip): one WORKGROUP
// per (image, group) adds nchunk x (C / G) x 2 values (thread = chunk, the group's channels contiguous: full-line reads),
// fp64 from the wave level up.  MODE 0: (mean, rstd) (+ table);  MODE 1: (s1 / n, s2 / n).
#ifndef GN_FC_VEC
#define GN_FC_VEC 1      // measurement builds: 0 = the scalar column loop
#endif
template <int MODE>
__global__ __launch_bounds__(256) void gn_finalize_cols_kernel(const float* __restrict__ cs, float* __restrict__ out,
                                                                float* __restrict__ table, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ film,
                                                                long long ldf, int B, int G, int C, int nchunk, double n,
                                                                float eps) {
  __shared__ double red[2][4];
  __shared__ float bc[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x, b = blockIdx.y;
  const int gs = C / G;
  const float* base = cs + (long long)b * nchunk * 2 * C + g * gs;
  float f1 = 0.f, f2 = 0.f;
  if (GN_FC_VEC && (gs & 3) == 0 && (C & 3) == 0 && (reinterpret_cast<size_t>(cs) & 15) == 0) {
    // item = (chunk, 4-column vector): up to eight 16-byte loads of a thread in flight at once (as a scalar loop over the group's
    // columns every addition waited for its own load: gs dependent round trips, 7.9 us per launch against the ~5 us floor)
    const int vpg = gs >> 2, items = nchunk * vpg;
    for (int it0 = tid; it0 < items; it0 += 4 * 256) {
      float4 a1[4], a2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int it = it0 + u * 256;
        a1[u] = a2[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (it < items) {
          const int ch = it / vpg, v = it - ch * vpg;
          const float* p1 = base + (long long)ch * 2 * C + 4 * v;
          a1[u] = *reinterpret_cast<const float4*>(p1);
          a2[u] = *reinterpret_cast<const float4*>(p1 + C);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f1 += (a1[u].x + a1[u].y) + (a1[u].z + a1[u].w);
        f2 += (a2[u].x + a2[u].y) + (a2[u].z + a2[u].w);
      }
    }
  } else {
    for (int ch = tid; ch < nchunk; ch += 256) {
      const float* p1 = base + (long long)ch * 2 * C;
      for (int e = 0; e < gs; ++e) {
        f1 += p1[e];
        f2 += p1[C + e];
      }
    }
  }
  double s1 = (double)f1, s2 = (double)f2;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if (lane == 0) {
    red[0][wave] = s1;
    red[1][wave] = s2;
  }
  __syncthreads();
  if (tid == 0) {
    const double t1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const double t2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    float o0, o1;
    if (MODE == 1) {
      o0 = (float)(t1 / n);
      o1 = (float)(t2 / n);
    } else {
      const double mu = t1 / n;
      double var = t2 / n - mu * mu;
      if (var < 0.0) var = 0.0;
      o0 = (float)mu;
      o1 = (float)(1.0 / sqrt(var + (double)eps));
    }
    out[(b * G + g) * 2] = o0;
    out[(b * G + g) * 2 + 1] = o1;
    bc[0] = o0;
    bc[1] = o1;
  }
  if (MODE == 1 || !table) return;
  __syncthreads();
  const float mean = bc[0], rstd = bc[1];
  float* t = table + (long long)b * 4 * C;
  for (int e = tid; e < gs; e += 256) {
    const int c = g * gs + e;
    const float sc = film ? film[(long long)b * ldf + c] : 0.f;
    const float sh = film ? film[(long long)b * ldf + C + c] : 0.f;
    t[c] = mean;
    t[C + c] = rstd;
    t[2 * C + c] = gamma[c] * (1.0f + sc);
    t[3 * C + c] = beta[c] * (1.0f + sc) + sh;
  }
}

// MODE 0: y = act(GN(x)).   MODE 1: dx = dGN(dy) (+ addend)
// Same (chunk, image) grid and (column-vector, row) thread mapping as the reduce pass: a thread owns
// fixed channel vector(s), so gamma/beta/FiLM/statistics are loaded once and the row loop is pure
// streaming (no integer division in the hot loop).
template <int VEC, int MODE>
__global__ __launch_bounds__(256) void gn_apply_kernel(GNArgs a) {
  const int tid = threadIdx.x;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int vpr = a.C / VEC;
  const int colT = vpr < 256 ? vpr : 256;
  const int rowT = 256 / colT;
  const int nj = (vpr + 255) / 256;
  const int tc = tid % colT, tr = tid / colT;
  // fused finalize (nchunk <= GN_FUSE_CHUNKS): every workgroup combines the chunk partials of its image in
  // the same fixed order (fp64), so all of them see identical statistics; chunk 0 publishes them.
  __shared__ float sst[2 * 256];
  if (a.fuse) {
    const int lane = tid & 63;
    for (int g = tid >> 6; g < a.G; g += 4) {
      double s1 = 0.0, s2 = 0.0;
      for (int k = lane; k < a.nchunk; k += 64) {
        const float2 pp = *reinterpret_cast<const float2*>(a.part + (((long long)b * a.nchunk + k) * a.G + g) * 2);
        s1 += (double)pp.x;
        s2 += (double)pp.y;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
      }
      if (lane == 0) {
        float o0, o1;
        if (MODE == 0) {
          const double mu = s1 / a.n;
          double var = s2 / a.n - mu * mu;
          if (var < 0.0) var = 0.0;
          o0 = (float)mu;
          o1 = (float)(1.0 / sqrt(var + (double)a.eps));
        } else {
          o0 = (float)(s1 / a.n);
          o1 = (float)(s2 / a.n);
        }
        sst[2 * g] = o0;
        sst[2 * g + 1] = o1;
        if (chunk == 0) {
          a.fin[(b * a.G + g) * 2] = o0;
          a.fin[(b * a.G + g) * 2 + 1] = o1;
        }
      }
    }
    __syncthreads();
  }
  const bool live = tr < rowT;
  const int p0 = chunk * a.ppc;
  const int p1 = min(a.HW, p0 + a.ppc);
  const bool film = a.film != nullptr;
  float omax = 0.f;
  unsigned onan = 0u;
#pragma unroll
  for (int j = 0; j < NJMAX; ++j) {
    const int v = tc + colT * j;
    if (!live || j >= nj || v >= vpr) break;
    const int c = v * VEC;
    const int g = c / a.gs;
    const float mean = (MODE == 0 && a.fuse) ? sst[2 * g] : a.stats[(b * a.G + g) * 2];
    const float rstd = (MODE == 0 && a.fuse) ? sst[2 * g + 1] : a.stats[(b * a.G + g) * 2 + 1];
    float m1 = 0.f, m2 = 0.f;
    if (MODE == 1) {
      m1 = a.fuse ? sst[2 * g] : a.gstats[(b * a.G + g) * 2];
      m2 = a.fuse ? sst[2 * g + 1] : a.gstats[(b * a.G + g) * 2 + 1];
    }
    float ga[VEC], be[VEC], sc[VEC], sh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      ga[e] = a.gamma[c + e];
      be[e] = a.beta[c + e];
      sc[e] = film ? a.film[(long long)b * a.ldf + c + e] : 0.f;
      sh[e] = film ? a.film[(long long)b * a.ldf + a.C + c + e] : 0.f;
    }
    const act_t* xp = a.x + ((long long)b * a.HW + p0 + tr) * a.ldx + c;
    const act_t* dp = MODE == 1 ? a.dy + ((long long)b * a.HW + p0 + tr) * a.lddy + c : nullptr;
    const act_t* ap = (MODE == 1 && a.addend) ? a.addend + ((long long)b * a.HW + p0 + tr) * a.ldadd + c : nullptr;
    const act_t* ap2 = (MODE == 1 && a.addend2) ? a.addend2 + ((long long)b * a.HW + p0 + tr) * a.ldadd2 + c : nullptr;
    act_t* op = a.out + ((long long)b * a.HW + p0 + tr) * a.ldo + c;
    const long long sx = (long long)rowT * a.ldx, sd = (long long)rowT * a.lddy, sa = (long long)rowT * a.ldadd,
                    sa2 = (long long)rowT * a.ldadd2, so = (long long)rowT * a.ldo;
#pragma unroll 4
    for (int p = p0 + tr; p < p1; p += rowT) {
      float xv[VEC], dv[VEC], ov[VEC], av[VEC];
      osm::ldv<VEC>(xp, xv);
      if (MODE == 1) {
        osm::ldv<VEC>(dp, dv);
#pragma unroll
        for (int e = 0; e < VEC; ++e) av[e] = 0.f;
        if (ap) osm::ldv<VEC>(ap, av);
        if (ap2) {
          float a2[VEC];
          osm::ldv<VEC>(ap2, a2);
#pragma unroll
          for (int e = 0; e < VEC; ++e) av[e] += a2[e];
        }
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float xh, z;
        gn_fwd_elem(xv[e], mean, rstd, ga[e], be[e], film, sc[e], sh[e], xh, z);
        if (MODE == 0) {
          ov[e] = a.silu ? osm::silu_f(z) : z;
        } else {
          float dz = dv[e] * (a.silu ? osm::dsilu_f(z) : 1.0f);
          if (film) dz *= (1.0f + sc[e]);
          const float dxh = dz * ga[e];
          float r = rstd * (dxh - m1 - xh * m2);
          if (ap || ap2) r += av[e];
          ov[e] = r;
        }
        if (a.maxabs) {      // of the value as stored
          const float sv = (float)(act_t)ov[e];
          omax = fmaxf(omax, fabsf(sv));
          if (sv != sv) onan = 0x7fc00000u;
        }
      }
      osm::stv<VEC>(op, ov);
      xp += sx;
      op += so;
      if (MODE == 1) {
        dp += sd;
        if (ap) ap += sa;
        if (ap2) ap2 += sa2;
      }
    }
  }
  if (a.maxabs) gn_publish_max(a.maxabs, b, chunk, gridDim.x, omax, onan);
}

// Low-resolution tensors (HW <= 256: 8x8 and 16x16) are a few hundred KB: three launches (reduce, finalize,
// apply) were pure launch latency (~17 us forward, ~18 us backward per GroupNorm, ~100 GroupNorms per step).
// One workgroup per (image, group) does both passes in ONE launch; the second pass re-reads its 8-64 KB
// slice from L2.   MODE 0: stats -> a.fin, y = act(GN(x)).   MODE 1: (m1, m2) -> a.fin, dx = dGN(dy) (+ addend).
template <int MODE>
__global__ __launch_bounds__(256) void gn_small_kernel(GNArgs a) {
  __shared__ double red[2][4];
  __shared__ float bc[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x, b = blockIdx.y;
  const int vpg = a.gs >> 2;              // float4 vectors per pixel inside the group
  const int ppi = 256 / vpg;              // pixels per sweep
  const int tv = tid % vpg, tp = tid / vpg;
  const bool live = tp < ppi;
  const int c = g * a.gs + 4 * tv;
  const bool film = a.film != nullptr;
  float ga[4], be[4], sc[4], sh[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ga[e] = a.gamma[c + e];
    be[e] = a.beta[c + e];
    sc[e] = film ? a.film[(long long)b * a.ldf + c + e] : 0.f;
    sh[e] = film ? a.film[(long long)b * a.ldf + a.C + c + e] : 0.f;
  }
  float mean = 0.f, rstd = 0.f;
  if (MODE == 1) {
    mean = a.stats[(b * a.G + g) * 2];
    rstd = a.stats[(b * a.G + g) * 2 + 1];
  }
  const long long row0 = (long long)b * a.HW;
  float s1 = 0.f, s2 = 0.f;
  if (live) {
    for (int p = tp; p < a.HW; p += ppi) {
      const float4 t = osm::ld4(a.x + (row0 + p) * a.ldx + c);
      const float xv[4] = {t.x, t.y, t.z, t.w};
      if (MODE == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1 += xv[e];
          s2 += xv[e] * xv[e];
        }
      } else {
        const float4 u = osm::ld4(a.dy + (row0 + p) * a.lddy + c);
        const float dv[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float xh, z;
          gn_fwd_elem(xv[e], mean, rstd, ga[e], be[e], film, sc[e], sh[e], xh, z);
          float dz = dv[e] * (a.silu ? osm::dsilu_f(z) : 1.0f);
          if (film) dz *= (1.0f + sc[e]);
          const float dxh = dz * ga[e];
          s1 += dxh;
          s2 += dxh * xh;
        }
      }
    }
  }
  double d1 = (double)s1, d2 = (double)s2;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    d1 += __shfl_xor(d1, o, 64);
    d2 += __shfl_xor(d2, o, 64);
  }
  if (lane == 0) {
    red[0][wave] = d1;
    red[1][wave] = d2;
  }
  __syncthreads();
  if (tid == 0) {
    const double t1 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    const double t2 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    float o0, o1;
    if (MODE == 0) {
      const double mu = t1 / a.n;
      double var = t2 / a.n - mu * mu;
      if (var < 0.0) var = 0.0;
      o0 = (float)mu;
      o1 = (float)(1.0 / sqrt(var + (double)a.eps));
    } else {
      o0 = (float)(t1 / a.n);
      o1 = (float)(t2 / a.n);
    }
    a.fin[(b * a.G + g) * 2] = o0;
    a.fin[(b * a.G + g) * 2 + 1] = o1;
    bc[0] = o0;
    bc[1] = o1;
  }
  __syncthreads();
  if (!a.out) return;
  const float q0 = bc[0], q1 = bc[1];
  if (MODE == 0) {
    mean = q0;
    rstd = q1;
  }
  float omax = 0.f;
  unsigned onan = 0u;
  for (int p = tp; live && p < a.HW; p += ppi) {
    const float4 t = osm::ld4(a.x + (row0 + p) * a.ldx + c);
    const float xv[4] = {t.x, t.y, t.z, t.w};
    float ov[4];
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float xh, z;
        gn_fwd_elem(xv[e], mean, rstd, ga[e], be[e], film, sc[e], sh[e], xh, z);
        ov[e] = a.silu ? osm::silu_f(z) : z;
      }
    } else {
      const float4 u = osm::ld4(a.dy + (row0 + p) * a.lddy + c);
      const float dv[4] = {u.x, u.y, u.z, u.w};
      float av[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.addend) {
        const float4 w = osm::ld4(a.addend + (row0 + p) * a.ldadd + c);
        av[0] = w.x; av[1] = w.y; av[2] = w.z; av[3] = w.w;
      }
      if (a.addend2) {
        const float4 w = osm::ld4(a.addend2 + (row0 + p) * a.ldadd2 + c);
        av[0] += w.x; av[1] += w.y; av[2] += w.z; av[3] += w.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float xh, z;
        gn_fwd_elem(xv[e], mean, rstd, ga[e], be[e], film, sc[e], sh[e], xh, z);
        float dz = dv[e] * (a.silu ? osm::dsilu_f(z) : 1.0f);
        if (film) dz *= (1.0f + sc[e]);
        const float dxh = dz * ga[e];
        ov[e] = rstd * (dxh - q0 - xh * q1) + av[e];
      }
    }
    if (a.maxabs) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sv = (float)(act_t)ov[e];
        omax = fmaxf(omax, fabsf(sv));
        if (sv != sv) onan = 0x7fc00000u;
      }
    }
    osm::st4(a.out + (row0 + p) * a.ldo + c, make_float4(ov[0], ov[1], ov[2], ov[3]));
  }
  if (a.maxabs) gn_publish_max(a.maxabs, b, g, gridDim.x, omax, onan);
}

// Mid-size tensors (32 x 32 with 512 / 1024 channels, 64 x 64 with 256): three launches of the chunked path (reduce, finalize,
// apply: 15-18 us, ~36 of them per step) are launch latency, and the two-pass one-launch kernel above walks 16 dependent sweeps.
// Here one 512-thread workgroup per (image, group) loads its WHOLE slice -- <= 32768 elements = NV <= 16 float4 per thread, and dy
// as well in the backward -- into REGISTERS with every load in flight at once, reduces, and applies from the registers: one
// launch, one read of each operand.  512 % (gs / 4) == 0, so a thread's channel vector (gamma, beta, FiLM) is fixed.
// MODE 0: stats -> a.fin, y = act(GN(x)) (+ max |x| -> a.maxabs_in).   MODE 1: (m1, m2) -> a.fin, dx = dGN(dy) (+ addends).
template <int MODE, int NV>
__global__ __launch_bounds__(512) void gn_reg_kernel(GNArgs a) {
  __shared__ double red[2][8];
  __shared__ float bc[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x, b = blockIdx.y;
  const int vpg = a.gs >> 2;
  const int tv = tid % vpg, p0 = tid / vpg, pstep = 512 / vpg;
  const int c = g * a.gs + 4 * tv;
  const bool film = a.film != nullptr;
  const long long row0 = (long long)b * a.HW;
  // (backward: the addends are requested here as well -- after the two barriers of the reduction they were one more dependent
  // memory round trip of a ~7-16 us kernel; an addend that aliases `out` is read by the thread that later writes the element)
  float4 xr[NV], dr[MODE == 1 ? NV : 1], ar[MODE == 1 ? NV : 1];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int p = p0 + k * pstep;
    const bool live = p < a.HW;
    const long long row = row0 + (live ? p : 0);
    const float4 t = osm::ld4(a.x + row * a.ldx + c);
    xr[k] = live ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 1) {
      const float4 u = osm::ld4(a.dy + row * a.lddy + c);
      dr[k] = live ? u : make_float4(0.f, 0.f, 0.f, 0.f);
      ar[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a.addend) ar[k] = osm::ld4(a.addend + row * a.ldadd + c);
      if (a.addend2) {
        const float4 w = osm::ld4(a.addend2 + row * a.ldadd2 + c);
        ar[k].x += w.x; ar[k].y += w.y; ar[k].z += w.z; ar[k].w += w.w;
      }
    }
  }
  float ga[4], be[4], sc[4], sh[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    ga[e] = a.gamma[c + e];
    be[e] = a.beta[c + e];
    sc[e] = film ? a.film[(long long)b * a.ldf + c + e] : 0.f;
    sh[e] = film ? a.film[(long long)b * a.ldf + a.C + c + e] : 0.f;
  }
  float mean = 0.f, rstd = 0.f;
  if (MODE == 1) {
    mean = a.stats[(b * a.G + g) * 2];
    rstd = a.stats[(b * a.G + g) * 2 + 1];
  }
  float s1 = 0.f, s2 = 0.f, imax = 0.f;
  unsigned inan = 0u;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const float xv[4] = {xr[k].x, xr[k].y, xr[k].z, xr[k].w};
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1 += xv[e];
        s2 += xv[e] * xv[e];
        imax = fmaxf(imax, fabsf(xv[e]));
        if (xv[e] != xv[e]) inan = 0x7fc00000u;
      }
    } else {
      const float dv[4] = {dr[k].x, dr[k].y, dr[k].z, dr[k].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float xh, z;
        gn_fwd_elem(xv[e], mean, rstd, ga[e], be[e], film, sc[e], sh[e], xh, z);
        float dz = dv[e] * (a.silu ? osm::dsilu_f(z) : 1.0f);
        if (film) dz *= (1.0f + sc[e]);
        const float dxh = dz * ga[e];
        s1 += dxh;
        s2 += dxh * xh;
        if (e == 0) dr[k].x = dxh; else if (e == 1) dr[k].y = dxh; else if (e == 2) dr[k].z = dxh; else dr[k].w = dxh;   // keep dxh
      }
    }
  }
  double d1 = (double)s1, d2 = (double)s2;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    d1 += __shfl_xor(d1, o, 64);
    d2 += __shfl_xor(d2, o, 64);
  }
  if (lane == 0) {
    red[0][wave] = d1;
    red[1][wave] = d2;
  }
  __syncthreads();
  if (tid == 0) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      t1 += red[0][w];
      t2 += red[1][w];
    }
    float o0, o1;
    if (MODE == 0) {
      const double mu = t1 / a.n;
      double var = t2 / a.n - mu * mu;
      if (var < 0.0) var = 0.0;
      o0 = (float)mu;
      o1 = (float)(1.0 / sqrt(var + (double)a.eps));
    } else {
      o0 = (float)(t1 / a.n);
      o1 = (float)(t2 / a.n);
    }
    a.fin[(b * a.G + g) * 2] = o0;
    a.fin[(b * a.G + g) * 2 + 1] = o1;
    bc[0] = o0;
    bc[1] = o1;
  }
  __syncthreads();
  const float q0 = bc[0], q1 = bc[1];
  if (MODE == 0) {
    mean = q0;
    rstd = q1;
  }
  float omax = 0.f;
  unsigned onan = 0u;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int p = p0 + k * pstep;
    if (p >= a.HW) continue;
    const float xv[4] = {xr[k].x, xr[k].y, xr[k].z, xr[k].w};
    float ov[4];
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float xh, z;
        gn_fwd_elem(xv[e], mean, rstd, ga[e], be[e], film, sc[e], sh[e], xh, z);
        ov[e] = a.silu ? osm::silu_f(z) : z;
      }
    } else {
      const float dxv[4] = {dr[k].x, dr[k].y, dr[k].z, dr[k].w};      // dxh of the first sweep
      const float av[4] = {ar[k].x, ar[k].y, ar[k].z, ar[k].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xv[e] - mean) * rstd;
        ov[e] = rstd * (dxv[e] - q0 - xh * q1) + av[e];
      }
    }
    if (a.maxabs) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float sv = (float)(act_t)ov[e];
        omax = fmaxf(omax, fabsf(sv));
        if (sv != sv) onan = 0x7fc00000u;
      }
    }
    osm::st4(a.out + (row0 + p) * a.ldo + c, make_float4(ov[0], ov[1], ov[2], ov[3]));
  }
  if (a.maxabs) gn_publish_max(a.maxabs, b, g, gridDim.x, omax, onan);
  if (MODE == 0 && a.maxabs_in) {
    __syncthreads();            // gn_publish_max's staging words are reused
    gn_publish_max(a.maxabs_in, b, g, gridDim.x, imax, inan);
  }
}


constexpr int GN_SMALL_HW = 256;   // measured: 32 x 32 tensors are faster on the chunked three-launch path

bool use_vec4(const GNArgs& a) {
  return a.gs % 4 == 0 && a.ldx % 4 == 0 && osm::aligned_act4(a.x) &&
         (!a.dy || (a.lddy % 4 == 0 && osm::aligned_act4(a.dy))) &&
         (!a.out || (a.ldo % 4 == 0 && osm::aligned_act4(a.out))) &&
         (!a.addend || (a.ldadd % 4 == 0 && osm::aligned_act4(a.addend))) &&
         (!a.addend2 || (a.ldadd2 % 4 == 0 && osm::aligned_act4(a.addend2)));
}

int check_common(const GNArgs& a, const char* who) {
  OSM_REQUIRE(a.B > 0 && a.HW > 0 && a.C > 0 && a.G > 0, "%s: bad shape", who);
  OSM_REQUIRE(a.C % a.G == 0, "%s: C (%d) not divisible by G (%d)", who, a.C, a.G);
  OSM_REQUIRE(a.G <= 256, "%s: G must be <= 256", who);
  OSM_REQUIRE(a.C <= 256 * NJMAX * 4, "%s: C too large", who);
  return OSM_OK;
}

// (round 2: 0 = never.  Re-combining the chunk partials in every apply workgroup cost more than the 6 us finalize launch it
// saved: 4.39 -> 4.25 ms of GroupNorm per step; 128 / 256 / 512 / 1024 measured in that order of increasing cost)
#ifndef GN_FUSE_CHUNKS_V
#define GN_FUSE_CHUNKS_V 0
#endif
constexpr int GN_FUSE_CHUNKS = GN_FUSE_CHUNKS_V;   // apply kernels re-combine up to this many chunk partials themselves

template <int MODE>
int run_reduce(GNArgs& a, float* finalized, hipStream_t st, bool finalize = true) {
  a.ppc = gn_ppc(a.HW);
  a.nchunk = (a.HW + a.ppc - 1) / a.ppc;
  a.gs = a.C / a.G;
  dim3 grid(a.nchunk, a.B);
  const bool v4 = use_vec4(a);
  if (!v4) OSM_REQUIRE(a.C <= 256 * NJMAX, "GroupNorm scalar path: C too large");
  if (v4)
    hipLaunchKernelGGL((gn_reduce_kernel<4, MODE>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((gn_reduce_kernel<1, MODE>), grid, dim3(256), 0, st, a);
  int rc = osm::check_launch("gn_reduce_kernel");
  if (rc || !finalize) return rc;
  const int n = a.B * a.G;
  hipLaunchKernelGGL((gn_finalize_kernel<MODE>), dim3((n + 3) / 4), dim3(256), 0, st, a.part, finalized, a.B,
                     a.G, a.nchunk, (double)a.HW * a.gs, a.eps);
  return osm::check_launch("gn_finalize_kernel");
}

template <int MODE>
int run_apply(GNArgs& a, hipStream_t st) {
  a.gs = a.C / a.G;
  a.n = (double)a.HW * a.gs;
  a.ppc = gn_ppc(a.HW);
  a.nchunk = (a.HW + a.ppc - 1) / a.ppc;
  const bool v4 = use_vec4(a);
  if (!v4) OSM_REQUIRE(a.C <= 256 * NJMAX, "GroupNorm scalar path: C too large");
  dim3 grid(a.nchunk, a.B);
  if (v4)
    hipLaunchKernelGGL((gn_apply_kernel<4, MODE>), grid, dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((gn_apply_kernel<1, MODE>), grid, dim3(256), 0, st, a);
  return osm::check_launch("gn_apply_kernel");
}

bool small_path(const GNArgs& a) {
  return a.HW <= GN_SMALL_HW && use_vec4(a) && (a.C / a.G) % 4 == 0 && (a.C / a.G) <= 1024;
}

// register-resident one-launch kernel: (HW, gs) whose group slice fits 512 threads x 16 float4 and whose channel vector is fixed
// per thread; the chunked path keeps everything else (OSM_GN_REG=0: off)
bool reg_path(const GNArgs& a, int mode) {
  static const bool on = [] { const char* e = std::getenv("OSM_GN_REG"); return !(e && e[0] == '0'); }();
  const int gs = a.C / a.G;
  static const bool small_too = [] { const char* e = std::getenv("OSM_GN_REG_SMALL"); return !(e && e[0] == '0'); }();
  if (!(on && a.out && (small_too || a.HW > GN_SMALL_HW) && use_vec4(a) && gs % 4 == 0)) return false;
  const int vpg = gs / 4;
  // (the backward holds x and dy: 16 vectors of each spill, 8 do not)
  return 512 % vpg == 0 && (long long)a.HW * vpg <= 512LL * (mode == 1 ? 8 : 16) && a.G <= OSM_MAXABS_PARTS;
}

template <int MODE>
int run_reg(GNArgs& a, float* finalized, hipStream_t st) {
  a.gs = a.C / a.G;
  a.fin = finalized;
  a.n = (double)a.HW * a.gs;
  const long long items = (long long)a.HW * (a.gs / 4);
  const dim3 grid(a.G, a.B);
  if (items <= 512 * 4) hipLaunchKernelGGL((gn_reg_kernel<MODE, 4>), grid, dim3(512), 0, st, a);
  else if (items <= 512 * 8 || MODE == 1) hipLaunchKernelGGL((gn_reg_kernel<MODE, 8>), grid, dim3(512), 0, st, a);
  else hipLaunchKernelGGL((gn_reg_kernel<MODE, (MODE == 1 ? 8 : 16)>), grid, dim3(512), 0, st, a);
  return osm::check_launch("gn_reg_kernel");
}

template <int MODE>
int run_small(GNArgs& a, float* finalized, hipStream_t st) {
  a.gs = a.C / a.G;
  a.fin = finalized;
  a.n = (double)a.HW * a.gs;
  hipLaunchKernelGGL((gn_small_kernel<MODE>), dim3(a.G, a.B), dim3(256), 0, st, a);
  return osm::check_launch("gn_small_kernel");
}

}  // namespace

#ifndef OSM_ACT_F16
extern "C" int osm_gn_nchunk(int HW) { return (HW + gn_ppc(HW) - 1) / gn_ppc(HW); }
#endif
static int gn_nchunk_of(int HW) { return (HW + gn_ppc(HW) - 1) / gn_ppc(HW); }
// maxabs_out of the apply entry points: [B][OSM_MAXABS_PARTS] partial max |out| (fp32 family; the grid must fit the slots)
static int set_maxabs_in(GNArgs& a, float* maxabs_in, const char* who) {
  if (!maxabs_in) return OSM_OK;
  OSM_REQUIRE(!OSM_ACT_IS_F16, "%s: maxabs_in belongs to the fp32 family", who);
  OSM_REQUIRE(gn_nchunk_of(a.HW) <= OSM_MAXABS_PARTS && a.HW > GN_SMALL_HW,
              "%s: maxabs_in needs the chunked statistics pass (HW > %d) and osm_gn_nchunk(HW) <= OSM_MAXABS_PARTS", who, GN_SMALL_HW);
  a.maxabs_in = reinterpret_cast<unsigned*>(maxabs_in);
  return OSM_OK;
}
static int set_maxabs(GNArgs& a, float* maxabs_out, const char* who) {
  if (!maxabs_out) return OSM_OK;
  OSM_REQUIRE(!OSM_ACT_IS_F16, "%s: maxabs_out belongs to the fp32 family", who);
  OSM_REQUIRE(gn_nchunk_of(a.HW) <= OSM_MAXABS_PARTS && a.G <= OSM_MAXABS_PARTS,
              "%s: maxabs_out needs osm_gn_nchunk(HW) <= OSM_MAXABS_PARTS (use osm_maxabs)", who);
  a.maxabs = reinterpret_cast<unsigned*>(maxabs_out);
  return OSM_OK;
}

extern "C" int OSM_FN(osm_gn_stats)(const abi_act_t* x, long long ldx, int B, int HW, int C, int G, float eps,
                            float* part, float* stats, void* stream) {
  OSM_REQUIRE(x && part && stats, "osm_gn_stats: null pointer");
  GNArgs a{};
  a.x = OSM_CACT(x); a.ldx = ldx; a.B = B; a.HW = HW; a.C = C; a.G = G; a.eps = eps; a.part = part;
  int rc = check_common(a, "osm_gn_stats");
  if (rc) return rc;
  return run_reduce<0>(a, stats, (hipStream_t)stream);
}

extern "C" int OSM_FN(osm_gn_prep)(const abi_act_t* x, long long ldx, int B, int HW, int C, int G, float eps, float* part,
                           float* stats, const float* gamma, const float* beta, const float* film,
                           long long ldfilm, float* table, float* maxabs_in, void* stream) {
  OSM_REQUIRE(x && part && stats && gamma && beta && table, "osm_gn_prep: null pointer");
  OSM_REQUIRE(!film || ldfilm >= 2LL * C, "osm_gn_prep: ldfilm smaller than 2*C");
  GNArgs a{};
  a.x = OSM_CACT(x); a.ldx = ldx; a.B = B; a.HW = HW; a.C = C; a.G = G; a.eps = eps; a.part = part;
  int rc = check_common(a, "osm_gn_prep");
  if (rc) return rc;
  if ((rc = set_maxabs_in(a, maxabs_in, "osm_gn_prep"))) return rc;
  rc = run_reduce<0>(a, stats, (hipStream_t)stream, false);
  if (rc) return rc;
  const int n = B * G;
  hipLaunchKernelGGL(gn_finalize_table_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, part, stats, table,
                     gamma, beta, film, ldfilm, B, G, C, a.nchunk, (double)HW * (C / G), eps);
  return osm::check_launch("gn_finalize_table_kernel");
}

extern "C" int OSM_FN(osm_gn_apply)(const abi_act_t* x, long long ldx, abi_act_t* y, long long ldy, int B, int HW, int C,
                            int G, const float* stats, const float* gamma, const float* beta,
                            const float* film, long long ldfilm, int silu, float* maxabs_out, void* stream) {
  OSM_REQUIRE(x && y && stats && gamma && beta, "osm_gn_apply: null pointer");
  OSM_REQUIRE(!film || ldfilm >= 2LL * C, "osm_gn_apply: ldfilm smaller than 2*C");
  GNArgs a{};
  a.x = OSM_CACT(x); a.ldx = ldx; a.out = OSM_ACT(y); a.ldo = ldy; a.B = B; a.HW = HW; a.C = C; a.G = G;
  a.stats = stats; a.gamma = gamma; a.beta = beta; a.film = film; a.ldf = ldfilm; a.silu = silu;
  int rc = check_common(a, "osm_gn_apply");
  if (rc) return rc;
  if ((rc = set_maxabs(a, maxabs_out, "osm_gn_apply"))) return rc;
  return run_apply<0>(a, (hipStream_t)stream);
}

#ifndef OSM_ACT_F16
extern "C" int osm_gn_finalize_cols(const float* colsum, int nchunk, int B, int HW, int C, int G, float eps, int mode,
                                    float* stats, const float* gamma, const float* beta, const float* film,
                                    long long ldfilm, float* table, void* stream) {
  OSM_REQUIRE(colsum && stats && nchunk > 0 && B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0,
              "osm_gn_finalize_cols: bad argument");
  OSM_REQUIRE(mode == 0 || mode == 1, "osm_gn_finalize_cols: mode must be 0 or 1");
  OSM_REQUIRE(!table || (mode == 0 && gamma && beta), "osm_gn_finalize_cols: the table needs mode 0, gamma and beta");
  OSM_REQUIRE(!film || ldfilm >= 2LL * C, "osm_gn_finalize_cols: ldfilm smaller than 2*C");
  const double cnt = (double)HW * (C / G);
  if (mode == 0)
    hipLaunchKernelGGL((gn_finalize_cols_kernel<0>), dim3(G, B), dim3(256), 0, (hipStream_t)stream, colsum, stats,
                       table, gamma, beta, film, ldfilm, B, G, C, nchunk, cnt, eps);
  else
    hipLaunchKernelGGL((gn_finalize_cols_kernel<1>), dim3(G, B), dim3(256), 0, (hipStream_t)stream, colsum, stats,
                       (float*)nullptr, gamma, beta, film, ldfilm, B, G, C, nchunk, cnt, eps);
  return osm::check_launch("gn_finalize_cols_kernel");
}
#endif

extern "C" int OSM_FN(osm_gn_bwd_apply)(const abi_act_t* x, long long ldx, const abi_act_t* dy, long long lddy,
                                        abi_act_t* dx, long long lddx, const abi_act_t* addend, long long ldadd,
                                        const abi_act_t* addend2, long long ldadd2, int B,
                                        int HW, int C, int G, const float* stats, const float* gstats, const float* gamma,
                                        const float* beta, const float* film, long long ldfilm, int silu,
                                        float* maxabs_out, void* stream) {
  OSM_REQUIRE(x && dy && dx && stats && gstats && gamma && beta, "osm_gn_bwd_apply: null pointer");
  OSM_REQUIRE(!film || ldfilm >= 2LL * C, "osm_gn_bwd_apply: ldfilm smaller than 2*C");
  GNArgs a{};
  a.x = OSM_CACT(x); a.ldx = ldx;
  a.dy = OSM_CACT(dy); a.lddy = lddy; a.out = OSM_ACT(dx); a.ldo = lddx; a.addend = OSM_CACT(addend); a.ldadd = ldadd;
  a.addend2 = OSM_CACT(addend2); a.ldadd2 = ldadd2;
  a.B = B; a.HW = HW; a.C = C; a.G = G; a.stats = stats; a.gstats = gstats; a.gamma = gamma; a.beta = beta;
  a.film = film; a.ldf = ldfilm; a.silu = silu;
  int rc = check_common(a, "osm_gn_bwd_apply");
  if (rc) return rc;
  if ((rc = set_maxabs(a, maxabs_out, "osm_gn_bwd_apply"))) return rc;
  a.fuse = 0;
  return run_apply<1>(a, (hipStream_t)stream);
}

extern "C" int OSM_FN(osm_gn_bwd)(const abi_act_t* x, long long ldx, const abi_act_t* dy, long long lddy, abi_act_t* dx,
                          long long lddx, const abi_act_t* addend, long long ldadd, const abi_act_t* addend2, long long ldadd2,
                          int B, int HW, int C, int G,
                          const float* stats, const float* gamma, const float* beta, const float* film,
                          long long ldfilm, int silu, float* part, float* gstats, float* maxabs_out, void* stream) {
  OSM_REQUIRE(x && dy && dx && stats && gamma && beta && part && gstats, "osm_gn_bwd: null pointer");
  OSM_REQUIRE(!film || ldfilm >= 2LL * C, "osm_gn_bwd: ldfilm smaller than 2*C");
  GNArgs a{};
  a.x = OSM_CACT(x); a.ldx = ldx; a.dy = OSM_CACT(dy); a.lddy = lddy; a.out = OSM_ACT(dx); a.ldo = lddx; a.addend = OSM_CACT(addend); a.ldadd = ldadd;
  a.addend2 = OSM_CACT(addend2); a.ldadd2 = ldadd2;
  a.B = B; a.HW = HW; a.C = C; a.G = G; a.stats = stats; a.gstats = gstats; a.gamma = gamma; a.beta = beta;
  a.film = film; a.ldf = ldfilm; a.silu = silu; a.part = part;
  int rc = check_common(a, "osm_gn_bwd");
  if (rc) return rc;
  if ((rc = set_maxabs(a, maxabs_out, "osm_gn_bwd"))) return rc;
  if (reg_path(a, 1)) return run_reg<1>(a, gstats, (hipStream_t)stream);
  if (small_path(a)) return run_small<1>(a, gstats, (hipStream_t)stream);
  a.fuse = gn_nchunk_of(HW) <= GN_FUSE_CHUNKS;
  a.fin = gstats;
  rc = run_reduce<1>(a, gstats, (hipStream_t)stream, !a.fuse);
  if (rc) return rc;
  return run_apply<1>(a, (hipStream_t)stream);
}

extern "C" int OSM_FN(osm_gn_fwd)(const abi_act_t* x, long long ldx, abi_act_t* y, long long ldy, int B, int HW, int C, int G,
                          float eps, float* part, float* stats, const float* gamma, const float* beta,
                          const float* film, long long ldfilm, int silu, float* maxabs_out, float* maxabs_in, void* stream) {
  OSM_REQUIRE(x && y && part && stats && gamma && beta, "osm_gn_fwd: null pointer");
  OSM_REQUIRE(!film || ldfilm >= 2LL * C, "osm_gn_fwd: ldfilm smaller than 2*C");
  GNArgs a{};
  a.x = OSM_CACT(x); a.ldx = ldx; a.out = OSM_ACT(y); a.ldo = ldy; a.B = B; a.HW = HW; a.C = C; a.G = G; a.eps = eps; a.part = part;
  a.stats = stats; a.gamma = gamma; a.beta = beta; a.film = film; a.ldf = ldfilm; a.silu = silu;
  int rc = check_common(a, "osm_gn_fwd");
  if (rc) return rc;
  if ((rc = set_maxabs(a, maxabs_out, "osm_gn_fwd"))) return rc;
  if ((rc = set_maxabs_in(a, maxabs_in, "osm_gn_fwd"))) return rc;
  if (reg_path(a, 0)) return run_reg<0>(a, stats, (hipStream_t)stream);
  if (small_path(a)) return run_small<0>(a, stats, (hipStream_t)stream);
  a.fuse = gn_nchunk_of(HW) <= GN_FUSE_CHUNKS;
  a.fin = stats;
  rc = run_reduce<0>(a, stats, (hipStream_t)stream, !a.fuse);
  if (rc) return rc;
  return run_apply<0>(a, (hipStream_t)stream);
}
