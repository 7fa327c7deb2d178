// Implicit-GEMM convolution / batched GEMM for gfx950 (MI355X), exact fp32 on the matrix cores.
//
//   out[m][n] = alpha * sum_{tap} sum_{c} A[pix(m)+off(tap)][c] * Bp[tap][n][c]  (+bias[n] +res[m][n])
//
// m = NHWC pixel row, n = output channel, (tap,c) = reduction.  One kernel family covers the
// reference's 3x3 convs, 1x1 convs / conv1d projections, the attention einsums and -- with the
// flipped/transposed weight packing -- every data-gradient of those (guidance back-propagates
// through the whole UNet: condition_methods.py:186-194).
//
// CDNA4 mapping
//   * v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles/instr/SIMD, 157 TFLOP/s chip peak).
//   * 256 threads = 4 waves (one per SIMD), 2x2 waves over a 128x128 output tile, each wave 2x2
//     MFMA tiles of 32x32 -> 64 accumulator VGPRs; 2 workgroups per CU so one workgroup's
//     global->LDS staging overlaps the other's MFMA stream.
//   * K is consumed in chunks of 32 channels of one tap; chunk order is channel-major / tap-minor so
//     the 9 taps of a channel slab re-hit the XCD's L2.  Tiles are staged global->VGPR->LDS with
//     16-byte accesses, double-buffered in LDS (one barrier per chunk), rows padded to 36 floats so
//     both the ds_write_b128 staging stores and the ds_read_b128 fragment loads are conflict-free.
//   * Each lane feeds the MFMA from a float4 of 4 consecutive k; lanes 0-31 / 32-63 hold k-offsets
//     0-3 / 4-7 of every 8-wide k group, so step j contracts k={j, j+4}: A and B use the same
//     permutation, hence the sum is unchanged and every LDS fragment read is a single b128.
//   * blockIdx -> tile mapping is XCD-aware (bijective remap, N-tiles fastest) so workgroups that
//     share an input slab share an L2.
//   * Small-M layers (8x8..32x32) are weight-bandwidth bound: split-K over grid.y with fp32
//     partials and a deterministic reduce.
#include "osm_common.h"
#include "mfma_split.h"
#include <cstdlib>
#include <type_traits>

namespace {


constexpr int BM = 128, BN = 128, BK = 32;
[[maybe_unused]] constexpr int LDS_STRIDE = 36;      // floats per staged row (32 + 4 pad)
[[maybe_unused]] constexpr int LDS_KN_STRIDE = 132;  // floats per k-row of a [k][n] staged B tile

struct IGemmParams {
  const act_t* A;      // activations: fp32, or half in the OSM_ACT_F16 family
  const float* Bm;     // weights (fp32 image, or a split-bf16 / fp16 fragment image behind the same pointer)
  const float* bias;
  const act_t* res;
  act_t* C;
  float* ws;           // split-K partials are fp32 in both families
  int M, N, K;
  int H, W;
  int splitk;
  int accumulate;
  float alpha;
  long long lda, ldb, ldc, ldr;
  long long tapstrideB;
  long long planestrideB;
  int nt32, ksteps;     // split-bf16 fragment image: 32-column tiles, k16 steps per tap
  int nb1;
  long long sA1, sB1, sC1, sA2, sB2, sC2;
  int mtiles, ntiles;
  int nchunks;
  int nbatch;
  const float* gn_table;   // halo kernel: fused GroupNorm(+FiLM)(+SiLU) of the input, [B][4][K]
  int gn_silu;
  // optional per-column sums of the OUTPUT tensor, written by the epilogue that holds the final values (halo kernel, or
  // the split-K reduce): colsum[image][chunk][2][N].  stat_mode 1: (sum y, sum y^2) -> the statistics of the GroupNorm
  // that reads y next;  2: y is the gradient w.r.t. a GroupNorm output act(xh g + b), xh = (x - mean) rstd, and the sums
  // are (sum dxh, sum dxh xh), dxh = y act'(z) g -> the two reductions of that GroupNorm's backward (x = stat_x).
  float* colsum;
  int stat_mode, stat_silu, stat_chunks;
  const act_t* stat_x;
  long long ld_sx;
  const float* stat_table;   // mode 2: [B][4][N] mean | rstd | g | b (the table the forward convolution applied)
  const float* xmax;         // f16x3 Winograd image: per-image max |x| of the input (osm_maxabs), [B]
  const float* wscale;       // ... and the power of two its U planes were scaled by (stored behind the image)
};

// the contribution of one final output element v (already rounded to the storage type) to the two column sums
struct StatCol {
  float mean, rstd, g, b;
};
__device__ __forceinline__ void stat_add(int mode, int silu, const StatCol& c, float v, float xv, float& s1, float& s2) {
  if (mode == 1) {
    s1 += v;
    s2 += v * v;
  } else {
    const float xh = (xv - c.mean) * c.rstd;
    const float z = xh * c.g + c.b;
    const float dxh = v * (silu ? osm::dsilu_f(z) : 1.0f) * c.g;
    s1 += dxh;
    s2 += dxh * xh;
  }
}

// mode 2 of stat_add with the SiLU choice as a select (straight-line code for unrolled epilogues)
__device__ __forceinline__ void stat_add_bwd(bool silu, const StatCol& c, float v, float xv, float& s1, float& s2) {
  const float xh = (xv - c.mean) * c.rstd;
  const float z = xh * c.g + c.b;
  float d = osm::dsilu_f(z);
  asm("" : "+v"(d));            // computed unconditionally: no scalar branch around it
  const float dxh = v * (silu ? d : 1.0f) * c.g;
  s1 += dxh;
  s2 += dxh * xh;
}

#ifndef OSM_ACT_F16
// NARROW (N <= 64, e.g. attention P V with 64-wide heads): the four waves split the 128 rows (32 each) and
// every wave covers the 64 live columns, instead of 2 x 2 waves of 64 x 64 where half would multiply zeros.
template <int TAPS, bool B_KN, bool NARROW = false>
__global__ __launch_bounds__(256, 2) void igemm_f32_kernel(const float* __restrict__ Aglob,
                                                            const float* __restrict__ Bglob,
                                                            IGemmParams p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDS_STRIDE];
  float* As = smem;
  float* Bs = smem + 2 * BM * LDS_STRIDE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // ---- XCD-aware tile mapping (bijective for any tile count)
  const int nt = p.mtiles * p.ntiles;
  const int bid = blockIdx.x;
  const int q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int tile_n = id % p.ntiles, tile_m = id / p.ntiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int z = blockIdx.z;
  const int b1 = z % p.nb1, b2 = z / p.nb1;
  const float* __restrict__ A = Aglob + b1 * p.sA1 + b2 * p.sA2;
  const float* __restrict__ Bm = Bglob + b1 * p.sB1 + b2 * p.sB2;

  const int ks = blockIdx.y;
  const int per = (p.nchunks + p.splitk - 1) / p.splitk;
  const int kc0 = ks * per;
  const int kc1 = min(p.nchunks, kc0 + per);

  // ---- per-thread staging coordinates
  const int cg = tid & 7;   // float4 column group inside the 32-wide chunk
  const int r0 = tid >> 3;  // 0..31
  long long arow[4];
  unsigned amask[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + r0 + 32 * i;
    arow[i] = (long long)m * p.lda;
    unsigned mk = 0;
    if (m < p.M) {
      if (TAPS == 9) {
        const int w = m % p.W;
        const int h = (m / p.W) % p.H;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
          if (hh >= 0 && hh < p.H && ww >= 0 && ww < p.W) mk |= 1u << t;
        }
      } else {
        mk = 1u;
      }
    }
    amask[i] = mk;
  }
  long long brow[4];
  bool bvalid[4];
  if (!B_KN) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + r0 + 32 * i;
      brow[i] = (long long)n * p.ldb;
      bvalid[i] = n < p.N;
    }
  } else {
    const int n = n0 + 4 * (tid & 31);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      brow[i] = n;
      bvalid[i] = n < p.N;
    }
  }

  float4 ra[4], rb[4];
  unsigned okm = 0;  // validity bits of the staged registers (applied at LDS-store time)

  // unconditional 16-byte loads from a selected (always valid) offset, zeroed afterwards:
  // keeps the staging loads branch-free so they stay in flight under the MFMA stream.
#define OSM_LOAD_CHUNK(kc_)                                                                     \
  {                                                                                             \
    okm = 0;                       \
    const int cc_ = (kc_) / TAPS;                                                               \
    const int tap_ = (kc_) - cc_ * TAPS;                                                        \
    const int c0_ = cc_ * BK;                                                                   \
    long long toff_ = 0;                                                                        \
    if (TAPS == 9) toff_ = ((long long)(tap_ / 3 - 1) * p.W + (tap_ % 3 - 1)) * p.lda;          \
    const int c_ = c0_ + 4 * cg;                                                                \
    const bool cok_ = c_ < p.K;                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                             \
      const bool ok_ = cok_ && ((amask[i] >> tap_) & 1u);                                       \
      const float4 v_ = *reinterpret_cast<const float4*>(A + (ok_ ? arow[i] + toff_ + c_ : 0)); \
      ra[i] = v_; okm |= (ok_ ? 1u : 0u) << i;                                                                \
    }                                                                                           \
    if (!B_KN) {                                                                                \
      const long long boff_ = (long long)tap_ * p.tapstrideB + c_;                              \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
        const bool ok_ = cok_ && bvalid[i];                                                     \
        const float4 v_ = *reinterpret_cast<const float4*>(Bm + (ok_ ? boff_ + brow[i] : 0));   \
        rb[i] = v_; okm |= (ok_ ? 16u : 0u) << i;                                                              \
      }                                                                                         \
    } else {                                                                                    \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
        const int k_ = c0_ + (tid >> 5) + 8 * i;                                                \
        const bool ok_ = (k_ < p.K) && bvalid[i];                                               \
        const float4 v_ =                                                                       \
            *reinterpret_cast<const float4*>(Bm + (ok_ ? (long long)k_ * p.ldb + brow[i] : 0)); \
        rb[i] = v_; okm |= (ok_ ? 16u : 0u) << i;                                                              \
      }                                                                                         \
    }                                                                                           \
  }

  auto store_chunk = [&](int buf) {
    float* a = As + buf * BM * LDS_STRIDE;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4*>(a + (r0 + 32 * i) * LDS_STRIDE + 4 * cg) = sel4((okm >> i) & 1u, ra[i]);
    if (!B_KN) {
      float* b = Bs + buf * BN * LDS_STRIDE;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4*>(b + (r0 + 32 * i) * LDS_STRIDE + 4 * cg) = sel4((okm >> (4 + i)) & 1u, rb[i]);
    } else {
      float* b = Bs + buf * BK * LDS_KN_STRIDE;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4*>(b + ((tid >> 5) + 8 * i) * LDS_KN_STRIDE + 4 * (tid & 31)) =
            sel4((okm >> (4 + i)) & 1u, rb[i]);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wm = NARROW ? wave : wave >> 1, wn = NARROW ? 0 : wave & 1;
  constexpr int WROWS = NARROW ? 32 : 64;   // rows per wave
  const int lr = lane & 31, lk = lane >> 5;

  auto compute = [&](int buf) {
    const float* a = As + buf * BM * LDS_STRIDE + (WROWS * wm + lr) * LDS_STRIDE + 4 * lk;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float4 af[2], bf[2];
      af[0] = *reinterpret_cast<const float4*>(a + 8 * kk);
      af[1] = NARROW ? af[0] : *reinterpret_cast<const float4*>(a + 32 * LDS_STRIDE + 8 * kk);
      if (!B_KN) {
        const float* b = Bs + buf * BN * LDS_STRIDE + (64 * wn + lr) * LDS_STRIDE + 4 * lk;
        bf[0] = *reinterpret_cast<const float4*>(b + 8 * kk);
        bf[1] = *reinterpret_cast<const float4*>(b + 32 * LDS_STRIDE + 8 * kk);
      } else {
        const float* b = Bs + buf * BK * LDS_KN_STRIDE + (8 * kk + 4 * lk) * LDS_KN_STRIDE + 64 * wn + lr;
        bf[0] = make_float4(b[0], b[LDS_KN_STRIDE], b[2 * LDS_KN_STRIDE], b[3 * LDS_KN_STRIDE]);
        bf[1] = make_float4(b[32], b[LDS_KN_STRIDE + 32], b[2 * LDS_KN_STRIDE + 32],
                            b[3 * LDS_KN_STRIDE + 32]);
      }
      const float a0[4] = {af[0].x, af[0].y, af[0].z, af[0].w};
      const float a1[4] = {af[1].x, af[1].y, af[1].z, af[1].w};
      const float b0[4] = {bf[0].x, bf[0].y, bf[0].z, bf[0].w};
      const float b1v[4] = {bf[1].x, bf[1].y, bf[1].z, bf[1].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1v[j], acc[0][1], 0, 0, 0);
        if (!NARROW) {
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1v[j], acc[1][1], 0, 0, 0);
        }
      }
    }
  };

  // ---- main loop: register-staged, LDS double-buffered, one barrier per chunk
  const int nk = kc1 - kc0;
  if (nk > 0) {
    OSM_LOAD_CHUNK(kc0);
    store_chunk(0);
    __syncthreads();
    for (int it = 0; it < nk; ++it) {
      const int cur = it & 1;
      const bool more = it + 1 < nk;
      if (more) OSM_LOAD_CHUNK(kc0 + it + 1);
      compute(cur);
      if (more) store_chunk(cur ^ 1);
      __syncthreads();
    }
  }

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool partial = p.splitk > 1;
  float* Cb = partial ? p.ws + ((long long)(ks * p.nbatch + z) * p.M) * p.N
                      : p.C + b1 * p.sC1 + b2 * p.sC2;
  const float* Rb = (p.res && !partial) ? p.res + b1 * p.sC1 + b2 * p.sC2 : nullptr;
  const long long ldc = partial ? (long long)p.N : p.ldc;
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int n = n0 + 64 * wn + 32 * tn + lr;
    if (n >= p.N) continue;
    const float bv = (!partial && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
    for (int tm = 0; tm < (NARROW ? 1 : 2); ++tm) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + WROWS * wm + 32 * tm + (e & 3) + 8 * (e >> 2) + 4 * lk;
        if (m >= p.M) continue;
        float v = acc[tm][tn][e];
        if (!partial) {
          v = v * p.alpha + bv;
          if (Rb) v += Rb[(long long)m * p.ldr + n];
          if (p.accumulate) v += Cb[(long long)m * ldc + n];
        }
        Cb[(long long)m * ldc + n] = v;
      }
    }
  }
}

#endif   // !OSM_ACT_F16

// Deep splits of small outputs (the 8x8 / 16x16 layers: 16-32 partials of <= 1 MB each): one thread per float4 would leave
// 64-256 workgroups each walking 32 dependent-latency loads.  Here a workgroup owns 64 float4 and its four waves take the
// partials k = w, w + 4, ...; the four sums are folded through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void splitk_reduce_deep_kernel(IGemmParams p) {
  __shared__ float4 red[4][64];
  const int lane = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const long long slice = (long long)p.M * p.N;
  const long long i = (long long)blockIdx.x * 64 + lane;       // float4 index
  const bool live = i < slice / 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  // the finishing wave requests its residual / previous-output operands WITH its partials (after the LDS fold's barrier they were
  // one more dependent memory round trip of a ~5 us kernel)
  const int nv = p.N / 4;
  const int n = (int)(i % nv) * 4;
  const long long m = i / nv;
  act_t* c = p.C + m * p.ldc + n;
  float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f), c4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float bq[4] = {0.f, 0.f, 0.f, 0.f};
  if (live && kg == 0) {
    if (p.res) r4 = osm::ld4(p.res + m * p.ldr + n);
    if (p.accumulate) c4 = osm::ld4(c);
    if (p.bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bq[e] = p.bias[n + e];
    }
  }
  if (live) {
    const float* w = p.ws + i * 4;
#pragma unroll 4
    for (int k = kg; k < p.splitk; k += 4) {
      const float4 t = *reinterpret_cast<const float4*>(w + k * slice);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
  }
  red[kg][lane] = s;
  __syncthreads();
  if (kg != 0 || !live) return;
  const float4 a = red[0][lane], b = red[1][lane], c2 = red[2][lane], d = red[3][lane];
  float v[4] = {(a.x + b.x) + (c2.x + d.x), (a.y + b.y) + (c2.y + d.y), (a.z + b.z) + (c2.z + d.z), (a.w + b.w) + (c2.w + d.w)};
  const float rv[4] = {r4.x, r4.y, r4.z, r4.w}, cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = (v[e] * p.alpha + bq[e]) + rv[e] + cv[e];
  osm::st4(c, make_float4(v[0], v[1], v[2], v[3]));
}

// C = alpha * sum_s ws[s] + bias + res (+ C)      (fixed summation order: deterministic)
template <int VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(IGemmParams p) {
  const long long slice = (long long)p.nbatch * p.M * p.N;
  const long long total = slice / VEC;
  const int nv = p.N / VEC;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % nv) * VEC;
    const long long mz = i / nv;
    const int m = (int)(mz % p.M);
    const int z = (int)(mz / p.M);
    float s[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = 0.f;
    const float* w = p.ws + i * VEC;
    // residual / previous output are requested BEFORE the walk over the partials (behind it they were one more dependent round trip)
    const int b1 = z % p.nb1, b2 = z / p.nb1;
    const long long coff = b1 * p.sC1 + b2 * p.sC2;
    act_t* c = p.C + coff + (long long)m * p.ldc + n;
    const act_t* rs = p.res ? p.res + coff + (long long)m * p.ldr + n : nullptr;
    float rv[VEC], cv[VEC];
    if constexpr (VEC == 4) {
      const float4 r4 = rs ? osm::ld4(rs) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 c4 = p.accumulate ? osm::ld4(c) : make_float4(0.f, 0.f, 0.f, 0.f);
      rv[0] = r4.x; rv[1] = r4.y; rv[2] = r4.z; rv[3] = r4.w;
      cv[0] = c4.x; cv[1] = c4.y; cv[2] = c4.z; cv[3] = c4.w;
    } else {
      rv[0] = rs ? osm::ld1(rs) : 0.f;
      cv[0] = p.accumulate ? osm::ld1(c) : 0.f;
    }
    for (int k = 0; k < p.splitk; ++k) {
      if (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(w + k * slice);
        s[0] += t.x; s[1] += t.y; s[2] += t.z; s[3] += t.w;
      } else {
        s[0] += w[k * slice];
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[e] = (s[e] * p.alpha + (p.bias ? p.bias[n + e] : 0.f)) + rv[e] + cv[e];
    if constexpr (VEC == 4) osm::st4(c, make_float4(s[0], s[1], s[2], s[3]));
    else osm::st1(c, s[0]);
  }
}

// The same combine for 3x3 / 1x1 convolutions that also want the column sums of their result (IGemmParams::colsum):
// a workgroup owns 8 rows x 128 columns (thread = 4 columns of one row), folds its 8 rows through LDS in a fixed order
// and writes chunk (= 8-row block) partials colsum[m / 8][2][N].  Needs N % 4 == 0 and 16-byte aligned rows.
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(IGemmParams p) {
  __shared__ float red[8][2][128];
  const int t = threadIdx.x, cg = t & 31, r = t >> 5;
  const int n = blockIdx.y * 128 + cg * 4;
  const long long m = (long long)blockIdx.x * 8 + r;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < p.N && m < p.M) {
    const long long slice = (long long)p.M * p.N;
    const float* w = p.ws + m * p.N + n;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < p.splitk; ++k) {
      const float4 q = *reinterpret_cast<const float4*>(w + k * slice);
      v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
    }
    act_t* c = p.C + m * p.ldc + n;
    const float4 r4 = p.res ? osm::ld4(p.res + m * p.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 c4 = p.accumulate ? osm::ld4(c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float rv[4] = {r4.x, r4.y, r4.z, r4.w}, cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (v[e] * p.alpha + (p.bias ? p.bias[n + e] : 0.f)) + rv[e] + cv[e];
    osm::st4(c, make_float4(v[0], v[1], v[2], v[3]));
    float xv[4] = {0.f, 0.f, 0.f, 0.f};
    StatCol sc[4] = {};
    if (p.stat_mode == 2) {
      const float4 x4 = osm::ld4(p.stat_x + m * p.ld_sx + n);
      xv[0] = x4.x; xv[1] = x4.y; xv[2] = x4.z; xv[3] = x4.w;
      const float* tb = p.stat_table + (m / ((long long)p.H * p.W)) * 4 * p.N + n;
#pragma unroll
      for (int e = 0; e < 4; ++e) sc[e] = StatCol{tb[e], tb[p.N + e], tb[2 * p.N + e], tb[3 * p.N + e]};
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) stat_add(p.stat_mode, p.stat_silu, sc[e], (float)(act_t)v[e], xv[e], s1[e], s2[e]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    red[r][0][cg * 4 + e] = s1[e];
    red[r][1][cg * 4 + e] = s2[e];
  }
  __syncthreads();
  if (t < 128 && blockIdx.y * 128 + t < p.N) {
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      a1 += red[q][0][t];
      a2 += red[q][1][t];
    }
    float* o = p.colsum + (long long)blockIdx.x * 2 * p.N + blockIdx.y * 128 + t;
    o[0] = a1;
    o[p.N] = a2;
  }
}

#ifndef OSM_ACT_F16
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wf,
                                   float* __restrict__ wd, int Cout, int Cin, int k) {
  const long long total = (long long)Cout * Cin * k * k;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int kw = (int)(i % k);
    const int kh = (int)((i / k) % k);
    const int ci = (int)((i / (k * k)) % Cin);
    const int co = (int)(i / ((long long)k * k * Cin));
    const float v = w[i];
    if (wf) wf[((long long)(kh * k + kw) * Cout + co) * Cin + ci] = v;
    if (wd) wd[((long long)((k - 1 - kh) * k + (k - 1 - kw)) * Cin + ci) * Cout + co] = v;
  }
}

#endif   // !OSM_ACT_F16

#include "igemm_bf16s.inc.h"
#include "conv3_halo.inc.h"
#include "conv3_wino.inc.h"
#include "conv3_wino8.inc.h"
#ifdef OSM_WITH_WINO4     // experiment (round 6): one wave per SIMD, 96-column workgroup tile (tools/experiments/conv3_wino4.inc.h)
#include "../../tools/experiments/conv3_wino4.inc.h"
#endif
#if defined(OSM_WITH_WINO16) && !defined(OSM_ACT_F16)     // measurement builds only (tools/experiments/, profiles/NOTES_r05.md)
#include "../../tools/experiments/conv3_wino16.inc.h"
#endif

// OSM_CONV_HALO=0 selects the tap-chunked kernel for 3x3 layers too (A/B measurements only)
bool halo_enabled() {
  static const bool on = [] {
    const char* e = std::getenv("OSM_CONV_HALO");
    return !(e && e[0] == '0');
  }();
  return on;
}

// Winograd F(2x2, 3x3) path (conv3_wino.inc.h): 3x3, split-bf16 image in the Winograd domain, fp32 activations
// (an image is addressed by 32-bit byte offsets: H * W * ldx * sizeof(x) < 4 GiB is checked at launch; the shape query below
// refuses images whose DENSE size already reaches 1 GiB, which leaves room for row strides up to 4 Cin -- callers with wider
// strides compare H * W * ldx * sizeof(x) themselves, as engine._conv does, and take the direct kernel instead)
bool wino_shape_ok(int H, int W, int K, int N) {
  return H >= 16 && W >= 16 && K >= 16 && N >= 64 && N % 32 == 0 && (long long)H * W * K * 4 < (1LL << 30);
}

int launch(IGemmParams& p, int taps, bool b_kn, hipStream_t st, int wfmt = 0, bool wino = false) {
  p.mtiles = (p.M + BM - 1) / BM;
  p.ntiles = (p.N + BN - 1) / BN;
  p.nchunks = taps * ((p.K + BK - 1) / BK);
  if (p.splitk < 1) p.splitk = 1;
  if (p.splitk > p.nchunks) p.splitk = p.nchunks;
  dim3 grid(p.mtiles * p.ntiles, p.splitk, p.nbatch);
#ifdef OSM_ACT_F16
  if (wfmt != 1) return osm::fail(OSM_ERR_UNSUPPORTED, "the fp16 family takes wfmt 1 (fp16 fragment image) only, got %d", wfmt);
#else
  if (wfmt == 1) return osm::fail(OSM_ERR_UNSUPPORTED, "wfmt 1 (fp16 arithmetic) belongs to the fp16 family (osm_conv2d_nhwc_h)");
#endif
  if (wino) {
    if (!(taps == 9 && wfmt >= 1 && wfmt <= 4 && wino_shape_ok(p.H, p.W, p.K, p.N)))
      return osm::fail(OSM_ERR_UNSUPPORTED, "Winograd weight image: 3x3, wfmt 1 / 2 / 3 / 4, H, W >= 16, Cin >= 16, Cout >= 64 and a multiple of 32 only");
    if (wfmt == 4 && !(p.xmax && osm::aligned16(p.xmax)))
      return osm::fail(OSM_ERR_INVALID, "the f16x3 Winograd image (wfmt 4) needs x_maxabs (osm_maxabs of the input), 16-byte aligned");
    if (!(p.ldc % 4 == 0 && osm::aligned_act4(p.C) && (!p.res || (p.ldr % 4 == 0 && osm::aligned_act4(p.res))) &&
          (!p.bias || osm::aligned16(p.bias)) && (p.splitk <= 1 || osm::aligned16(p.ws))))
      return osm::fail(OSM_ERR_UNSUPPORTED, "the Winograd kernel stores 4-element vectors: ldy, ldr multiples of 4, aligned y / res / bias");
    const int nimg = p.M / (p.H * p.W);
    if ((long long)p.H * p.W * p.lda * (long long)sizeof(act_t) > 0xffffffffLL)
      return osm::fail(OSM_ERR_UNSUPPORTED, "the Winograd kernel addresses an image by 32-bit byte offsets: H * W * ldx * sizeof(x) must be < 4 GiB");
    p.mtiles = nimg * ((p.H + 15) / 16) * ((p.W + 15) / 16);
    p.ntiles = (p.N + 63) / 64;
    p.nchunks = p.ksteps;                       // 16-channel slabs
    if (p.splitk > p.nchunks) p.splitk = p.nchunks;
    p.stat_chunks = p.mtiles / nimg;            // column sums without split-K: one chunk per 16 x 16 patch
    if (p.colsum && p.stat_mode == 2 && !(p.ld_sx % 4 == 0 && osm::aligned_act4(p.stat_x) && osm::aligned16(p.stat_table)))
      return osm::fail(OSM_ERR_UNSUPPORTED, "the Winograd kernel reads stat_x / stat_table as 4-element vectors");
    // column tiles that share an input patch in time (conv3_wino.inc.h, tile mapping): the largest divisor of ntiles <= 4
    static const int ngrp = [] { const char* e = std::getenv("OSM_WINO_NGROUP"); return e ? atoi(e) : 4; }();
    p.nb1 = 1;
    for (int g = 2; g <= ngrp; ++g) if (p.ntiles % g == 0) p.nb1 = g;
    const unsigned short* Up = reinterpret_cast<const unsigned short*>(p.Bm);
    const dim3 gw(p.mtiles * p.ntiles, p.splitk, 1);
#define OSM_WINO_LAUNCH(NP_)                                                                               \
    if (p.gn_table) hipLaunchKernelGGL((conv3_wino8_kernel<NP_, true>), gw, dim3(512), 0, st, p.A, Up, p);  \
    else hipLaunchKernelGGL((conv3_wino8_kernel<NP_, false>), gw, dim3(512), 0, st, p.A, Up, p);
#ifdef OSM_ACT_F16
    if (wfmt != 1) return osm::fail(OSM_ERR_UNSUPPORTED, "fp16 family: Winograd image wfmt 1 only");
    OSM_WINO_LAUNCH(1)
#else
    if (wfmt == 4) {     // f16x3: two half planes behind a scale word
      p.wscale = reinterpret_cast<const float*>(Up + 2LL * 16 * p.ksteps * p.nt32 * 512);
      if (p.gn_table)    // x_maxabs is the range of x, not of act(GroupNorm(x)): the scale would be wrong
        return osm::fail(OSM_ERR_UNSUPPORTED, "the f16x3 Winograd image does not take a fused GroupNorm input (gn_table)");
      // the software-pipelined K loop: 16-channel slabs without a tail, and enough of them per workgroup to pay for its prologue
      const int per = (p.ksteps + p.splitk - 1) / p.splitk;
      // (its staging addresses activations by 32-bit buffer offsets relative to the image, out-of-range = padding: < 2 GiB per image)
      const long long img_bytes = (long long)p.H * p.W * p.lda * (long long)sizeof(act_t);
#ifdef OSM_WITH_WINO4
      {
        static const int wino4 = [] { const char* e = std::getenv("OSM_WINO4"); return e ? atoi(e) : 0; }();
        if (wino4 && (p.K & 15) == 0 && p.splitk == 1 && !p.colsum && img_bytes < 0x7fffffffLL) {
          IGemmParams q = p;
          q.ntiles = (p.N + 32 * W4_NCT - 1) / (32 * W4_NCT);
          if (wino4 == 2) hipLaunchKernelGGL((conv3_wino4p_kernel<W4_NCT>), dim3(q.mtiles * q.ntiles, 1, 1), dim3(256), 0, st, q.A, Up, q);
          else hipLaunchKernelGGL((conv3_wino4_kernel<W4_NCT>), dim3(q.mtiles * q.ntiles, 1, 1), dim3(256), 0, st, q.A, Up, q);
          return osm::check_launch("conv3_wino4_kernel");
        }
      }
#endif
#ifdef OSM_WITH_WINO16   // experiment: the 16-wave, one-xi-per-wave instance for the plain case (tools/experiments/conv3_wino16.inc.h)
#ifdef W16_FORCE
      static const bool wino16 = true;
#else
      static const bool wino16 = [] { const char* e = std::getenv("OSM_WINO16"); return e && atoi(e) == 1; }();
#endif
      if (wino16 && (p.K & 15) == 0 && p.splitk == 1 && !p.colsum && img_bytes < 0x7fffffffLL) {
        hipLaunchKernelGGL((conv3_wino16_kernel<0>), gw, dim3(1024), 0, st, p.A, Up, p);
        return osm::check_launch("conv3_wino16_kernel");
      }
#endif
      if (W8_PIPE && (p.K & 15) == 0 && per >= W8_PIPE_MIN && img_bytes < 0x7fffffffLL)
        hipLaunchKernelGGL((conv3_wino8_kernel<2, false, true, true>), gw, dim3(512), 0, st, p.A, Up, p);
      else
        hipLaunchKernelGGL((conv3_wino8_kernel<2, false, true, false>), gw, dim3(512), 0, st, p.A, Up, p);
    } else if (wfmt == 3) { OSM_WINO_LAUNCH(3) } else { OSM_WINO_LAUNCH(2) }
#endif
#undef OSM_WINO_LAUNCH
  } else
  {
  const bool halo_path = wfmt != 0 && taps == 9 && p.W >= 8 && p.H >= 8 && halo_enabled();
  if (p.colsum && (wfmt == 0 || !(halo_path || p.splitk > 1)))
    return osm::fail(OSM_ERR_UNSUPPORTED, "column sums are produced by the halo-tile kernel or the split-K combine only "
                                          "(ask osm_conv_stat_chunks first)");
  if (wfmt != 0 && taps == 9 && p.W >= 8 && p.H >= 8 && halo_enabled()) {
    // halo-tile kernel: M-tiles are 8 x 16 (W >= 16) or 8 x 8 pixel patches, K is consumed in 32-channel slabs of all 9 taps
    const unsigned short* Bp = reinterpret_cast<const unsigned short*>(p.Bm);
    const bool wide = p.W >= 16;
    const int nimg = p.M / (p.H * p.W);
    p.mtiles = nimg * ((p.H + 7) / 8) * (wide ? (p.W + 15) / 16 : (p.W + 7) / 8);
    p.nchunks = (p.K + BK - 1) / BK;
    if (p.splitk > p.nchunks) p.splitk = p.nchunks;
    p.stat_chunks = p.mtiles / nimg;
    const dim3 g2(p.mtiles * p.ntiles, p.splitk, 1);
    if (wfmt < 1 || wfmt > 4) return osm::fail(OSM_ERR_UNSUPPORTED, "unknown weight format %d", wfmt);
    // <= 32 output columns (head / stem data-gradient): the waves split the row blocks instead of the column tiles
    static const bool narrow_on = [] { const char* e = std::getenv("OSM_NARROW"); return !(e && e[0] == '0'); }();
    const bool narrow = narrow_on && wide && p.N <= 32 && !p.colsum;
#define OSM_HALO_LAUNCH(NP_, GN_, PW_, BR_) \
    hipLaunchKernelGGL((conv3_halo_bf16s_kernel<NP_, GN_, PW_, BR_>), g2, dim3(256), 0, st, p.A, Bp, p)
#define OSM_HALO_PICK(NP_)                                                                                  \
    if (narrow) { if (p.gn_table) hipLaunchKernelGGL((conv3_halo_bf16s_kernel<NP_, true, 16, 3, 8, true>), g2, dim3(256), 0, st, p.A, Bp, p); \
                  else hipLaunchKernelGGL((conv3_halo_bf16s_kernel<NP_, false, 16, 3, 8, true>), g2, dim3(256), 0, st, p.A, Bp, p); }       \
    else if (wide)      { if (p.gn_table) OSM_HALO_LAUNCH(NP_, true, 16, 3); else OSM_HALO_LAUNCH(NP_, false, 16, 3); } \
    else                { if (p.gn_table) OSM_HALO_LAUNCH(NP_, true, 8, 3);  else OSM_HALO_LAUNCH(NP_, false, 8, 3); }
#ifdef OSM_ACT_F16
    // two column tiles per wave (128 rows x 256 columns per workgroup) where the layer still fills the chip with them: the
    // one-MFMA-per-product arithmetic is bound by its LDS fragment reads (conv3_halo.inc.h, NT).  Aligned shapes only (its
    // epilogue has the vector form only)
    const char* nt2_env = std::getenv("OSM_HALO_NT2");      // 0: off, 1 (default): where it fills the chip, 2: wherever the shape allows (tests)
    const int nt2_mode = nt2_env ? atoi(nt2_env) : 1;
    const auto al = [](const void* q_, unsigned bytes) { return (reinterpret_cast<unsigned long long>(q_) & (bytes - 1)) == 0; };
    const bool nt2 = nt2_mode > 0 && wide && !narrow && p.N % 256 == 0 &&
                     (nt2_mode == 2 || (long long)p.mtiles * (p.N / 256) * p.splitk >= 512) &&
                     (p.splitk > 1 ? al(p.ws, 16)
                                   : ((p.ldc & 3) == 0 && al(p.C, 4 * ACT_B) && (!p.bias || al(p.bias, 16)) &&
                                      (!p.res || ((p.ldr & 3) == 0 && al(p.res, 4 * ACT_B))) &&
                                      (!p.colsum || p.stat_mode == 1 ||
                                       ((p.ld_sx & 3) == 0 && al(p.stat_x, 4 * ACT_B) && al(p.stat_table, 16)))));
    if (nt2) {
      p.ntiles = p.N / 256;
      const dim3 g3(p.mtiles * p.ntiles, p.splitk, 1);
      if (p.gn_table) hipLaunchKernelGGL((conv3_halo_bf16s_kernel<1, true, 16, 3, 8, false, 2>), g3, dim3(256), 0, st, p.A, Bp, p);
      else hipLaunchKernelGGL((conv3_halo_bf16s_kernel<1, false, 16, 3, 8, false, 2>), g3, dim3(256), 0, st, p.A, Bp, p);
    } else {
    OSM_HALO_PICK(1)
    }
#else
    if (wfmt == 4) {      // direct f16x3 (conv3_halo.inc.h, HP): two half planes behind a scale word, the input's range from the caller
      if (!(p.xmax && osm::aligned16(p.xmax)))
        return osm::fail(OSM_ERR_INVALID, "the f16x3 image (wfmt 4) needs x_maxabs (osm_maxabs of the input), 16-byte aligned");
      if (p.gn_table) return osm::fail(OSM_ERR_UNSUPPORTED, "the f16x3 image does not take a fused GroupNorm input (gn_table)");
      p.wscale = reinterpret_cast<const float*>(Bp + 2LL * 9 * p.ksteps * p.nt32 * 512);
      if (narrow) hipLaunchKernelGGL((conv3_halo_bf16s_kernel<2, false, 16, 3, 8, true, 1, true>), g2, dim3(256), 0, st, p.A, Bp, p);
      else if (wide) hipLaunchKernelGGL((conv3_halo_bf16s_kernel<2, false, 16, 3, 8, false, 1, true>), g2, dim3(256), 0, st, p.A, Bp, p);
      else hipLaunchKernelGGL((conv3_halo_bf16s_kernel<2, false, 8, 3, 8, false, 1, true>), g2, dim3(256), 0, st, p.A, Bp, p);
    } else if (wfmt == 3) { OSM_HALO_PICK(3) } else { OSM_HALO_PICK(2) }
#endif
#undef OSM_HALO_PICK
#undef OSM_HALO_LAUNCH
  } else if (wfmt != 0) {
    const unsigned short* Bp = reinterpret_cast<const unsigned short*>(p.Bm);
    const dim3 g2(p.mtiles * p.ntiles, p.splitk, 1);
#ifdef OSM_ACT_F16
    if (taps == 9)
      hipLaunchKernelGGL((igemm_bf16s_kernel<9, 1>), g2, dim3(256), 0, st, p.A, Bp, p);
    else
      hipLaunchKernelGGL((igemm_bf16s_kernel<1, 1>), g2, dim3(256), 0, st, p.A, Bp, p);
#else
    if (wfmt == 4) {     // f16x3 image of a 1x1 layer: two half planes behind a scale word, the input's range from the caller
      if (taps != 1) return osm::fail(OSM_ERR_UNSUPPORTED, "the direct f16x3 image (wfmt 4 without OSM_WFMT_WINOGRAD): 3x3 layers need H, W >= 8 (the halo-tile kernel)");
      if (!(p.xmax && osm::aligned16(p.xmax)))
        return osm::fail(OSM_ERR_INVALID, "the f16x3 image (wfmt 4) needs x_maxabs (osm_maxabs of the input), 16-byte aligned");
      if ((p.H * p.W) % BM != 0)
        return osm::fail(OSM_ERR_UNSUPPORTED, "the direct f16x3 kernel needs H * W to be a multiple of %d (one image per tile)", BM);
      p.wscale = reinterpret_cast<const float*>(Bp + 2LL * p.ksteps * p.nt32 * 512);
      hipLaunchKernelGGL((igemm_bf16s_kernel<1, 2, true>), g2, dim3(256), 0, st, p.A, Bp, p);
    } else if (wfmt == 3 && taps == 9)
      hipLaunchKernelGGL((igemm_bf16s_kernel<9, 3>), g2, dim3(256), 0, st, p.A, Bp, p);
    else if (wfmt == 3)
      hipLaunchKernelGGL((igemm_bf16s_kernel<1, 3>), g2, dim3(256), 0, st, p.A, Bp, p);
    else if (wfmt == 2 && taps == 9)
      hipLaunchKernelGGL((igemm_bf16s_kernel<9, 2>), g2, dim3(256), 0, st, p.A, Bp, p);
    else if (wfmt == 2)
      hipLaunchKernelGGL((igemm_bf16s_kernel<1, 2>), g2, dim3(256), 0, st, p.A, Bp, p);
    else
      return osm::fail(OSM_ERR_UNSUPPORTED, "unknown weight format %d", wfmt);
#endif
  }
#ifndef OSM_ACT_F16
  else if (taps == 9) {
    if (b_kn) return osm::fail(OSM_ERR_UNSUPPORTED, "3x3 conv needs [n][k] weights");
    hipLaunchKernelGGL((igemm_f32_kernel<9, false>), grid, dim3(256), 0, st, p.A, p.Bm, p);
  } else if (b_kn && p.N <= 64) {
    hipLaunchKernelGGL((igemm_f32_kernel<1, true, true>), grid, dim3(256), 0, st, p.A, p.Bm, p);
  } else if (b_kn) {
    hipLaunchKernelGGL((igemm_f32_kernel<1, true>), grid, dim3(256), 0, st, p.A, p.Bm, p);
  } else if (p.N <= 64) {
    hipLaunchKernelGGL((igemm_f32_kernel<1, false, true>), grid, dim3(256), 0, st, p.A, p.Bm, p);
  } else {
    hipLaunchKernelGGL((igemm_f32_kernel<1, false>), grid, dim3(256), 0, st, p.A, p.Bm, p);
  }
#endif
  }
  int rc = osm::check_launch("igemm kernel");
  if (rc) return rc;
  if (p.splitk > 1 && p.colsum) {
    if (!(p.N % 4 == 0 && p.ldc % 4 == 0 && (!p.res || p.ldr % 4 == 0) && osm::aligned_act4(p.C) && osm::aligned16(p.ws) &&
          (!p.res || osm::aligned_act4(p.res)) && p.M % 8 == 0 && (p.H * p.W) % 8 == 0 && p.nbatch == 1))
      return osm::fail(OSM_ERR_UNSUPPORTED, "column sums with split-K need N %% 4 == 0, 8-row aligned images, aligned rows");
    hipLaunchKernelGGL(splitk_reduce_stats_kernel, dim3(p.M / 8, (p.N + 127) / 128), dim3(256), 0, st, p);
    return osm::check_launch("splitk_reduce_stats_kernel");
  }
  if (p.splitk > 1) {
    const bool v4 = p.N % 4 == 0 && p.ldc % 4 == 0 && (!p.res || p.ldr % 4 == 0) && osm::aligned_act4(p.C) &&
                    osm::aligned16(p.ws) && (!p.res || osm::aligned_act4(p.res)) && p.sC1 % 4 == 0 && p.sC2 % 4 == 0;
    const long long total = (long long)p.nbatch * p.M * p.N / (v4 ? 4 : 1);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (v4 && p.nbatch == 1 && p.splitk >= 8 && total <= 64 * 1024) {
      hipLaunchKernelGGL(splitk_reduce_deep_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, p);
      return osm::check_launch("splitk_reduce_deep_kernel");
    }
    if (v4) hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(blocks), dim3(256), 0, st, p);
    rc = osm::check_launch("splitk_reduce_kernel");
  }
  return rc;
}

}  // namespace

#ifndef OSM_ACT_F16
extern "C" int osm_splitk_hint(int M, int N, int K, int taps, int nbatch) {
  const long long tiles = (long long)((M + BM - 1) / BM) * ((N + BN - 1) / BN) * (nbatch > 0 ? nbatch : 1);
  const int nchunks = taps * ((K + BK - 1) / BK);
  static const int target = [] {   // OSM_SPLITK_TARGET: workgroups to aim for (tuning knob)
    const char* e = std::getenv("OSM_SPLITK_TARGET");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 256;
  }();
  if (tiles >= target) return 1;   // one workgroup per CU already (measured: 256 beats 384 / 512 / 768)
  long long s = (target + tiles - 1) / tiles;  // aim at ~1 workgroup per CU: fewer fp32 partials to write and re-read
  const int max_by_chunks = nchunks / 4 > 0 ? nchunks / 4 : 1;  // >= 4 chunks per slice
  if (s > max_by_chunks) s = max_by_chunks;
  if (s > 64) s = 64;
  return (int)(s < 1 ? 1 : s);
}

// split-K factor osm_conv2d_nhwc(_h) wants for a layer (callers size the fp32 workspace splitk * M * Cout from it)
extern "C" int osm_conv_splitk(int B, int H, int W, int Cin, int Cout, int ksize, int wfmt, int has_gn_table) {
  const int M = B * H * W;
  if (wfmt & OSM_WFMT_WINOGRAD) {   // workgroups of 16 x 16 pixels x 64 columns, 16-channel slabs, one workgroup per CU
    const long long tiles = (long long)B * ((H + 15) / 16) * ((W + 15) / 16) * ((Cout + 63) / 64);
    const int nslab = 2 * ((Cin + 31) / 32);
    static const int wtarget = [] { const char* e = std::getenv("OSM_WINO_SPLIT_TARGET"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
    if (tiles >= (3 * wtarget) / 4) return 1;
    long long s = (wtarget + tiles - 1) / tiles;
    if (s > nslab / 4) s = nslab / 4;
    return (int)(s < 1 ? 1 : (s > 32 ? 32 : s));
  }
  return osm_splitk_hint(M, Cout, Cin, ksize * ksize, 1);
}
extern "C" int osm_conv_kernel_kind(int B, int H, int W, int Cin, int Cout, int ksize, int wfmt) {
  (void)B;
  if (wfmt & OSM_WFMT_WINOGRAD) return 5;
  if (wfmt == 0) return 0;
  if (ksize == 3 && W >= 8 && H >= 8 && halo_enabled()) {
    static const bool narrow_on = [] { const char* e = std::getenv("OSM_NARROW"); return !(e && e[0] == '0'); }();
    if (W >= 16) return (narrow_on && Cout <= 32) ? 4 : 2;
    return 3;
  }
  return 1;
}
// chunks per image of the column sums a layer can emit (osm_conv_desc.colsum), 0 = that layer's kernel cannot
extern "C" int osm_conv_stat_chunks(int B, int H, int W, int Cin, int Cout, int ksize, int wfmt, int splitk,
                                    int has_gn_table) {
  (void)B; (void)has_gn_table;
  if (wfmt & OSM_WFMT_WINOGRAD) {
    const int nslab = 2 * ((Cin + 31) / 32);
    if (splitk > nslab) splitk = nslab;
    if (splitk > 1) return (Cout % 4 == 0 && (H * W) % 8 == 0) ? H * W / 8 : 0;
    return ((H + 15) / 16) * ((W + 15) / 16);   // the kernel's epilogue: one chunk per 16 x 16 patch
  }
  if (wfmt == 0) return 0;
  const bool halo = ksize == 3 && W >= 8 && H >= 8 && halo_enabled();
  // the split launch() will really use: it is clamped to the number of K chunks (32-channel slabs of the halo kernel /
  // 32-channel chunks per tap of the tap-chunked kernel)
  const int nchunks = (halo ? 1 : ksize * ksize) * ((Cin + BK - 1) / BK);
  if (splitk > nchunks) splitk = nchunks;
  if (splitk > 1) return (Cout % 4 == 0 && (H * W) % 8 == 0) ? H * W / 8 : 0;
  if (halo) return ((H + 7) / 8) * (W >= 16 ? (W + 15) / 16 : (W + 7) / 8);
  return 0;
}

// 1 when a layer may be given a Winograd weight image (OSM_WFMT_WINOGRAD): 3x3, stride 1, wfmt 2 / 3, fp32 family
extern "C" int osm_conv_winograd_ok(int H, int W, int Cin, int Cout, int ksize, int wfmt) {
  return ksize == 3 && wfmt >= 1 && wfmt <= 4 && wino_shape_ok(H, W, Cin, Cout) ? 1 : 0;
}
extern "C" long long osm_winograd_weight_elems(int Cout, int Cin, int wfmt, int dgrad) {
  const int N = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
  const long long per_plane = 16LL * (2 * ((K + 31) / 32)) * ((N + 31) / 32) * 512;   // 16-bit elements
  if (wfmt == 4) return 2 * per_plane + 8;      // f16x3: two half planes + 16 bytes holding the scale (one float)
  return wfmt * per_plane;
}
extern "C" int osm_pack_conv_weight_winograd(const float* w, void* w_fwd, void* w_dgrad, int Cout, int Cin, int wfmt,
                                             void* stream) {
  OSM_REQUIRE(w && (w_fwd || w_dgrad), "osm_pack_conv_weight_winograd: null pointer");
  OSM_REQUIRE(wfmt >= 1 && wfmt <= 4, "osm_pack_conv_weight_winograd: wfmt must be 1 (fp16), 2 or 3 (bf16 planes), 4 (f16x3)");
  for (int dg = 0; dg < 2; ++dg) {
    unsigned short* out = reinterpret_cast<unsigned short*>(dg ? w_dgrad : w_fwd);
    if (!out) continue;
    const long long per_plane = (osm_winograd_weight_elems(Cout, Cin, wfmt, dg) - (wfmt == 4 ? 8 : 0)) / (wfmt == 4 ? 2 : wfmt);
    int blocks = (int)((per_plane + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (wfmt == 4) {     // pass 1: max |U| (as uint bits) into the scale word behind the planes; pass 2 scales by it
      unsigned* sw = reinterpret_cast<unsigned*>(out + 2 * per_plane);
      hipError_t e = hipMemsetAsync(sw, 0, 16, (hipStream_t)stream);
      if (e != hipSuccess) return osm::fail(OSM_ERR_LAUNCH, "osm_pack_conv_weight_winograd: memset failed");
      hipLaunchKernelGGL(wino_umax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, sw, Cout, Cin, dg);
      hipLaunchKernelGGL(wino_scale_word_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sw, 0);
    }
    hipLaunchKernelGGL(pack_weight_wino_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin, wfmt, dg);
    if (wfmt == 4)
      hipLaunchKernelGGL(wino_scale_word_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream,
                         reinterpret_cast<unsigned*>(out + 2 * per_plane), 1);
    int rc = osm::check_launch("pack_weight_wino_kernel");
    if (rc) return rc;
  }
  return OSM_OK;
}
#endif   // !OSM_ACT_F16

#ifdef OSM_ACT_F16
extern "C" int osm_conv2d_nhwc_h(const osm_conv_desc_h* d, void* stream) {
#else
extern "C" int osm_conv2d_nhwc(const osm_conv_desc* d, void* stream) {
#endif
  OSM_REQUIRE(d && d->x && d->w && d->y, "osm_conv2d_nhwc: null pointer");
  OSM_REQUIRE(d->ksize == 1 || d->ksize == 3, "osm_conv2d_nhwc: ksize must be 1 or 3 (got %d)", d->ksize);
  OSM_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "osm_conv2d_nhwc: bad shape");
  OSM_REQUIRE(d->Cin % 4 == 0 && d->ldx % 4 == 0, "osm_conv2d_nhwc: Cin and ldx must be multiples of 4");
  OSM_REQUIRE(d->ldx >= d->Cin && d->ldy >= d->Cout, "osm_conv2d_nhwc: ld smaller than channels");
  OSM_REQUIRE(osm::aligned_act4(d->x) && osm::aligned16(d->w), "osm_conv2d_nhwc: x (4 elements) / w (16 bytes) misaligned");
  OSM_REQUIRE(d->splitk <= 1 || d->splitk_ws, "osm_conv2d_nhwc: splitk>1 needs a workspace");
  OSM_REQUIRE(!d->res || d->ldr >= d->Cout, "osm_conv2d_nhwc: ldr smaller than Cout");
  IGemmParams p{};
  p.A = OSM_CACT(d->x); p.Bm = reinterpret_cast<const float*>(d->w); p.bias = d->bias; p.res = OSM_CACT(d->res);
  p.C = OSM_ACT(d->y); p.ws = d->splitk_ws;
  p.M = d->B * d->H * d->W; p.N = d->Cout; p.K = d->Cin; p.H = d->H; p.W = d->W;
  p.splitk = d->splitk; p.accumulate = d->accumulate; p.alpha = 1.f;
  p.lda = d->ldx; p.ldb = d->Cin; p.ldc = d->ldy; p.ldr = d->ldr;
  p.tapstrideB = (long long)d->Cout * d->Cin;
  p.nb1 = 1; p.nbatch = 1;
  if (d->gn_table) {
    OSM_REQUIRE((d->wfmt & ~OSM_WFMT_WINOGRAD) != 0 && d->ksize == 3 && d->W >= 8 && d->H >= 8 && halo_enabled(),
                "osm_conv2d_nhwc: gn_table needs the halo-tile kernel (3x3, split-bf16 weights, W >= 8, H >= 8)");
    OSM_REQUIRE(osm::aligned16(d->gn_table), "osm_conv2d_nhwc: gn_table must be 16-byte aligned");
    p.gn_table = d->gn_table;
    p.gn_silu = d->gn_silu;
  }
  if (d->colsum) {
    OSM_REQUIRE(d->stat_mode == 1 || d->stat_mode == 2, "osm_conv2d_nhwc: stat_mode must be 1 or 2 with colsum");
    OSM_REQUIRE(d->stat_mode == 1 || (d->stat_x && d->stat_table && d->ld_sx >= d->Cout && d->ld_sx % 4 == 0),
                "osm_conv2d_nhwc: stat_mode 2 needs stat_x (ld >= Cout, multiple of 4) and stat_table");
    p.colsum = d->colsum; p.stat_mode = d->stat_mode; p.stat_silu = d->stat_silu;
    p.stat_x = OSM_CACT(d->stat_x); p.ld_sx = d->ld_sx; p.stat_table = d->stat_table;
  }
  const bool wino = (d->wfmt & OSM_WFMT_WINOGRAD) != 0;
  const int wfmt = d->wfmt & ~OSM_WFMT_WINOGRAD;
  OSM_REQUIRE(!wino || (d->ksize == 3 && wfmt >= 1 && wfmt <= 4),
              "osm_conv2d_nhwc: OSM_WFMT_WINOGRAD goes with ksize 3 and wfmt 1 / 2 / 3 / 4");
  p.xmax = d->x_maxabs;
  if (wfmt != 0) {   // split-bf16 fragment image [plane][tap][k16-step][Cout/32][lane][8]
    OSM_REQUIRE(wfmt >= 1 && wfmt <= 4, "osm_conv2d_nhwc: wfmt must be 0 (f32), 1 (fp16), 2 (bf16x3), 3 (bf16x6), or 4 (f16x3)");
    p.nt32 = (d->Cout + 31) / 32;
    p.ksteps = 2 * ((d->Cin + 31) / 32);
  }
  return launch(p, d->ksize * d->ksize, false, (hipStream_t)stream, wfmt, wino);
}

#ifndef OSM_ACT_F16
extern "C" int osm_gemm(const osm_gemm_desc* d, void* stream) {
  OSM_REQUIRE(d && d->A && d->Bm && d->C, "osm_gemm: null pointer");
  OSM_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->nb1 > 0 && d->nb2 > 0, "osm_gemm: bad shape");
  OSM_REQUIRE(d->K % 4 == 0 && d->lda % 4 == 0 && d->ldb % 4 == 0, "osm_gemm: K, lda, ldb must be multiples of 4");
  OSM_REQUIRE(!d->b_kn || d->N % 4 == 0, "osm_gemm: [k][n] B needs N %% 4 == 0");
  OSM_REQUIRE(osm::aligned16(d->A) && osm::aligned16(d->Bm), "osm_gemm: A/B must be 16-byte aligned");
  OSM_REQUIRE(d->sA1 % 4 == 0 && d->sB1 % 4 == 0 && d->sA2 % 4 == 0 && d->sB2 % 4 == 0,
              "osm_gemm: batch strides of A/B must be multiples of 4");
  IGemmParams p{};
  OSM_REQUIRE(d->splitk <= 1 || d->splitk_ws, "osm_gemm: splitk>1 needs a workspace");
  p.A = d->A; p.Bm = d->Bm; p.bias = d->bias; p.res = d->res; p.C = d->C; p.ws = d->splitk_ws;
  p.M = d->M; p.N = d->N; p.K = d->K; p.H = 1; p.W = d->M;
  p.splitk = d->splitk > 1 ? d->splitk : 1; p.accumulate = d->accumulate; p.alpha = d->alpha;
  p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr;
  p.tapstrideB = 0;
  p.nb1 = d->nb1; p.nbatch = d->nb1 * d->nb2;
  p.sA1 = d->sA1; p.sB1 = d->sB1; p.sC1 = d->sC1; p.sA2 = d->sA2; p.sB2 = d->sB2; p.sC2 = d->sC2;
  return launch(p, 1, d->b_kn != 0, (hipStream_t)stream);
}

extern "C" int osm_pack_conv_weight(const float* w, float* wf, float* wd, int Cout, int Cin, int k,
                                    void* stream) {
  OSM_REQUIRE(w && (wf || wd), "osm_pack_conv_weight: null pointer");
  OSM_REQUIRE(k == 1 || k == 3, "osm_pack_conv_weight: ksize must be 1 or 3");
  const long long total = (long long)Cout * Cin * k * k;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, wf, wd, Cout, Cin, k);
  return osm::check_launch("pack_weight_kernel");
}

extern "C" long long osm_packed_weight_elems(int Cout, int Cin, int k, int wfmt, int dgrad) {
  if (wfmt == 0) return (long long)k * k * Cout * Cin;                         // floats
  // wfmt 1 (one fp16 plane), 2, 3 (bf16 planes): 16-bit elements; 4 (f16x3): two half planes + 16 bytes for the scale word
  const int N = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
  const long long per_plane = (long long)k * k * (2 * ((K + 31) / 32)) * ((N + 31) / 32) * 512;
  if (wfmt == 4) return 2 * per_plane + 8;
  return wfmt * per_plane;   // bf16 (uint16) elements
}

extern "C" int osm_pack_conv_weight_bf16s(const float* w, void* w_fwd, void* w_dgrad, int Cout, int Cin, int k,
                                          int wfmt, void* stream) {
  OSM_REQUIRE(w && (w_fwd || w_dgrad), "osm_pack_conv_weight_bf16s: null pointer");
  OSM_REQUIRE(k == 1 || k == 3, "osm_pack_conv_weight_bf16s: ksize must be 1 or 3");
  OSM_REQUIRE(wfmt >= 1 && wfmt <= 4, "osm_pack_conv_weight_bf16s: wfmt must be 1 (fp16), 2 or 3 (bf16 planes), or 4 (f16x3)");
  for (int dg = 0; dg < 2; ++dg) {
    unsigned short* out = reinterpret_cast<unsigned short*>(dg ? w_dgrad : w_fwd);
    if (!out) continue;
    const long long per_plane = (osm_packed_weight_elems(Cout, Cin, k, wfmt, dg) - (wfmt == 4 ? 8 : 0)) / (wfmt == 4 ? 2 : wfmt);
    int blocks = (int)((per_plane + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (wfmt == 4) {     // pass 1: max |w| (as uint bits) into the scale word behind the planes; the pack pass scales by it
      unsigned* sw = reinterpret_cast<unsigned*>(out + 2 * per_plane);
      hipError_t e = hipMemsetAsync(sw, 0, 16, (hipStream_t)stream);
      if (e != hipSuccess) return osm::fail(OSM_ERR_LAUNCH, "osm_pack_conv_weight_bf16s: memset failed");
      hipLaunchKernelGGL(wmax_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, w, sw, (long long)Cout * Cin * k * k);
      hipLaunchKernelGGL(wino_scale_word_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, sw, 0);
    }
    hipLaunchKernelGGL(pack_weight_bf16s_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin,
                       k, wfmt, dg);
    if (wfmt == 4)
      hipLaunchKernelGGL(wino_scale_word_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream,
                         reinterpret_cast<unsigned*>(out + 2 * per_plane), 1);
    int rc = osm::check_launch("pack_weight_bf16s_kernel");
    if (rc) return rc;
  }
  return OSM_OK;
}
#endif   // !OSM_ACT_F16

#if defined(W8_STAMP) && !defined(OSM_ACT_F16)
extern "C" int osm_debug_w8_stamps(unsigned long long* host) {   // measurement build only (tools/w8_stamps.py)
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(osm_w8_stamps), sizeof(unsigned long long) * 64 * 8 * 16);
}
extern "C" int osm_debug_w8_fine_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(osm_w8_fine_stamps), sizeof(unsigned long long) * 8 * 32 * 16);
}
extern "C" int osm_debug_w8_slab_stamps(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(osm_w8_slab_stamps), sizeof(unsigned long long) * 64 * 8 * 160);
}
#endif
