// Flash-style attention core on the matrix cores for the AttentionBlocks with 64-wide heads (T = 64, 256, 1024 tokens:
// the 8x8, 16x16 and 32x32 levels).  Reference: QKVAttentionLegacy.forward / QKVAttention.forward (unet.py:416-433, 459-467):
//     w = softmax(einsum(q * s, k * s)) (in fp32, :431),  a = einsum(w, v),  s = ch^-1/4
// and its autograd (the reference checkpoints the block, unet.py:376 / nn.py:142-170: P is recomputed in backward;
// so is it here, from the saved log-sum-exp).
//
// Round 1 ran this as GEMM -> softmax -> transpose -> GEMM through HBM (5 + 9 launches, logits and probabilities
// written and re-read: 1.8 ms per step).  Here S and P never leave the registers of the wave that made them:
//   * arithmetic: fp32 operands split exactly into 3 bf16 planes, 6 bf16 MFMAs per product (mfma_split.h, the same
//     fp32-class arithmetic as the convolutions); softmax in fp32 on the VALU;
//   * a workgroup = (32 query rows | 32 key rows, head, image); its four waves split the OTHER sequence axis four ways
//     (flash-decoding style), so that B = 1 still yields 256 (T = 1024) / 128 (T = 256) workgroups, and combine
//     through LDS (running max / sum for the forward, plain sums for the gradients);
//   * logits are produced TRANSPOSED (keys x queries for the forward / dq pass): in the 32x32 MFMA C layout a lane then
//     holds 16 keys of ONE query, so the row max / sum of the softmax are in-lane reductions plus one cross-half
//     shuffle, and the same registers ARE the B-operand fragment of the following P V product (the 16 key slots of a
//     k16-step are a permutation of 16 consecutive keys; the V gather uses the same permutation) -- no LDS
//     round trip, no cross-lane traffic between the two GEMMs;
//   * operands come straight from the qkv matrix in L2 (<= 6 MB): row-pattern fragments are two 16-byte loads,
//     gather-pattern fragments (the operand that is contracted over its ROW index) eight 4-byte loads of
//     128-byte-coalesced rows.
#include "osm_common.h"
#include "mfma_split.h"
#include <cstdlib>
#include <type_traits>

namespace {

// NP (template parameter of everything below): 3 = bf16x6 (fp32 operands split exactly into three bf16 planes, six MFMAs per product:
// the fp32-class arithmetic of the fp32 family);  1 = ONE IEEE-half plane, one fp16 MFMA per product with fp32 accumulation: what the
// reference's use_fp16 attention computes (QKVAttentionLegacy on half tensors: fp16 einsum, softmax in fp32, unet.py:426-433) --
// osm_attn_desc.arith = 1, the fp16-storage family (round 4: the core was bf16x6 there too, 6x the matrix work the reference asks for).
constexpr int DH = 64;     // head width

struct FlashArgs {
  const float* qkv;
  long long ldqkv;
  int q_off, k_off, v_off, hs;
  float* out;            // forward: [B*T][ldout], head h at columns h * 64
  long long ldout;
  float* lse;            // [B*heads][T]: max + log(sum) of the scaled logits of a query row
  const float* o;        // backward: the forward output (for delta = rowsum(dO * O))
  long long ldo;
  const float* dout;     // [B*T][lddout]
  long long lddout;
  float* delta;          // [B*heads][T]
  float* dqkv;           // [B*T][lddqkv], same column layout as qkv
  long long lddqkv;
  int B, T, heads;
  int nw;                // waves that split the other sequence axis (nw_of)
  float scale;
};

// X[row][col + 16 s + 8 h + e], e = 0..7  ->  NP fragments (A operand: rows x k; or B operand: k x columns)
template <int NP>
__device__ __forceinline__ void frag_row(const float* __restrict__ X, long long ld, long long row, int col, int s, int h,
                                         float scale, uint4 (&f)[NP]) {
  const float* p = X + row * ld + col + 16 * s + 8 * h;
  float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale;
  b.x *= scale; b.y *= scale; b.z *= scale; b.w *= scale;
  split_frag8<NP>(a, b, f);
}
// X[row0 + 16 s + (e & 3) + 8 (e >> 2) + 4 h][col], e = 0..7: the operand contracted over its row index, in the
// slot order in which a C-layout accumulator hands over the other operand
template <int NP>
__device__ __forceinline__ void frag_gather(const float* __restrict__ X, long long ld, long long row0, int col, int s,
                                            int h, uint4 (&f)[NP]) {
  const float* p = X + (row0 + 16 * s + 4 * h) * ld + col;
  const float4 a = make_float4(p[0], p[ld], p[2 * ld], p[3 * ld]);
  const float4 b = make_float4(p[8 * ld], p[9 * ld], p[10 * ld], p[11 * ld]);
  split_frag8<NP>(a, b, f);
}
// accumulator registers [8 s .. 8 s + 7] -> fragment of step s
template <int NP>
__device__ __forceinline__ void frag_acc(const f32x16& c, int s, uint4 (&f)[NP]) {
  if (s == 0)
    split_frag8<NP>(make_float4(c[0], c[1], c[2], c[3]), make_float4(c[4], c[5], c[6], c[7]), f);
  else
    split_frag8<NP>(make_float4(c[8], c[9], c[10], c[11]), make_float4(c[12], c[13], c[14], c[15]), f);
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int e = 0; e < 16; ++e) z[e] = 0.f;
  return z;
}

// ------------------------------------------------------------------------------------------------ forward
// grid (T / 32, heads, B).  NW = min(4, T / 32) waves split the keys: wave w < NW takes keys [w T/NW, (w+1) T/NW),
// 32 NSUB at a time (NSUB = 2 independent 32-key logit tiles where the range allows: two MFMA accumulator chains).
// WV = waves per workgroup (4, or 8 = two per SIMD: at B = 1 the grid is one workgroup per CU, so the second wave per SIMD is the
// only latency hiding there is -- every fragment goes global load -> split -> MFMA with nothing prefetched).
template <int NP, int NSUB, int WV>
__global__ __launch_bounds__(64 * WV, 1) void flash_fwd_kernel(FlashArgs a) {
  __shared__ float os[WV][DH][33];
  __shared__ float ms[WV][32];
  __shared__ float ls[WV][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * 32, hd = blockIdx.y;
  const long long rb = (long long)blockIdx.z * a.T;
  const float* __restrict__ Q = a.qkv + a.q_off + hd * a.hs;
  const float* __restrict__ K = a.qkv + a.k_off + hd * a.hs;
  const float* __restrict__ V = a.qkv + a.v_off + hd * a.hs;
  const long long ld = a.ldqkv;

  uint4 qf[4][NP];   // B operand of S^T = K (scale Q)^T: k = d, column = this lane's query
#pragma unroll
  for (int s = 0; s < 4; ++s) frag_row(Q, ld, rb + q0 + lr, 0, s, h, a.scale, qf[s]);

  f32x16 o0 = zero16(), o1 = zero16();   // O^T: rows d (0..31 | 32..63), column = query
  float m = -INFINITY, l = 0.f;
  const int kw = a.T / a.nw;
  const int kend = wave < a.nw ? (wave + 1) * kw : 0;
  for (int kb = wave * kw; kb < kend; kb += 32 * NSUB) {
    f32x16 sa[NSUB];   // S^T of keys kb + 32 u .. + 31
#pragma unroll
    for (int u = 0; u < NSUB; ++u) sa[u] = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 kf[NSUB][NP];
#pragma unroll
      for (int u = 0; u < NSUB; ++u) frag_row(K, ld, rb + kb + 32 * u + lr, 0, s, h, 1.f, kf[u]);
#pragma unroll
      for (int pa = NP - 1; pa >= 0; --pa)
#pragma unroll
        for (int pb = NP - 1 - pa; pb >= 0; --pb)
#pragma unroll
          for (int u = 0; u < NSUB; ++u) sa[u] = mma16<NP>(kf[u][pa], qf[s][pb], sa[u]);
    }
    // online softmax of this lane's query over its 16 NSUB keys (the other 16 NSUB sit in lane ^ 32)
    float mx = sa[0][0];
#pragma unroll
    for (int u = 0; u < NSUB; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sa[u][e]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float alpha = __expf(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int u = 0; u < NSUB; ++u)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        sa[u][e] = __expf(sa[u][e] - mn);
        ps += sa[u][e];
      }
    l = l * alpha + ps;
    m = mn;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      o0[e] *= alpha;
      o1[e] *= alpha;
    }
    // O^T += V^T P^T : A = V^T (rows d, key slots), B = P^T straight from the S^T registers
#pragma unroll
    for (int t = 0; t < 2 * NSUB; ++t) {
      uint4 pf[NP], v0[NP], v1[NP];
      frag_acc(sa[t >> 1], t & 1, pf);
      frag_gather(V, ld, rb + kb + 32 * (t >> 1), lr, t & 1, h, v0);
      frag_gather(V, ld, rb + kb + 32 * (t >> 1), 32 + lr, t & 1, h, v1);
#pragma unroll
      for (int pa = NP - 1; pa >= 0; --pa)
#pragma unroll
        for (int pb = NP - 1 - pa; pb >= 0; --pb) {
          o0 = mma16<NP>(v0[pa], pf[pb], o0);
          o1 = mma16<NP>(v1[pa], pf[pb], o1);
        }
    }
  }
  // ---- combine the key ranges.  O^T C layout: column = query lr, row d = 32 mb + (e&3) + 8 (e>>2) + 4 h
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int d = (e & 3) + 8 * (e >> 2) + 4 * h;
    os[wave][d][lr] = o0[e];
    os[wave][32 + d][lr] = o1[e];
  }
  if (h == 0) ms[wave][lr] = m;
  ls[wave][lane] = l;
  __syncthreads();
  const int d = tid & 63, qg = tid >> 6;
#pragma unroll
  for (int i = 0; i < 32 / WV; ++i) {
    const int q = qg * (32 / WV) + i;
    float M = ms[0][q];
#pragma unroll
    for (int w = 1; w < WV; ++w) M = fmaxf(M, ms[w][q]);
    float L = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < WV; ++w) {
      const float sc = __expf(ms[w][q] - M);     // idle waves: exp(-inf) = 0
      L += (ls[w][q] + ls[w][q + 32]) * sc;
      o += os[w][d][q] * sc;
    }
    a.out[(rb + q0 + q) * a.ldout + hd * DH + d] = o / L;
    if (d == 0) a.lse[((long long)blockIdx.z * a.heads + hd) * a.T + q0 + q] = M + logf(L);
  }
}

// ------------------------------------------------------------------------------------------------ backward
// dq: grid (T / 32, heads, B): workgroup = 32 queries, waves split the keys.
//   S^T = K (scale Q)^T, P^T = exp(S^T - lse[q]);  dP^T = V dO^T;  dS^T = P^T (dP^T - delta[q]);
//   dq^T[d][q] = scale * sum_key K^T[d][key] dS^T[key][q]
template <int NP, int WV>
__global__ __launch_bounds__(64 * WV, 1) void flash_bwd_q_kernel(FlashArgs a) {
  __shared__ float os[WV][DH][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * 32, hd = blockIdx.y;
  const long long rb = (long long)blockIdx.z * a.T;
  const float* __restrict__ Q = a.qkv + a.q_off + hd * a.hs;
  const float* __restrict__ K = a.qkv + a.k_off + hd * a.hs;
  const float* __restrict__ V = a.qkv + a.v_off + hd * a.hs;
  const float* __restrict__ dO = a.dout + hd * DH;
  const long long ld = a.ldqkv;
  const long long stat = ((long long)blockIdx.z * a.heads + hd) * a.T + q0 + lr;
  const float lse = a.lse[stat];

  uint4 qf[4][NP], gf[4][NP];   // B operands (k = d, column = query): scale * Q and dO
  // delta[q] = sum_d dO[q][d] O[q][d] of this lane's query: the lane reads 32 of its 64 dims for the dO fragments anyway (the other
  // 32 sit in lane ^ 32), so every wave forms it for itself and wave 0 leaves it in a.delta for the dk / dv kernel that follows --
  // a launch of its own (flash_delta_kernel, ~5 us x 16 per step) is not needed
  float dl = 0.f;
  {
    const float* __restrict__ Op = a.o + (rb + q0 + lr) * a.ldo + hd * DH;
    const float* __restrict__ Gp = dO + (rb + q0 + lr) * a.lddout;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 o0 = *reinterpret_cast<const float4*>(Op + 16 * s + 8 * h), o1 = *reinterpret_cast<const float4*>(Op + 16 * s + 8 * h + 4);
      const float4 g0 = *reinterpret_cast<const float4*>(Gp + 16 * s + 8 * h), g1 = *reinterpret_cast<const float4*>(Gp + 16 * s + 8 * h + 4);
      dl += (g0.x * o0.x + g0.y * o0.y) + (g0.z * o0.z + g0.w * o0.w) + (g1.x * o1.x + g1.y * o1.y) + (g1.z * o1.z + g1.w * o1.w);
    }
    dl += __shfl_xor(dl, 32, 64);
    if (wave == 0 && h == 0) a.delta[stat] = dl;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    frag_row(Q, ld, rb + q0 + lr, 0, s, h, a.scale, qf[s]);
    frag_row(dO, a.lddout, rb + q0 + lr, 0, s, h, 1.f, gf[s]);
  }
  f32x16 g0 = zero16(), g1 = zero16();   // dq^T rows d (0..31 | 32..63), column = query
  const int kw = a.T / a.nw;
  const int kend = wave < a.nw ? (wave + 1) * kw : 0;
  for (int kb = wave * kw; kb < kend; kb += 32) {
    f32x16 st = zero16(), dp = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 kf[NP], vf[NP];
      frag_row(K, ld, rb + kb + lr, 0, s, h, 1.f, kf);
      frag_row(V, ld, rb + kb + lr, 0, s, h, 1.f, vf);
#pragma unroll
      for (int pa = NP - 1; pa >= 0; --pa)
#pragma unroll
        for (int pb = NP - 1 - pa; pb >= 0; --pb) {
          st = mma16<NP>(kf[pa], qf[s][pb], st);
          dp = mma16<NP>(vf[pa], gf[s][pb], dp);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = __expf(st[e] - lse) * (dp[e] - dl);   // dS^T
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      uint4 sf[NP], k0[NP], k1[NP];
      frag_acc(st, t, sf);
      frag_gather(K, ld, rb + kb, lr, t, h, k0);
      frag_gather(K, ld, rb + kb, 32 + lr, t, h, k1);
#pragma unroll
      for (int pa = NP - 1; pa >= 0; --pa)
#pragma unroll
        for (int pb = NP - 1 - pa; pb >= 0; --pb) {
          g0 = mma16<NP>(k0[pa], sf[pb], g0);
          g1 = mma16<NP>(k1[pa], sf[pb], g1);
        }
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int d = (e & 3) + 8 * (e >> 2) + 4 * h;
    os[wave][d][lr] = g0[e];
    os[wave][32 + d][lr] = g1[e];
  }
  __syncthreads();
  const int d = tid & 63, qg = tid >> 6;
#pragma unroll
  for (int i = 0; i < 32 / WV; ++i) {
    const int q = qg * (32 / WV) + i;
    float v = (os[0][d][q] + os[1][d][q]) + (os[2][d][q] + os[3][d][q]);
    if (WV == 8) v += (os[WV - 4][d][q] + os[WV - 3][d][q]) + (os[WV - 2][d][q] + os[WV - 1][d][q]);
    a.dqkv[(rb + q0 + q) * a.lddqkv + a.q_off + hd * a.hs + d] = v * a.scale;
  }
}

// dk, dv: grid (T / 32, heads, B): workgroup = 32 keys, waves split the queries.
//   S = (scale Q) K^T (queries x keys: column = this lane's key), P = exp(S - lse[q]);  dP = dO V^T;  dS = P (dP - delta[q]);
//   dv^T[d][key] = sum_q dO^T[d][q] P[q][key];   dk^T[d][key] = scale * sum_q Q^T[d][q] dS[q][key]
// KVL: the K / V fragments of the workgroup's 32 keys (2 x 4 k16-steps x NP planes, 24 KB for bf16x6) live in LDS -- aliased with the
// combine buffer, which is only written after the loop -- instead of 96 registers per lane: the bf16x6 instance then fits two waves
// per SIMD (WV = 8: 250 registers; with the fragments resident it needs 368 and runs one wave per SIMD with nothing to hide its loads).
template <int NP, int WV, bool KVL = false>
__global__ __launch_bounds__(64 * WV, 1) void flash_bwd_kv_kernel(FlashArgs a) {
  __shared__ __attribute__((aligned(16))) float os[WV][2 * DH][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, h = lane >> 5;
  const int k0r = blockIdx.x * 32, hd = blockIdx.y;
  const long long rb = (long long)blockIdx.z * a.T;
  const float* __restrict__ Q = a.qkv + a.q_off + hd * a.hs;
  const float* __restrict__ K = a.qkv + a.k_off + hd * a.hs;
  const float* __restrict__ V = a.qkv + a.v_off + hd * a.hs;
  const float* __restrict__ dO = a.dout + hd * DH;
  const long long ld = a.ldqkv;
  const float* __restrict__ lsep = a.lse + ((long long)blockIdx.z * a.heads + hd) * a.T;
  const float* __restrict__ dlp = a.delta + ((long long)blockIdx.z * a.heads + hd) * a.T;

  uint4 kf[KVL ? 1 : 4][NP], vf[KVL ? 1 : 4][NP];   // B operands (k = d, column = key); KVL: one k16-step at a time, from LDS
  uint4* kvs = reinterpret_cast<uint4*>(&os[0][0][0]);      // KVL: [K | V][k16-step][plane][lane]
  if constexpr (KVL) {
    static_assert(sizeof(os) >= 2 * 4 * NP * 64 * sizeof(uint4), "the fragments alias the combine buffer");
    for (int f = wave; f < 8; f += WV) {           // fragment f = (K | V, k16-step): split once per workgroup, by one wave each
      uint4 t[NP];
      frag_row(f < 4 ? K : V, ld, rb + k0r + lr, 0, f & 3, h, 1.f, t);
#pragma unroll
      for (int q = 0; q < NP; ++q) kvs[(f * NP + q) * 64 + lane] = t[q];
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      frag_row(K, ld, rb + k0r + lr, 0, s, h, 1.f, kf[s]);
      frag_row(V, ld, rb + k0r + lr, 0, s, h, 1.f, vf[s]);
    }
  }
  f32x16 dv0 = zero16(), dv1 = zero16(), dk0 = zero16(), dk1 = zero16();
  const int qw = a.T / a.nw;
  const int qend = wave < a.nw ? (wave + 1) * qw : 0;
  for (int qb = wave * qw; qb < qend; qb += 32) {
    f32x16 sc = zero16(), dp = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      uint4 qf[NP], gf[NP];
      frag_row(Q, ld, rb + qb + lr, 0, s, h, a.scale, qf);
      frag_row(dO, a.lddout, rb + qb + lr, 0, s, h, 1.f, gf);
      if constexpr (KVL) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          kf[0][q] = kvs[(s * NP + q) * 64 + lane];
          vf[0][q] = kvs[((4 + s) * NP + q) * 64 + lane];
        }
      }
      const int si = KVL ? 0 : s;
#pragma unroll
      for (int pa = NP - 1; pa >= 0; --pa)
#pragma unroll
        for (int pb = NP - 1 - pa; pb >= 0; --pb) {
          sc = mma16<NP>(qf[pa], kf[si][pb], sc);
          dp = mma16<NP>(gf[pa], vf[si][pb], dp);
        }
    }
    // rows of the C layout are queries: q = qb + (e&3) + 8 (e>>2) + 4 h
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 l4 = *reinterpret_cast<const float4*>(lsep + qb + 8 * j + 4 * h);
      const float4 d4 = *reinterpret_cast<const float4*>(dlp + qb + 8 * j + 4 * h);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pr = __expf(sc[4 * j + i] - lv[i]);
        sc[4 * j + i] = pr;                          // P
        dp[4 * j + i] = pr * (dp[4 * j + i] - dv[i]);   // dS
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      {   // dv^T += dO^T P        (the two halves of a step kept apart: fewer fragments live at once -- no spill at 256 registers)
        uint4 pf[NP], g0[NP], g1[NP];
        frag_acc(sc, t, pf);
        frag_gather(dO, a.lddout, rb + qb, lr, t, h, g0);
        frag_gather(dO, a.lddout, rb + qb, 32 + lr, t, h, g1);
#pragma unroll
        for (int pa = NP - 1; pa >= 0; --pa)
#pragma unroll
          for (int pb = NP - 1 - pa; pb >= 0; --pb) {
            dv0 = mma16<NP>(g0[pa], pf[pb], dv0);
            dv1 = mma16<NP>(g1[pa], pf[pb], dv1);
          }
      }
      if constexpr (KVL) __builtin_amdgcn_sched_barrier(0);
      {   // dk^T += Q^T dS
        uint4 sf[NP], q0f[NP], q1f[NP];
        frag_acc(dp, t, sf);
        frag_gather(Q, ld, rb + qb, lr, t, h, q0f);
        frag_gather(Q, ld, rb + qb, 32 + lr, t, h, q1f);
#pragma unroll
        for (int pa = NP - 1; pa >= 0; --pa)
#pragma unroll
          for (int pb = NP - 1 - pa; pb >= 0; --pb) {
            dk0 = mma16<NP>(q0f[pa], sf[pb], dk0);
            dk1 = mma16<NP>(q1f[pa], sf[pb], dk1);
          }
      }
      if constexpr (KVL) __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (KVL) __syncthreads();     // every wave's last fragment read is over before the combine buffer overwrites them
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int d = (e & 3) + 8 * (e >> 2) + 4 * h;
    os[wave][d][lr] = dv0[e];
    os[wave][32 + d][lr] = dv1[e];
    os[wave][64 + d][lr] = dk0[e];
    os[wave][96 + d][lr] = dk1[e];
  }
  __syncthreads();
  const int d = tid & 63, kg = tid >> 6;
#pragma unroll
  for (int i = 0; i < 32 / WV; ++i) {
    const int k = kg * (32 / WV) + i;
    float dvv = (os[0][d][k] + os[1][d][k]) + (os[2][d][k] + os[3][d][k]);
    float dkv = (os[0][64 + d][k] + os[1][64 + d][k]) + (os[2][64 + d][k] + os[3][64 + d][k]);
    if (WV == 8) {
      dvv += (os[WV - 4][d][k] + os[WV - 3][d][k]) + (os[WV - 2][d][k] + os[WV - 1][d][k]);
      dkv += (os[WV - 4][64 + d][k] + os[WV - 3][64 + d][k]) + (os[WV - 2][64 + d][k] + os[WV - 1][64 + d][k]);
    }
    float* row = a.dqkv + (rb + k0r + k) * a.lddqkv + hd * a.hs;
    row[a.v_off + d] = dvv;
    row[a.k_off + d] = dkv * a.scale;
  }
}

// ================================================================================================ f16x3 ("HP") instances
// The same three kernels in the convolutions' default arithmetic (mfma_split.h, round 3): every fp32 operand is scaled by a power
// of two into the fp16 range and split into TWO IEEE-half terms (~22 bits), a product is three fp16 MFMAs (h0 g0 + h0 g1 + h1 g0)
// instead of six bf16 ones, and a split costs 6 VALU per 4 values instead of 22.  osm_attn_desc.arith = 2.
// Operand ranges are found IN the kernels (no side channel from the producers): a 32 x 64 operand tile is spread over exactly
// one wave, so its max |.| is the lane's max over its 32 values and one 6-step butterfly; the power-of-two scale is undone
// exactly on the accumulator.  Products that ACCUMULATE over tiles whose scales differ (O += P V, dq += dS K, dv += P^T dO,
// dk += dS^T Q) keep a running scale exponent -- the smallest seen, i.e. the largest operand -- and rescale the accumulator by an
// exact power of two when it falls, as the online softmax does with its running max (forward: folded into its alpha).  P is in
// [0, 1] and needs no scale; dS = P (dP - delta) is scaled per COLUMN of the C layout (one query / key = one lane and its
// partner lane ^ 32: a column scale of a B operand factors out of the product per accumulator column, lane-locally).
__device__ __forceinline__ unsigned abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ unsigned wave_max_bits(unsigned v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
  return v;
}
// e with max * 2^e in [2^11, 2^12); 0 for zero / denormal / non-finite maxima (a NaN or Inf operand then poisons the result)
__device__ __forceinline__ int scale_exp_of(unsigned maxbits) {
  const int ex = (int)(maxbits >> 23);
  if (ex == 0 || ex == 255) return 0;
  return min(max(11 - (ex - 127), -100), 100);
}
__device__ __forceinline__ float pow2i(int e) { return __uint_as_float((unsigned)(127 + min(max(e, -126), 127)) << 23); }
__device__ __forceinline__ unsigned max8_bits(const float4& a, const float4& b, unsigned m) {
  m = max(m, max(max(abs_bits(a.x), abs_bits(a.y)), max(abs_bits(a.z), abs_bits(a.w))));
  return max(m, max(max(abs_bits(b.x), abs_bits(b.y)), max(abs_bits(b.z), abs_bits(b.w))));
}
struct Raw8 { float4 a, b; };
__device__ __forceinline__ Raw8 raw_row(const float* __restrict__ X, long long ld, long long row, int col, int s, int h) {
  const float* p = X + row * ld + col + 16 * s + 8 * h;
  return Raw8{*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4)};
}
__device__ __forceinline__ Raw8 raw_gather(const float* __restrict__ X, long long ld, long long row0, int col, int s, int h) {
  const float* p = X + (row0 + 16 * s + 4 * h) * ld + col;
  return Raw8{make_float4(p[0], p[ld], p[2 * ld], p[3 * ld]), make_float4(p[8 * ld], p[9 * ld], p[10 * ld], p[11 * ld])};
}
// 8 fp32 x sc -> the two half planes of one fragment
__device__ __forceinline__ void split_hp(const Raw8& r, float sc, uint4 (&f)[2]) {
  uint2 lo[2], hi[2];
  split_f16x2(make_float4(r.a.x * sc, r.a.y * sc, r.a.z * sc, r.a.w * sc), lo);
  split_f16x2(make_float4(r.b.x * sc, r.b.y * sc, r.b.z * sc, r.b.w * sc), hi);
  f[0] = make_uint4(lo[0].x, lo[0].y, hi[0].x, hi[0].y);
  f[1] = make_uint4(lo[1].x, lo[1].y, hi[1].x, hi[1].y);
}
// accumulator registers [8 s .. 8 s + 7] x sc -> fragment of step s
__device__ __forceinline__ void split_acc_hp(const f32x16& c, int s, float sc, uint4 (&f)[2]) {
  const Raw8 r = s == 0 ? Raw8{make_float4(c[0], c[1], c[2], c[3]), make_float4(c[4], c[5], c[6], c[7])}
                        : Raw8{make_float4(c[8], c[9], c[10], c[11]), make_float4(c[12], c[13], c[14], c[15])};
  split_hp(r, sc, f);
}
// c += (a0 + a1)(b0 + b1) without a1 b1: smallest terms first
__device__ __forceinline__ f32x16 mma_hp(const uint4 (&a)[2], const uint4 (&b)[2], f32x16 c) {
  c = mma16h(a[1], b[0], c);
  c = mma16h(a[0], b[1], c);
  return mma16h(a[0], b[0], c);
}

template <int WV>
__global__ __launch_bounds__(64 * WV, 1) void flash_fwd_hp_kernel(FlashArgs a) {
  __shared__ float os[WV][DH][33];
  __shared__ float ms[WV][32];
  __shared__ float ls[WV][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * 32, hd = blockIdx.y;
  const long long rb = (long long)blockIdx.z * a.T;
  const float* __restrict__ Q = a.qkv + a.q_off + hd * a.hs;
  const float* __restrict__ K = a.qkv + a.k_off + hd * a.hs;
  const float* __restrict__ V = a.qkv + a.v_off + hd * a.hs;
  const long long ld = a.ldqkv;

  uint4 qf[4][2];   // B operand of S^T = K (scale Q)^T: k = d, column = this lane's query; x 2^eq
  int eq;
  {
    Raw8 r[4];
    unsigned m = 0u;
#pragma unroll
    for (int s = 0; s < 4; ++s) { r[s] = raw_row(Q, ld, rb + q0 + lr, 0, s, h); m = max8_bits(r[s].a, r[s].b, m); }
    eq = scale_exp_of(abs_bits(__uint_as_float(wave_max_bits(m)) * a.scale));
    const float sc = a.scale * pow2i(eq);
#pragma unroll
    for (int s = 0; s < 4; ++s) split_hp(r[s], sc, qf[s]);
  }
  f32x16 o0 = zero16(), o1 = zero16();   // O^T x 2^ev: rows d (0..31 | 32..63), column = query
  int ev = 100;                          // running scale exponent of the V operand (falls as larger |V| tiles arrive)
  float m = -INFINITY, l = 0.f;
  const int kw = a.T / a.nw;
  const int kend = wave < a.nw ? (wave + 1) * kw : 0;
  for (int kb = wave * kw; kb < kend; kb += 32) {
    f32x16 sa = zero16();   // S^T of keys kb .. kb + 31, x 2^(ek + eq)
    float cs;               // 2^-(ek + eq)
    {
      Raw8 r[4];
      unsigned mk = 0u;
#pragma unroll
      for (int s = 0; s < 4; ++s) { r[s] = raw_row(K, ld, rb + kb + lr, 0, s, h); mk = max8_bits(r[s].a, r[s].b, mk); }
      const int ek = scale_exp_of(wave_max_bits(mk));
      const float sc = pow2i(ek);
      cs = pow2i(-(ek + eq));
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        uint4 kf[2];
        split_hp(r[s], sc, kf);
        sa = mma_hp(kf, qf[s], sa);
      }
    }
    // the V tile (both k16-steps, both halves of d) goes out before the softmax: its latency runs under it
    Raw8 rv[2][2];
    unsigned mv = 0u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      rv[t][0] = raw_gather(V, ld, rb + kb, lr, t, h);
      rv[t][1] = raw_gather(V, ld, rb + kb, 32 + lr, t, h);
    }
    // online softmax of this lane's query over its 16 keys (the other 16 sit in lane ^ 32)
    float mx = sa[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) mx = fmaxf(mx, sa[e]);
    mx *= cs;
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    float alpha = __expf(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      sa[e] = __expf(fmaf(sa[e], cs, -mn));
      ps += sa[e];
    }
    l = l * alpha + ps;
    m = mn;
#pragma unroll
    for (int t = 0; t < 2; ++t) mv = max8_bits(rv[t][1].a, rv[t][1].b, max8_bits(rv[t][0].a, rv[t][0].b, mv));
    const int evn = min(ev, scale_exp_of(wave_max_bits(mv)));
    alpha *= pow2i(evn - ev);          // the accumulator moves to the new (smaller) scale with the softmax's own rescale
    ev = evn;
    const float sv = pow2i(ev);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      o0[e] *= alpha;
      o1[e] *= alpha;
    }
    // O^T += V^T P^T : A = V^T (rows d, key slots) x 2^ev, B = P^T straight from the S^T registers (in [0, 1]: no scale)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      uint4 pf[2], v0[2], v1[2];
      split_acc_hp(sa, t, 1.f, pf);
      split_hp(rv[t][0], sv, v0);
      split_hp(rv[t][1], sv, v1);
      o0 = mma_hp(v0, pf, o0);
      o1 = mma_hp(v1, pf, o1);
    }
  }
  const float un = pow2i(-ev);           // (an idle wave: 0 x 2^-100)
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int d = (e & 3) + 8 * (e >> 2) + 4 * h;
    os[wave][d][lr] = o0[e] * un;
    os[wave][32 + d][lr] = o1[e] * un;
  }
  if (h == 0) ms[wave][lr] = m;
  ls[wave][lane] = l;
  __syncthreads();
  const int d = tid & 63, qg = tid >> 6;
#pragma unroll
  for (int i = 0; i < 32 / WV; ++i) {
    const int q = qg * (32 / WV) + i;
    float M = ms[0][q];
#pragma unroll
    for (int w = 1; w < WV; ++w) M = fmaxf(M, ms[w][q]);
    float L = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < WV; ++w) {
      const float sc = __expf(ms[w][q] - M);     // idle waves: exp(-inf) = 0
      L += (ls[w][q] + ls[w][q + 32]) * sc;
      o += os[w][d][q] * sc;
    }
    a.out[(rb + q0 + q) * a.ldout + hd * DH + d] = o / L;
    if (d == 0) a.lse[((long long)blockIdx.z * a.heads + hd) * a.T + q0 + q] = M + logf(L);
  }
}

template <int WV>
__global__ __launch_bounds__(64 * WV, 1) void flash_bwd_q_hp_kernel(FlashArgs a) {
  __shared__ float os[WV][DH][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, h = lane >> 5;
  const int q0 = blockIdx.x * 32, hd = blockIdx.y;
  const long long rb = (long long)blockIdx.z * a.T;
  const float* __restrict__ Q = a.qkv + a.q_off + hd * a.hs;
  const float* __restrict__ K = a.qkv + a.k_off + hd * a.hs;
  const float* __restrict__ V = a.qkv + a.v_off + hd * a.hs;
  const float* __restrict__ dO = a.dout + hd * DH;
  const long long ld = a.ldqkv;
  const long long stat = ((long long)blockIdx.z * a.heads + hd) * a.T + q0 + lr;
  const float lse = a.lse[stat];

  uint4 qf[4][2], gf[4][2];   // B operands (k = d, column = query): scale * Q x 2^eq and dO x 2^eg
  int eq, eg;
  float dl = 0.f;
  {
    const float* __restrict__ Op = a.o + (rb + q0 + lr) * a.ldo + hd * DH;
    Raw8 rq[4], rg[4];
    unsigned mq = 0u, mg = 0u;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      rq[s] = raw_row(Q, ld, rb + q0 + lr, 0, s, h);
      rg[s] = raw_row(dO, a.lddout, rb + q0 + lr, 0, s, h);
      mq = max8_bits(rq[s].a, rq[s].b, mq);
      mg = max8_bits(rg[s].a, rg[s].b, mg);
      const float4 o0 = *reinterpret_cast<const float4*>(Op + 16 * s + 8 * h), o1 = *reinterpret_cast<const float4*>(Op + 16 * s + 8 * h + 4);
      const float4 g0 = rg[s].a, g1 = rg[s].b;
      dl += (g0.x * o0.x + g0.y * o0.y) + (g0.z * o0.z + g0.w * o0.w) + (g1.x * o1.x + g1.y * o1.y) + (g1.z * o1.z + g1.w * o1.w);
    }
    dl += __shfl_xor(dl, 32, 64);
    if (wave == 0 && h == 0) a.delta[stat] = dl;
    eq = scale_exp_of(abs_bits(__uint_as_float(wave_max_bits(mq)) * a.scale));
    eg = scale_exp_of(wave_max_bits(mg));
    const float sq = a.scale * pow2i(eq), sg = pow2i(eg);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      split_hp(rq[s], sq, qf[s]);
      split_hp(rg[s], sg, gf[s]);
    }
  }
  f32x16 g0 = zero16(), g1 = zero16();   // dq^T x 2^er: rows d (0..31 | 32..63), column = query
  int er = 200;                          // running scale exponent of this lane's column (K tile scale + dS column scale)
  const int kw = a.T / a.nw;
  const int kend = wave < a.nw ? (wave + 1) * kw : 0;
  for (int kb = wave * kw; kb < kend; kb += 32) {
    f32x16 st = zero16(), dp = zero16();
    Raw8 rk[4];
    int ek;
    float c1, c2;
    {
      Raw8 rvv[4];
      unsigned mk = 0u, mv = 0u;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        rk[s] = raw_row(K, ld, rb + kb + lr, 0, s, h);
        rvv[s] = raw_row(V, ld, rb + kb + lr, 0, s, h);
        mk = max8_bits(rk[s].a, rk[s].b, mk);
        mv = max8_bits(rvv[s].a, rvv[s].b, mv);
      }
      ek = scale_exp_of(wave_max_bits(mk));
      const int ev = scale_exp_of(wave_max_bits(mv));
      const float sk = pow2i(ek), sv = pow2i(ev);
      c1 = pow2i(-(ek + eq));
      c2 = pow2i(-(ev + eg));
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        uint4 kf[2], vf[2];
        split_hp(rk[s], sk, kf);
        split_hp(rvv[s], sv, vf);
        st = mma_hp(kf, qf[s], st);
        dp = mma_hp(vf, gf[s], dp);
      }
    }
    // the gather fragments of the K tile (the same 32 x 64 values, contracted over the keys) go out before the exponentials
    Raw8 rg2[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      rg2[t][0] = raw_gather(K, ld, rb + kb, lr, t, h);
      rg2[t][1] = raw_gather(K, ld, rb + kb, 32 + lr, t, h);
    }
    unsigned md = 0u;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      st[e] = __expf(fmaf(st[e], c1, -lse)) * fmaf(dp[e], c2, -dl);   // dS^T
      md = max(md, abs_bits(st[e]));
    }
    md = max(md, (unsigned)__shfl_xor((int)md, 32, 64));               // the column's other 16 keys
    const int ern = min(er, ek + scale_exp_of(md));
    if (__any(ern != er)) {
      const float f = pow2i(ern - er);
#pragma unroll
      for (int e = 0; e < 16; ++e) { g0[e] *= f; g1[e] *= f; }
    }
    er = ern;
    const float sd = pow2i(er - ek), sk = pow2i(ek);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      uint4 sf[2], k0[2], k1[2];
      split_acc_hp(st, t, sd, sf);
      split_hp(rg2[t][0], sk, k0);
      split_hp(rg2[t][1], sk, k1);
      g0 = mma_hp(k0, sf, g0);
      g1 = mma_hp(k1, sf, g1);
    }
  }
  const float un = pow2i(-er);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int d = (e & 3) + 8 * (e >> 2) + 4 * h;
    os[wave][d][lr] = g0[e] * un;
    os[wave][32 + d][lr] = g1[e] * un;
  }
  __syncthreads();
  const int d = tid & 63, qg = tid >> 6;
#pragma unroll
  for (int i = 0; i < 32 / WV; ++i) {
    const int q = qg * (32 / WV) + i;
    float v = (os[0][d][q] + os[1][d][q]) + (os[2][d][q] + os[3][d][q]);
    if (WV == 8) v += (os[WV - 4][d][q] + os[WV - 3][d][q]) + (os[WV - 2][d][q] + os[WV - 1][d][q]);
    a.dqkv[(rb + q0 + q) * a.lddqkv + a.q_off + hd * a.hs + d] = v * a.scale;
  }
}

// dk, dv (f16x3): the K / V fragments of the workgroup's 32 keys are split once and live in LDS (16 KB), as in the KVL form above
template <int WV>
__global__ __launch_bounds__(64 * WV, 1) void flash_bwd_kv_hp_kernel(FlashArgs a) {
  __shared__ __attribute__((aligned(16))) float os[WV][2 * DH][33];
  __shared__ unsigned kvmax[2][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, h = lane >> 5;
  const int k0r = blockIdx.x * 32, hd = blockIdx.y;
  const long long rb = (long long)blockIdx.z * a.T;
  const float* __restrict__ Q = a.qkv + a.q_off + hd * a.hs;
  const float* __restrict__ K = a.qkv + a.k_off + hd * a.hs;
  const float* __restrict__ V = a.qkv + a.v_off + hd * a.hs;
  const float* __restrict__ dO = a.dout + hd * DH;
  const long long ld = a.ldqkv;
  const float* __restrict__ lsep = a.lse + ((long long)blockIdx.z * a.heads + hd) * a.T;
  const float* __restrict__ dlp = a.delta + ((long long)blockIdx.z * a.heads + hd) * a.T;

  uint4* kvs = reinterpret_cast<uint4*>(&os[0][0][0]);      // [K | V][k16-step][plane][lane]
  static_assert(sizeof(os) >= 2 * 4 * 2 * 64 * sizeof(uint4), "the fragments alias the combine buffer");
  int ek, ev;
  {
    // fragment f = (K | V, k16-step): loaded by wave f % WV; its max goes through LDS so that every wave knows both tile maxima
    static_assert(WV == 4 || WV == 8, "fragments per wave");
    constexpr int PER = 8 / WV;
    Raw8 r[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int f = wave + j * WV;
      if (f < 8) {
        r[j] = raw_row(f < 4 ? K : V, ld, rb + k0r + lr, 0, f & 3, h);
        const unsigned m = wave_max_bits(max8_bits(r[j].a, r[j].b, 0u));
        if (lane == 0) kvmax[f >> 2][f & 3] = m;
      }
    }
    __syncthreads();
    ek = scale_exp_of(max(max(kvmax[0][0], kvmax[0][1]), max(kvmax[0][2], kvmax[0][3])));
    ev = scale_exp_of(max(max(kvmax[1][0], kvmax[1][1]), max(kvmax[1][2], kvmax[1][3])));
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int f = wave + j * WV;
      if (f < 8) {
        uint4 t[2];
        split_hp(r[j], pow2i(f < 4 ? ek : ev), t);
        kvs[(f * 2 + 0) * 64 + lane] = t[0];
        kvs[(f * 2 + 1) * 64 + lane] = t[1];
      }
    }
    __syncthreads();
  }
  f32x16 dv0 = zero16(), dv1 = zero16(), dk0 = zero16(), dk1 = zero16();
  int erv = 200;        // running scale exponent of the dv accumulators (the dO tile's; wave-uniform)
  int erk = 200;        // ... of this lane's column of the dk accumulators (Q tile scale + dS column scale)
  const int qw = a.T / a.nw;
  const int qend = wave < a.nw ? (wave + 1) * qw : 0;
  for (int qb = wave * qw; qb < qend; qb += 32) {
    f32x16 sc = zero16(), dp = zero16();
    int eq, eg;
    float c1, c2;
    {
      Raw8 rq[4], rg[4];
      unsigned mq = 0u, mg = 0u;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        rq[s] = raw_row(Q, ld, rb + qb + lr, 0, s, h);
        rg[s] = raw_row(dO, a.lddout, rb + qb + lr, 0, s, h);
        mq = max8_bits(rq[s].a, rq[s].b, mq);
        mg = max8_bits(rg[s].a, rg[s].b, mg);
      }
      eq = scale_exp_of(wave_max_bits(mq));          // of the unscaled Q tile (the gather below uses the same exponent)
      eg = scale_exp_of(wave_max_bits(mg));
      const float sq = a.scale * pow2i(eq), sg = pow2i(eg);
      c1 = pow2i(-(eq + ek));
      c2 = pow2i(-(eg + ev));
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        uint4 qf[2], gf[2], kf[2], vf[2];
        split_hp(rq[s], sq, qf);
        split_hp(rg[s], sg, gf);
        kf[0] = kvs[(s * 2 + 0) * 64 + lane]; kf[1] = kvs[(s * 2 + 1) * 64 + lane];
        vf[0] = kvs[((4 + s) * 2 + 0) * 64 + lane]; vf[1] = kvs[((4 + s) * 2 + 1) * 64 + lane];
        sc = mma_hp(qf, kf, sc);
        dp = mma_hp(gf, vf, dp);
      }
    }
    __builtin_amdgcn_sched_barrier(0);     // (the row-pattern tiles are dead here: the gathers below must not be hoisted over them)
    // gather fragments of the dO tile (contracted over the queries) go out before the exponentials; the Q tile's follow once the
    // dO ones are in use (all eight at once do not fit 256 registers beside the four accumulators)
    Raw8 rgo[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      rgo[t][0] = raw_gather(dO, a.lddout, rb + qb, lr, t, h);
      rgo[t][1] = raw_gather(dO, a.lddout, rb + qb, 32 + lr, t, h);
    }
    // rows of the C layout are queries: q = qb + (e&3) + 8 (e>>2) + 4 h
    unsigned md = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 l4 = *reinterpret_cast<const float4*>(lsep + qb + 8 * j + 4 * h);
      const float4 d4 = *reinterpret_cast<const float4*>(dlp + qb + 8 * j + 4 * h);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pr = __expf(fmaf(sc[4 * j + i], c1, -lv[i]));
        sc[4 * j + i] = pr;                                            // P
        dp[4 * j + i] = pr * fmaf(dp[4 * j + i], c2, -dv[i]);          // dS
        md = max(md, abs_bits(dp[4 * j + i]));
      }
    }
    md = max(md, (unsigned)__shfl_xor((int)md, 32, 64));
    const int ervn = min(erv, eg);
    if (ervn != erv) {                     // wave-uniform
      const float f = pow2i(ervn - erv);
#pragma unroll
      for (int e = 0; e < 16; ++e) { dv0[e] *= f; dv1[e] *= f; }
    }
    erv = ervn;
    const int erkn = min(erk, eq + scale_exp_of(md));
    if (__any(erkn != erk)) {
      const float f = pow2i(erkn - erk);
#pragma unroll
      for (int e = 0; e < 16; ++e) { dk0[e] *= f; dk1[e] *= f; }
    }
    erk = erkn;
    const float sgo = pow2i(erv), sgq = pow2i(eq), sds = pow2i(erk - eq);
    Raw8 rgq[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {          // dv^T += dO^T P
      uint4 pf[2], g0[2], g1[2];
      rgq[t][0] = raw_gather(Q, ld, rb + qb, lr, t, h);
      rgq[t][1] = raw_gather(Q, ld, rb + qb, 32 + lr, t, h);
      split_acc_hp(sc, t, 1.f, pf);
      split_hp(rgo[t][0], sgo, g0);
      split_hp(rgo[t][1], sgo, g1);
      dv0 = mma_hp(g0, pf, dv0);
      dv1 = mma_hp(g1, pf, dv1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 2; ++t) {          // dk^T += Q^T dS
      uint4 sf[2], q0f[2], q1f[2];
      split_acc_hp(dp, t, sds, sf);
      split_hp(rgq[t][0], sgq, q0f);
      split_hp(rgq[t][1], sgq, q1f);
      dk0 = mma_hp(q0f, sf, dk0);
      dk1 = mma_hp(q1f, sf, dk1);
    }
  }
  __syncthreads();     // every wave's last fragment read is over before the combine buffer overwrites them
  const float unv = pow2i(-erv), unk = pow2i(-erk);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int d = (e & 3) + 8 * (e >> 2) + 4 * h;
    os[wave][d][lr] = dv0[e] * unv;
    os[wave][32 + d][lr] = dv1[e] * unv;
    os[wave][64 + d][lr] = dk0[e] * unk;
    os[wave][96 + d][lr] = dk1[e] * unk;
  }
  __syncthreads();
  const int d = tid & 63, kg = tid >> 6;
#pragma unroll
  for (int i = 0; i < 32 / WV; ++i) {
    const int k = kg * (32 / WV) + i;
    float dvv = (os[0][d][k] + os[1][d][k]) + (os[2][d][k] + os[3][d][k]);
    float dkv = (os[0][64 + d][k] + os[1][64 + d][k]) + (os[2][64 + d][k] + os[3][64 + d][k]);
    if (WV == 8) {
      dvv += (os[WV - 4][d][k] + os[WV - 3][d][k]) + (os[WV - 2][d][k] + os[WV - 1][d][k]);
      dkv += (os[WV - 4][64 + d][k] + os[WV - 3][64 + d][k]) + (os[WV - 2][64 + d][k] + os[WV - 1][64 + d][k]);
    }
    float* row = a.dqkv + (rb + k0r + k) * a.lddqkv + hd * a.hs;
    row[a.v_off + d] = dvv;
    row[a.k_off + d] = dkv * a.scale;
  }
}

// waves that split the other sequence axis: up to 8 (two per SIMD) from T = 256.  OSM_FLASH_WAVES=4: round 2's four
int max_waves() {
  static const int v = [] { const char* e = std::getenv("OSM_FLASH_WAVES"); return (e && atoi(e) == 4) ? 4 : 8; }();
  return v;
}
int nw_of(int T) { return T >= 256 && T % 256 == 0 && max_waves() == 8 ? 8 : (T >= 128 ? 4 : (T >= 64 ? 2 : 1)); }

int check(const osm_attn_desc* d, const char* who) {
  OSM_REQUIRE(d && d->qkv, "%s: null pointer", who);
  OSM_REQUIRE(d->ch == DH && d->T >= 64 && d->T % (32 * nw_of(d->T)) == 0,
              "%s: needs 64-wide heads and T a multiple of 64 (of 128 from T = 128) (got ch %d, T %d)", who, d->ch, d->T);
  OSM_REQUIRE(d->B > 0 && d->heads > 0, "%s: bad shape", who);
  OSM_REQUIRE(d->arith >= 0 && d->arith <= 2, "%s: arith must be 0 (bf16x6), 1 (fp16) or 2 (f16x3)", who);
  OSM_REQUIRE(d->ldqkv % 4 == 0 && d->q_off % 4 == 0 && d->k_off % 4 == 0 && d->v_off % 4 == 0 && d->head_stride % 4 == 0 &&
              osm::aligned16(d->qkv), "%s: qkv columns must be 16-byte aligned", who);
  return OSM_OK;
}

FlashArgs to_args(const osm_attn_desc* d) {
  FlashArgs a{};
  a.qkv = d->qkv; a.ldqkv = d->ldqkv; a.q_off = d->q_off; a.k_off = d->k_off; a.v_off = d->v_off; a.hs = d->head_stride;
  a.out = d->out; a.ldout = d->ldout; a.dout = d->dout; a.lddout = d->lddout; a.dqkv = d->dqkv; a.lddqkv = d->lddqkv;
  a.B = d->B; a.T = d->T; a.heads = d->heads; a.scale = d->scale;
  a.nw = nw_of(d->T);
  return a;
}

}  // namespace

extern "C" int osm_attn_flash_supported(int T, int ch) { return ch == DH && T >= 64 && T % (32 * nw_of(T)) == 0; }

extern "C" int osm_attn_flash_fwd(const osm_attn_desc* d, float* lse, void* stream) {
  int rc = check(d, "osm_attn_flash_fwd");
  if (rc) return rc;
  OSM_REQUIRE(d->out && lse, "osm_attn_flash_fwd: null pointer");
  FlashArgs a = to_args(d);
  a.lse = lse;
  const dim3 g(d->T / 32, d->heads, d->B);
  const bool two = (d->T / a.nw) % 64 == 0;
  const hipStream_t st = (hipStream_t)stream;
  auto launch = [&](auto np) {
    constexpr int P = decltype(np)::value;
    if (a.nw == 8) {
      // (the one-plane instance with two logit tiles in flight spills 48 registers at two waves per SIMD: one tile there)
      if (two && P != 1) hipLaunchKernelGGL((flash_fwd_kernel<P, 2, 8>), g, dim3(512), 0, st, a);
      else hipLaunchKernelGGL((flash_fwd_kernel<P, 1, 8>), g, dim3(512), 0, st, a);
    } else if (two) {
      hipLaunchKernelGGL((flash_fwd_kernel<P, 2, 4>), g, dim3(256), 0, st, a);
    } else {
      hipLaunchKernelGGL((flash_fwd_kernel<P, 1, 4>), g, dim3(256), 0, st, a);
    }
  };
  if (d->arith == 2) {       // f16x3: two half planes per operand, ranges found in the kernel
    if (a.nw == 8) hipLaunchKernelGGL((flash_fwd_hp_kernel<8>), g, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((flash_fwd_hp_kernel<4>), g, dim3(256), 0, st, a);
  } else if (d->arith == 1) launch(std::integral_constant<int, 1>{});
  else launch(std::integral_constant<int, 3>{});
  return osm::check_launch("flash_fwd_kernel");
}

extern "C" int osm_attn_flash_bwd(const osm_attn_desc* d, const float* out, long long ldout, const float* lse, float* delta,
                                  void* stream) {
  int rc = check(d, "osm_attn_flash_bwd");
  if (rc) return rc;
  OSM_REQUIRE(d->dout && d->dqkv && out && lse && delta, "osm_attn_flash_bwd: null pointer");
  OSM_REQUIRE(d->lddout % 4 == 0 && osm::aligned16(d->dout), "osm_attn_flash_bwd: dout must be 16-byte aligned");
  FlashArgs a = to_args(d);
  a.o = out; a.ldo = ldout; a.lse = const_cast<float*>(lse); a.delta = delta;
  OSM_REQUIRE(ldout % 4 == 0 && osm::aligned16(out), "osm_attn_flash_bwd: out must be 16-byte aligned");
  const dim3 g(d->T / 32, d->heads, d->B);       // (delta is formed by the dq kernel and read by the dk / dv kernel after it)
  const hipStream_t st = (hipStream_t)stream;
  auto launch = [&](auto np) {
    constexpr int P = decltype(np)::value;
    if (a.nw == 8) hipLaunchKernelGGL((flash_bwd_q_kernel<P, 8>), g, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((flash_bwd_q_kernel<P, 4>), g, dim3(256), 0, st, a);
    // the dk / dv kernel: with K, V fragments resident it needs 368 registers (bf16x6) = one wave per SIMD; with the fragments in
    // LDS (KVL) eight waves = two per SIMD split the queries (round 5: attention class 0.99 -> 0.95 ms; OSM_FLASH_KV_LDS=0: the old form)
    static const bool kv_lds = [] { const char* e = std::getenv("OSM_FLASH_KV_LDS"); return !(e && atoi(e) == 0); }();
    FlashArgs b = a;
    if (kv_lds && b.nw == 8) {
      hipLaunchKernelGGL((flash_bwd_kv_kernel<P, 8, true>), g, dim3(512), 0, st, b);
    } else {
      if (b.nw > 4) b.nw = 4;
      hipLaunchKernelGGL((flash_bwd_kv_kernel<P, 4>), g, dim3(256), 0, st, b);
    }
  };
  if (d->arith == 2) {
    if (a.nw == 8) {
      hipLaunchKernelGGL((flash_bwd_q_hp_kernel<8>), g, dim3(512), 0, st, a);
      hipLaunchKernelGGL((flash_bwd_kv_hp_kernel<8>), g, dim3(512), 0, st, a);
    } else {
      hipLaunchKernelGGL((flash_bwd_q_hp_kernel<4>), g, dim3(256), 0, st, a);
      hipLaunchKernelGGL((flash_bwd_kv_hp_kernel<4>), g, dim3(256), 0, st, a);
    }
  } else if (d->arith == 1) launch(std::integral_constant<int, 1>{});
  else launch(std::integral_constant<int, 3>{});
  return osm::check_launch("flash_bwd kernels");
}
