// Fused attention core for the low-resolution AttentionBlocks (T = 64 or 256 tokens, 16/32/64-wide heads).
// Reference: QKVAttentionLegacy.forward / QKVAttention.forward (unet.py:416-433, 459-467):
//     w = softmax(einsum(q * s, k * s)), a = einsum(w, v),  s = ch^-1/4  (logits scaled by 1/sqrt(ch)).
//
// At 8x8 and 16x16 the per-head problem is tiny (<= 256 x 256 x 64) and the unfused pipeline -- QK^T GEMM,
// softmax, transpose, P V GEMM (+ split-K reduce); backward: dP GEMM, softmax-bwd, transpose, three GEMMs --
// is 5 + 9 launches of 13-20 us each, 12 blocks per step.  Here the logits never leave the CU:
//   forward  (1 launch):  workgroup = (64 query rows, head, image): K -> LDS, S = scale Q K^T in registers -> LDS,
//                         row softmax in LDS (same expf / summation order as softmax_rows_kernel), V -> LDS, O = P V.
//   backward (2 launches): (i) per 64 query rows: recompute P, dP = dO V^T, dS = P (dP - rowsum(dP P)),
//                         dq = scale dS K; P and dS tiles go to a scratch buffer;  (ii) per 64 key rows:
//                         dv = P^T dO, dk = scale dS^T Q from the scratch tiles.
// Arithmetic is plain fp32 FMA (the fp32 MFMA has the same peak as the vector pipe on gfx950, and these
// kernels are latency-, not throughput-bound); nothing is kept from the forward (P is recomputed).
#include "osm_common.h"

namespace {

struct AttnArgs {
  const float* qkv;
  const float* dout;
  float* out;
  float* dqkv;
  float* ws;
  long long ldqkv, ldout, lddout, lddqkv;
  int q_off, k_off, v_off, hs;
  int B, T, heads;
  float scale;
};

// rows [row0, row0 + nrows) x CH columns starting at column `col` of a row-major matrix -> LDS [nrows][CH + 1]
template <int CH>
__device__ __forceinline__ void load_rows(float* dst, const float* __restrict__ src, long long ld, long long row0,
                                          int col, int nrows) {
  constexpr int V = CH / 4;
  for (int v = threadIdx.x; v < nrows * V; v += 256) {
    const int r = v / V, c4 = v - r * V;
    const float4 t = *reinterpret_cast<const float4*>(src + (row0 + r) * ld + col + 4 * c4);
    float* d = dst + r * (CH + 1) + 4 * c4;
    d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
  }
}

// acc[a][jj] = sum_d A[ty*4 + a][d] * Bm[tx + 16 jj][d]      (A: [64][CH+1], Bm: [16 NJ][CH+1], both in LDS)
template <int CH, int NJ>
__device__ __forceinline__ void tile_abt(const float* As, const float* Bs, int ty, int tx, float (&acc)[4][NJ]) {
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) acc[a][jj] = 0.f;
#pragma unroll 4
  for (int d = 0; d < CH; ++d) {
    float av[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) av[a] = As[(ty * 4 + a) * (CH + 1) + d];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const float bv = Bs[(tx + 16 * jj) * (CH + 1) + d];
#pragma unroll
      for (int a = 0; a < 4; ++a) acc[a][jj] = fmaf(av[a], bv, acc[a][jj]);
    }
  }
}

// o[a][e] = sum_{j < n} L[(ty*4 + a) * ldl + j] * R[j * (CH+1) + tx*EC + e]
template <int CH>
__device__ __forceinline__ void tile_ab(const float* L, int ldl, const float* R, int n, int ty, int tx,
                                        float (&o)[4][CH / 16]) {
  constexpr int EC = CH / 16;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int e = 0; e < EC; ++e) o[a][e] = 0.f;
#pragma unroll 4
  for (int j = 0; j < n; ++j) {
    float lv[4], rv[EC];
#pragma unroll
    for (int a = 0; a < 4; ++a) lv[a] = L[(ty * 4 + a) * ldl + j];
#pragma unroll
    for (int e = 0; e < EC; ++e) rv[e] = R[j * (CH + 1) + tx * EC + e];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int e = 0; e < EC; ++e) o[a][e] = fmaf(lv[a], rv[e], o[a][e]);
  }
}

// row softmax of Ss[64][T+1] in place; wave w owns rows 16 w .. 16 w + 15 (lane-strided like softmax_rows_kernel)
template <int T>
__device__ __forceinline__ void softmax_tile(float* Ss) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int rr = 0; rr < 16; ++rr) {
    float* s = Ss + (wave * 16 + rr) * (T + 1);
    float mx = -INFINITY;
    for (int i = lane; i < T; i += 64) mx = fmaxf(mx, s[i]);
    mx = osm::wave_max(mx);
    float sum = 0.f;
    for (int i = lane; i < T; i += 64) sum += expf(s[i] - mx);
    sum = osm::wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int i = lane; i < T; i += 64) s[i] = expf(s[i] - mx) * inv;
  }
}

template <int CH, int NJ>
__global__ __launch_bounds__(256) void attn_small_fwd_kernel(AttnArgs a) {
  constexpr int T = 16 * NJ, EC = CH / 16;
  extern __shared__ float sm[];
  float* Ks = sm;                      // [T][CH+1]: K, then V
  float* Qs = Ks + T * (CH + 1);       // [64][CH+1]
  float* Ss = Qs + 64 * (CH + 1);      // [64][T+1]
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int i0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const long long row0 = (long long)b * T;
  load_rows<CH>(Ks, a.qkv, a.ldqkv, row0, a.k_off + h * a.hs, T);
  load_rows<CH>(Qs, a.qkv, a.ldqkv, row0 + i0, a.q_off + h * a.hs, 64);
  __syncthreads();
  float acc[4][NJ];
  tile_abt<CH, NJ>(Qs, Ks, ty, tx, acc);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) Ss[(ty * 4 + r) * (T + 1) + tx + 16 * jj] = acc[r][jj] * a.scale;
  __syncthreads();
  softmax_tile<T>(Ss);
  load_rows<CH>(Ks, a.qkv, a.ldqkv, row0, a.v_off + h * a.hs, T);   // K is dead: the softmax only touches Ss
  __syncthreads();
  float o[4][EC];
  tile_ab<CH>(Ss, T + 1, Ks, T, ty, tx, o);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int e = 0; e < EC; ++e)
      a.out[(row0 + i0 + ty * 4 + r) * a.ldout + h * CH + tx * EC + e] = o[r][e];
}

// backward, query-row tiles: dq, and the P / dS tiles for the key-row pass
template <int CH, int NJ>
__global__ __launch_bounds__(256) void attn_small_bwd_q_kernel(AttnArgs a) {
  constexpr int T = 16 * NJ, EC = CH / 16;
  extern __shared__ float sm[];
  float* Ks = sm;
  float* Qs = Ks + T * (CH + 1);
  float* Ss = Qs + 64 * (CH + 1);
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int i0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const long long row0 = (long long)b * T;
  load_rows<CH>(Ks, a.qkv, a.ldqkv, row0, a.k_off + h * a.hs, T);
  load_rows<CH>(Qs, a.qkv, a.ldqkv, row0 + i0, a.q_off + h * a.hs, 64);
  __syncthreads();
  float acc[4][NJ];
  tile_abt<CH, NJ>(Qs, Ks, ty, tx, acc);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) Ss[(ty * 4 + r) * (T + 1) + tx + 16 * jj] = acc[r][jj] * a.scale;
  __syncthreads();
  softmax_tile<T>(Ss);
  __syncthreads();
  float pr[4][NJ];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) pr[r][jj] = Ss[(ty * 4 + r) * (T + 1) + tx + 16 * jj];
  load_rows<CH>(Ks, a.qkv, a.ldqkv, row0, a.v_off + h * a.hs, T);             // V
  load_rows<CH>(Qs, a.dout, a.lddout, row0 + i0, h * CH, 64);                  // dO tile
  __syncthreads();
  tile_abt<CH, NJ>(Qs, Ks, ty, tx, acc);                                       // dP = dO V^T
  float* wsP = a.ws + ((long long)(b * a.heads + h) * T + i0) * T;
  float* wsS = wsP + (long long)a.B * a.heads * T * T;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float dot = 0.f;
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) dot = fmaf(pr[r][jj], acc[r][jj], dot);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);            // the 16 tx lanes of this row
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
      const float ds = pr[r][jj] * (acc[r][jj] - dot);
      const int c = tx + 16 * jj;
      Ss[(ty * 4 + r) * (T + 1) + c] = ds;
      wsP[(long long)(ty * 4 + r) * T + c] = pr[r][jj];
      wsS[(long long)(ty * 4 + r) * T + c] = ds;
    }
  }
  __syncthreads();                                                             // V / dO reads done, dS visible
  load_rows<CH>(Ks, a.qkv, a.ldqkv, row0, a.k_off + h * a.hs, T);             // K again
  __syncthreads();
  float o[4][EC];
  tile_ab<CH>(Ss, T + 1, Ks, T, ty, tx, o);
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int e = 0; e < EC; ++e)
      a.dqkv[(row0 + i0 + ty * 4 + r) * a.lddqkv + a.q_off + h * a.hs + tx * EC + e] = o[r][e] * a.scale;
}

// backward, key-row tiles:  dv[j] = sum_i P[i][j] dO[i],  dk[j] = scale sum_i dS[i][j] Q[i]
template <int CH, int NJ>
__global__ __launch_bounds__(256) void attn_small_bwd_k_kernel(AttnArgs a) {
  constexpr int T = 16 * NJ, EC = CH / 16;
  extern __shared__ float sm[];
  float* Ps = sm;                      // [T][65]: column tile (64 keys) of P, then of dS   (row i, local key jl)
  float* Xs = Ps + T * 65;             // [T][CH+1]: dO, then Q
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
  const int j0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const long long row0 = (long long)b * T;
  const float* wsP = a.ws + (long long)(b * a.heads + h) * T * T;
  const float* wsS = wsP + (long long)a.B * a.heads * T * T;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    const float* src = pass == 0 ? wsP : wsS;
    for (int v = tid; v < T * 16; v += 256) {
      const int i = v >> 4, c4 = v & 15;
      const float4 t = *reinterpret_cast<const float4*>(src + (long long)i * T + j0 + 4 * c4);
      float* d = Ps + i * 65 + 4 * c4;
      d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
    }
    if (pass == 0) load_rows<CH>(Xs, a.dout, a.lddout, row0, h * CH, T);
    else load_rows<CH>(Xs, a.qkv, a.ldqkv, row0, a.q_off + h * a.hs, T);
    __syncthreads();
    float o[4][EC];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int e = 0; e < EC; ++e) o[r][e] = 0.f;
#pragma unroll 4
    for (int i = 0; i < T; ++i) {
      float lv[4], rv[EC];
#pragma unroll
      for (int r = 0; r < 4; ++r) lv[r] = Ps[i * 65 + ty * 4 + r];
#pragma unroll
      for (int e = 0; e < EC; ++e) rv[e] = Xs[i * (CH + 1) + tx * EC + e];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < EC; ++e) o[r][e] = fmaf(lv[r], rv[e], o[r][e]);
    }
    const int col = (pass == 0 ? a.v_off : a.k_off) + h * a.hs;
    const float mul = pass == 0 ? 1.0f : a.scale;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int e = 0; e < EC; ++e)
        a.dqkv[(row0 + j0 + ty * 4 + r) * a.lddqkv + col + tx * EC + e] = o[r][e] * mul;
    __syncthreads();
  }
}

size_t lds_qtile(int T, int CH) { return sizeof(float) * ((size_t)T * (CH + 1) + 64 * (CH + 1) + 64 * (T + 1)); }
size_t lds_ktile(int T, int CH) { return sizeof(float) * ((size_t)T * 65 + (size_t)T * (CH + 1)); }

bool supported(int T, int ch) { return (T == 64 || T == 256) && (ch == 16 || ch == 32 || ch == 64); }

template <int CH, int NJ>
int run(const AttnArgs& a, int which, hipStream_t st) {
  const int T = 16 * NJ;
  const dim3 grid(T / 64, a.heads, a.B);
  // LDS above the 64 KB default needs an opt-in per kernel instantiation (done once, thread-safe static init)
  static const hipError_t attr_rc = [] {
    constexpr int T_ = 16 * NJ;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_small_fwd_kernel<CH, NJ>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_qtile(T_, CH));
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_small_bwd_q_kernel<CH, NJ>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_qtile(T_, CH));
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_small_bwd_k_kernel<CH, NJ>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ktile(T_, CH));
    return e;
  }();
  if (attr_rc != hipSuccess)
    return osm::fail(OSM_ERR_LAUNCH, "osm_attn_small: cannot raise the dynamic LDS limit: %s", hipGetErrorString(attr_rc));
  if (which == 0) {
    hipLaunchKernelGGL((attn_small_fwd_kernel<CH, NJ>), grid, dim3(256), lds_qtile(T, CH), st, a);
    return osm::check_launch("attn_small_fwd_kernel");
  }
  hipLaunchKernelGGL((attn_small_bwd_q_kernel<CH, NJ>), grid, dim3(256), lds_qtile(T, CH), st, a);
  int rc = osm::check_launch("attn_small_bwd_q_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL((attn_small_bwd_k_kernel<CH, NJ>), grid, dim3(256), lds_ktile(T, CH), st, a);
  return osm::check_launch("attn_small_bwd_k_kernel");
}

int dispatch(const osm_attn_desc* d, int which, hipStream_t st) {
  AttnArgs a{};
  a.qkv = d->qkv; a.dout = d->dout; a.out = d->out; a.dqkv = d->dqkv; a.ws = d->ws;
  a.ldqkv = d->ldqkv; a.ldout = d->ldout; a.lddout = d->lddout; a.lddqkv = d->lddqkv;
  a.q_off = d->q_off; a.k_off = d->k_off; a.v_off = d->v_off; a.hs = d->head_stride;
  a.B = d->B; a.T = d->T; a.heads = d->heads; a.scale = d->scale;
  const int key = d->ch * 1000 + d->T;
  switch (key) {
    case 64 * 1000 + 256: return run<64, 16>(a, which, st);
    case 64 * 1000 + 64: return run<64, 4>(a, which, st);
    case 32 * 1000 + 256: return run<32, 16>(a, which, st);
    case 32 * 1000 + 64: return run<32, 4>(a, which, st);
    case 16 * 1000 + 256: return run<16, 16>(a, which, st);
    case 16 * 1000 + 64: return run<16, 4>(a, which, st);
  }
  return osm::fail(OSM_ERR_UNSUPPORTED, "osm_attn_small: T=%d ch=%d not supported (T in {64,256}, ch in {16,32,64})",
                   d->T, d->ch);
}

int check(const osm_attn_desc* d, bool bwd, const char* who) {
  OSM_REQUIRE(d && d->qkv, "%s: null pointer", who);
  OSM_REQUIRE(d->B > 0 && d->heads > 0 && supported(d->T, d->ch), "%s: unsupported shape (T=%d, ch=%d)", who,
              d ? d->T : 0, d ? d->ch : 0);
  OSM_REQUIRE(d->ldqkv % 4 == 0 && d->q_off % 4 == 0 && d->k_off % 4 == 0 && d->v_off % 4 == 0 &&
                  d->head_stride % 4 == 0 && osm::aligned16(d->qkv),
              "%s: qkv layout must be 16-byte aligned per head row", who);
  if (!bwd) {
    OSM_REQUIRE(d->out && d->ldout >= (long long)d->heads * d->ch, "%s: bad output", who);
  } else {
    OSM_REQUIRE(d->dout && d->dqkv && d->ws, "%s: null pointer", who);
    OSM_REQUIRE(d->lddout % 4 == 0 && osm::aligned16(d->dout) && osm::aligned16(d->ws), "%s: dout / ws alignment", who);
  }
  return OSM_OK;
}

}  // namespace

extern "C" int osm_attn_small_supported(int T, int ch) { return supported(T, ch) ? 1 : 0; }

extern "C" int osm_attn_small_fwd(const osm_attn_desc* d, void* stream) {
  int rc = check(d, false, "osm_attn_small_fwd");
  if (rc) return rc;
  return dispatch(d, 0, (hipStream_t)stream);
}

extern "C" int osm_attn_small_bwd(const osm_attn_desc* d, void* stream) {
  int rc = check(d, true, "osm_attn_small_bwd");
  if (rc) return rc;
  return dispatch(d, 1, (hipStream_t)stream);
}
