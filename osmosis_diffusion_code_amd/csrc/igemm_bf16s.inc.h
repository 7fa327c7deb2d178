// Split-bf16 implicit-GEMM convolution (included inside igemm.hip's anonymous namespace).
//
// fp32 has no fast matrix path on gfx950 (no xf32; the f32 MFMA runs at the vector rate, 1/16 of the
// bf16 MFMA).  An fp32 value splits EXACTLY into three bf16 terms  a = a0 + a1 + a2  (8+8+8 mantissa
// bits), so  a*b = a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0) + O(2^-24 |ab|):
//   NP = 3 planes, 6 bf16 MFMAs per product group  -> fp32-class accuracy  ("bf16x6", 16/6 = 2.7x the f32 roof)
//   NP = 2 planes, 3 bf16 MFMAs                    -> ~2^-16 relative       ("bf16x3", 5.3x the f32 roof)
// Products are exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16; only the dropped terms and
// the fp32 accumulation differ from the exact-f32 kernel.
//
// Data flow per 128x128x32 chunk:
//   A (activations, fp32 in HBM): global -> VGPR (16-byte loads, branch-free) -> RNE split on the VALU
//       (v_cvt_pk_bf16_f32) -> NP bf16 planes in LDS.   Padding / channel masks are applied here.
//   B (weights): pre-split ONCE at pack time into NP bf16 planes [plane][tap][Cout][Kp] in HBM ->
//       16-byte loads -> LDS, no VALU work.
//   Fragments: one ds_read_b128 per (tile, plane) = 8 consecutive k of one row, which is exactly the
//       v_mfma_f32_32x32x16_bf16 operand (lane = row + 32 * k-octet).  Rows are 80 B apart (64 + 16 pad):
//       the 16 rows of a ds_read_b128 lane group land on 16 distinct 16-byte slots -> conflict-free.
//   One LDS stage (NP * 20 KB), two workgroups per CU: one workgroup's split+store phase overlaps the
//   other's MFMA phase; next-chunk global loads are in flight under the current chunk's MFMAs.

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int S_ROWB = 80;              // bytes per staged row (32 bf16 + 16 B pad)
constexpr int S_PLANE = 128 * S_ROWB;   // bytes per 128-row plane

template <int NP>
__device__ __forceinline__ void split_planes(f32x4_t x, bf16x4_t (&pl)[NP]) {
  pl[0] = __builtin_convertvector(x, bf16x4_t);
  f32x4_t r = x - __builtin_convertvector(pl[0], f32x4_t);
  pl[1] = __builtin_convertvector(r, bf16x4_t);
  if (NP == 3) {
    r = r - __builtin_convertvector(pl[1], f32x4_t);
    pl[NP - 1] = __builtin_convertvector(r, bf16x4_t);
  }
}

template <int TAPS, int NP>
__global__ __launch_bounds__(256, 2) void igemm_bf16s_kernel(const float* __restrict__ Aglob,
                                                              const unsigned short* __restrict__ Bglob,
                                                              IGemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * NP * S_PLANE];
  unsigned char* As = smem;
  unsigned char* Bs = smem + NP * S_PLANE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  const int nt = p.mtiles * p.ntiles;
  const int bid = blockIdx.x;
  const int q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int tile_n = id % p.ntiles, tile_m = id / p.ntiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const float* __restrict__ A = Aglob;
  const unsigned short* __restrict__ Bm = Bglob;

  const int ks = blockIdx.y;
  const int per = (p.nchunks + p.splitk - 1) / p.splitk;
  const int kc0 = ks * per;
  const int kc1 = min(p.nchunks, kc0 + per);

  // ---- staging coordinates.  A: float4 column group cg of rows r0+32i.  B: 16-byte segment seg of rows rb0+64i.
  const int cg = tid & 7, r0 = tid >> 3;
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
  const int seg = tid & 3, rb0 = tid >> 2;
  long long brow[2];
  bool bvalid[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = n0 + rb0 + 64 * i;
    brow[i] = (long long)n * p.ldb;
    bvalid[i] = n < p.N;
  }

  float4 ra[4];
  uint4 rb[NP][2];
  unsigned okm = 0;

#define OSM_S_LOAD(kc_)                                                                              \
  {                                                                                                  \
    okm = 0;                                                                                         \
    const int cc_ = (kc_) / TAPS;                                                                    \
    const int tap_ = (kc_) - cc_ * TAPS;                                                             \
    const int c0_ = cc_ * BK;                                                                        \
    long long toff_ = 0;                                                                             \
    if (TAPS == 9) toff_ = ((long long)(tap_ / 3 - 1) * p.W + (tap_ % 3 - 1)) * p.lda;               \
    const int c_ = c0_ + 4 * cg;                                                                     \
    const bool cok_ = c_ < p.K;                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                  \
      const bool ok_ = cok_ && ((amask[i] >> tap_) & 1u);                                            \
      ra[i] = *reinterpret_cast<const float4*>(A + (ok_ ? arow[i] + toff_ + c_ : 0));                \
      okm |= (ok_ ? 1u : 0u) << i;                                                                   \
    }                                                                                                \
    const int kb_ = c0_ + 8 * seg;                                                                   \
    const bool kok_ = kb_ < p.ldb;                                                                   \
    const long long boff_ = (long long)tap_ * p.tapstrideB + kb_;                                    \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                  \
      const bool ok_ = kok_ && bvalid[i];                                                            \
      okm |= (ok_ ? 16u : 0u) << i;                                                                  \
      _Pragma("unroll") for (int pl = 0; pl < NP; ++pl)                                              \
        rb[pl][i] = *reinterpret_cast<const uint4*>(Bm + (ok_ ? pl * p.planestrideB + boff_ + brow[i] : 0)); \
    }                                                                                                \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lk = lane >> 5;
  const unsigned char* a_rd = As + (64 * wm + lr) * S_ROWB + 16 * lk;
  const unsigned char* b_rd = Bs + (64 * wn + lr) * S_ROWB + 16 * lk;

  const int nk = kc1 - kc0;
  if (nk > 0) OSM_S_LOAD(kc0);
  for (int it = 0; it < nk; ++it) {
    // ---- split + store the staged registers (validity masks applied here)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = sel4((okm >> i) & 1u, ra[i]);
      f32x4_t x = {v.x, v.y, v.z, v.w};
      bf16x4_t pl[NP];
      split_planes<NP>(x, pl);
#pragma unroll
      for (int q2 = 0; q2 < NP; ++q2)
        *reinterpret_cast<bf16x4_t*>(As + q2 * S_PLANE + (r0 + 32 * i) * S_ROWB + 8 * cg) = pl[q2];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = (okm >> (4 + i)) & 1u;
#pragma unroll
      for (int q2 = 0; q2 < NP; ++q2) {
        uint4 v = rb[q2][i];
        if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(Bs + q2 * S_PLANE + (rb0 + 64 * i) * S_ROWB + 16 * seg) = v;
      }
    }
    __syncthreads();
    if (it + 1 < nk) OSM_S_LOAD(kc0 + it + 1);
    // ---- MFMA phase: 2 k16-steps x (2x2 tiles) x (NP==3 ? 6 : 3) products
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      bf16x8_t af[2][NP], bf[2][NP];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q2 = 0; q2 < NP; ++q2) {
          af[t][q2] = *reinterpret_cast<const bf16x8_t*>(a_rd + q2 * S_PLANE + t * 32 * S_ROWB + 32 * st);
          bf[t][q2] = *reinterpret_cast<const bf16x8_t*>(b_rd + q2 * S_PLANE + t * 32 * S_ROWB + 32 * st);
        }
#pragma unroll
      for (int pa = NP - 1; pa >= 0; --pa)
#pragma unroll
        for (int pb = NP - 1 - pa; pb >= 0; --pb) {   // all (pa, pb) with pa + pb <= NP - 1, small terms first
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][pa], bf[0][pb], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][pa], bf[1][pb], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][pa], bf[0][pb], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][pa], bf[1][pb], acc[1][1], 0, 0, 0);
        }
    }
    __syncthreads();
  }
#undef OSM_S_LOAD

  // ---- epilogue (identical C/D mapping to the f32 kernel)
  const bool partial = p.splitk > 1;
  float* Cb = partial ? p.ws + ((long long)ks * p.M) * p.N : p.C;
  const float* Rb = (p.res && !partial) ? p.res : nullptr;
  const long long ldc = partial ? (long long)p.N : p.ldc;
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) {
    const int n = n0 + 64 * wn + 32 * tn + lr;
    if (n >= p.N) continue;
    const float bv = (!partial && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + 64 * wm + 32 * tm + (e & 3) + 8 * (e >> 2) + 4 * lk;
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

// OIHW fp32 -> NP bf16 planes, forward [plane][tap][Cout][Kpf] and data-gradient [plane][tap][Cin][Kpd]
// (taps flipped, channels transposed); K padded with zeros to a multiple of 8 (16-byte rows).
__device__ __forceinline__ void split_scalar(float x, int np, unsigned short* out) {
  float r = x;
  for (int q = 0; q < np; ++q) {
    const __bf16 b = (__bf16)r;
    out[q] = __builtin_bit_cast(unsigned short, b);
    r -= (float)b;
  }
}

__global__ void pack_weight_bf16s_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout,
                                         int Cin, int k, int np, int dgrad) {
  // output index space: [tap][rows][Kp]; forward rows = Cout, K = Cin ; dgrad rows = Cin, K = Cout
  const int rows = dgrad ? Cin : Cout;
  const int K = dgrad ? Cout : Cin;
  const int Kp = (K + 7) & ~7;
  const long long per_plane = (long long)k * k * rows * Kp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_plane;
       i += (long long)gridDim.x * blockDim.x) {
    const int kk = (int)(i % Kp);
    const int rr = (int)((i / Kp) % rows);
    const int tap = (int)(i / ((long long)Kp * rows));
    float v = 0.f;
    if (kk < K) {
      int kh = tap / k, kw = tap % k;
      int co = rr, ci = kk;
      if (dgrad) {
        kh = k - 1 - kh;
        kw = k - 1 - kw;
        co = kk;
        ci = rr;
      }
      v = w[(((long long)co * Cin + ci) * k + kh) * k + kw];
    }
    unsigned short pl[3];
    split_scalar(v, np, pl);
    for (int q = 0; q < np; ++q) out[q * per_plane + i] = pl[q];
  }
}
