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
// Measured (rocprofv3 PMC + ablation builds, round 1): with both operands staged through LDS the kernel
// was bound by the VGPR->LDS store path and by LDS queueing (ds_write removed: -37 % time; MFMA pipe 48 %
// busy) -- not by the matrix cores.  Hence:
//   B (weights) never touches LDS.  It is pre-split AND pre-swizzled at pack time into MFMA-fragment
//       order  [plane][tap][k16-step][n/32][lane][8 bf16] : a wave fetches the B operand of one 32-column
//       tile with ONE fully coalesced 1 KB global load per (k16-step, plane), straight into the VGPRs the
//       MFMA reads.  Weights are shared by every M-tile, so these loads are L2 hits.
//   A (activations, fp32 in HBM): global -> VGPR (16-byte loads, branch-free; masked lanes read a zero
//       page) -> RNE split on the VALU (v_cvt_pk_bf16_f32, 22 ops per float4) -> NP bf16 planes in LDS,
//       rows 80 B apart so ds_read_b128 fragment reads are conflict-free.
//   Waves are 1(M) x 4(N): each wave owns 128 rows x 32 columns (4 MFMA tiles), so every staged A byte is
//       reused by all four waves and no B byte is fetched twice inside a workgroup.
//   PING-PONG workgroup: 8 waves = two 4-wave groups, each owning one 128x128 output tile and its own A
//       stage.  Both run the same two-phase loop (split+store | MFMA) shifted by ONE barrier, so in every
//       barrier interval one group feeds the matrix pipe while the other stages.

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

constexpr int S_ROWB = 80;              // bytes per staged A row (32 bf16 + 16 B pad)
constexpr int S_PLANE = 128 * S_ROWB;   // bytes per 128-row plane

// 32 B of zeros every masked staging load is redirected to (halo / ragged edge / channel tail).
__device__ uint4 g_zero_page[2];   // zero-initialised device global (never written)

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {   // RNE, v_cvt_pk_bf16_f32
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// 4 fp32 -> NP planes of 4 bf16 (2 packed dwords per plane); 22 VALU ops for NP = 3.
template <int NP>
__device__ __forceinline__ void split_planes(float4 x, uint2 (&pl)[NP]) {
  unsigned a = cvt_pk_bf16(x.x, x.y), b = cvt_pk_bf16(x.z, x.w);
  pl[0] = make_uint2(a, b);
  float r0 = x.x - bf_lo(a), r1 = x.y - bf_hi(a), r2 = x.z - bf_lo(b), r3 = x.w - bf_hi(b);
  a = cvt_pk_bf16(r0, r1);
  b = cvt_pk_bf16(r2, r3);
  pl[1] = make_uint2(a, b);
  if (NP == 3) {
    r0 -= bf_lo(a); r1 -= bf_hi(a); r2 -= bf_lo(b); r3 -= bf_hi(b);
    pl[NP - 1] = make_uint2(cvt_pk_bf16(r0, r1), cvt_pk_bf16(r2, r3));
  }
}

__device__ __forceinline__ bf16x8_t as_frag(uint4 v) { return __builtin_bit_cast(bf16x8_t, v); }

// timeline probe (profiling builds only, OSM_DBG=9): s_memtime at the phase boundaries of workgroup 0
__device__ unsigned long long g_dbg_stamps[8 * 32 * 8];
#define OSM_STAMP(slot_)                                                                             \
  if (DBG >= 9 && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && (it_s) < 32)     \
    g_dbg_stamps[((threadIdx.x >> 6) * 32 + (it_s)) * 8 + (slot_)] = __builtin_amdgcn_s_memtime();

template <int TAPS, int NP, int DBG = 0>
__global__ __launch_bounds__(512, 2) void igemm_bf16s_kernel(const float* __restrict__ Aglob,
                                                              const unsigned short* __restrict__ Bglob,
                                                              IGemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * NP * S_PLANE];
  const int grp = threadIdx.x >> 8;
  unsigned char* As = smem + grp * (NP * S_PLANE);

  const int tid = threadIdx.x & 255;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  const int nt = p.mtiles * p.ntiles;
  const int npair = (nt + 1) >> 1;
  const int bid = blockIdx.x;
  const int q = npair >> 3, r = npair & 7, xcd = bid & 7, idx = bid >> 3;
  const int pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int id = 2 * pid + grp;
  const bool tile_ok = id < nt;          // odd tile count: the last group only keeps the barriers company
  const int tile_n = id % p.ntiles, tile_m = id / p.ntiles;
  const int m0 = tile_ok ? tile_m * BM : p.M, n0 = tile_ok ? tile_n * BN : p.N;
  const float* __restrict__ A = Aglob;
  const unsigned short* __restrict__ Bm = Bglob;

  const int ks = blockIdx.y;
  const int per = (p.nchunks + p.splitk - 1) / p.splitk;
  const int kc0 = ks * per;
  const int kc1 = min(p.nchunks, kc0 + per);

  // ---- A staging coordinates: float4 column group cg of rows r0 + 32 i
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
  // byte offsets from the operand bases to the zero page (global memory is one flat 64-bit space)
  const long long zoffA = reinterpret_cast<const char*>(g_zero_page) - reinterpret_cast<const char*>(A);
  const long long zoffB = reinterpret_cast<const char*>(g_zero_page) - reinterpret_cast<const char*>(Bm);

  // ---- B fragment addressing: image [plane][tap][k16-step][n/32][lane][8]
  const int jn = (n0 >> 5) + wave;                 // this wave's 32-column tile
  const bool b_ok = jn < p.nt32;
  const long long b_lane = ((long long)jn * 64 + lane) * 16;          // bytes inside one (plane, tap, step) slab
  const long long b_step = (long long)p.nt32 * 64 * 16;               // bytes per k16-step
  const long long b_tap = b_step * p.ksteps;                          // bytes per tap
  const long long b_plane = b_tap * TAPS;                             // bytes per plane

  // Two register sets for each operand: global loads are issued >= one full chunk (~3000 cycles)
  // before they are consumed.  (Measured: with a half-chunk distance the ~1500-cycle loaded latency
  // sat on the critical path twice per chunk -- the no-MFMA ablation still ran at 52 % of the time.)
  float4 raE[4], raO[4];
  uint4 bE00, bE01, bE02, bE10, bE11, bE12;   // [k16-step][plane] B fragments, even chunks
  uint4 bO00, bO01, bO02, bO10, bO11, bO12;   // odd chunks
  bE02 = bE12 = bO02 = bO12 = make_uint4(0u, 0u, 0u, 0u);

#define OSM_S_LOAD_A(ra_, kc_)                                                                       \
  {                                                                                                  \
    const int cc_ = (kc_) / TAPS;                                                                    \
    const int tap_ = (kc_) - cc_ * TAPS;                                                             \
    long long toff_ = 0;                                                                             \
    if (TAPS == 9) toff_ = ((long long)(tap_ / 3 - 1) * p.W + (tap_ % 3 - 1)) * p.lda;               \
    const int c_ = cc_ * BK + 4 * cg;                                                                \
    const bool cok_ = c_ < p.K;                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                  \
      const bool ok_ = cok_ && ((amask[i] >> tap_) & 1u);                                            \
      const long long o_ = ok_ ? (arow[i] + toff_ + c_) * 4 : zoffA;                                 \
      ra_[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(A) + o_);              \
    }                                                                                                \
  }
#define OSM_S_LOAD_B(b_, kc_)                                                                        \
  {                                                                                                  \
    const int cc_ = (kc_) / TAPS;                                                                    \
    const int tap_ = (kc_) - cc_ * TAPS;                                                             \
    const long long o_ = b_ok ? tap_ * b_tap + (2 * cc_) * b_step + b_lane : zoffB;                  \
    const long long st_ = b_ok ? b_step : 0;                                                         \
    const long long pl_ = b_ok ? b_plane : 0;                                                        \
    const char* bp_ = reinterpret_cast<const char*>(Bm) + o_;                                        \
    b_##00 = *reinterpret_cast<const uint4*>(bp_);                                                   \
    b_##10 = *reinterpret_cast<const uint4*>(bp_ + st_);                                             \
    b_##01 = *reinterpret_cast<const uint4*>(bp_ + pl_);                                             \
    b_##11 = *reinterpret_cast<const uint4*>(bp_ + pl_ + st_);                                       \
    if (NP == 3) {                                                                                   \
      b_##02 = *reinterpret_cast<const uint4*>(bp_ + 2 * pl_);                                       \
      b_##12 = *reinterpret_cast<const uint4*>(bp_ + 2 * pl_ + st_);                                 \
    }                                                                                                \
  }
#define OSM_S_READ(f_, st_, half_)                                                                   \
  _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                      \
    _Pragma("unroll") for (int q2 = 0; q2 < NP; ++q2)                                                \
      if (DBG == 10) f_[t][q2] = as_frag(make_uint4(t, q2, st_, half_));                             \
      else f_[t][q2] = *reinterpret_cast<const bf16x8_t*>(a_rd + q2 * S_PLANE +                      \
                                                     (64 * (half_) + 32 * t) * S_ROWB + 32 * (st_));
#define OSM_S_MMA(f_, b_, st_, half_)                                                                \
  {                                                                                                  \
    bf16x8_t bf[NP];                                                                                 \
    bf[0] = as_frag((st_) ? b_##10 : b_##00);                                                        \
    bf[1] = as_frag((st_) ? b_##11 : b_##01);                                                        \
    if (NP == 3) bf[NP - 1] = as_frag((st_) ? b_##12 : b_##02);                                      \
    _Pragma("unroll") for (int pa = NP - 1; pa >= 0; --pa)                                           \
      _Pragma("unroll") for (int pb = NP - 1 - pa; pb >= 0; --pb) {                                  \
        acc[2 * (half_)] =                                                                           \
            __builtin_amdgcn_mfma_f32_32x32x16_bf16(f_[0][pa], bf[pb], acc[2 * (half_)], 0, 0, 0);   \
        acc[2 * (half_) + 1] =                                                                       \
            __builtin_amdgcn_mfma_f32_32x32x16_bf16(f_[1][pa], bf[pb], acc[2 * (half_) + 1], 0, 0, 0); \
      }                                                                                              \
  }
// one chunk: phase 1 = prefetch next B set, split + store this chunk's A registers;
//            phase 2 = refill this A register set two chunks ahead, 2 k16-steps x 4 tiles x (6|3) MFMAs
#define OSM_S_CHUNK(ra_, bc_, bn_, it_)                                                              \
  {                                                                                                  \
    const int it_s = (it_);                                                                          \
    OSM_STAMP(0)                                                                                     \
    /* prefetches are UNCONDITIONAL (tail chunks re-load the last chunk): with conditional loads the   \
       compiler's vmcnt bookkeeping must assume the shortest path and waits for the newest loads */   \
    OSM_S_LOAD_B(bn_, min(kc0 + (it_) + 1, kc1 - 1));                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                  \
      uint2 pl[NP];                                                                                  \
      if (DBG == 11) { asm volatile("" ::"v"(ra_[i].x), "v"(ra_[i].y), "v"(ra_[i].z), "v"(ra_[i].w)); continue; } \
      split_planes<NP>(ra_[i], pl);                                                                  \
      _Pragma("unroll") for (int q2 = 0; q2 < NP; ++q2)                                              \
        *reinterpret_cast<uint2*>(As + q2 * S_PLANE + (r0 + 32 * i) * S_ROWB + 8 * cg) = pl[q2];     \
    }                                                                                                \
    OSM_STAMP(1)                                                                                     \
    __syncthreads();                                                                                 \
    OSM_STAMP(2)                                                                                     \
    OSM_S_LOAD_A(ra_, min(kc0 + (it_) + 2, kc1 - 1));                                          \
    /* software-pipelined fragment reads: the 6 ds_read_b128 of quarter q+1 are issued BEFORE the 12   \
       MFMAs of quarter q (the compiler otherwise sinks every read next to its first use and exposes   \
       the LDS latency ~12 times per chunk: measured 2450 cycles for 1536 cycles of MFMA) */           \
    bf16x8_t fx[2][NP], fy[2][NP];                                                                   \
    OSM_S_READ(fx, 0, 0)                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    OSM_STAMP(4)                                                                                     \
    OSM_S_READ(fy, 0, 1)                                                                             \
    OSM_S_MMA(fx, bc_, 0, 0)                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    OSM_STAMP(5)                                                                                     \
    OSM_S_READ(fx, 1, 0)                                                                             \
    OSM_S_MMA(fy, bc_, 0, 1)                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    OSM_STAMP(6)                                                                                     \
    OSM_S_READ(fy, 1, 1)                                                                             \
    OSM_S_MMA(fx, bc_, 1, 0)                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    OSM_STAMP(7)                                                                                     \
    OSM_S_MMA(fy, bc_, 1, 1)                                                                         \
    OSM_STAMP(3)                                                                                     \
    __syncthreads();                                                                                 \
  }

  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  const int lr = lane & 31, lk = lane >> 5;
  const unsigned char* a_rd = As + lr * S_ROWB + 16 * lk;

  const int nk = kc1 - kc0;
  if (nk > 0) {
    OSM_S_LOAD_A(raE, kc0);
    OSM_S_LOAD_B(bE, kc0);
    OSM_S_LOAD_A(raO, min(kc0 + 1, kc1 - 1));
  }
  if (grp == 1) __syncthreads();   // phase shift of group 1 (group 0 pays the matching barrier after the loop)
  for (int it = 0; it < nk; it += 2) {
    OSM_S_CHUNK(raE, bE, bO, it);
    if (it + 1 < nk) OSM_S_CHUNK(raO, bO, bE, it + 1);
  }
  if (grp == 0) __syncthreads();
#undef OSM_S_LOAD_A
#undef OSM_S_LOAD_B
#undef OSM_S_CHUNK
#undef OSM_S_READ
#undef OSM_S_MMA
  if (!tile_ok) return;

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const bool partial = p.splitk > 1;
  float* Cb = partial ? p.ws + ((long long)ks * p.M) * p.N : p.C;
  const float* Rb = (p.res && !partial) ? p.res : nullptr;
  const long long ldc = partial ? (long long)p.N : p.ldc;
  const int n = n0 + 32 * wave + lr;
  if (n >= p.N) return;
  const float bv = (!partial && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
  for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + 32 * tm + (e & 3) + 8 * (e >> 2) + 4 * lk;
      if (m >= p.M) continue;
      float v = acc[tm][e];
      if (!partial) {
        v = v * p.alpha + bv;
        if (Rb) v += Rb[(long long)m * p.ldr + n];
        if (p.accumulate) v += Cb[(long long)m * ldc + n];
      }
      Cb[(long long)m * ldc + n] = v;
    }
  }
}

// OIHW fp32 -> NP bf16 planes in MFMA-fragment order
//   [plane][tap][k16-step s][n/32 j][lane l][e],  n = 32 j + (l & 31),  k = 16 s + 8 (l >> 5) + e
// forward: n = Cout, k = Cin ; data-gradient: n = Cin, k = Cout, taps flipped.  Out-of-range (n, k) are zero;
// the step count is even (2 per 32-wide chunk) so a chunk never reads past the image.
__global__ void pack_weight_bf16s_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout,
                                         int Cin, int k, int np, int dgrad) {
  const int N = dgrad ? Cin : Cout;
  const int K = dgrad ? Cout : Cin;
  const int nt32 = (N + 31) / 32;
  const int ksteps = 2 * ((K + 31) / 32);
  const long long per_plane = (long long)k * k * ksteps * nt32 * 512;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_plane;
       i += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7);
    const int l = (int)((i >> 3) & 63);
    long long rest = i >> 9;
    const int j = (int)(rest % nt32);
    rest /= nt32;
    const int s = (int)(rest % ksteps);
    const int tap = (int)(rest / ksteps);
    const int nn = 32 * j + (l & 31);
    const int kk = 16 * s + 8 * (l >> 5) + e;
    float v = 0.f;
    if (nn < N && kk < K) {
      int kh = tap / k, kw = tap % k;
      int co = nn, ci = kk;
      if (dgrad) {
        kh = k - 1 - kh;
        kw = k - 1 - kw;
        co = kk;
        ci = nn;
      }
      v = w[(((long long)co * Cin + ci) * k + kh) * k + kw];
    }
    float rr = v;
    for (int qq = 0; qq < np; ++qq) {
      const __bf16 b = (__bf16)rr;
      out[qq * per_plane + i] = __builtin_bit_cast(unsigned short, b);
      rr -= (float)b;
    }
  }
}
