// Small-M ("skinny") implicit GEMM for the low-resolution layers (included inside igemm.hip's anonymous namespace,
// after igemm_bf16s.inc.h whose split helpers / fragment-order weight image it shares).
//
// At batch 1 the 8x8 ... 32x32 levels of the UNet have M = 64 ... 1024 pixel rows against 1-57 MB of weights per
// layer: their time is set by how fast the weight image streams from HBM, not by the matrix cores.  The tiled kernels
// (128-column workgroup tiles, split-K over grid.y + a reduce launch, weight fragments fetched two steps ahead) ran
// those layers at 0.3-2 TB/s of weight stream (round 1: 8x8 1024->1024 3x3 in 28 us for 57 MB).  This kernel is built
// for that regime instead:
//   * a WAVE owns one 32-column tile and a contiguous range of k16-steps of the [tap][k16-step] reduction; it keeps
//     ALL rows of its M-tile (RB <= 8 row blocks of 32 rows = up to 256 rows) in accumulators, so every weight byte is
//     read exactly once per M-tile and the loads of D (= 3 ... 8) steps are in flight at any time;
//   * no LDS staging and no barrier in the main loop: the A operand (<= 2 MB, L2 resident) is gathered per lane
//     straight into MFMA-fragment order (8 consecutive channels of the tap-shifted pixel = one or two 16-byte loads,
//     masked outside the image), split into bf16 planes in registers (fp32 family) and used at once;
//   * the four waves of a workgroup take adjacent k-ranges of the SAME column tile and add their accumulators
//     through LDS, so a 32-way split of K costs 8 (not 32) fp32 partials per output element -- or none when the
//     in-workgroup split alone yields enough waves; grid.y > 1 hands fp32 partials to splitk_reduce_kernel.
// Rows: M-tile z covers rows [z * RB * 32, ...); columns: blockIdx.x; k: blockIdx.y * 4 + wave.

#if OSM_ACT_IS_F16
typedef uint4 rawa_t;            // 8 halfs = one 16-byte load, already the fragment
__device__ __forceinline__ rawa_t ld_rawa(const char* p) { return *reinterpret_cast<const uint4*>(p); }
template <int NP>
__device__ __forceinline__ void rawa_planes(const rawa_t& r, bool ok, uint4 (&pl)[NP]) {
  pl[0] = ok ? r : make_uint4(0u, 0u, 0u, 0u);
}
#else
struct rawa_t { float4 a, b; };  // 8 floats
__device__ __forceinline__ rawa_t ld_rawa(const char* p) {
  rawa_t r;
  r.a = *reinterpret_cast<const float4*>(p);
  r.b = *reinterpret_cast<const float4*>(p + 16);
  return r;
}
template <int NP>
__device__ __forceinline__ void rawa_planes(const rawa_t& r, bool ok, uint4 (&pl)[NP]) {
  split_frag8<NP>(sel4(ok, r.a), sel4(ok, r.b), pl);
}
#endif

template <int NP, int RB, int TAPS>
__global__ __launch_bounds__(256, 1) void skinny_kernel(const act_t* __restrict__ Aglob,
                                                         const unsigned short* __restrict__ Bglob, IGemmParams p) {
  // Prefetch distance in k16-steps, the SAME for the weight fragments and the A gathers of a step: vmcnt retires
  // loads in issue order, so what a wave may keep in flight while it waits for step s is exactly what it issued after
  // the loads of step s -- two streams with different distances would cap both at the shorter one.  A step is
  // 6 RB MFMAs (192 RB cycles); D steps cover ~1.5-2 us of HBM latency under load.
  constexpr int D = RB <= 2 ? 8 : (RB <= 4 ? 6 : 3);
  __shared__ __attribute__((aligned(16))) float red[4 * RB * 1024];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lk = lane >> 5;
  const int jn = blockIdx.x;
  const int m0 = blockIdx.z * (RB * 32);

  // ---- this wave's range of k16-steps (step s = tap * ksteps + ks)
  const int nsteps = TAPS * p.ksteps;
  const int nsplit = 4 * gridDim.y;
  const int per = (nsteps + nsplit - 1) / nsplit;
  const int s0 = min(nsteps, (int)(blockIdx.y * 4 + wave) * per);
  const int s1 = min(nsteps, s0 + per);

  // ---- A gather coordinates: row rb * 32 + lr, channels 16 ks + 8 lk .. + 7 of the tap-shifted pixel
  const long long rowB = (long long)p.lda * ACT_B;
  const char* __restrict__ abase = reinterpret_cast<const char*>(Aglob) + 8 * ACT_B * lk;
  unsigned aoff[RB], amask[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int m = m0 + rb * 32 + lr;
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
    amask[rb] = mk;
    aoff[rb] = (unsigned)((m < p.M ? (long long)m : 0LL) * rowB);
  }

  // ---- B fragment addressing: image [plane][step][n/32][lane][8] (steps of all taps are contiguous)
  const unsigned b_lane = (unsigned)((jn * 64 + lane) * 16);
  const unsigned b_step = (unsigned)p.nt32 * 1024u;
  const unsigned b_plane = b_step * (unsigned)nsteps;
  const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(Bglob)), 0, 0x7fffffff, 0x00020000);

  f32x16 acc[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  uint4 bq[D][NP];
  rawa_t ra[D][RB];

  const int slast = max(s0, s1 - 1);   // every prefetch is unconditional; past the end it re-reads the last step
#define OSM_K_LOAD_B(slot_, s_)                                                              \
  {                                                                                          \
    const unsigned so_ = (unsigned)min((s_), slast) * b_step;                                \
    _Pragma("unroll") for (int q2 = 0; q2 < NP; ++q2)                                        \
      bq[slot_][q2] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(      \
          brsrc, (int)b_lane, (int)(so_ + q2 * b_plane), 0));                                \
  }
#define OSM_K_LOAD_A(slot_, s_)                                                              \
  {                                                                                          \
    const int sc_ = min((s_), slast);                                                        \
    const int tap_ = sc_ / p.ksteps, ks_ = sc_ - tap_ * p.ksteps;                            \
    int toff_ = 0;                                                                           \
    if (TAPS == 9) toff_ = (int)(((tap_ / 3 - 1) * p.W + (tap_ % 3 - 1)) * rowB);            \
    const char* pa_ = abase + (long long)ks_ * (16 * ACT_B);                                 \
    _Pragma("unroll") for (int rb = 0; rb < RB; ++rb) {                                      \
      const bool ok_ = (amask[rb] >> tap_) & 1u;                                             \
      ra[slot_][rb] = ld_rawa(pa_ + (long long)(int)(aoff[rb] + (unsigned)(ok_ ? toff_ : 0))); \
    }                                                                                        \
  }
#define OSM_K_COMPUTE(aslot_, bslot_, s_)                                                    \
  {                                                                                          \
    const int tap_ = (s_) / p.ksteps;                                                        \
    _Pragma("unroll") for (int rb = 0; rb < RB; rb += 2) {                                   \
      uint4 p0[NP], p1[NP];                                                                  \
      rawa_planes<NP>(ra[aslot_][rb], (amask[rb] >> tap_) & 1u, p0);                         \
      rawa_planes<NP>(ra[aslot_][rb + 1], (amask[rb + 1] >> tap_) & 1u, p1);                 \
      _Pragma("unroll") for (int pa = NP - 1; pa >= 0; --pa)                                 \
        _Pragma("unroll") for (int pb = NP - 1 - pa; pb >= 0; --pb) {                        \
          acc[rb] = mma16<NP>(p0[pa], bq[bslot_][pb], acc[rb]);                              \
          acc[rb + 1] = mma16<NP>(p1[pa], bq[bslot_][pb], acc[rb + 1]);                      \
        }                                                                                    \
    }                                                                                        \
  }

  if (s1 > s0) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      OSM_K_LOAD_B(d, s0 + d)
      OSM_K_LOAD_A(d, s0 + d)
    }
    for (int sb = s0; sb < s1; sb += D) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const int s = sb + i;
        if (s < s1) OSM_K_COMPUTE(i, i, s)
        OSM_K_LOAD_B(i, s + D)
        OSM_K_LOAD_A(i, s + D)
      }
    }
  }
#undef OSM_K_LOAD_A
#undef OSM_K_LOAD_B
#undef OSM_K_COMPUTE

  // ---- add the four waves' accumulators through LDS.  C/D layout: col = lane & 31, row = (e&3) + 8 (e>>2) + 4 (lane>>5)
  float* mine = red + wave * (RB * 1024);
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int e = 0; e < 16; ++e)
      mine[(rb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk) * 32 + lr] = acc[rb][e];
  __syncthreads();

  const bool partial = gridDim.y > 1;
  const int n0 = jn * 32;
#pragma unroll
  for (int i = 0; i < RB; ++i) {          // RB * 256 float4 groups: 8 per row
    const int idx = tid + 256 * i;
    const int row = idx >> 3, c4 = (idx & 7) * 4;
    const float4 v0 = *reinterpret_cast<const float4*>(red + row * 32 + c4);
    const float4 v1 = *reinterpret_cast<const float4*>(red + RB * 1024 + row * 32 + c4);
    const float4 v2 = *reinterpret_cast<const float4*>(red + 2 * RB * 1024 + row * 32 + c4);
    const float4 v3 = *reinterpret_cast<const float4*>(red + 3 * RB * 1024 + row * 32 + c4);
    float v[4] = {(v0.x + v1.x) + (v2.x + v3.x), (v0.y + v1.y) + (v2.y + v3.y), (v0.z + v1.z) + (v2.z + v3.z),
                  (v0.w + v1.w) + (v2.w + v3.w)};
    const int m = m0 + row, n = n0 + c4;
    if (m >= p.M || n >= p.N) continue;
    if (partial) {
      float* w = p.ws + ((long long)blockIdx.y * p.M + m) * p.N + n;
      if (n + 3 < p.N && (p.N & 3) == 0) {
        *reinterpret_cast<float4*>(w) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        for (int e = 0; e < 4 && n + e < p.N; ++e) w[e] = v[e];
      }
    } else {
      act_t* c = p.C + (long long)m * p.ldc + n;
      const act_t* r = p.res ? p.res + (long long)m * p.ldr + n : nullptr;
      for (int e = 0; e < 4 && n + e < p.N; ++e) {
        float o = v[e] * p.alpha + (p.bias ? p.bias[n + e] : 0.f);
        if (r) o += osm::ld1(r + e);
        if (p.accumulate) o += osm::ld1(c + e);
        osm::st1(c + e, o);
      }
    }
  }
}
