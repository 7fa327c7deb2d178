// 3x3 split-bf16 convolution on HALO tiles (included inside igemm.hip's anonymous namespace, after
// igemm_bf16s.inc.h whose split helpers / fragment-order weight image it shares).
//
// igemm_bf16s_kernel treats the 9 taps as 9 independent K-chunks: every activation is fetched from
// L2/HBM, split into bf16 planes on the VALU and stored to LDS nine times per N-tile, with one barrier
// per 48 MFMAs.  Measured (round 1): MFMA pipe ~50 % busy, 3x HBM over-fetch.  Here a workgroup owns a
// SPATIAL patch of 8 x 16 output pixels (= the 128 rows of the GEMM tile) and, per 32-channel slab,
// stages the (8+2) x (16+2) halo ONCE: 180 pixels x 32 channels -> NP bf16 planes in LDS.  The nine taps
// are then nine shifted windows of the same LDS image:
//     A-fragment address = lane base + compile-time constant (tap shift, k16 half, plane, row block),
// so the MFMA stream of a slab is 9 taps x 2 k16-steps x 24 MFMAs = 432 MFMAs per wave between two
// barriers, fed by ds_read_b128 (immediate offsets, no address VALU) and by the pre-swizzled weight
// fragments streamed straight from L2 into VGPRs.  Per slab and thread the staging work is 6 float4
// loads + 6 splits (was 9 x 4), i.e. ~7x less VALU / LDS-store work and 9x fewer A reads.
//
// Row r of MFMA row-block i (0..3) is patch pixel (py, px) = (i + 4 (r >> 4), r & 15).  LDS pixel rows are
// 80 B apart (32 bf16 + pad) and the halo row pitch is 20 pixels: with lanes 16-31 four halo rows (= 80
// LDS rows = 5 x 16) below lanes 0-15, every ds_read_b128 lane group of gfx950 ({0-3,12-15,20-27}, ...) sees
// 16 distinct bank quads for every tap shift -> 4 LDS cycles per fragment read (pitch 18 / adjacent rows: 8).

// PW = patch width: 16 (8 x 16 patch = 128 GEMM rows, 4 MFMA row blocks) for W >= 16, or 8 (8 x 8 patch = 64 rows,
// 2 row blocks: the 8x8 layers, where an 8 x 16 patch would be mostly halo).
// PH = patch height: 8, or 16 with PW = 16 (a 16 x 16 patch = 256 GEMM rows per wave, 8 row blocks: every weight fragment
// then feeds 48 instead of 24 MFMAs -- the weight-fragment loads are what the CU's vector-memory path spends half its time
// on at 8 x 16 -- at the price of one workgroup per CU: 88 KB of LDS, accumulators in AGPRs).
template <int PW, int PH = 8> struct HaloGeom {
  static constexpr int HW = PW + 2;                 // staged pixels per halo row
  static constexpr int HP = PW == 16 ? 20 : 10;     // LDS pitch of a halo row, in pixels
  static constexpr int PIX = (PH + 2) * HW;         // staged pixels
  static constexpr int NJ = (PIX * 8 + 255) / 256;  // float4 staging loads per thread and slab
  static constexpr int ROWS = (PH + 2) * HP + 8;    // LDS pixel rows (+ a dump row for unused staging slots)
  static constexpr int PLANE = ROWS * S_ROWB;       // bytes per plane
  static constexpr int RB = PW * PH / 32;           // MFMA row blocks (32 rows each)
};
// BR = weight-fragment register sets (a divisor of the 18 steps of a slab); a fragment is loaded BR - 1 steps before
// its use.  3 on the MFMA-bound layers; deeper rings (6 / 9) keep more of the weight stream in flight for the
// weight-bandwidth-bound low-resolution layers (few rows per weight byte: the loads, not the MFMAs, set the pace).

// NARROW (8 x 16 patches only): layers with <= 32 output columns (the network's head: 256 -> 8, and the stem's data-gradient:
// 256 -> 4).  The four waves then split the four row blocks of the patch instead of the (empty) column tiles: 108 instead of
// 432 MFMAs per wave and slab.
// NT = 32-column tiles per wave (1, or 2 on 8 x 16 patches): with ONE MFMA per product (the fp16 family) every A fragment read from
// LDS feeds a single MFMA per column tile -- 1 KB of LDS reads per MFMA, i.e. the CU's whole LDS bandwidth at the matrix rate.  Two
// column tiles per wave (workgroup = 128 rows x 256 columns, 8 accumulators per wave) halve that; the weight-fragment traffic per
// MFMA stays what it was (a fragment still feeds the 4 row blocks).
// HP ("f16x3", NP = 2; round 5: the 8 x 8 layers, where the Winograd kernel does not apply): the two planes are IEEE halves of
// x * 2^ex (2^ex from the per-image max |x|, IGemmParams::xmax) against a weight image of halves of w * 2^ew behind its scale word
// -- three fp16 MFMAs per product instead of six bf16 ones and 4 instead of 6 bytes of weight stream per weight; the accumulators
// are rescaled (exactly, by a power of two) before the epilogue.  A patch belongs to one image, so one scale per workgroup.
template <int NP, bool GNF, int PW = 16, int BR = 3, int PH = 8, bool NARROW = false, int NT = 1, bool HP = false>
__global__ __launch_bounds__(256, PH == 16 ? 1 : 2) void conv3_halo_bf16s_kernel(const act_t* __restrict__ Aglob,
                                                                   const unsigned short* __restrict__ Bglob,
                                                                   IGemmParams p) {
  constexpr int B_RING = BR, B_DIST = BR - 1;
  static_assert(PH == 8, "patches are 8 x 8 or 8 x 16 (the 16 x 16 variant of round 2 lost everywhere but one layer and was removed)");
  static_assert(!NARROW || (PW == 16 && PH == 8), "the narrow variant works on 8 x 16 patches");
  static_assert(NT == 1 || (NT == 2 && PW == 16 && !NARROW), "two column tiles per wave: 8 x 16 patches only");
  static_assert(!HP || (NP == 2 && !GNF && !OSM_ACT_IS_F16), "f16x3: two half planes, fp32 activations, no fused GroupNorm (x_maxabs is the range of x)");
  using GEO = HaloGeom<PW, PH>;
  constexpr int HALO_W = GEO::HW, HALO_P = GEO::HP, HALO_PIX = GEO::PIX, H_PLANE = GEO::PLANE, NJ = GEO::NJ,
                RB = GEO::RB;
  __shared__ __attribute__((aligned(16))) unsigned char As[NP * H_PLANE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // ---- XCD-aware tile mapping; M-tiles enumerate (image, patch row, patch col), N-tiles fastest
  const int nt = p.mtiles * p.ntiles;
  const int bid = blockIdx.x;
  const int q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int tile_n = id % p.ntiles, tile_m = id / p.ntiles;
  const int tpx = (p.W + PW - 1) / PW, tpy = (p.H + PH - 1) / PH;
  const int tx = tile_m % tpx, ty = (tile_m / tpx) % tpy, img = tile_m / (tpx * tpy);
  const int x0 = tx * PW, y0 = ty * PH, n0 = tile_n * (BN * NT);

  const int ks = blockIdx.y;
  const int nslab = (p.K + BK - 1) / BK;
  const int per = (nslab + p.splitk - 1) / p.splitk;
  const int kc0 = ks * per;
  const int kc1 = min(nslab, kc0 + per);

  float xscale = 1.f, oscale = 1.f;
  if constexpr (HP) {      // as igemm_bf16s_kernel<1, 2, true>: fold the image's partial maxima, x * 2^ex into [2^13, 2^14)
    static_assert(OSM_MAXABS_PARTS == 1024, "four partial maxima per thread");
    const unsigned* xm = reinterpret_cast<const unsigned*>(p.xmax) + (long long)img * OSM_MAXABS_PARTS;
    unsigned mb = max(max(xm[tid], xm[tid + 256]), max(xm[tid + 512], xm[tid + 768]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, o, 64));
    unsigned* red_u = reinterpret_cast<unsigned*>(As);
    if (lane == 0) red_u[wave] = mb;
    __syncthreads();
    mb = max(max(red_u[0], red_u[1]), max(red_u[2], red_u[3]));
    __syncthreads();              // the staging stores that follow reuse the LDS
    const float mx = __uint_as_float(mb);
    int ex = 0;
    if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &ex); ex = min(14 - ex, 100); }
    xscale = mx == mx ? ldexpf(1.f, ex) : mx;       // a NaN in the input poisons the output
    oscale = ldexpf(1.f, -ex) / p.wscale[0];
  }

  // ---- halo staging coordinates: slot s = tid + 256 j -> halo pixel (tid >> 3) + 32 j, float4 group tid & 7
  const int cg = tid & 7;
  const long long rowB = (long long)p.lda * ACT_B;
  const char* __restrict__ sbaseA = reinterpret_cast<const char*>(Aglob) + (long long)img * p.H * p.W * rowB;
  unsigned voff[NJ], woff[NJ], vmask = 0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int hp = (tid >> 3) + 32 * j;
    const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
    woff[j] = (unsigned)((hp < HALO_PIX ? hy * HALO_P + hx : (PH + 2) * HALO_P) * S_ROWB + 8 * cg);
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = hp < HALO_PIX && y >= 0 && y < p.H && x >= 0 && x < p.W;
    const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.W - 1);
    voff[j] = (unsigned)((long long)(yc * p.W + xc) * rowB);
    vmask |= (ok ? 1u : 0u) << j;
  }

  // ---- B fragment addressing (image [plane][tap][k16-step][n/32][lane][8])
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int jn = (n0 >> 5) + (NARROW ? 0 : NT * wave_u);      // this wave's first 32-column tile
  const bool b_ok = jn < p.nt32;
  const unsigned b_nt = (NT == 2 && jn + 1 < p.nt32) ? 1024u : 0u;   // byte offset of its second tile (or the first again)
  const unsigned b_lane = b_ok ? (unsigned)((jn * 64 + lane) * 16) : (unsigned)(lane & 1) * 16u;
  // uniform 32-bit byte offsets (the largest image, 2048 -> 1024 channels, is 113 MB)
  const unsigned b_step = b_ok ? (unsigned)p.nt32 * 1024u : 0u;
  const unsigned b_tap = b_step * (unsigned)p.ksteps;
  const unsigned b_plane = b_tap * 9u;
  const char* __restrict__ sbaseB0 =
      b_ok ? reinterpret_cast<const char*>(Bglob) : reinterpret_cast<const char*>(g_zero_page);
  // buffer resource over the weight image: loads are `buffer_load_dwordx4 v, v_lane, s[rsrc], s_step offen` -- the
  // per-step offset stays in an SGPR and the only address VGPR is the (constant) lane offset
  const __amdgpu_buffer_rsrc_t brsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(sbaseB0), 0, 0x7fffffff, 0x00020000);

  f32x16 acc[RB * NT];     // [column tile][row block]
#pragma unroll
  for (int i = 0; i < RB * NT; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  // lane row lr of row block i: PW = 16 -> patch pixel (i + PH/2 (lr >> 4), lr & 15);  PW = 8 -> (4 i + (lr >> 3), lr & 7)
  const int lr = lane & 31, lk = lane >> 5;
  constexpr int RB_STRIDE = (PW == 16 ? HALO_P : 4 * HALO_P) * S_ROWB;   // bytes between row blocks
  const unsigned char* a_rd =
      As + (PW == 16 ? (lr >> 4) * ((PH / 2) * HALO_P) + (lr & 15) : (lr >> 3) * HALO_P + (lr & 7)) * S_ROWB + 16 * lk +
      (NARROW ? wave_u * RB_STRIDE : 0);

  float4 ra[NJ];
  float4 gm, gr, gg, gb;   // GNF: mean | rstd | g | b of this thread's 4 channels of the current slab
  gm = gr = gg = gb = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* __restrict__ gtab = GNF ? p.gn_table + (long long)img * 4 * p.K : nullptr;
  unsigned okm = 0;
  uint4 bq[B_RING][NT][NP];   // rolling weight-fragment sets: step g lives in bq[g % B_RING]

#define OSM_H_LOAD_A(cc_)                                                                  \
  {                                                                                        \
    const bool cok_ = (cc_) * BK + 4 * cg < p.K;                                           \
    const unsigned d_ = (unsigned)((cc_) * (BK * ACT_B) + 4 * ACT_B * cg);                 \
    okm = cok_ ? vmask : 0u;                                                               \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                         \
      ra[j] = osm::ld4(reinterpret_cast<const act_t*>(sbaseA + (voff[j] + (cok_ ? d_ : 0u)))); \
    if (GNF) {                                                                             \
      const float* gt_ = gtab + (cok_ ? (cc_) * BK + 4 * cg : 0);                          \
      gm = *reinterpret_cast<const float4*>(gt_);                                          \
      gr = *reinterpret_cast<const float4*>(gt_ + p.K);                                    \
      gg = *reinterpret_cast<const float4*>(gt_ + 2 * p.K);                                \
      gb = *reinterpret_cast<const float4*>(gt_ + 3 * p.K);                                \
    }                                                                                      \
  }
// weight fragments of step s_ (tap s_/2, k16 half s_&1) of slab cc_
#define OSM_H_LOAD_B(slot_, cc_, s_)                                                       \
  {                                                                                        \
    const unsigned so_ = (unsigned)((s_) >> 1) * b_tap + (unsigned)(2 * (cc_) + ((s_) & 1)) * b_step; \
    _Pragma("unroll") for (int t2 = 0; t2 < NT; ++t2)                                      \
      _Pragma("unroll") for (int q2 = 0; q2 < NP; ++q2)                                    \
        bq[slot_][t2][q2] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128( \
            brsrc, (int)b_lane, (int)(so_ + q2 * b_plane + (unsigned)t2 * b_nt), 0));     \
  }
// A fragments of half-step hs_ (step hs_/2, row blocks 2 (hs_&1) and 2 (hs_&1) + 1)
#define OSM_H_READ(f_, hs_)                                                                \
  {                                                                                        \
    constexpr int t_ = (hs_) >> 2, kk_ = ((hs_) >> 1) & 1, h_ = (hs_) & 1;                 \
    constexpr int off_ = ((t_ / 3) * HALO_P + (t_ % 3)) * S_ROWB + 32 * kk_;               \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                          \
      _Pragma("unroll") for (int q2 = 0; q2 < NP; ++q2)                                    \
        f_[t][q2] = *reinterpret_cast<const uint4*>(a_rd + q2 * H_PLANE +                  \
                                                       (2 * h_ + t) * RB_STRIDE + off_);             \
  }
#define OSM_H_MMA(f_, slot_, h_)                                                           \
  _Pragma("unroll") for (int pa = NP - 1; pa >= 0; --pa)                                   \
    _Pragma("unroll") for (int pb = NP - 1 - pa; pb >= 0; --pb)                            \
      _Pragma("unroll") for (int t2 = 0; t2 < NT; ++t2) {                                  \
        acc[RB * t2 + 2 * (h_)] = HP ? mma16h(f_[0][pa], bq[slot_][t2][pb], acc[RB * t2 + 2 * (h_)])          \
                                     : mma16<NP>(f_[0][pa], bq[slot_][t2][pb], acc[RB * t2 + 2 * (h_)]);      \
        acc[RB * t2 + 2 * (h_) + 1] = HP ? mma16h(f_[1][pa], bq[slot_][t2][pb], acc[RB * t2 + 2 * (h_) + 1])  \
                                         : mma16<NP>(f_[1][pa], bq[slot_][t2][pb], acc[RB * t2 + 2 * (h_) + 1]); \
      }

  if (kc1 > kc0) {
    OSM_H_LOAD_A(kc0);
#pragma unroll
    for (int s0 = 0; s0 < B_DIST; ++s0) OSM_H_LOAD_B(s0, kc0, s0);
    for (int c = kc0; c < kc1; ++c) {
      // split the staged halo of slab c into LDS (masked lanes / slots store zeros)
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        uint2 pl[NP];
        float4 v = ra[j];
        if (GNF) {   // GroupNorm(+FiLM)(+SiLU) of the input on the fly; the zero padding applies to the result
          v.x = ((v.x - gm.x) * gr.x) * gg.x + gb.x;
          v.y = ((v.y - gm.y) * gr.y) * gg.y + gb.y;
          v.z = ((v.z - gm.z) * gr.z) * gg.z + gb.z;
          v.w = ((v.w - gm.w) * gr.w) * gg.w + gb.w;
          if (p.gn_silu) {
            v.x = osm::silu_f(v.x); v.y = osm::silu_f(v.y); v.z = osm::silu_f(v.z); v.w = osm::silu_f(v.w);
          }
        }
        if constexpr (HP) {
          const float4 z = sel4((okm >> j) & 1u, v);
          uint2 ph[2];
          split_f16x2(make_float4(z.x * xscale, z.y * xscale, z.z * xscale, z.w * xscale), ph);
          pl[0] = ph[0]; pl[NP - 1] = ph[1];
        } else {
          split_planes<NP>(sel4((okm >> j) & 1u, v), pl);
        }
#pragma unroll
        for (int q2 = 0; q2 < NP; ++q2)
          *reinterpret_cast<uint2*>(As + q2 * H_PLANE + woff[j]) = pl[q2];
      }
      const int cn = min(c + 1, kc1 - 1);
      OSM_H_LOAD_A(cn);                      // next slab's halo: in flight during the MFMA stream
      __syncthreads();
      uint4 fx[2][NP], fy[2][NP];
      if constexpr (NARROW) {
        // one row block (= this wave's) per step, two accumulators alternate so that consecutive MFMAs are independent; the A
        // fragments of step s + 1 are read before the MFMAs of step s (fx[0] / fx[1] by step parity): with 6 MFMAs per step the LDS
        // latency was exposed in every step (PMC: matrix pipe 36 % busy, 65 % of the wave cycles waiting)
#define OSM_H_NREAD(s_)                                                                    \
        {                                                                                  \
          constexpr int t_ = (s_) >> 1, kk_ = (s_) & 1;                                    \
          constexpr int off_ = ((t_ / 3) * HALO_P + (t_ % 3)) * S_ROWB + 32 * kk_;         \
          _Pragma("unroll") for (int q2 = 0; q2 < NP; ++q2)                                \
            fx[(s_) & 1][q2] = *reinterpret_cast<const uint4*>(a_rd + q2 * H_PLANE + off_); \
        }
#define OSM_H_NSTEP(s_)                                                                    \
        {                                                                                  \
          OSM_H_LOAD_B(((s_) + B_DIST) % B_RING, ((s_) + B_DIST < 18 ? c : cn), ((s_) + B_DIST) % 18); \
          if ((s_) + 1 < 18) OSM_H_NREAD(((s_) + 1) % 18)                                  \
          int cnt = 0;                                                                     \
          _Pragma("unroll") for (int pa = NP - 1; pa >= 0; --pa)                           \
            _Pragma("unroll") for (int pb = NP - 1 - pa; pb >= 0; --pb, ++cnt)             \
              acc[cnt & 1] = HP ? mma16h(fx[(s_) & 1][pa], bq[(s_) % B_RING][0][pb], acc[cnt & 1]) \
                                : mma16<NP>(fx[(s_) & 1][pa], bq[(s_) % B_RING][0][pb], acc[cnt & 1]); \
          __builtin_amdgcn_sched_barrier(0);                                               \
        }
        OSM_H_NREAD(0)
        OSM_H_NSTEP(0) OSM_H_NSTEP(1) OSM_H_NSTEP(2) OSM_H_NSTEP(3) OSM_H_NSTEP(4) OSM_H_NSTEP(5)
        OSM_H_NSTEP(6) OSM_H_NSTEP(7) OSM_H_NSTEP(8) OSM_H_NSTEP(9) OSM_H_NSTEP(10) OSM_H_NSTEP(11)
        OSM_H_NSTEP(12) OSM_H_NSTEP(13) OSM_H_NSTEP(14) OSM_H_NSTEP(15) OSM_H_NSTEP(16) OSM_H_NSTEP(17)
#undef OSM_H_NSTEP
#undef OSM_H_NREAD
      } else if constexpr (PW == 16) {
        OSM_H_READ(fx, 0)
#define OSM_H_STEP(s_)                                                                     \
        OSM_H_LOAD_B(((s_) + B_DIST) % B_RING, ((s_) + B_DIST < 18 ? c : cn), ((s_) + B_DIST) % 18); \
        OSM_H_READ(fy, 2 * (s_) + 1)                                                       \
        OSM_H_MMA(fx, (s_) % B_RING, 0)                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        if ((s_) + 1 < 18) OSM_H_READ(fx, (2 * (s_) + 2) % 36)                             \
        OSM_H_MMA(fy, (s_) % B_RING, 1)                                                    \
        __builtin_amdgcn_sched_barrier(0);
        OSM_H_STEP(0) OSM_H_STEP(1) OSM_H_STEP(2) OSM_H_STEP(3) OSM_H_STEP(4) OSM_H_STEP(5)
        OSM_H_STEP(6) OSM_H_STEP(7) OSM_H_STEP(8) OSM_H_STEP(9) OSM_H_STEP(10) OSM_H_STEP(11)
        OSM_H_STEP(12) OSM_H_STEP(13) OSM_H_STEP(14) OSM_H_STEP(15) OSM_H_STEP(16) OSM_H_STEP(17)
#undef OSM_H_STEP
      } else {
        OSM_H_READ(fx, 0)
        // 64-row patch: one pair of row blocks per step; fx serves the even steps, fy the odd ones
#define OSM_H_STEP2(s_)                                                                    \
        OSM_H_LOAD_B(((s_) + B_DIST) % B_RING, ((s_) + B_DIST < 18 ? c : cn), ((s_) + B_DIST) % 18); \
        OSM_H_READ(fy, 2 * ((s_) + 1))                                                     \
        OSM_H_MMA(fx, (s_) % B_RING, 0)                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        OSM_H_LOAD_B(((s_) + 1 + B_DIST) % B_RING, ((s_) + 1 + B_DIST < 18 ? c : cn), ((s_) + 1 + B_DIST) % 18); \
        if ((s_) + 2 < 18) OSM_H_READ(fx, 2 * ((s_) + 2))                                  \
        OSM_H_MMA(fy, ((s_) + 1) % B_RING, 0)                                              \
        __builtin_amdgcn_sched_barrier(0);
        OSM_H_STEP2(0) OSM_H_STEP2(2) OSM_H_STEP2(4) OSM_H_STEP2(6) OSM_H_STEP2(8) OSM_H_STEP2(10)
        OSM_H_STEP2(12) OSM_H_STEP2(14) OSM_H_STEP2(16)
#undef OSM_H_STEP2
      }
      __syncthreads();                       // every wave is done reading this slab's LDS image
    }
  }
#undef OSM_H_LOAD_A
#undef OSM_H_LOAD_B
#undef OSM_H_READ
#undef OSM_H_MMA
  if constexpr (HP) {      // undo both power-of-two scales (exact)
#pragma unroll
    for (int i = 0; i < RB * NT; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] *= oscale;
  }

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  // Vector form (see igemm_bf16s.inc.h: stores are issue-bound): each wave transposes its (RB x 32) x 32 tile through its
  // own slice of the staging LDS, RP rows per pass, so that a lane owns 4 consecutive columns of a pixel.
  if constexpr (!NARROW) {
    const bool partial_ = p.splitk > 1;
    const auto al = [](const void* q_, unsigned bytes) { return (reinterpret_cast<unsigned long long>(q_) & (bytes - 1)) == 0; };
    const bool vec = (p.N & 3) == 0 &&
                     (partial_ ? al(p.ws, 16)
                               : ((p.ldc & 3) == 0 && al(p.C, 4 * ACT_B) && (!p.bias || al(p.bias, 16)) &&
                                  (!p.res || ((p.ldr & 3) == 0 && al(p.res, 4 * ACT_B))) &&
                                  (!p.colsum || p.stat_mode == 1 || ((p.ld_sx & 3) == 0 && al(p.stat_x, 4 * ACT_B) && al(p.stat_table, 16)))));
    if (vec) {
      constexpr int LDSB = NP * H_PLANE;
      constexpr int RP = LDSB >= 4 * 64 * 128 ? 64 : (LDSB >= 4 * 32 * 128 ? 32 : (LDSB >= 4 * 16 * 128 ? 16 : 8));
      static_assert(LDSB >= 4 * RP * 128 && (RB * 32) % RP == 0, "transpose slices must fit the staging LDS");
      float* tb = reinterpret_cast<float*>(As) + wave * (RP * 32);
      const int c4 = 4 * (lane & 7), rr = lane >> 3;
#pragma unroll
      for (int t2 = 0; t2 < NT; ++t2) {
      const int nb = n0 + 32 * (NT * wave + t2) + c4;
      const bool nok = nb < p.N;
      float4 bv4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!partial_ && p.bias && nok) bv4 = *reinterpret_cast<const float4*>(p.bias + nb);
      const bool stats_ = p.colsum != nullptr && !partial_;
      float q1[4] = {0.f, 0.f, 0.f, 0.f}, q2[4] = {0.f, 0.f, 0.f, 0.f};
      StatCol sc[4] = {};
      if (stats_ && p.stat_mode == 2 && nok) {
        const float* tbl = p.stat_table + (long long)img * 4 * p.N + nb;
        const float4 tm_ = *reinterpret_cast<const float4*>(tbl), tr_ = *reinterpret_cast<const float4*>(tbl + p.N);
        const float4 tg_ = *reinterpret_cast<const float4*>(tbl + 2 * p.N), tb_ = *reinterpret_cast<const float4*>(tbl + 3 * p.N);
        sc[0] = StatCol{tm_.x, tr_.x, tg_.x, tb_.x}; sc[1] = StatCol{tm_.y, tr_.y, tg_.y, tb_.y};
        sc[2] = StatCol{tm_.z, tr_.z, tg_.z, tb_.z}; sc[3] = StatCol{tm_.w, tr_.w, tg_.w, tb_.w};
      }
      const long long pimg = (long long)img * p.H * p.W;
#pragma unroll
      for (int pass = 0; pass < RB * 32 / RP; ++pass) {
#pragma unroll
        for (int tm = 0; tm < RB; ++tm)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int R0 = 32 * tm + 8 * (e >> 2);                       // first row of this element's 8-row group
            if (R0 / RP != pass) continue;                               // compile-time
            tb[(R0 % RP + (e & 3) + 4 * lk) * 32 + lr] = acc[RB * t2 + tm][e];
          }
#pragma unroll
        for (int it = 0; it < RP / 8; ++it) {
          const int row = 8 * it + rr;
          float4 v = *reinterpret_cast<const float4*>(tb + row * 32 + c4);
          const int R = pass * RP + row, tm = R >> 5, r = R & 31;
          const int dy = PW == 16 ? tm + (PH / 2) * (r >> 4) : 4 * tm + (r >> 3);
          const int dx = PW == 16 ? (r & 15) : (r & 7);
          if (y0 + dy >= p.H || x0 + dx >= p.W || !nok) continue;
          const long long pix = pimg + (long long)(y0 + dy) * p.W + (x0 + dx);
          if (partial_) {
            *reinterpret_cast<float4*>(p.ws + ((long long)ks * p.M + pix) * p.N + nb) = v;
          } else {
            act_t* c = p.C + pix * p.ldc + nb;
            v = make_float4(v.x * p.alpha + bv4.x, v.y * p.alpha + bv4.y, v.z * p.alpha + bv4.z, v.w * p.alpha + bv4.w);
            if (p.res) {
              const float4 r4 = osm::ld4(p.res + pix * p.ldr + nb);
              v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
            }
            if (p.accumulate) {
              const float4 r4 = osm::ld4(c);
              v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
            }
            osm::st4(c, v);
            if (stats_) {
              float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
              if (p.stat_mode == 2) xv = osm::ld4(p.stat_x + pix * p.ld_sx + nb);
              stat_add(p.stat_mode, p.stat_silu, sc[0], (float)(act_t)v.x, xv.x, q1[0], q2[0]);
              stat_add(p.stat_mode, p.stat_silu, sc[1], (float)(act_t)v.y, xv.y, q1[1], q2[1]);
              stat_add(p.stat_mode, p.stat_silu, sc[2], (float)(act_t)v.z, xv.z, q1[2], q2[2]);
              stat_add(p.stat_mode, p.stat_silu, sc[3], (float)(act_t)v.w, xv.w, q1[3], q2[3]);
            }
          }
        }
      }
      if (stats_) {      // lanes with the same (lane & 7) hold the same 4 columns: fold them, lanes 0-7 write
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int m_ = 8; m_ < 64; m_ <<= 1) {
            q1[k] += __shfl_xor(q1[k], m_, 64);
            q2[k] += __shfl_xor(q2[k], m_, 64);
          }
        if (lane < 8 && nok) {
          float* o = p.colsum + ((long long)img * p.stat_chunks + (ty * tpx + tx)) * 2 * p.N + nb;
          *reinterpret_cast<float4*>(o) = make_float4(q1[0], q1[1], q1[2], q1[3]);
          *reinterpret_cast<float4*>(o + p.N) = make_float4(q2[0], q2[1], q2[2], q2[3]);
        }
      }
      }   // column tile t2
      return;
    }
  }
  static_assert(NT == 1 || true, "");
  if constexpr (NT == 2) {   // (two column tiles per wave are launched on aligned shapes only: launch())
    return;
  }
  // scalar form (unaligned / odd shapes, and the narrow variant)
  const bool partial = p.splitk > 1;
  const long long ldc = partial ? (long long)p.N : p.ldc;
  const int n = n0 + (NARROW ? 0 : 32 * wave) + lr;
  if (n >= p.N) return;
  const float bv = (!partial && p.bias) ? p.bias[n] : 0.f;
  // element e of row block tm is patch pixel  PW = 16: (tm + PH/2 (e >> 3), (e & 3) + 8 ((e >> 2) & 1) + 4 lk)
  //                                            PW = 8 : (4 tm + (e >> 2), (e & 3) + 4 lk)
  const int xl = x0 + 4 * lk;
  const long long pix0 = (long long)img * p.H * p.W + (long long)y0 * p.W + xl;
  float* __restrict__ wp = p.ws + ((long long)ks * p.M) * p.N + pix0 * ldc + n;   // fp32 partials (split-K only)
  act_t* __restrict__ cp = p.C + pix0 * ldc + n;
  const act_t* __restrict__ rp = (p.res && !partial) ? p.res + pix0 * p.ldr + n : nullptr;
  const long long crow = (long long)p.W * ldc, rrow = (long long)p.W * p.ldr;
  // optional column sums of the final values (IGemmParams::colsum); the split-K case is served by the combine kernel
  const bool stats = p.colsum != nullptr && !partial;
  float s1 = 0.f, s2 = 0.f;
  StatCol scol = {};
  const act_t* __restrict__ sxp = nullptr;
  const long long srow = (long long)p.W * p.ld_sx;
  if (stats && p.stat_mode == 2) {
    const float* tb = p.stat_table + (long long)img * 4 * p.N + n;
    scol = StatCol{tb[0], tb[p.N], tb[2 * p.N], tb[3 * p.N]};
    sxp = p.stat_x + pix0 * p.ld_sx + n;
  }
#pragma unroll
  for (int tm0 = 0; tm0 < (NARROW ? 1 : RB); ++tm0) {
    const int tm = NARROW ? wave : tm0;        // narrow: this wave's row block, its two accumulators summed
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int dy = PW == 16 ? tm + (PH / 2) * (e >> 3) : 4 * tm + (e >> 2);
      const int dx = PW == 16 ? (e & 3) + 8 * ((e >> 2) & 1) : (e & 3);
      if (y0 + dy >= p.H || xl + dx >= p.W) continue;
      float v = NARROW ? acc[0][e] + acc[1][e] : acc[tm0][e];
      if (partial) {
        wp[dy * crow + dx * ldc] = v;
      } else {
        act_t* c = cp + dy * crow + dx * ldc;
        v = v * p.alpha + bv;
        if (rp) v += osm::ld1(rp + dy * rrow + dx * p.ldr);
        if (p.accumulate) v += osm::ld1(c);
        osm::st1(c, v);
        if (stats)
          stat_add(p.stat_mode, p.stat_silu, scol, (float)(act_t)v,
                   p.stat_mode == 2 ? osm::ld1(sxp + dy * srow + dx * p.ld_sx) : 0.f, s1, s2);
      }
    }
  }
  if (stats) {   // fold the two half-waves (same column, other rows); one (sum, sum) pair per column and patch
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (lk == 0) {
      float* o = p.colsum + ((long long)img * p.stat_chunks + (ty * tpx + tx)) * 2 * p.N + n;
      o[0] = s1;
      o[p.N] = s2;
    }
  }
}
