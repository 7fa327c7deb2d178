// Winograd F(2x2, 3x3) convolution kernel, two waves per SIMD (included inside igemm.hip's anonymous namespace, after
// conv3_wino.inc.h, which has the algorithm, the staging layout and the weight packing).
//
// Workgroup = 16 x 16 output pixels (8 x 8 tiles = two 32-tile blocks) x 64 output channels (two 32-column tiles), 16 input
// channels per slab, 8 waves x 8 accumulators.  The 18 x 18 x 16-channel halo lands once per slab as raw fp32 in LDS
// (double-buffered, one barrier per slab); every wave forms the V fragments it multiplies IN the registers that feed the MFMA
// (lane = (tile, channel half) is the lane that holds that row of the A operand), U fragments come straight from L2 into
// registers two units ahead.  Round 2's kernel ran one wave per SIMD with 16 accumulators and a hand-interleaved instruction
// stream; with two waves per SIMD the compiler-scheduled stream below reaches the same time (profiles/r03_wino_power.txt: both
// draw ~1300 W and sit at the clock the power manager grants), in a third of the code.
//
//   wave (r, h): xi row r = wave & 3 of B^T d (two input rows of a tile, as before) and xi COLUMN PAIR h = wave >> 2:
//          h = 0: V0 = t0 - t2, V1 = t1 + t2;   h = 1: V2 = t2 - t1, V3 = t1 - t3     (t = the wave's row of B^T d)
//          for both 32-tile blocks and both 32-column tiles: acc[2 xi][2 tile blocks][2 column tiles].
//   unit   = (local xi jj, tile block tb): 12 MFMAs (2 column tiles x 6 plane pairs) on one A fragment triple va; while they
//          run, the NEXT unit's va is built (8 LDS reads, 24 FMAs, the 44-op split).  U fragments of a xi (24 registers) are
//          loaded two units ahead into the registers the previous slab's same xi just released.
//   out    the xi columns are now spread over the wave pair: h = 0 contributes (M0 + M1, M1), h = 1 (M2, -M2 - M3) to
//          (s0, s1); all eight waves write their two partial column transforms to LDS (aliasing the staging buffers), and
//          wave (oy, ox, column tile) sums 3 xi rows x 2 halves and stores output pixel (oy, ox) of every tile, 16 bytes a lane.
#ifndef W8_ABL
#define W8_ABL 0        // measurement builds (tools/wino8_ablate.sh; results are then wrong): 1 no activation loads, 2 no U loads,
                        // 4 no epilogue, 8 no LDS reads in the V build, 16 no split, 32 MFMAs replaced by one VALU op each,
                        // 64 activation loads re-read slab 0 (cache hits), 128 U loads re-read one 6 KB piece (cache hits), 1024 no barrier in the slab loop
#endif
#ifndef W8_DBL
#define W8_DBL 0        // measurement builds: 1 the build's LDS reads issued twice, 2 the split done twice, 4 every MFMA issued twice
#endif
#ifndef W8_TSHARE
#define W8_TSHARE 1      // 1: the t column the two xi of a wave share is formed once per slab and tile block
#endif
#ifndef W8_ORDER
#define W8_ORDER 0       // measurement builds: 1 = a unit's MFMAs are written BEFORE the build of the next unit's A fragments; 2 = as 1, and the
                         // unit is pinned to "8 LDS reads, then 6 x (1 MFMA, W8_SGB_VALU VALU)" with sched_group_barrier
#endif
#ifndef W8_SGB_VALU
#define W8_SGB_VALU 8
#endif
#ifndef W8_PRIO
#define W8_PRIO 0        // measurement builds: s_setprio level around the MFMA group of a unit (0: none)
#endif
#ifndef W8_PRIO_BUILD
#define W8_PRIO_BUILD 0  // measurement builds: s_setprio level around the V build of a unit (0: none)
#endif
#ifndef W8_PIPE
#define W8_PIPE 1        // f16x3: the LDS reads of a V build are issued ONE UNIT before its arithmetic (inline-asm ds_read_b128, waited
                         // for by hand), so their latency runs under the previous unit's MFMAs instead of stalling the wave
#endif
#ifndef W8_YPRIO
#define W8_YPRIO 3        // units (bit mask) in which the younger half of the workgroup (waves 4-7) runs at s_setprio 1: both waves of a SIMD
                         // issue by age otherwise, the older half reached the slab barrier ~25 % earlier and waited there (time stamps:
                         // barrier wait 17 -> 12 % of the loop); 0 = off
#endif
#ifndef W8_PIPE_MIN
#define W8_PIPE_MIN 4
#endif
constexpr int W8_NJ = 3;                     // raw staging pieces per thread and slab (1296 pieces, 512 threads)

// HP ("f16x3", NP = 2): the two planes of V and of U are IEEE halfs of the operands scaled into the fp16 range -- V by
// 2^ex from the per-image max |x| the caller supplies (IGemmParams::xmax; |V| <= 4 max |x| -> 2^14), U by the power of two
// stored behind its image (pack_weight_wino_kernel) -- three fp16 MFMAs per product, the result rescaled (exactly) in the
// epilogue.
#ifdef W8_STAMP   // measurement build: s_memtime (100 MHz) at the phase boundaries of the first 64 workgroups' waves -> osm_w8_stamps
__device__ unsigned long long osm_w8_stamps[64 * 8 * 16];
__device__ unsigned long long osm_w8_slab_stamps[64 * 8 * 160];   // [workgroup][wave][slab][unit 0..3 start, barrier passed]
__device__ unsigned long long osm_w8_fine_stamps[8 * 32 * 16];     // workgroup 0: [wave][slab][step inside unit 0]
#define OSM_W8_FINE_STAMP(c_, k_) if (blockIdx.x == 0 && lane == 0 && (c_) - kc0 < 32) osm_w8_fine_stamps[(wave * 32 + ((c_) - kc0)) * 16 + (k_)] = __builtin_readcyclecounter();
#define OSM_W8_SLAB_STAMP(c_, k_) if (blockIdx.x < 64 && lane == 0 && (c_) - kc0 < 32) osm_w8_slab_stamps[((blockIdx.x * 8 + wave) * 32 + ((c_) - kc0)) * 5 + (k_)] = __builtin_readcyclecounter();
#define OSM_W8_STAMP(k_) if (HP && blockIdx.x < 64 && lane == 0) osm_w8_stamps[(blockIdx.x * 8 + wave) * 16 + (k_)] = __builtin_readcyclecounter();
#else
#define OSM_W8_STAMP(k_)
#define OSM_W8_SLAB_STAMP(c_, k_)
#endif
template <int NP, bool GNF, bool HP = false, bool PIPE = false>
__global__ __launch_bounds__(512, 2) void conv3_wino8_kernel(const act_t* __restrict__ Aglob,
                                                              const unsigned short* __restrict__ Uglob, IGemmParams p) {
  // 128 KB: the two raw slabs (46 KB) during the slab loop, the 8-wave exchange buffer of the epilogue after it
  __shared__ __attribute__((aligned(16))) float smem[8 * 2 * 2 * 16 * 64];
  float4* raw = reinterpret_cast<float4*>(smem);
  float* red = smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave & 3, wh = wave >> 2;
  OSM_W8_STAMP(0)
  const int lr = lane & 31, lk = lane >> 5;

  // ---- XCD-aware tile mapping.  Workgroups that share an XCD (a contiguous id range) run in step; ids enumerate
  // (column-tile group, M-tile, column tile within the group): the p.nb1 column tiles of a group work on the SAME input
  // patch at the same time, so the patch comes from HBM once and from the XCD's L2 p.nb1 - 1 times.
  const int nt = p.mtiles * p.ntiles;
  const int bid = blockIdx.x;
  const int qq = nt >> 3, rr8 = nt & 7, xcd = bid & 7, idx8 = bid >> 3;
  const int id = (xcd < rr8 ? xcd * (qq + 1) : rr8 * (qq + 1) + (xcd - rr8) * qq) + idx8;
  const int grp_sz = p.nb1 * p.mtiles;
  const int tile_n = (id / grp_sz) * p.nb1 + id % p.nb1, tile_m = (id % grp_sz) / p.nb1;
  const int tpx = (p.W + 15) >> 4, tpy = (p.H + 15) >> 4;
  const int tx = tile_m % tpx, ty = (tile_m / tpx) % tpy, img = tile_m / (tpx * tpy);
  const int x0 = tx * 16, y0 = ty * 16;

  const int ks = blockIdx.y;
  const int nslab = p.ksteps;
  const int per = (nslab + p.splitk - 1) / p.splitk;
  const int kc0 = ks * per;
  const int kc1 = min(nslab, kc0 + per);

  // ---- raw staging coordinates: piece s = tid + 512 j -> channel quad tid & 3, halo pixel (tid >> 2) + 128 j
  const int q4 = tid & 3;
  const long long rowB = (long long)p.lda * ACT_B;
  const char* __restrict__ sbaseA = reinterpret_cast<const char*>(Aglob) + (long long)img * p.H * p.W * rowB;
  unsigned voff[W8_NJ], woff[W8_NJ], vmask = 0;
#pragma unroll
  for (int j = 0; j < W8_NJ; ++j) {
    const int pix = (tid >> 2) + 128 * j;
    const int r = pix / 18, col = pix - r * 18;
    const bool in = pix < 324;
    woff[j] = (unsigned)(q4 * WN_QP + (in ? r * WN_ROWP + (col & 1) * 10 + (col >> 1) : 17 * WN_ROWP + 19));
    const int y = y0 - 1 + r, x = x0 - 1 + col;
    const bool ok = in && y >= 0 && y < p.H && x >= 0 && x < p.W;
    const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.W - 1);
    voff[j] = (W8_ABL & 256) ? (unsigned)((tid + 512 * j) * 16) : (unsigned)((long long)(yc * p.W + xc) * rowB);
    vmask |= (ok ? 1u : 0u) << j;
  }

  // ---- U fragments: image [plane][xi][slab][n/32][lane][8]; this wave reads xi = 4 wr + 2 wh + jj
  const int jn0 = 2 * tile_n;
  const unsigned u_lane = (unsigned)((jn0 * 64 + lane) * 16);
  const unsigned u_nt = jn0 + 1 < p.nt32 ? 1024u : 0u;
  const unsigned u_slab = (unsigned)p.nt32 * 1024u;
  const unsigned u_xi = u_slab * (unsigned)nslab;
  const unsigned u_plane = u_xi * 16u;
  const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(Uglob)), 0, 0x7fffffff, 0x00020000);

  // accumulators [local xi jj][tile block][column tile], cleared by eight MFMAs on zero operands with the literal 0 as C
  // (one matrix instruction writes the 16 registers 16 v_mov would: 128 VALU issues less per wave)
  f32x16 acc[2][2][2];
  {
    const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
    f32x16 z16;
#pragma unroll
    for (int e = 0; e < 16; ++e) z16[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          uint4 z = z4;
          asm("" : "+v"(z.x));       // opaque to the optimiser: eight separate instructions, not one result copied 7 x 16 times
          acc[j][a][b] = mma16h(z, z4, z16);
        }
  }

  // ---- this wave's row of B^T d:  t = x + sg y  with (x, y) = input rows (0, 2) | (1, 2) | (2, 1) | (1, 3) of the 4 x 4
  // tile and sg = -1 | +1 | -1 | -1.  Lane = tile (lr >> 3, lr & 7) of a 32-tile block, channel half lk.
  const int tyl = lr >> 3, txl = lr & 7;
  const int rx = wr == 0 ? 0 : (wr == 2 ? 2 : 1), ry = wr == 2 ? 1 : (wr == 3 ? 3 : 2);
  const float sg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(wr == 1 ? 0x3f800000 : (int)0xbf800000));
  const osm::floatx4_t* t_x = reinterpret_cast<const osm::floatx4_t*>(raw) + (2 * lk) * WN_QP + (2 * tyl + rx) * WN_ROWP + txl;
  const osm::floatx4_t* t_y = reinterpret_cast<const osm::floatx4_t*>(raw) + (2 * lk) * WN_QP + (2 * tyl + ry) * WN_ROWP + txl;

  // (requesting the first U fragments up here as well was measured: hipcc then spills ~30 registers around the fold and inside
  // the K loop -- +7 % on the class; the 12 activation registers alone fit)
  float4 ra[W8_NJ];
  uint4 uq[2][2][NP];       // [local xi][column tile][plane]: loaded two units ahead into the registers the previous slab's
                            // same xi released (a second slab-deep set was measured: +8..12 % time, it spills)
  constexpr bool use_pipe = HP && W8_PIPE && !GNF && PIPE;    // the host picks PIPE: K % 16 == 0 and >= W8_PIPE_MIN slabs per workgroup
  // raw staging of the pipelined loop: buffer loads relative to the image (padding pixels carry an out-of-range offset and
  // come back as zeros: no mask; the slab's channel offset is the scalar offset: no address arithmetic), scale from an SGPR,
  // buffer parity a compile-time constant (the loop is unrolled by two: every LDS address is base + immediate).  K % 16 == 0.
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(sbaseA), 0, (int)min((long long)p.H * p.W * rowB, 0x7fffffffLL), 0x00020000);
  unsigned aoff[W8_NJ];
#pragma unroll
  for (int j = 0; j < W8_NJ; ++j) aoff[j] = ((vmask >> j) & 1u) ? voff[j] + (unsigned)(q4 * 4 * ACT_B) : 0x80000000u;

  if constexpr (use_pipe) {   // the first slab's activation loads go out ahead of the max |x| fold: one global round trip instead of two
#pragma unroll
    for (int j = 0; j < W8_NJ; ++j)
      ra[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, (int)aoff[j], kc0 * (16 * ACT_B), 0));
  }
  // HP: x 2^ex brings the largest |x| of the image (OSM_MAXABS_PARTS partial maxima, as bit patterns) to [2^11, 2^12)
  float xscale = 1.f, oscale = 1.f;
  if (HP) {
    static_assert(OSM_MAXABS_PARTS == 1024, "two partial maxima per thread");
    const unsigned* xm = reinterpret_cast<const unsigned*>(p.xmax) + (long long)img * OSM_MAXABS_PARTS;
    unsigned mb = max(xm[tid], xm[tid + 512]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, o, 64));
    unsigned* red_u = reinterpret_cast<unsigned*>(smem);
    if (lane == 0) red_u[wave] = mb;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) mb = max(mb, red_u[q]);
    __syncthreads();              // the staging stores that follow reuse smem
    const float mx = __uint_as_float(mb);
    int ex = 0;
    if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &ex); ex = min(12 - ex, 100); }   // (denormal maxima: 2^ex stays finite)
    xscale = mx == mx ? ldexpf(1.f, ex) : mx;       // a NaN in the input poisons the output
    oscale = ldexpf(1.f, -ex) / p.wscale[0];
    xscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, xscale)));   // wave-uniform: SGPRs
    oscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, oscale)));
  }
  float mscale[W8_NJ];
#pragma unroll
  for (int j = 0; j < W8_NJ; ++j) mscale[j] = ((vmask >> j) & 1u) ? xscale : 0.f;
  float4 gm, gs, gb;        // fused GroupNorm: mean | rstd * gamma | beta of this thread's channel quad
  gm = gs = gb = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* __restrict__ gtab = GNF ? p.gn_table + (long long)img * 4 * p.K : nullptr;
  // the shared t column costs 16 registers: the 3-plane arithmetic has none to spare (it would spill 16)
  constexpr int TSH = (W8_TSHARE && (HP || NP < 3)) ? 1 : 0;
  uint4 va[2][NP];          // A fragments, [unit parity][plane]
  float4 ts[2][2];          // the t column both xi of this wave use, [tile block][channel quad] (W8_TSHARE)

#define OSM_W8_LOAD_RAW(cc_, j_)                                                           \
  {                                                                                        \
    const bool cok_ = (cc_) * 16 + 4 * q4 < p.K;                                           \
    const unsigned d_ = (unsigned)(((cc_) * 16 + 4 * q4) * ACT_B);                         \
    ra[j_] = (W8_ABL & 1) ? make_float4(1.f, 2.f, 3.f, (float)d_)                          \
                          : osm::ld4(reinterpret_cast<const act_t*>(sbaseA + (voff[j_] + ((cok_ && !(W8_ABL & 64)) ? d_ : 0u)))); \
  }
#define OSM_W8_LOAD_TAB(cc_)                                                               \
  if (GNF) {                                                                               \
    const bool cok_ = (cc_) * 16 + 4 * q4 < p.K;                                           \
    const float* gt_ = gtab + (cok_ ? (cc_) * 16 + 4 * q4 : 0);                            \
    gm = *reinterpret_cast<const float4*>(gt_);                                            \
    const float4 gr_ = *reinterpret_cast<const float4*>(gt_ + p.K);                        \
    const float4 gg_ = *reinterpret_cast<const float4*>(gt_ + 2 * p.K);                    \
    gs = make_float4(gr_.x * gg_.x, gr_.y * gg_.y, gr_.z * gg_.z, gr_.w * gg_.w);          \
    gb = *reinterpret_cast<const float4*>(gt_ + 3 * p.K);                                  \
  }
#define OSM_W8_STORE_RAW(cc_, j_)                                                          \
  {                                                                                        \
    const unsigned okm_ = ((cc_) * 16 + 4 * q4 < p.K) ? vmask : 0u;                        \
    float4 v = ra[j_];                                                                     \
    if (W8_ABL & 512) { asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); v = make_float4(1.f, 2.f, 3.f, (float)(cc_)); } \
    if (GNF) {                                                                             \
      v.x = (v.x - gm.x) * gs.x + gb.x;                                                    \
      v.y = (v.y - gm.y) * gs.y + gb.y;                                                    \
      v.z = (v.z - gm.z) * gs.z + gb.z;                                                    \
      v.w = (v.w - gm.w) * gs.w + gb.w;                                                    \
      if (p.gn_silu) {                                                                     \
        v.x = osm::silu_f(v.x); v.y = osm::silu_f(v.y); v.z = osm::silu_f(v.z); v.w = osm::silu_f(v.w); \
      }                                                                                    \
    }                                                                                      \
    if (HP) {   /* the scale doubles as the padding mask (0 outside the image); channels past K meet zero weights */ \
      v.x *= mscale[j_]; v.y *= mscale[j_]; v.z *= mscale[j_]; v.w *= mscale[j_];          \
      raw[((cc_) & 1) * (4 * WN_QP) + woff[j_]] = v;                                       \
    } else                                                                                 \
    raw[((cc_) & 1) * (4 * WN_QP) + woff[j_]] = sel4((okm_ >> (j_)) & 1u, v);             \
  }
#define OSM_W8_LOAD_U(cc_, jj_)                                                            \
  {                                                                                        \
    const unsigned so_ = (W8_ABL & 128) ? (unsigned)((cc_) & 1) * 64u                      \
                                        : (unsigned)(4 * wr + 2 * wh + (jj_)) * u_xi + (unsigned)(cc_) * u_slab; \
    _Pragma("unroll") for (int b = 0; b < 2; ++b)                                          \
      _Pragma("unroll") for (int q2 = 0; q2 < NP; ++q2)                                    \
        uq[jj_][b][q2] = (W8_ABL & 2) ? make_uint4(so_, u_lane, q2, b)                     \
            : __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(             \
                  ursrc, (int)u_lane, (int)(so_ + (unsigned)q2 * u_plane + (unsigned)b * u_nt), 0)); \
  }
// one U fragment (column tile b_, plane q2_) of local xi jj_ for slab cc_
#define OSM_W8_LOAD_U1(cc_, jj_, b_, q2_)                                                  \
  uq[jj_][b_][q2_] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(     \
      ursrc, (int)u_lane, (int)((unsigned)(4 * wr + 2 * wh + (jj_)) * u_xi + (unsigned)(cc_) * u_slab + (unsigned)(q2_) * u_plane + (unsigned)(b_) * u_nt), 0));
// A fragments of unit (local xi jj_, tile block tb_) from the raw slab at slot offset bo_: V = t[ca] + sb t[cb],
// t[c] = x[c] + sg y[c] (column c of the tile = parity c & 1, slot txl + (c >> 1)), split into NP planes.
// The two xi of a wave share one t column (h = 0: t2, h = 1: t1): the FIRST build of a tile block (local xi 0, where the
// shared column is the `cb` operand) keeps it in ts[tb], the second takes it from there -- keep_ 1: save t[cb] to ts[tb_];
// use_ 1: t[cb] comes from ts[tb_], 2: t[ca] comes from ts[tb_] (no LDS reads / FMAs for that column).
#define OSM_W8_BUILD(par_, ca_, cb_, sb_, tb_, bo_, keep_, use_)                           \
  {                                                                                        \
    if (W8_PRIO_BUILD) __builtin_amdgcn_s_setprio(W8_PRIO_BUILD);                          \
    uint2 vh_[2][NP];                                                                      \
    _Pragma("unroll") for (int hq = 0; hq < 2; ++hq) {                                     \
      const int oa_ = (bo_) + hq * WN_QP + 8 * (tb_) * WN_ROWP + ((ca_) & 1) * 10 + ((ca_) >> 1); \
      const int ob_ = (bo_) + hq * WN_QP + 8 * (tb_) * WN_ROWP + ((cb_) & 1) * 10 + ((cb_) >> 1); \
      float4 ta_, tb2_;                                                                    \
      if (W8_ABL & 8) { ta_ = tb2_ = make_float4((float)oa_, sg, (float)ob_, 1.f); }       \
      else {                                                                               \
        if ((use_) == 2) ta_ = ts[tb_][hq];                                                \
        else {                                                                             \
          osm::floatx4_t xa_ = t_x[oa_], ya_ = t_y[oa_];                                   \
          if (W8_DBL & 1) {   /* measurement: every LDS read of the build issued twice */ \
            asm volatile("" :: "v"(xa_), "v"(ya_));                                        \
            xa_ = *(volatile const osm::floatx4_t*)(t_x + oa_); ya_ = *(volatile const osm::floatx4_t*)(t_y + oa_); \
          }                                                                                \
          ta_ = make_float4(fmaf(sg, ya_[0], xa_[0]), fmaf(sg, ya_[1], xa_[1]), fmaf(sg, ya_[2], xa_[2]), fmaf(sg, ya_[3], xa_[3])); \
        }                                                                                  \
        if ((use_) == 1) tb2_ = ts[tb_][hq];                                               \
        else {                                                                             \
          osm::floatx4_t xb_ = t_x[ob_], yb_ = t_y[ob_];                                   \
          if (W8_DBL & 1) {                                                                \
            asm volatile("" :: "v"(xb_), "v"(yb_));                                        \
            xb_ = *(volatile const osm::floatx4_t*)(t_x + ob_); yb_ = *(volatile const osm::floatx4_t*)(t_y + ob_); \
          }                                                                                \
          tb2_ = make_float4(fmaf(sg, yb_[0], xb_[0]), fmaf(sg, yb_[1], xb_[1]), fmaf(sg, yb_[2], xb_[2]), fmaf(sg, yb_[3], xb_[3])); \
        }                                                                                  \
        if (keep_) ts[tb_][hq] = tb2_;                                                     \
      }                                                                                    \
      const float4 v_ = make_float4(ta_.x + (sb_) * tb2_.x, ta_.y + (sb_) * tb2_.y, ta_.z + (sb_) * tb2_.z, ta_.w + (sb_) * tb2_.w); \
      if (W8_ABL & 16) { _Pragma("unroll") for (int q2 = 0; q2 < NP; ++q2)                  \
          vh_[hq][q2] = make_uint2(__float_as_uint(v_.x) + q2, __float_as_uint(v_.y) ^ __float_as_uint(v_.z) ^ __float_as_uint(v_.w)); } \
      else if constexpr (HP) {                                                             \
        split_f16x2(v_, vh_[hq]);                                                          \
        if (W8_DBL & 2) {   /* measurement: the split done twice */                        \
          asm volatile("" :: "v"(vh_[hq][0].x), "v"(vh_[hq][0].y), "v"(vh_[hq][1].x), "v"(vh_[hq][1].y)); \
          float4 v2_ = v_; asm volatile("" : "+v"(v2_.x), "+v"(v2_.y), "+v"(v2_.z), "+v"(v2_.w)); \
          split_f16x2(v2_, vh_[hq]);                                                       \
        }                                                                                  \
      }                                                                                    \
      else split_planes<NP>(v_, vh_[hq]);                                                  \
    }                                                                                      \
    _Pragma("unroll") for (int q2 = 0; q2 < NP; ++q2)                                      \
      va[par_][q2] = make_uint4(vh_[0][q2].x, vh_[0][q2].y, vh_[1][q2].x, vh_[1][q2].y);   \
    if (W8_PRIO_BUILD) __builtin_amdgcn_s_setprio(0);                                      \
  }
// the 12 MFMAs of unit (jj_, tb_): smallest plane-pair terms first
#define OSM_W8_MMA(par_, jj_, tb_)                                                         \
  if (W8_PRIO) __builtin_amdgcn_s_setprio(W8_PRIO);                                        \
  OSM_W8_MMA_BODY(par_, jj_, tb_)                                                          \
  if (W8_PRIO) __builtin_amdgcn_s_setprio(0);
#define OSM_W8_MMA_BODY(par_, jj_, tb_)                                                    \
  _Pragma("unroll") for (int pa = NP - 1; pa >= 0; --pa)                                   \
    _Pragma("unroll") for (int pb = NP - 1 - pa; pb >= 0; --pb)                            \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                                        \
        if (W8_ABL & 32) acc[jj_][tb_][b][(pa * 3 + pb) & 15] += __uint_as_float((va[par_][pa].x ^ uq[jj_][b][pb].y) + (va[par_][pa].z ^ uq[jj_][b][pb].w)); \
        else if constexpr (HP) {                                                           \
          acc[jj_][tb_][b] = mma16h(va[par_][pa], uq[jj_][b][pb], acc[jj_][tb_][b]);       \
          if (W8_DBL & 4) acc[jj_][tb_][b] = mma16h(va[par_][pa], uq[jj_][b][pb], acc[jj_][tb_][b]); \
        }                                                                                  \
        else acc[jj_][tb_][b] = mma16<NP>(va[par_][pa], uq[jj_][b][pb], acc[jj_][tb_][b]);
// one unit: the build of the NEXT unit's A fragments and the MFMAs of the current one (independent of each other)
#if W8_ORDER == 0
#define OSM_W8_UNIT(build_, mma_) build_ mma_
#elif W8_ORDER == 1
#define OSM_W8_UNIT(build_, mma_) mma_ build_
#else
#define OSM_W8_UNIT(build_, mma_)                                                          \
  mma_ build_                                                                              \
  __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   /* the build's LDS reads first */   \
  _Pragma("unroll") for (int sg_ = 0; sg_ < 6; ++sg_) {                                    \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          /* one MFMA */             \
    __builtin_amdgcn_sched_group_barrier(0x002, W8_SGB_VALU, 0); /* VALU of the build */    \
  }
#endif
// nothing crosses a unit boundary (keeps the prefetch distance of the loads and the live ranges of va / uq as written)
#ifndef W8_FENCE_MODE
#define W8_FENCE_MODE 0
#endif
#define OSM_W8_FENCE()                                                                     \
  if (W8_FENCE_MODE != 2) asm volatile("" ::: "memory");                                   \
  if (W8_FENCE_MODE == 0) __builtin_amdgcn_sched_barrier(0);

  // one K loop per column pair: the (column, sign) constants of V differ, everything else is shared
  auto slab_loop = [&](auto hc) __attribute__((always_inline)) {
    constexpr int H = decltype(hc)::value;
    // local xi 0: V = t[CA0] - t[CB0];  local xi 1: V = t[CA1] + SB1 t[CB1]
    constexpr int CA0 = H ? 2 : 0, CB0 = H ? 1 : 2;
    constexpr int CA1 = 1, CB1 = H ? 3 : 2;
    constexpr float SB1 = H ? -1.f : 1.f;
    // the shared t column (CB0) is operand b of the second xi for h = 0 (t1 + t2) and operand a for h = 1 (t1 - t3)
    constexpr int USE1 = TSH ? (H ? 2 : 1) : 0;
    static_assert(CB0 == (H ? CA1 : CB1), "the shared column");
    const int k1 = min(kc0 + 1, kc1 - 1);
    OSM_W8_LOAD_RAW(kc0, 0) OSM_W8_LOAD_RAW(kc0, 1) OSM_W8_LOAD_RAW(kc0, 2)
    OSM_W8_LOAD_TAB(kc0)
    OSM_W8_LOAD_U(kc0, 0)
    OSM_W8_STORE_RAW(kc0, 0) OSM_W8_STORE_RAW(kc0, 1) OSM_W8_STORE_RAW(kc0, 2)
    OSM_W8_LOAD_RAW(k1, 0) OSM_W8_LOAD_RAW(k1, 1) OSM_W8_LOAD_RAW(k1, 2)
    OSM_W8_LOAD_TAB(k1)
    __syncthreads();
    OSM_W8_BUILD(0, CA0, CB0, -1.f, 0, (kc0 & 1) * (4 * WN_QP), TSH, 0)
    OSM_W8_FENCE()
    for (int c = kc0; c < kc1; ++c) {
      const int c1 = min(c + 1, kc1 - 1), c2 = min(c + 2, kc1 - 1);
      const int bo = (c & 1) * (4 * WN_QP), bn = ((c + 1) & 1) * (4 * WN_QP);
      // unit 0 = (xi 0, block 0) | builds (xi 0, block 1); U of xi 1 for this slab
      OSM_W8_LOAD_U(c, 1)
      OSM_W8_STORE_RAW(c + 1, 0) OSM_W8_LOAD_RAW(c2, 0)
      OSM_W8_UNIT(OSM_W8_BUILD(1, CA0, CB0, -1.f, 1, bo, TSH, 0), OSM_W8_MMA(0, 0, 0))
      OSM_W8_FENCE()
      // unit 1 = (xi 0, block 1) | builds (xi 1, block 0)
      OSM_W8_STORE_RAW(c + 1, 1) OSM_W8_LOAD_RAW(c2, 1)
      OSM_W8_UNIT(OSM_W8_BUILD(0, CA1, CB1, SB1, 0, bo, 0, USE1), OSM_W8_MMA(1, 0, 1))
      OSM_W8_FENCE()
      // unit 2 = (xi 1, block 0) | builds (xi 1, block 1); U of xi 0 for the next slab
      OSM_W8_LOAD_U(c1, 0)
      OSM_W8_STORE_RAW(c + 1, 2) OSM_W8_LOAD_RAW(c2, 2) OSM_W8_LOAD_TAB(c2)
      OSM_W8_UNIT(OSM_W8_BUILD(1, CA1, CB1, SB1, 1, bo, 0, USE1), OSM_W8_MMA(0, 1, 0))
      OSM_W8_FENCE()
      if (!(W8_ABL & 1024)) __syncthreads();          // raw(c + 1) is complete in its buffer; nobody reads raw(c) any more
      // unit 3 = (xi 1, block 1) | builds (xi 0, block 0) of slab c + 1
      OSM_W8_UNIT(OSM_W8_BUILD(0, CA0, CB0, -1.f, 0, bn, TSH, 0), OSM_W8_MMA(1, 1, 1))
      OSM_W8_FENCE()
    }
  };

  // ---- W8_PIPE: the same units, software-pipelined at HALF-build granularity.  A build is two independent halves (channel quads
  // hq = 0 | 1 of the lane's 8 channels -> the .xy | .zw words of both A planes); the LDS reads of a half are issued half a unit
  // before its arithmetic, into the four registers the previous half just released, and three MFMAs of the running unit sit
  // between them.  The reads are inline asm -- the compiler neither sinks them down to their use nor counts them (LDS returns in
  // order, so its own lgkmcnt waits only become more conservative); OSM_W8P_WAIT is the hand-placed wait, tied to the
  // destination registers by "+v" operands (cdna_hip_programming.md, asm loads, form ii).
#define OSM_W8P_RD(dst_, addr_, slot_)                                                     \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "n"((slot_) * 16));
  // reads of half hq_ of the build of tile block tb_ from the raw slab with x / y row base addresses ax_ / ay_:
  // column ca_ -> rd[0..1] (if na_), column cb_ -> rd[2..3] (if nb_)
#define OSM_W8P_READS(hq_, ax_, ay_, buf_, tb_, ca_, cb_, na_, nb_)                              \
  if (na_) {                                                                               \
    OSM_W8P_RD(rd[0], ax_, ((buf_) * 4 + (hq_)) * WN_QP + 8 * (tb_) * WN_ROWP + ((ca_) & 1) * 10 + ((ca_) >> 1)) \
    OSM_W8P_RD(rd[1], ay_, ((buf_) * 4 + (hq_)) * WN_QP + 8 * (tb_) * WN_ROWP + ((ca_) & 1) * 10 + ((ca_) >> 1)) \
  }                                                                                        \
  if (nb_) {                                                                               \
    OSM_W8P_RD(rd[2], ax_, ((buf_) * 4 + (hq_)) * WN_QP + 8 * (tb_) * WN_ROWP + ((cb_) & 1) * 10 + ((cb_) >> 1)) \
    OSM_W8P_RD(rd[3], ay_, ((buf_) * 4 + (hq_)) * WN_QP + 8 * (tb_) * WN_ROWP + ((cb_) & 1) * 10 + ((cb_) >> 1)) \
  }
#define OSM_W8P_WAIT() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rd[0]), "+v"(rd[1]), "+v"(rd[2]), "+v"(rd[3]));
  // arithmetic of half hq_ from rd: V = t[a] + sb t[b] -> words 2 hq_, 2 hq_ + 1 of va[par_][plane];
  // keep_ 1: t[b] -> ts[tb_][hq_]; use_ 1: t[b] from ts[tb_][hq_], 2: t[a] from there
#define OSM_W8P_MATH(hq_, par_, sb_, tb_, keep_, use_)                                     \
  {                                                                                        \
    OSM_W8P_WAIT()                                                                         \
    float4 ta_, tb2_;                                                                      \
    if ((use_) == 2) ta_ = ts[tb_][hq_];                                                   \
    else ta_ = make_float4(fmaf(sg, rd[1][0], rd[0][0]), fmaf(sg, rd[1][1], rd[0][1]),     \
                           fmaf(sg, rd[1][2], rd[0][2]), fmaf(sg, rd[1][3], rd[0][3]));    \
    if ((use_) == 1) tb2_ = ts[tb_][hq_];                                                  \
    else tb2_ = make_float4(fmaf(sg, rd[3][0], rd[2][0]), fmaf(sg, rd[3][1], rd[2][1]),    \
                            fmaf(sg, rd[3][2], rd[2][2]), fmaf(sg, rd[3][3], rd[2][3]));   \
    if (keep_) ts[tb_][hq_] = tb2_;                                                        \
    const float4 v_ = make_float4(ta_.x + (sb_) * tb2_.x, ta_.y + (sb_) * tb2_.y, ta_.z + (sb_) * tb2_.z, ta_.w + (sb_) * tb2_.w); \
    uint2 vh_[2];                                                                          \
    split_f16x2(v_, vh_);                                                                  \
    if ((hq_) == 0) { va[par_][0].x = vh_[0].x; va[par_][0].y = vh_[0].y; va[par_][1].x = vh_[1].x; va[par_][1].y = vh_[1].y; } \
    else            { va[par_][0].z = vh_[0].x; va[par_][0].w = vh_[0].y; va[par_][1].z = vh_[1].x; va[par_][1].w = vh_[1].y; } \
  }
  // half of the 6 MFMAs of unit (jj_, tb_): plane pairs (1,0), then (0,1) on column tile 0 | (0,1) on tile 1, then (0,0)
#define OSM_W8P_MMA(half_, par_, jj_, tb_)                                                 \
  if ((half_) == 0) {                                                                      \
    acc[jj_][tb_][0] = mma16h(va[par_][1], uq[jj_][0][0], acc[jj_][tb_][0]);               \
    acc[jj_][tb_][1] = mma16h(va[par_][1], uq[jj_][1][0], acc[jj_][tb_][1]);               \
    acc[jj_][tb_][0] = mma16h(va[par_][0], uq[jj_][0][1], acc[jj_][tb_][0]);               \
  } else {                                                                                 \
    acc[jj_][tb_][1] = mma16h(va[par_][0], uq[jj_][1][1], acc[jj_][tb_][1]);               \
    acc[jj_][tb_][0] = mma16h(va[par_][0], uq[jj_][0][0], acc[jj_][tb_][0]);               \
    acc[jj_][tb_][1] = mma16h(va[par_][0], uq[jj_][1][0], acc[jj_][tb_][1]);               \
  }
#define OSM_W8P_LOAD_RAW(cc_, j_)                                                          \
  ra[j_] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, (int)aoff[j_], (cc_) * (16 * ACT_B), 0));
#define OSM_W8P_STORE_RAW(buf_, j_)                                                        \
  raw[(buf_) * (4 * WN_QP) + woff[j_]] = make_float4(ra[j_].x * xscale, ra[j_].y * xscale, ra[j_].z * xscale, ra[j_].w * xscale);
  auto slab_loop_p = [&](auto hc) __attribute__((always_inline)) {
    constexpr int H = decltype(hc)::value;
    constexpr int CA0 = H ? 2 : 0, CB0 = H ? 1 : 2;
    constexpr int CA1 = 1, CB1 = H ? 3 : 2;
    constexpr float SB1 = H ? -1.f : 1.f;
    constexpr int USE1 = H ? 2 : 1;            // the shared column is operand b of the second xi for h = 0, operand a for h = 1
    constexpr int NA1 = H ? 0 : 1, NB1 = H ? 1 : 0;     // second xi: only the column that is not shared is read
    osm::floatx4_t rd[4];
    const unsigned txa = (unsigned)(size_t)t_x, tya = (unsigned)(size_t)t_y;
    const int k1 = min(kc0 + 1, kc1 - 1);
    OSM_W8_LOAD_U(kc0, 0)                      // (the activations of slab kc0 were requested before the max |x| fold)
    OSM_W8P_STORE_RAW(0, 0) OSM_W8P_STORE_RAW(0, 1) OSM_W8P_STORE_RAW(0, 2)
    OSM_W8P_LOAD_RAW(k1, 0) OSM_W8P_LOAD_RAW(k1, 1) OSM_W8P_LOAD_RAW(k1, 2)
    __syncthreads();
    OSM_W8P_READS(0, txa, tya, 0, 0, CA0, CB0, 1, 1)
    OSM_W8P_MATH(0, 0, -1.f, 0, 1, 0)
    OSM_W8P_READS(1, txa, tya, 0, 0, CA0, CB0, 1, 1)
    OSM_W8P_MATH(1, 0, -1.f, 0, 1, 0)
    OSM_W8P_READS(0, txa, tya, 0, 1, CA0, CB0, 1, 1)
    OSM_W8_FENCE()
    // one slab; P = the parity of its LDS buffer (slab c - kc0 lives in buffer (c - kc0) & 1)
    auto slab = [&](auto pc, const int c) __attribute__((always_inline)) {
      constexpr int P = decltype(pc)::value, Q = 1 - P;
      const int c1 = min(c + 1, kc1 - 1), c2 = min(c + 2, kc1 - 1);
      if constexpr (H == 1 && W8_YPRIO != 0) __builtin_amdgcn_s_setprio((W8_YPRIO >> 0) & 1);
      OSM_W8_SLAB_STAMP(c, 0)
      // unit 0 = MFMAs of (xi 0, block 0) | arithmetic of (xi 0, block 1); U of xi 1 for this slab
      // (the eight U fragments of a slab are requested one per half unit, in the order the MFMAs use them, each into the register
      // its last MFMA just released: as two bursts of four, all eight waves queued at the address unit at the same moment --
      // ~350 cycles per burst before a wave got its loads out, measured with time stamps)
      OSM_W8_LOAD_U1(c, 1, 0, 0)
      OSM_W8P_STORE_RAW(Q, 0)
      OSM_W8P_MATH(0, 1, -1.f, 1, 1, 0)
      OSM_W8P_READS(1, txa, tya, P, 1, CA0, CB0, 1, 1)
      OSM_W8P_MMA(0, 0, 0, 0)
      OSM_W8_FENCE()
      OSM_W8_LOAD_U1(c, 1, 1, 0) OSM_W8P_LOAD_RAW(c2, 0)
      OSM_W8P_MATH(1, 1, -1.f, 1, 1, 0)
      OSM_W8P_READS(0, txa, tya, P, 0, CA1, CB1, NA1, NB1)
      OSM_W8P_MMA(1, 0, 0, 0)
      OSM_W8_FENCE()
#ifdef W8_XBAR
      __syncthreads();          // measurement build: a second barrier per slab (what does the lockstep cost?)
#endif
      if constexpr (H == 1 && W8_YPRIO != 0) __builtin_amdgcn_s_setprio((W8_YPRIO >> 1) & 1);
      OSM_W8_SLAB_STAMP(c, 1)
      // unit 1 = (xi 0, block 1) | arithmetic of (xi 1, block 0)
      OSM_W8_LOAD_U1(c, 1, 0, 1)
      OSM_W8P_STORE_RAW(Q, 1)
      OSM_W8P_MATH(0, 0, SB1, 0, 0, USE1)
      OSM_W8P_READS(1, txa, tya, P, 0, CA1, CB1, NA1, NB1)
      OSM_W8P_MMA(0, 1, 0, 1)
      OSM_W8_FENCE()
      OSM_W8_LOAD_U1(c, 1, 1, 1) OSM_W8P_LOAD_RAW(c2, 1)
      OSM_W8P_MATH(1, 0, SB1, 0, 0, USE1)
      OSM_W8P_READS(0, txa, tya, P, 1, CA1, CB1, NA1, NB1)
      OSM_W8P_MMA(1, 1, 0, 1)
      OSM_W8_FENCE()
      if constexpr (H == 1 && W8_YPRIO != 0) __builtin_amdgcn_s_setprio((W8_YPRIO >> 2) & 1);
      OSM_W8_SLAB_STAMP(c, 2)
      // unit 2 = (xi 1, block 0) | arithmetic of (xi 1, block 1) | barrier | first reads of slab c + 1; U of xi 0 for the next slab
      OSM_W8_LOAD_U1(c1, 0, 0, 0)
      OSM_W8P_STORE_RAW(Q, 2)
      OSM_W8P_MATH(0, 1, SB1, 1, 0, USE1)
      OSM_W8P_READS(1, txa, tya, P, 1, CA1, CB1, NA1, NB1)
      OSM_W8P_MMA(0, 0, 1, 0)
      OSM_W8_FENCE()
      OSM_W8_LOAD_U1(c1, 0, 1, 0) OSM_W8P_LOAD_RAW(c2, 2)
      OSM_W8P_MATH(1, 1, SB1, 1, 0, USE1)
      OSM_W8_FENCE()
      OSM_W8_SLAB_STAMP(c, 3)
      __syncthreads();          // raw(c + 1) is complete in its buffer; every read of raw(c) has returned (the waits above)
      OSM_W8_SLAB_STAMP(c, 4)
      OSM_W8P_READS(0, txa, tya, Q, 0, CA0, CB0, 1, 1)
      OSM_W8P_MMA(1, 0, 1, 0)
      OSM_W8_FENCE()
      if constexpr (H == 1 && W8_YPRIO != 0) __builtin_amdgcn_s_setprio((W8_YPRIO >> 3) & 1);
      // unit 3 = (xi 1, block 1) | arithmetic of (xi 0, block 0) of slab c + 1
      OSM_W8_LOAD_U1(c1, 0, 0, 1)
      OSM_W8P_MATH(0, 0, -1.f, 0, 1, 0)
      OSM_W8P_READS(1, txa, tya, Q, 0, CA0, CB0, 1, 1)
      OSM_W8P_MMA(0, 1, 1, 1)
      OSM_W8_FENCE()
      OSM_W8_LOAD_U1(c1, 0, 1, 1)
      OSM_W8P_MATH(1, 0, -1.f, 0, 1, 0)
      OSM_W8P_READS(0, txa, tya, Q, 1, CA0, CB0, 1, 1)
      OSM_W8P_MMA(1, 1, 1, 1)
      OSM_W8_FENCE()
    };
    for (int c = kc0; c < kc1; c += 2) {
      slab(std::integral_constant<int, 0>{}, c);
      if (c + 1 >= kc1) break;
      slab(std::integral_constant<int, 1>{}, c + 1);
    }
    OSM_W8P_WAIT()               // the reads issued by the last half unit land before their registers are reused
    if constexpr (H == 1 && W8_YPRIO != 0) __builtin_amdgcn_s_setprio(0);
  };
  OSM_W8_STAMP(1)
  if (kc1 > kc0) {
    // (the host launches this instance from W8_PIPE_MIN slabs per workgroup: its prologue is two dependent LDS round trips longer)
    if constexpr (use_pipe) {
      if (wh == 0) slab_loop_p(std::integral_constant<int, 0>{});
      else slab_loop_p(std::integral_constant<int, 1>{});
    } else {
      if (wh == 0) slab_loop(std::integral_constant<int, 0>{});
      else slab_loop(std::integral_constant<int, 1>{});
    }
  }
#undef OSM_W8P_RD
#undef OSM_W8P_LOAD_RAW
#undef OSM_W8P_STORE_RAW
#undef OSM_W8P_READS
#undef OSM_W8P_MMA
#undef OSM_W8P_WAIT
#undef OSM_W8P_MATH
#undef OSM_W8_LOAD_RAW
#undef OSM_W8_LOAD_TAB
#undef OSM_W8_STORE_RAW
#undef OSM_W8_LOAD_U
#undef OSM_W8_LOAD_U1
#undef OSM_W8_BUILD
#undef OSM_W8_MMA
#undef OSM_W8_MMA_BODY
#undef OSM_W8_UNIT
#undef OSM_W8_FENCE

  OSM_W8_STAMP(2)
  if ((W8_ABL & 4) && p.alpha != 12345.f) return;      // measurement build: no epilogue
  // ---- Y = A^T M A.  xi columns: s0 = M0 + M1 + M2, s1 = M1 - M2 - M3 -> this wave's partials p0 | p1:
  //   h = 0 (M0, M1): (M0 + M1, M1);   h = 1 (M2, M3): (M2, M2 + M3), the latter subtracted.
  // xi rows: Y[0][.] = s(0) + s(1) + s(2), Y[1][.] = s(1) - s(2) - s(3).  One tile block per round.
  // red: [wave = 4 h + r][ox][column tile][e][lane = 32 lk + column]; finishing wave f = (oy, ox, column tile b).
  const int oy = wave >> 2, ox = (wave >> 1) & 1, fb = wave & 1;
  const bool partial = p.splitk > 1;
  const long long pix0 = (long long)img * p.H * p.W + (long long)(y0 + oy) * p.W + (x0 + ox);
  float* __restrict__ wbase = p.ws + ((long long)ks * p.M + pix0) * p.N;
  act_t* __restrict__ obase = p.C + pix0 * p.ldc;
  const act_t* __restrict__ rbase = (!partial && p.res) ? p.res + pix0 * p.ldr : nullptr;
  const int c4 = 4 * (lane & 7), lk2 = (lane >> 3) & 1, e_lo = lane >> 4;
  const int dx = 2 * (e_lo + 4 * lk2);
  const bool xok = x0 + ox + dx < p.W;
  const float* red_rd = red + ((ox * 2 + fb) * 16 + e_lo) * 64 + lk2 * 32 + c4;   // + ((wave' * 4) * 16 + 4 i) * 64
  const bool stats = p.colsum != nullptr && !partial;
  float st1[4], st2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) st1[k] = st2[k] = 0.f;
  const act_t* __restrict__ sxbase = (stats && p.stat_mode == 2) ? p.stat_x + pix0 * p.ld_sx : nullptr;
  const int n = (jn0 + fb) * 32 + c4;
  const bool nok = n < p.N && (fb == 0 || u_nt != 0u);
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!partial && p.bias && nok) bv = *reinterpret_cast<const float4*>(p.bias + n);
  StatCol sc[4] = {};
  if (sxbase && nok) {
    const float* tb = p.stat_table + (long long)img * 4 * p.N + n;
    const float4 tm = *reinterpret_cast<const float4*>(tb), tr = *reinterpret_cast<const float4*>(tb + p.N);
    const float4 tg = *reinterpret_cast<const float4*>(tb + 2 * p.N), tbb = *reinterpret_cast<const float4*>(tb + 3 * p.N);
    sc[0] = StatCol{tm.x, tr.x, tg.x, tbb.x}; sc[1] = StatCol{tm.y, tr.y, tg.y, tbb.y};
    sc[2] = StatCol{tm.z, tr.z, tg.z, tbb.z}; sc[3] = StatCol{tm.w, tr.w, tg.w, tbb.w};
  }
  // Straight-line code: the wave-uniform choices (column pair h, the signs of the row / column sums, residual / accumulate /
  // statistics) are made ONCE per round, not per element -- as per-element scalar branches the epilogue was ~260 branches per
  // workgroup and cost a quarter of the kernel (measured by running it twice).
  // signs of the six partial planes a finishing wave sums: xi rows + + + (oy = 0) | + - - (oy = 1); the second column partial
  // of the h = 1 waves enters negated
  float sgn[3][2];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
      sgn[k][h2] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(
          ((oy == 0 || k == 0) != (h2 == 1 && ox == 1)) ? 0x3f800000 : (int)0xbf800000));
  const bool ok = nok && xok;
  // write phase of round a_: h = 0: (M0 + M1, M1);  h = 1: (M2, M2 + M3)
  auto put = [&](auto whc, const int a_) __attribute__((always_inline)) {
    constexpr int WH = decltype(whc)::value;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float m0 = a_ == 0 ? acc[0][0][b][e] : acc[0][1][b][e], m1 = a_ == 0 ? acc[1][0][b][e] : acc[1][1][b][e];
        red[(((wave * 2 + 0) * 2 + b) * 16 + e) * 64 + lane] = WH == 0 ? m0 + m1 : m0;
        red[(((wave * 2 + 1) * 2 + b) * 16 + e) * 64 + lane] = WH == 0 ? m1 : m0 + m1;
      }
  };
  // sum of tile row 4 a_ + i_ (e' = 4 i_ + e_lo) for this wave's (oy, ox, column tile), rescaled
  auto gather = [&](const int i_) __attribute__((always_inline)) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const float4 s = *reinterpret_cast<const float4*>(red_rd + (((h2 * 4 + oy + k) * 4) * 16 + 4 * i_) * 64);
        v.x = fmaf(sgn[k][h2], s.x, v.x); v.y = fmaf(sgn[k][h2], s.y, v.y);
        v.z = fmaf(sgn[k][h2], s.z, v.z); v.w = fmaf(sgn[k][h2], s.w, v.w);
      }
    if (HP) { v.x *= oscale; v.y *= oscale; v.z *= oscale; v.w *= oscale; }
    return v;
  };
  // the residual / accumulate operands of a round are requested BEFORE its LDS exchange (their global round trips, one per
  // tile row and serial when issued where they are used, then run under the exchange); zeros where there is nothing to add
  // (MODE 3, backward statistics: so is the GroupNorm input the reductions need -- as four dependent round trips inside the
  // finishing phase it made OSM_FUSE_STATS=all a net loss)
  float4 pre_r[4], pre_a[4], pre_x[4];
  const bool silu_on = p.stat_silu != 0;
  auto prefetch = [&](auto modec, const int a_) __attribute__((always_inline)) {
    constexpr int MODE = decltype(modec)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int dy = 8 * a_ + 2 * i;
      const int po = dy * p.W + dx;
      const bool live = ok && y0 + oy + dy < p.H;
      pre_r[i] = pre_a[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (MODE != 0) {      // (a split-K partial round adds nothing: no loads -- ADVICE r04)
        if (rbase && live) pre_r[i] = osm::ld4(rbase + po * (int)p.ldr + n);
        if (p.accumulate && live) pre_a[i] = osm::ld4(obase + po * (int)p.ldc + n);
      }
      if constexpr (MODE == 3) {
        pre_x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (sxbase && live) pre_x[i] = osm::ld4(sxbase + po * (int)p.ld_sx + n);
      }
    }
  };
  // finishing phase of round a_.  MODE 0: split-K partial;  1: y = alpha v + bias + residual + previous y;  2: the same with
  // forward column statistics;  3: with the backward ones
  auto finish = [&](auto modec, const int a_) __attribute__((always_inline)) {
    constexpr int MODE = decltype(modec)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = gather(i);
      const int dy = 8 * a_ + 2 * i;
      if (y0 + oy + dy >= p.H) continue;          // wave-uniform
      const int po = dy * p.W + dx;
      if constexpr (MODE == 0) {
        if (ok) *reinterpret_cast<float4*>(wbase + po * p.N + n) = v;
      } else {
        act_t* __restrict__ op = obase + po * (int)p.ldc + n;
        v = make_float4(v.x * p.alpha + bv.x, v.y * p.alpha + bv.y, v.z * p.alpha + bv.z, v.w * p.alpha + bv.w);
        v.x += pre_r[i].x; v.y += pre_r[i].y; v.z += pre_r[i].z; v.w += pre_r[i].w;       // (in this order: residual, then y)
        v.x += pre_a[i].x; v.y += pre_a[i].y; v.z += pre_a[i].z; v.w += pre_a[i].w;
        if (ok) {
          osm::st4(op, v);
          if constexpr (MODE == 2) {          // forward statistics of the tensor as stored
            stat_add(1, 0, sc[0], (float)(act_t)v.x, 0.f, st1[0], st2[0]);
            stat_add(1, 0, sc[1], (float)(act_t)v.y, 0.f, st1[1], st2[1]);
            stat_add(1, 0, sc[2], (float)(act_t)v.z, 0.f, st1[2], st2[2]);
            stat_add(1, 0, sc[3], (float)(act_t)v.w, 0.f, st1[3], st2[3]);
          } else if constexpr (MODE == 3) {   // the two backward reductions (no per-element scalar branches: learned 42)
            const float4 xv = pre_x[i];
            stat_add_bwd(silu_on, sc[0], (float)(act_t)v.x, xv.x, st1[0], st2[0]);
            stat_add_bwd(silu_on, sc[1], (float)(act_t)v.y, xv.y, st1[1], st2[1]);
            stat_add_bwd(silu_on, sc[2], (float)(act_t)v.z, xv.z, st1[2], st2[2]);
            stat_add_bwd(silu_on, sc[3], (float)(act_t)v.w, xv.w, st1[3], st2[3]);
          }
        }
      }
    }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;
#ifndef W8_EPI_REP
#define W8_EPI_REP 1        // measurement builds: the epilogue executed this many times
#endif
  for (int rep_ = 0; rep_ < W8_EPI_REP; ++rep_)
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    auto round = [&](auto modec) __attribute__((always_inline)) {
      prefetch(modec, a);
      OSM_W8_STAMP(3 + 5 * a)
      __syncthreads();     // a = 0: the slab loop's reads of raw are over;  a = 1: the previous round's reads of red
      OSM_W8_STAMP(4 + 5 * a)
      if (wh == 0) put(I0{}, a); else put(I1{}, a);
      OSM_W8_STAMP(5 + 5 * a)
      __syncthreads();
      OSM_W8_STAMP(6 + 5 * a)
      finish(modec, a);
      OSM_W8_STAMP(7 + 5 * a)
    };
    if (partial) round(I0{});
    else if (stats && sxbase) round(I3{});
    else if (stats) round(I2{});
    else round(I1{});
  }
  if (stats) {       // workgroup-uniform
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int m = 8; m < 64; m <<= 1) {          // lanes with the same (lane & 7) hold the same columns
        st1[k] += __shfl_xor(st1[k], m, 64);
        st2[k] += __shfl_xor(st2[k], m, 64);
      }
    }
    __syncthreads();            // the last round's reads of red are over
    if (lane < 8) {             // red: [wave][which sum][32 columns]
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        red[(wave * 2 + 0) * 32 + 4 * lane + k] = st1[k];
        red[(wave * 2 + 1) * 32 + 4 * lane + k] = st2[k];
      }
    }
    __syncthreads();
    if (tid < 128) {            // (column tile b, which sum, column): the four (oy, ox) waves of column tile b are 2 q + b
      const int b = tid >> 6, w2 = (tid >> 5) & 1, col = tid & 31;
      const int nn = (jn0 + b) * 32 + col;
      if (nn < p.N && (b == 0 || u_nt != 0u)) {
        const float v = (red[((0 + b) * 2 + w2) * 32 + col] + red[((2 + b) * 2 + w2) * 32 + col]) +
                        (red[((4 + b) * 2 + w2) * 32 + col] + red[((6 + b) * 2 + w2) * 32 + col]);
        p.colsum[(((long long)img * p.stat_chunks + (ty * tpx + tx)) * 2 + w2) * p.N + nn] = v;
      }
    }
  }
}
