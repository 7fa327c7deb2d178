// Winograd F(2x2, 3x3) form of the 3x3 convolution (included inside igemm.hip's anonymous namespace, after
// conv3_halo.inc.h): split-bf16 arithmetic on fp32 activations (NP = 3 / 2 planes), or -- fp16 family -- ONE half plane
// per operand on half activations (NP = 1; the transforms run in fp32 either way).
//
// Why: the direct halo-tile kernel keeps the matrix pipe 77-80 % busy and runs the chip into its 1400 W power cap
// (profiles/r02_power_under_conv.txt): with six bf16 MFMAs per fp32 product the only lever left is to issue fewer MFMAs
// per output.  F(2x2, 3x3) computes a 2 x 2 output tile from a 4 x 4 input tile with 16 instead of 36 multiplications
// per (input channel, output channel):
//     V = B^T d B (input: adds only),  U = G g G^T (weights: precomputed),  M_xi = sum_c V_xi[c] U_xi[c][n],  Y = A^T M A
// i.e. 16 independent GEMMs [tiles x Cin] . [Cin x Cout]: 2.25x fewer MFMAs for the same result.  (cuDNN picks the same
// algorithm for these layers of the reference.)
//
// What limits it here is not the matrix pipe but everything else per MFMA, which grows 2.25x in relative terms: every V
// value must be split into three bf16 planes on the VALU (7.5 VALU ops per value), and 16 accumulator tiles are needed
// per (32 tiles x 32 columns), so a workgroup (4 waves x 16 accumulators = all 256 AGPRs) covers 64 tiles x 64 columns
// and runs one wave per SIMD: nothing overlaps unless the instruction stream of that wave makes it overlap.
//
// Mapping: workgroup = 16 x 16 output pixels (8 x 8 tiles) x 64 output channels, 16 input channels per slab.
//   stage  the 18 x 18 x 16-channel input halo lands ONCE per slab as raw fp32 in LDS (double-buffered; fused
//          GroupNorm/SiLU applied on the way), laid out [channel quad][row][column parity][column / 2]: the 16 lanes of
//          a ds_read_b128 group (neighbouring tiles, pixel stride 2) then read 16 consecutive 16-byte slots.
//   wave w owns xi ROW w (xi = 4 w + j, j = 0..3) for both 32-tile blocks and both 32-column tiles.  Row w of B^T d needs
//          only two input rows of a tile (d0 - d2 | d1 + d2 | d2 - d1 | d1 - d3), and lane (tile, channel half) of the
//          wave is exactly the lane that holds row `tile`, k = 8 (lane >> 5) + e of the MFMA A operand: the V fragments
//          are produced in the registers that consume them -- no V round trip through LDS, one barrier per slab.
//   per j  T(j): the column(s) of t this V_xi needs are read (V0 = t0 - t2, V1 = t1 + t2, V2 = t2 - t1, V3 = t1 - t3),
//          V is formed and split into NP planes;  M(j): 4 accumulators x 6 MFMAs with the U fragments (buffer loads
//          two xi ahead).  A V fragment lives for one j only.
//   out    Y = A^T M A: the xi columns are inside a wave (4 accumulators -> 2 values), the xi rows are the four waves:
//          they exchange through LDS (transposing to 4 consecutive columns per lane) and wave (oy, ox) finishes output
//          pixel (oy, ox) of every tile with 16-byte stores (bias, residual, accumulate, or a split-K partial).

#ifndef WN_NT_STORE
#define WN_NT_STORE 0
#endif
#ifndef WN_VPM
#define WN_VPM 5      // VALU instructions asked for after every MFMA of a region
#endif
#ifndef WN_ABL
#define WN_ABL 0        // measurement builds (tools/wino_ablate.sh): 1 no activation loads, 2 no U loads, 4 no epilogue,
                        // 8 no slab loop, 16 epilogue without its global stores (results are then wrong, of course)
#endif
constexpr int WN_ROWP = 20;                  // 16-byte slots per staged row: [column parity 2][10 (9 used)]
constexpr int WN_QP = 18 * WN_ROWP + 1;      // slots per channel-quad plane (+1: the 4 quads of a pixel hit 4 bank groups)
constexpr int WN_NJ = 6;                     // raw staging pieces per thread and slab (18 x 18 pixels x 4 quads = 1296)

template <int NP, bool GNF>
__global__ __launch_bounds__(256, 1) void conv3_wino_kernel(const act_t* __restrict__ Aglob,
                                                             const unsigned short* __restrict__ Uglob, IGemmParams p) {
  __shared__ __attribute__((aligned(16))) float4 raw[2 * 4 * WN_QP];       // two slabs: 46 KB
  __shared__ __attribute__((aligned(16))) float red[4 * 4 * 16 * 64];      // epilogue exchange: 64 KB

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lk = lane >> 5;

  // ---- XCD-aware tile mapping.  Workgroups that share an XCD (a contiguous id range) run in step; ids enumerate
  // (column-tile group, M-tile, column tile within the group): the p.nb1 column tiles of a group work on the SAME input
  // patch at the same time, so the patch comes from HBM once and from the XCD's L2 p.nb1 - 1 times, while the U stream
  // of a group is walked by all its workgroups together.  M-tiles enumerate (image, patch row, patch col).
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
  const int nslab = p.ksteps;                       // 16-channel slabs (the weight image is zero padded to 32 channels)
  const int per = (nslab + p.splitk - 1) / p.splitk;
  const int kc0 = ks * per;
  int kc1_ = min(nslab, kc0 + per);

  // ---- raw staging coordinates: piece s = tid + 256 j -> channel quad tid & 3, halo pixel (tid >> 2) + 64 j
  const int q4 = tid & 3;
  const long long rowB = (long long)p.lda * ACT_B;
  const char* __restrict__ sbaseA = reinterpret_cast<const char*>(Aglob) + (long long)img * p.H * p.W * rowB;
  unsigned voff[WN_NJ], woff[WN_NJ], vmask = 0;
#pragma unroll
  for (int j = 0; j < WN_NJ; ++j) {
    const int pix = (tid >> 2) + 64 * j;
    const int r = pix / 18, col = pix - r * 18;
    const bool in = pix < 324;
    // unused pieces land in slot (row 17, parity 1, entry 9) of their quad plane, which no tile reads
    woff[j] = (unsigned)(q4 * WN_QP + (in ? r * WN_ROWP + (col & 1) * 10 + (col >> 1) : 17 * WN_ROWP + 19));
    const int y = y0 - 1 + r, x = x0 - 1 + col;
    const bool ok = in && y >= 0 && y < p.H && x >= 0 && x < p.W;
    const int yc = min(max(y, 0), p.H - 1), xc = min(max(x, 0), p.W - 1);
    voff[j] = (unsigned)((long long)(yc * p.W + xc) * rowB);
    vmask |= (ok ? 1u : 0u) << j;
  }

  // ---- U fragments: image [plane][xi][slab][n/32][lane][8]; this wave reads xi = 4 wave + j, column tiles 2 tile_n (+1)
  const int jn0 = 2 * tile_n;
  const unsigned u_lane = (unsigned)((jn0 * 64 + lane) * 16);
  const unsigned u_nt = jn0 + 1 < p.nt32 ? 1024u : 0u;      // odd tile count: the last workgroup reads tile jn0 twice
  const unsigned u_slab = (unsigned)p.nt32 * 1024u;
  const unsigned u_xi = u_slab * (unsigned)nslab;
  const unsigned u_plane = u_xi * 16u;
  const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(Uglob)), 0, 0x7fffffff, 0x00020000);

  f32x16 acc[4][2][2];      // [xi column j][tile block][column tile]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][a][b][e] = 0.f;

  // ---- this wave's row of B^T d:  t = x + sg y  with (x, y) = input rows (0, 2) | (1, 2) | (2, 1) | (1, 3) of the 4 x 4
  // tile and sg = -1 | +1 | -1 | -1.  Lane = tile (lr >> 3, lr & 7) of a 32-tile block, channel half lk.
  const int tyl = lr >> 3, txl = lr & 7;
  const int rx = wave == 0 ? 0 : (wave == 2 ? 2 : 1), ry = wave == 2 ? 1 : (wave == 3 ? 3 : 2);
  const float sg = wave == 1 ? 1.f : -1.f;
  const osm::floatx4_t* t_x = reinterpret_cast<const osm::floatx4_t*>(raw) + (2 * lk) * WN_QP + (2 * tyl + rx) * WN_ROWP + txl;
  const osm::floatx4_t* t_y = reinterpret_cast<const osm::floatx4_t*>(raw) + (2 * lk) * WN_QP + (2 * tyl + ry) * WN_ROWP + txl;

  float4 ra[WN_NJ];
  float4 gm, gr, gg, gb;
  gm = gr = gg = gb = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* __restrict__ gtab = GNF ? p.gn_table + (long long)img * 4 * p.K : nullptr;
  uint4 uq[4][2][NP];       // [xi column j][column tile][plane]: filled two xi ahead of their use

// activation pieces j0_ .. j0_ + n_ - 1 of slab cc_ -> registers; the GroupNorm table rows of the slab with piece 0 ... 
#define OSM_W_LOAD_RAW(cc_, j0_, n_)                                                       \
  {                                                                                        \
    const bool cok_ = (cc_) * 16 + 4 * q4 < p.K;                                           \
    const unsigned d_ = (unsigned)(((cc_) * 16 + 4 * q4) * ACT_B);                         \
    _Pragma("unroll") for (int j = (j0_); j < (j0_) + (n_); ++j)                           \
      ra[j] = (WN_ABL & 1) ? make_float4(1.f, 2.f, 3.f, (float)d_)                         \
                           : osm::ld4(reinterpret_cast<const act_t*>(sbaseA + (voff[j] + (cok_ ? d_ : 0u)))); \
  }
// ... are loaded separately: they must outlive the stores of the previous slab
#define OSM_W_LOAD_TAB(cc_)                                                                \
  if (GNF) {                                                                               \
    const bool cok_ = (cc_) * 16 + 4 * q4 < p.K;                                           \
    const float* gt_ = gtab + (cok_ ? (cc_) * 16 + 4 * q4 : 0);                            \
    gm = *reinterpret_cast<const float4*>(gt_);                                            \
    gr = *reinterpret_cast<const float4*>(gt_ + p.K);                                      \
    gg = *reinterpret_cast<const float4*>(gt_ + 2 * p.K);                                  \
    gb = *reinterpret_cast<const float4*>(gt_ + 3 * p.K);                                  \
  }
// pieces j0_ .. of slab cc_ (in registers) -> LDS buffer cc_ & 1, GroupNorm(+SiLU) applied, zero outside image / channels
#define OSM_W_STORE_RAW(cc_, j0_, n_)                                                      \
  {                                                                                        \
    const unsigned okm_ = ((cc_) * 16 + 4 * q4 < p.K) ? vmask : 0u;                        \
    _Pragma("unroll") for (int j = (j0_); j < (j0_) + (n_); ++j) {                         \
      float4 v = ra[j];                                                                    \
      if (GNF) {                                                                           \
        v.x = ((v.x - gm.x) * gr.x) * gg.x + gb.x;                                         \
        v.y = ((v.y - gm.y) * gr.y) * gg.y + gb.y;                                         \
        v.z = ((v.z - gm.z) * gr.z) * gg.z + gb.z;                                         \
        v.w = ((v.w - gm.w) * gr.w) * gg.w + gb.w;                                         \
        if (p.gn_silu) {                                                                   \
          v.x = osm::silu_f(v.x); v.y = osm::silu_f(v.y); v.z = osm::silu_f(v.z); v.w = osm::silu_f(v.w); \
        }                                                                                  \
      }                                                                                    \
      raw[((cc_) & 1) * (4 * WN_QP) + woff[j]] = sel4((okm_ >> j) & 1u, v);                \
    }                                                                                      \
  }
#define OSM_W_LOAD_U(cc_, j_)                                                              \
  {                                                                                        \
    const unsigned so_ = (unsigned)(4 * wave + (j_)) * u_xi + (unsigned)(cc_) * u_slab;    \
    _Pragma("unroll") for (int b = 0; b < 2; ++b)                                          \
      _Pragma("unroll") for (int q2 = 0; q2 < NP; ++q2)                                    \
        uq[j_][b][q2] = (WN_ABL & 2) ? make_uint4(so_, u_lane, q2, b)                      \
            : __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(             \
                  ursrc, (int)u_lane, (int)(so_ + (unsigned)q2 * u_plane + (unsigned)b * u_nt), 0)); \
  }
// t column j_ of both tile blocks: 8 LDS reads (2 rows x 2 channel quads x 2 blocks) into qx / qy ...
#define OSM_W_TRD(j_, bo_)                                                                 \
  _Pragma("unroll") for (int a = 0; a < 2; ++a)                                            \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                        \
      const int o_ = (bo_) + h * WN_QP + 8 * a * WN_ROWP + ((j_) & 1) * 10 + ((j_) >> 1);  \
      qx[a][h] = t_x[o_];                                                                  \
      qy[a][h] = t_y[o_];                                                                  \
    }
// ... and t = x + sg y
#define OSM_W_TFMA(j_)                                                                     \
  _Pragma("unroll") for (int a = 0; a < 2; ++a)                                            \
    _Pragma("unroll") for (int h = 0; h < 2; ++h)                                          \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) tc[j_][a][4 * h + e] = fmaf(sg, qy[a][h][e], qx[a][h][e]);
// V of tile block a_: f = sa_ t[ja_] + sb_ t[jb_] ...
#define OSM_W_VADD(a_, ja_, sa_, jb_, sb_)                                                 \
  _Pragma("unroll") for (int e = 0; e < 8; ++e) vf[a_][e] = (sa_) * tc[ja_][a_][e] + (sb_) * tc[jb_][a_][e];
// ... split: half h_ (4 channels) of tile block a_ -> NP x 2 packed dwords
#define OSM_W_VSPL(a_, h_)                                                                 \
  split_planes<NP>(make_float4(vf[a_][4 * (h_)], vf[a_][4 * (h_) + 1], vf[a_][4 * (h_) + 2], vf[a_][4 * (h_) + 3]), vh[a_][h_]);
// ... the two halves are one 16-byte A fragment per plane
#define OSM_W_VFIN(par_, a_)                                                               \
  _Pragma("unroll") for (int q2 = 0; q2 < NP; ++q2)                                        \
    va[par_][a_][q2] = make_uint4(vh[a_][0][q2].x, vh[a_][0][q2].y, vh[a_][1][q2].x, vh[a_][1][q2].y);
// the u_-th plane pair of the product group (smallest terms first): NP = 3: six pairs, 2: three, 1: one
#define OSM_W_MMA(j_, par_, u_)                                                            \
  if ((u_) < (NP == 3 ? 6 : (NP == 2 ? 3 : 1))) {                                          \
    constexpr int pa_ = NP == 3 ? (u_ == 0 ? 2 : (u_ <= 2 ? 1 : 0)) : (NP == 2 ? (u_ == 0 ? 1 : 0) : 0);              \
    constexpr int pb_ = NP == 3 ? (u_ == 0 ? 0 : (u_ == 1 ? 1 : (u_ == 2 ? 0 : 5 - u_))) : (NP == 2 ? (u_ == 1 ? 1 : 0) : 0); \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                          \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                                        \
        acc[j_][a][b] = mma16<NP>(va[par_][a][pa_], uq[j_][b][pb_], acc[j_][a][b]);        \
  }
// end of a unit = { 4 MFMAs (one plane pair) + a piece of the preparation of the next xi column }: ask for the VALU
// instructions BETWEEN the MFMAs (one wave per SIMD: <= 5-6 single-issue instructions are free per 32-cycle MFMA),
// and let nothing cross the unit boundary
#define OSM_W_UNIT()                                                                       \
  {                                                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                     \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   \
      __builtin_amdgcn_sched_group_barrier(0x002, WN_VPM, 0);                              \
    }                                                                                      \
    asm volatile("" ::: "memory");                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                     \
  }
// region = the six units of xi column j_ (operands va[par_]); P0_ .. P5_ = what each unit does besides its MFMAs
#define OSM_W_REGION(j_, par_, P0_, P1_, P2_, P3_, P4_, P5_)                               \
  OSM_W_MMA(j_, par_, 0) P0_ OSM_W_UNIT()                                                  \
  OSM_W_MMA(j_, par_, 1) P1_ OSM_W_UNIT()                                                  \
  OSM_W_MMA(j_, par_, 2) P2_ OSM_W_UNIT()                                                  \
  OSM_W_MMA(j_, par_, 3) P3_ OSM_W_UNIT()                                                  \
  OSM_W_MMA(j_, par_, 4) P4_ OSM_W_UNIT()                                                  \
  OSM_W_MMA(j_, par_, 5) P5_ OSM_W_UNIT()

  // Software pipeline: region r_j = the 24 MFMAs of xi column j, and between them everything that prepares xi column
  // j + 1 (LDS reads of the t column it needs, the V transform and split, ~100 VALU) plus a share of the slab's memory
  // instructions (U fragments two xi ahead; two of the six activation pieces of slab c + 1 into LDS and of slab c + 2 into
  // their registers).  r_3 prepares column 0 of the NEXT slab, after the slab's only barrier.
  float tc[4][2][8];                  // [input column][tile block][channel]: t of the current slab
  osm::floatx4_t qx[2][2], qy[2][2];  // [tile block][channel quad]: the two input rows of one t column
  float vf[2][8];                     // V of the xi column being prepared
  uint2 vh[2][2][NP];                 // its planes, [tile block][half]
  uint4 va[2][2][NP];                 // A fragments, [xi column parity][tile block][plane]
  if ((WN_ABL & 8) && p.alpha != 12345.f) kc1_ = kc0;      // measurement build: no slab loop
  if (kc1_ > kc0) {
    const int kc1 = kc1_;
    const int k1 = min(kc0 + 1, kc1 - 1);
    OSM_W_LOAD_RAW(kc0, 0, WN_NJ)
    OSM_W_LOAD_TAB(kc0)
    OSM_W_LOAD_U(kc0, 0)
    OSM_W_LOAD_U(kc0, 1)
    OSM_W_STORE_RAW(kc0, 0, WN_NJ)
    OSM_W_LOAD_RAW(k1, 0, WN_NJ)
    OSM_W_LOAD_TAB(k1)
    __syncthreads();
    {
      const int bo = (kc0 & 1) * (4 * WN_QP);
      OSM_W_TRD(0, bo) OSM_W_TFMA(0)
      OSM_W_TRD(2, bo) OSM_W_TFMA(2)
      OSM_W_VADD(0, 0, 1.f, 2, -1.f) OSM_W_VADD(1, 0, 1.f, 2, -1.f)
      OSM_W_VSPL(0, 0) OSM_W_VSPL(0, 1) OSM_W_VSPL(1, 0) OSM_W_VSPL(1, 1)
      OSM_W_VFIN(0, 0) OSM_W_VFIN(0, 1)
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    for (int c = kc0; c < kc1; ++c) {
      const int c1 = min(c + 1, kc1 - 1), c2 = min(c + 2, kc1 - 1);
      const int bo = (c & 1) * (4 * WN_QP), bn = ((c + 1) & 1) * (4 * WN_QP);
      // r0: xi column 0 | prepares V1 = t1 + t2
      OSM_W_REGION(0, 0,
                   OSM_W_LOAD_U(c, 2) OSM_W_TRD(1, bo) OSM_W_STORE_RAW(c + 1, 0, 2) OSM_W_LOAD_RAW(c2, 0, 2),
                   OSM_W_TFMA(1) OSM_W_VADD(0, 1, 1.f, 2, 1.f),
                   OSM_W_VSPL(0, 0),
                   OSM_W_VSPL(0, 1) OSM_W_VADD(1, 1, 1.f, 2, 1.f),
                   OSM_W_VSPL(1, 0),
                   OSM_W_VSPL(1, 1) OSM_W_VFIN(1, 0) OSM_W_VFIN(1, 1))
      // r1: xi column 1 | prepares V2 = t2 - t1, reads t column 3
      OSM_W_REGION(1, 1,
                   OSM_W_LOAD_U(c, 3) OSM_W_TRD(3, bo) OSM_W_STORE_RAW(c + 1, 2, 2) OSM_W_LOAD_RAW(c2, 2, 2),
                   OSM_W_VADD(0, 2, 1.f, 1, -1.f) OSM_W_TFMA(3),
                   OSM_W_VSPL(0, 0),
                   OSM_W_VSPL(0, 1) OSM_W_VADD(1, 2, 1.f, 1, -1.f),
                   OSM_W_VSPL(1, 0),
                   OSM_W_VSPL(1, 1) OSM_W_VFIN(0, 0) OSM_W_VFIN(0, 1))
      // r2: xi column 2 | prepares V3 = t1 - t3
      OSM_W_REGION(2, 0,
                   OSM_W_LOAD_U(c1, 0) OSM_W_STORE_RAW(c + 1, 4, 2) OSM_W_LOAD_RAW(c2, 4, 2) OSM_W_LOAD_TAB(c2),
                   OSM_W_VADD(0, 1, 1.f, 3, -1.f),
                   OSM_W_VSPL(0, 0),
                   OSM_W_VSPL(0, 1) OSM_W_VADD(1, 1, 1.f, 3, -1.f),
                   OSM_W_VSPL(1, 0),
                   OSM_W_VSPL(1, 1) OSM_W_VFIN(1, 0) OSM_W_VFIN(1, 1))
      __syncthreads();          // raw(c + 1) is complete in its buffer; nobody reads raw(c) any more
      // r3: xi column 3 | prepares V0 = t0 - t2 of slab c + 1
      OSM_W_REGION(3, 1,
                   OSM_W_LOAD_U(c1, 1) OSM_W_TRD(0, bn),
                   OSM_W_TFMA(0) OSM_W_TRD(2, bn),
                   OSM_W_TFMA(2) OSM_W_VADD(0, 0, 1.f, 2, -1.f) OSM_W_VSPL(0, 0),
                   OSM_W_VSPL(0, 1) OSM_W_VADD(1, 0, 1.f, 2, -1.f),
                   OSM_W_VSPL(1, 0),
                   OSM_W_VSPL(1, 1) OSM_W_VFIN(0, 0) OSM_W_VFIN(0, 1))
    }
  }
#undef OSM_W_LOAD_RAW
#undef OSM_W_LOAD_TAB
#undef OSM_W_STORE_RAW
#undef OSM_W_LOAD_U
#undef OSM_W_TRD
#undef OSM_W_TFMA
#undef OSM_W_VADD
#undef OSM_W_VSPL
#undef OSM_W_VFIN
#undef OSM_W_MMA
#undef OSM_W_UNIT
#undef OSM_W_REGION

  if ((WN_ABL & 4) && p.alpha != 12345.f) return;      // measurement build: no epilogue
  // ---- Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1].  xi columns (in this wave): s0 = M0 + M1 + M2, s1 = M1 - M2 - M3;
  // xi rows (= waves): Y[0][.] = s(0) + s(1) + s(2), Y[1][.] = s(1) - s(2) - s(3).  One tile block per round.
  // Stores are issue-bound (one 4-byte store per lane and instruction costs as much as a 16-byte one), so the exchange
  // also transposes: the finishing wave reads (tile, 4 consecutive columns) per lane -- 16 contiguous bytes of red --
  // and writes dwordx4: 8 lanes cover the 128 B of one pixel's 32-column tile.
  // red: [wave = xi row][ox][column tile][e][lane = 32 lk + column]
  const int oy = wave >> 1, ox = wave & 1;
  const bool partial = p.splitk > 1;
  const long long pix0 = (long long)img * p.H * p.W + (long long)(y0 + oy) * p.W + (x0 + ox);
  float* __restrict__ wbase = p.ws + ((long long)ks * p.M + pix0) * p.N;            // split-K partials: fp32, ld = N
  act_t* __restrict__ obase = p.C + pix0 * p.ldc;
  const act_t* __restrict__ rbase = (!partial && p.res) ? p.res + pix0 * p.ldr : nullptr;
  // item it = 64 i + lane of a (tile block, column tile): columns 4 (it & 7) .. + 3 of accumulator element
  // e' = it >> 4 in lane half lk' = (it >> 3) & 1, i.e. tile (4 a + (e' >> 2), (e' & 3) + 4 lk') of the patch
  const int c4 = 4 * (lane & 7), lk2 = (lane >> 3) & 1, e_lo = lane >> 4;      // e' = 4 i + e_lo
  const int dx = 2 * (e_lo + 4 * lk2);                                          // pixel column offset in the patch
  const bool xok = x0 + ox + dx < p.W;
  const float* red_rd = red + ((ox * 2) * 16 + e_lo) * 64 + lk2 * 32 + c4;      // + ((row * 4 + b) * 16 + 4 i) * 64
  // optional column sums of the final values (IGemmParams::colsum; the split-K case is served by the combine kernel):
  // every lane accumulates its 4 columns over the pixels it stores, the lanes of a wave and the four waves are folded
  // at the end -> one (sum, sum) pair per column and workgroup patch
  const bool stats = p.colsum != nullptr && !partial;
  float st1[2][4], st2[2][4];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int k = 0; k < 4; ++k) st1[b][k] = st2[b][k] = 0.f;
  const act_t* __restrict__ sxbase = (stats && p.stat_mode == 2) ? p.stat_x + pix0 * p.ld_sx : nullptr;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    if (a) __syncthreads();     // the previous round's reads of red are over
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        red[(((wave * 2 + 0) * 2 + b) * 16 + e) * 64 + lane] = acc[0][a][b][e] + acc[1][a][b][e] + acc[2][a][b][e];
        red[(((wave * 2 + 1) * 2 + b) * 16 + e) * 64 + lane] = acc[1][a][b][e] - acc[2][a][b][e] - acc[3][a][b][e];
      }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int n = (jn0 + b) * 32 + c4;
      const bool nok = n < p.N && (b == 0 || u_nt != 0u);
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
#pragma unroll
      for (int i = 0; i < 4; ++i) {                 // e' = 4 i + e_lo: tile row 4 a + i of the patch
        const float4 s0 = *reinterpret_cast<const float4*>(red_rd + (((oy + 0) * 4 + b) * 16 + 4 * i) * 64);
        const float4 s1 = *reinterpret_cast<const float4*>(red_rd + (((oy + 1) * 4 + b) * 16 + 4 * i) * 64);
        const float4 s2 = *reinterpret_cast<const float4*>(red_rd + (((oy + 2) * 4 + b) * 16 + 4 * i) * 64);
        float4 v;
        if (oy == 0) {
          v = make_float4(s0.x + s1.x + s2.x, s0.y + s1.y + s2.y, s0.z + s1.z + s2.z, s0.w + s1.w + s2.w);
        } else {
          v = make_float4(s0.x - s1.x - s2.x, s0.y - s1.y - s2.y, s0.z - s1.z - s2.z, s0.w - s1.w - s2.w);
        }
        const int dy = 8 * a + 2 * i;
        if (y0 + oy + dy >= p.H) continue;          // wave-uniform
        const bool ok = nok && xok;
        const int po = dy * p.W + dx;               // pixel offset inside the image
        if (partial) {
          if (ok && !((WN_ABL & 16) && p.alpha != 12345.f)) *reinterpret_cast<float4*>(wbase + po * p.N + n) = v;
        } else {
          act_t* __restrict__ op = obase + po * (int)p.ldc + n;
          v = make_float4(v.x * p.alpha + bv.x, v.y * p.alpha + bv.y, v.z * p.alpha + bv.z, v.w * p.alpha + bv.w);
          if (rbase && ok) {
            const float4 r = osm::ld4(rbase + po * (int)p.ldr + n);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
          }
          if (p.accumulate && ok) {
            const float4 r = osm::ld4(op);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
          }
#if WN_NT_STORE && !OSM_ACT_IS_F16
          if (ok) __builtin_nontemporal_store(osm::floatx4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<osm::floatx4_t*>(op));
#else
          if (ok && !((WN_ABL & 16) && p.alpha != 12345.f)) osm::st4(op, v);     // 16: measurement build without the stores
#endif
          if (stats && ok) {
            float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sxbase) xv = osm::ld4(sxbase + po * (int)p.ld_sx + n);
            // the sums are over the values as stored (rounded to the storage type)
            stat_add(p.stat_mode, p.stat_silu, sc[0], (float)(act_t)v.x, xv.x, st1[b][0], st2[b][0]);
            stat_add(p.stat_mode, p.stat_silu, sc[1], (float)(act_t)v.y, xv.y, st1[b][1], st2[b][1]);
            stat_add(p.stat_mode, p.stat_silu, sc[2], (float)(act_t)v.z, xv.z, st1[b][2], st2[b][2]);
            stat_add(p.stat_mode, p.stat_silu, sc[3], (float)(act_t)v.w, xv.w, st1[b][3], st2[b][3]);
          }
        }
      }
    }
  }
  if (stats) {       // wave-uniform
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int m = 8; m < 64; m <<= 1) {        // lanes with the same (lane & 7) hold the same columns
          st1[b][k] += __shfl_xor(st1[b][k], m, 64);
          st2[b][k] += __shfl_xor(st2[b][k], m, 64);
        }
      }
    __syncthreads();            // the last round's reads of red are over
    if (lane < 8) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          red[((wave * 2 + b) * 2 + 0) * 32 + 4 * lane + k] = st1[b][k];
          red[((wave * 2 + b) * 2 + 1) * 32 + 4 * lane + k] = st2[b][k];
        }
    }
    __syncthreads();
    if (tid < 128) {            // (column tile b, which sum, column)
      const int b = tid >> 6, w2 = (tid >> 5) & 1, col = tid & 31;
      const int n = (jn0 + b) * 32 + col;
      if (n < p.N && (b == 0 || u_nt != 0u)) {
        const float v = (red[((0 * 2 + b) * 2 + w2) * 32 + col] + red[((1 * 2 + b) * 2 + w2) * 32 + col]) +
                        (red[((2 * 2 + b) * 2 + w2) * 32 + col] + red[((3 * 2 + b) * 2 + w2) * 32 + col]);
        p.colsum[(((long long)img * p.stat_chunks + (ty * tpx + tx)) * 2 + w2) * p.N + n] = v;
      }
    }
  }
}

#ifndef OSM_ACT_F16
// OIHW fp32 -> Winograd-domain weights U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], as np bf16 planes (np = 1: ONE
// plane of IEEE half, the fp16 family's image) in
// MFMA-fragment order [plane][xi = 4 a + b][16-channel slab s][n/32 j][lane l][e]: n = 32 j + (l & 31),
// k = 16 s + 8 (l >> 5) + e.  forward: n = Cout, k = Cin; data-gradient: n = Cin, k = Cout, taps flipped.
// U is formed in double, rounded once to fp32, and that fp32 value is split exactly into the planes.
// U = G g G^T of one (n, k, xi) in double
__device__ __forceinline__ double wino_u(const float* __restrict__ w, int Cout, int Cin, int N, int K, int nn, int kk, int xi,
                                         int dgrad) {
  double u = 0.0;
  if (nn < N && kk < K) {
    const int co = dgrad ? kk : nn, ci = dgrad ? nn : kk;
    const float* g = w + ((long long)co * Cin + ci) * 9;
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    const int a = xi >> 2, b = xi & 3;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        const float gv = dgrad ? g[(2 - r) * 3 + (2 - c)] : g[r * 3 + c];
        u += G[a][r] * (double)gv * G[b][c];
      }
  }
  return u;
}
// f16x3 image, pass 1: max |U| over the whole tensor, as the bit pattern of a non-negative float (atomicMax on uint)
__global__ void wino_umax_kernel(const float* __restrict__ w, unsigned* __restrict__ umax, int Cout, int Cin, int dgrad) {
  const int N = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
  const long long total = 16LL * N * K;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int xi = (int)(i & 15);
    const long long r = i >> 4;
    m = fmaxf(m, fabsf((float)wino_u(w, Cout, Cin, N, K, (int)(r / K), (int)(r % K), xi, dgrad)));
  }
  m = osm::wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(umax, __float_as_uint(m));
}
__global__ void pack_weight_wino_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout, int Cin,
                                        int np, int dgrad) {
  const int N = dgrad ? Cin : Cout;
  const int K = dgrad ? Cout : Cin;
  const int nt32 = (N + 31) / 32;
  const int ksteps = 2 * ((K + 31) / 32);
  const long long per_plane = 16LL * ksteps * nt32 * 512;
  // np = 4 (f16x3): two IEEE-half planes of U * 2^ew, 2^ew = the power of two that brings max |U| (left behind the planes
  // by wino_umax_kernel) to [2^13, 2^14); the word is rewritten as that scale (a float) by the last statement
  float wsc = 1.f;
  if (np == 4) {
    const float mx = __uint_as_float(*reinterpret_cast<const unsigned*>(out + 2 * per_plane + 2));   // copy made below
    int e = 0;
    if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &e); e = 14 - e; }
    wsc = ldexpf(1.f, e);
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_plane;
       i += (long long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7);
    const int l = (int)((i >> 3) & 63);
    long long rest = i >> 9;
    const int j = (int)(rest % nt32);
    rest /= nt32;
    const int s = (int)(rest % ksteps);
    const int xi = (int)(rest / ksteps);
    const int nn = 32 * j + (l & 31);
    const int kk = 16 * s + 8 * (l >> 5) + e;
    const double u = wino_u(w, Cout, Cin, N, K, nn, kk, xi, dgrad);
    float rr = (float)u;
    if (np == 1) {
      out[i] = __builtin_bit_cast(unsigned short, (_Float16)rr);
      continue;
    }
    if (np == 4) {
      rr *= wsc;
      const _Float16 h0 = (_Float16)rr;
      out[i] = __builtin_bit_cast(unsigned short, h0);
      out[per_plane + i] = __builtin_bit_cast(unsigned short, (_Float16)(rr - (float)h0));
      continue;
    }
    for (int q2 = 0; q2 < np; ++q2) {
      const __bf16 bb = (__bf16)rr;
      out[q2 * per_plane + i] = __builtin_bit_cast(unsigned short, bb);
      rr -= (float)bb;
    }
  }
}
// f16x3 image: the scale word.  [0] = max |U| bits from pass 1 -> copied to [1] (read by the pack pass), then [0] = the scale
__global__ void wino_scale_word_kernel(unsigned* __restrict__ word, int stage) {
  if (stage == 0) { word[1] = word[0]; return; }
  const float mx = __uint_as_float(word[1]);
  int e = 0;
  if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &e); e = 14 - e; }
  word[0] = __float_as_uint(ldexpf(1.f, e));
}


#endif   // !OSM_ACT_F16
