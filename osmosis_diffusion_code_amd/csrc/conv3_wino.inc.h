// Winograd F(2x2, 3x3) form of the 3x3 convolution, split-bf16 arithmetic, fp32 activations (included inside
// igemm.hip's anonymous namespace, after conv3_halo.inc.h; fp32 entry-point family only).
//
// Why: the direct halo-tile kernel keeps the matrix pipe 77-80 % busy and runs the chip into its 1400 W power cap
// (profiles/r02_power_under_conv.txt): with six bf16 MFMAs per fp32 product the only lever left is to issue fewer MFMAs
// per output.  F(2x2, 3x3) computes a 2 x 2 output tile from a 4 x 4 input tile with 16 instead of 36 multiplications
// per (input channel, output channel):
//     V = B^T d B (input: adds only),  U = G g G^T (weights: precomputed),  M_xi = sum_c V_xi[c] U_xi[c][n],  Y = A^T M A
// i.e. 16 independent GEMMs [tiles x Cin] . [Cin x Cout]: 2.25x fewer MFMAs for the same result.  (cuDNN picks the same
// algorithm for these layers of the reference.)
//
// What limits it here is not the matrix pipe but everything else per MFMA, which grows 2.25x in relative terms:
//   * every V value must be split into three bf16 planes on the VALU (7.5 VALU ops per value): it has to be done ONCE per
//     workgroup, not per wave or per 32-column tile -> the V planes of a slab live in LDS, in MFMA-fragment order;
//   * 16 accumulator tiles per (32 tiles x 32 columns): a workgroup (4 waves x 16 accumulators = all 256 AGPRs) covers
//     64 tiles x 64 columns, and each wave takes 4 transform positions x 2 tile blocks x 2 column tiles, so that every
//     fragment it fetches (V from LDS, U from L2) feeds two accumulators: 31 B/clk/CU on each path.
//
// Mapping: workgroup = 16 x 16 output pixels (8 x 8 tiles) x 64 output channels, 16 input channels per slab:
//   stage  the 18 x 18 x 16-channel input halo lands ONCE per slab as raw fp32 in LDS (fused GroupNorm/SiLU applied on
//          the way), laid out [channel quad][row][column parity][column / 2]: the 16 lanes of a ds_read_b128 group
//          (neighbouring tiles, pixel stride 2) then read 16 consecutive 16-byte slots -- conflict-free;
//   T      thread (tile block mb, half hf, tile, channel half lk) reads 3 x 4 pixels of its tile, forms the 8 V_xi of
//          xi rows 2 hf, 2 hf + 1 for its 8 channels in registers, splits them and writes the NP planes to LDS as the
//          A fragments [xi][mb][plane][lane] (1 KB contiguous per wave store);
//   M      wave w owns xi = 4 w .. 4 w + 3: per xi it reads 2 x NP V fragments, fetches 2 x NP U fragments (buffer loads
//          one xi ahead) and issues 4 x 6 MFMAs;
//   out    Y = A^T M A: the xi columns are inside a wave (4 accumulators -> 2 values), the xi rows are the four waves:
//          they exchange through LDS and wave (oy, ox) finishes output pixel (oy, ox) of every tile (bias, residual,
//          accumulate, or the fp32 partial of a split-K slice).
// Two barriers per slab; the raw image of slab c + 1 is written while slab c is in its M phase.

#ifndef OSM_ACT_F16

#ifndef WN_ABL
#define WN_ABL 0        // ablation builds (tools/wino_ablate.sh): 1 no activation loads, 2 no U loads, 4 no T phase, 8 no MFMAs
#endif
constexpr int WN_ROWP = 20;                  // 16-byte slots per staged row: [column parity 2][10 (9 used)]
constexpr int WN_QP = 18 * WN_ROWP + 1;      // slots per channel-quad plane (+1: the 4 quads of a pixel hit 4 bank groups)
constexpr int WN_NJ = 6;                     // raw staging pieces per thread and slab (18 x 18 pixels x 4 quads = 1296)

#if WN_ABL & 64
__device__ unsigned long long g_wn_prof[8];     // [stage, T, M, barrier A wait, barrier B wait, epilogue, total, waves]
#define WN_T(var_) const unsigned long long var_ = __builtin_amdgcn_s_memtime()
#define WN_ACC(i_, a_, b_) prof[i_] += (b_) - (a_)
#else
#define WN_T(var_)
#define WN_ACC(i_, a_, b_)
#endif

template <int NP, bool GNF>
__global__ __launch_bounds__(256, 1) void conv3_wino_kernel(const float* __restrict__ Aglob,
                                                             const unsigned short* __restrict__ Uglob, IGemmParams p) {
  __shared__ __attribute__((aligned(16))) float4 raw[4 * WN_QP];
  __shared__ __attribute__((aligned(16))) uint4 vp[16 * 2 * NP * 64];     // NP = 3: 96 KB
  static_assert(sizeof(uint4) * 16 * 2 * NP * 64 >= 4 * 4 * 16 * 64 * 4, "the epilogue exchange buffer aliases vp");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lk = lane >> 5;
#if WN_ABL & 64
  unsigned long long prof[6] = {0, 0, 0, 0, 0, 0};
#endif
  WN_T(t_begin);
#if WN_ABL & 64
  unsigned long long t_prev = t_begin;
#endif

  // ---- XCD-aware tile mapping; M-tiles enumerate (image, patch row, patch col) and run FASTEST: the workgroups that
  // share an XCD (a contiguous id range) then work on one or two 64-column tiles at a time, in step, so the U stream of
  // that column tile (16/9 of the direct kernel's weight bytes) is fetched into the XCD's L2 once and hit 30 times.
  // (With N fastest an XCD walks all of U at once: 6-25 MB against 4 MB of L2 -- measured: 40 % of the time in vmcnt.)
  const int nt = p.mtiles * p.ntiles;
  const int bid = blockIdx.x;
  const int qq = nt >> 3, rr8 = nt & 7, xcd = bid & 7, idx8 = bid >> 3;
  const int id = (xcd < rr8 ? xcd * (qq + 1) : rr8 * (qq + 1) + (xcd - rr8) * qq) + idx8;
  const int tile_m = id % p.mtiles, tile_n = id / p.mtiles;
  const int tpx = (p.W + 15) >> 4, tpy = (p.H + 15) >> 4;
  const int tx = tile_m % tpx, ty = (tile_m / tpx) % tpy, img = tile_m / (tpx * tpy);
  const int x0 = tx * 16, y0 = ty * 16;

  const int ks = blockIdx.y;
  const int nslab = p.ksteps;                       // 16-channel slabs (the weight image is zero padded to 32 channels)
  const int per = (nslab + p.splitk - 1) / p.splitk;
  const int kc0 = ks * per;
  const int kc1 = min(nslab, kc0 + per);

  // ---- raw staging coordinates: piece s = tid + 256 j -> channel quad tid & 3, halo pixel (tid >> 2) + 64 j
  const int q4 = tid & 3;
  const long long rowB = (long long)p.lda * 4;
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

  // ---- T-phase role: tile block mb, xi-row half hf; lane = tile (lr >> 3, lr & 7) of the block, channel half lk
  const int mb = wave >> 1, hf = wave & 1;
  const int tyl = lr >> 3, txl = lr & 7;
  const float4* a_t = raw + (2 * lk) * WN_QP + (2 * (4 * mb + tyl)) * WN_ROWP + txl;   // row 0 of the 4 x 4 input tile
  const float4* a_x = a_t + (hf ? 2 : 0) * WN_ROWP;
  const float4* a_y = a_t + (hf ? 1 : 2) * WN_ROWP;
  const float4* a_z = a_t + (hf ? 3 : 1) * WN_ROWP;
  const float sg = hf ? -1.f : 1.f;
  uint4* v_wr = vp + ((8 * hf) * 2 + mb) * NP * 64 + lane;           // xi = 8 hf + 4 i + j
  const uint4* v_rd = vp + ((4 * wave) * 2) * NP * 64 + lane;        // M phase: xi = 4 wave + j

  float4 ra[WN_NJ];
  float4 gm, gr, gg, gb;
  gm = gr = gg = gb = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* __restrict__ gtab = GNF ? p.gn_table + (long long)img * 4 * p.K : nullptr;
  unsigned okm = 0;
  uint4 uq[4][2][NP];       // [xi column j][column tile][plane]: the U fragments of one slab

#define OSM_W_LOAD_RAW(cc_)                                                                \
  {                                                                                        \
    const bool cok_ = (cc_) * 16 + 4 * q4 < p.K;                                           \
    const unsigned d_ = (unsigned)((cc_) * 64 + 16 * q4);                                  \
    okm = cok_ ? vmask : 0u;                                                               \
    _Pragma("unroll") for (int j = 0; j < WN_NJ; ++j)                                      \
      ra[j] = (WN_ABL & 1) ? make_float4(1.f, 2.f, 3.f, (float)d_)                         \
                           : *reinterpret_cast<const float4*>(sbaseA + (voff[j] + (cok_ ? d_ : 0u))); \
    if (GNF) {                                                                             \
      const float* gt_ = gtab + (cok_ ? (cc_) * 16 + 4 * q4 : 0);                          \
      gm = *reinterpret_cast<const float4*>(gt_);                                          \
      gr = *reinterpret_cast<const float4*>(gt_ + p.K);                                    \
      gg = *reinterpret_cast<const float4*>(gt_ + 2 * p.K);                                \
      gb = *reinterpret_cast<const float4*>(gt_ + 3 * p.K);                                \
    }                                                                                      \
  }
#define OSM_W_STORE_RAW()                                                                  \
  {                                                                                        \
    _Pragma("unroll") for (int j = 0; j < WN_NJ; ++j) {                                    \
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
      raw[woff[j]] = sel4((okm >> j) & 1u, v);                                             \
    }                                                                                      \
  }
#define OSM_W_LOAD_U(slot_, cc_, j_)                                                       \
  {                                                                                        \
    const unsigned so_ = (unsigned)(4 * wave + (j_)) * u_xi + (unsigned)(cc_) * u_slab;    \
    _Pragma("unroll") for (int b = 0; b < 2; ++b)                                          \
      _Pragma("unroll") for (int q2 = 0; q2 < NP; ++q2)                                    \
        uq[slot_][b][q2] = (WN_ABL & 2) ? make_uint4(so_, u_lane, q2, b)                   \
            : __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(             \
                  ursrc, (int)u_lane, (int)(so_ + (unsigned)q2 * u_plane + (unsigned)b * u_nt), 0)); \
  }

  if (kc1 > kc0) {
    OSM_W_LOAD_RAW(kc0);
#pragma unroll
    for (int j = 0; j < 4; ++j) OSM_W_LOAD_U(j, kc0, j);
    OSM_W_STORE_RAW();
    OSM_W_LOAD_RAW(min(kc0 + 1, kc1 - 1));
    for (int c = kc0; c < kc1; ++c) {
      __syncthreads();          // raw(c) is complete; every wave has left the M phase of slab c - 1 (vp is free)
      WN_T(t1);
      WN_ACC(2, t_prev, t1);        // M phase of the previous slab + wait at barrier A

      // ---- T: V_xi of this lane's tile, 8 channels (quads 2 lk, 2 lk + 1), xi rows 2 hf and 2 hf + 1.
      // The two rows of B^T d this half owns, per input column j: hf = 0: t0 = d0 - d2, t1 = d1 + d2;  hf = 1:
      // t2 = d2 - d1 and -t3 = d3 - d1: both halves are  tA = x - y, tB = z + sg y  with (x, y, z) = input rows
      // (0, 2, 1), sg = +1  or  (2, 1, 3), sg = -1 -- no per-element select.  xi row 3 is therefore NEGATED in V; the
      // output transform adds instead of subtracts it.  Then V[.][0] = t0 - t2, [1] = t1 + t2, [2] = t2 - t1,
      // [3] = t1 - t3 along the columns.
      // Staged by hand: LDS reads queue behind this CU's 16-byte stores (13 cycles each) and one wave per SIMD has
      // nobody to cover their latency, so every column's reads are issued a stage before they are needed:
      //   reads col 0, 2 | reads col 1 | V[.][0] (6 stores) | reads col 3 | V[.][1], V[.][2] (12 stores) | V[.][3] (6)
      {
        osm::floatx4_t qx[4][2], qy[4][2], qz[4][2];      // [column j][channel quad]
        float tA[4][8], tB[4][8];
        const osm::floatx4_t* b_x = reinterpret_cast<const osm::floatx4_t*>(a_x);
        const osm::floatx4_t* b_y = reinterpret_cast<const osm::floatx4_t*>(a_y);
        const osm::floatx4_t* b_z = reinterpret_cast<const osm::floatx4_t*>(a_z);
#define OSM_W_RD(j_)                                                                               \
  {                                                                                                \
    const int o_ = ((j_) & 1) * 10 + ((j_) >> 1);                                                  \
    qx[j_][0] = b_x[o_]; qx[j_][1] = b_x[WN_QP + o_];                                              \
    qy[j_][0] = b_y[o_]; qy[j_][1] = b_y[WN_QP + o_];                                              \
    qz[j_][0] = b_z[o_]; qz[j_][1] = b_z[WN_QP + o_];                                              \
  }
// nothing that uses column j_ may start, and no LDS access may move, across this point (the compiler otherwise hoists
// the arithmetic up to the first loads and sinks the later loads down to their uses)
#define OSM_W_PIN(j_)                                                                              \
  asm volatile("" : "+v"(qx[j_][0]), "+v"(qx[j_][1]), "+v"(qy[j_][0]), "+v"(qy[j_][1]), "+v"(qz[j_][0]), \
               "+v"(qz[j_][1]) : : "memory");
#define OSM_W_TR1(j_, h_, e_, c_)                                                                  \
  tA[j_][e_] = qx[j_][h_].c_ - qy[j_][h_].c_;                                                      \
  tB[j_][e_] = fmaf(sg, qy[j_][h_].c_, qz[j_][h_].c_);
#define OSM_W_TR(j_)                                                                               \
  OSM_W_TR1(j_, 0, 0, x) OSM_W_TR1(j_, 0, 1, y) OSM_W_TR1(j_, 0, 2, z) OSM_W_TR1(j_, 0, 3, w)      \
  OSM_W_TR1(j_, 1, 4, x) OSM_W_TR1(j_, 1, 5, y) OSM_W_TR1(j_, 1, 6, z) OSM_W_TR1(j_, 1, 7, w)
// V[i][jv_] = sa_ * t[i][ja_] + sb_ * t[i][jb_] for both xi rows i of this half: split, store the NP planes
#define OSM_W_VOUT(jv_, ja_, sa_, jb_, sb_)                                                        \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                  \
    float f_[8];                                                                                   \
    _Pragma("unroll") for (int e = 0; e < 8; ++e)                                                  \
      f_[e] = (sa_) * (i ? tB[ja_][e] : tA[ja_][e]) + (sb_) * (i ? tB[jb_][e] : tA[jb_][e]);       \
    uint4 pl_[NP];                                                                                 \
    split_frag8<NP>(make_float4(f_[0], f_[1], f_[2], f_[3]), make_float4(f_[4], f_[5], f_[6], f_[7]), pl_); \
    _Pragma("unroll") for (int q2 = 0; q2 < NP; ++q2) v_wr[(((4 * i + (jv_)) * 2) * NP + q2) * 64] = pl_[q2]; \
  }
        OSM_W_RD(0)
        OSM_W_RD(2)
        OSM_W_RD(1)
        OSM_W_PIN(0)
        OSM_W_PIN(2)
        OSM_W_TR(0)
        OSM_W_TR(2)
        OSM_W_VOUT(0, 0, 1.f, 2, -1.f)
        OSM_W_RD(3)
        OSM_W_PIN(1)
        OSM_W_TR(1)
        OSM_W_VOUT(1, 1, 1.f, 2, 1.f)
        OSM_W_VOUT(2, 2, 1.f, 1, -1.f)
        OSM_W_PIN(3)
        OSM_W_TR(3)
        OSM_W_VOUT(3, 1, 1.f, 3, -1.f)
#undef OSM_W_RD
#undef OSM_W_PIN
#undef OSM_W_TR1
#undef OSM_W_TR
#undef OSM_W_VOUT
      }
      __syncthreads();          // V(c) is complete; raw is free
      __builtin_amdgcn_sched_barrier(0);
      WN_T(t3);
      WN_ACC(1, t1, t3);            // T phase + wait at barrier B
#if WN_ABL & 64
      t_prev = t3;
#endif
      // ---- M: xi = 4 wave + j.  Hand-placed stream: everything else this wave has to issue per slab -- the stores of
      // raw(c + 1), the loads of raw(c + 2), the refill of each U register set with the next slab's fragments as soon as
      // its last MFMA has issued, the V fragments of the next xi -- goes BETWEEN the MFMA groups, a few instructions at a
      // time, so that the matrix pipe never waits behind a burst of memory instructions (one wave per SIMD: nobody
      // else would fill the gap).  Product groups run smallest planes first: (0,2) | (1,1) (0,1) | (2,0) (1,0) (0,0).
      const int cn = min(c + 1, kc1 - 1);
      uint4 va[2][2][NP];       // [j parity][tile block][plane]
#define OSM_W_READ_V(j_, a_)                                                                       \
  _Pragma("unroll") for (int q2 = 0; q2 < NP; ++q2) va[(j_) & 1][a_][q2] = v_rd[(((j_) * 2 + (a_)) * NP + q2) * 64];
#define OSM_W_MMA(j_, pa_, pb_)                                                                    \
  if ((pa_) < NP && (pb_) < NP) {                                                                  \
    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                  \
      _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                \
        acc[j_][a][b] = mma16<NP>(va[(j_) & 1][a][pa_], uq[j_][b][pb_], acc[j_][a][b]);            \
  }
#define OSM_W_REFILL(j_, pb_)                                                                      \
  if ((pb_) < NP) {                                                                                \
    const unsigned so_ = (unsigned)(4 * wave + (j_)) * u_xi + (unsigned)cn * u_slab + (unsigned)(pb_) * u_plane; \
    _Pragma("unroll") for (int b = 0; b < 2; ++b)                                                  \
      uq[j_][b][pb_] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(            \
          ursrc, (int)u_lane, (int)(so_ + (unsigned)b * u_nt), 0));                                \
  }
#define OSM_W_FENCE() __builtin_amdgcn_sched_barrier(0)
      OSM_W_READ_V(0, 0)
      OSM_W_READ_V(0, 1)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        OSM_W_MMA(j, 0, 2)
        OSM_W_FENCE();
        OSM_W_REFILL(j, 2)
        if (j < 3) { OSM_W_READ_V(j + 1, 0) }
        if (j == 0) { OSM_W_STORE_RAW(); }          // raw(c + 1), loaded one M phase ago
        OSM_W_FENCE();
        OSM_W_MMA(j, 1, 1)
        OSM_W_MMA(j, 0, 1)
        OSM_W_FENCE();
        OSM_W_REFILL(j, 1)
        if (j < 3) { OSM_W_READ_V(j + 1, 1) }
        if (j == 1) { OSM_W_LOAD_RAW(min(c + 2, kc1 - 1)); }
        OSM_W_FENCE();
        OSM_W_MMA(j, 2, 0)
        OSM_W_MMA(j, 1, 0)
        OSM_W_MMA(j, 0, 0)
        OSM_W_FENCE();
        OSM_W_REFILL(j, 0)
      }
      OSM_W_FENCE();
#undef OSM_W_READ_V
#undef OSM_W_MMA
#undef OSM_W_REFILL
#undef OSM_W_FENCE
    }
  }
  WN_T(t_loop);
#undef OSM_W_LOAD_RAW
#undef OSM_W_STORE_RAW
#undef OSM_W_LOAD_U

  // ---- Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1].  xi columns (in this wave): s0 = M0 + M1 + M2, s1 = M1 - M2 - M3;
  // xi rows (= waves): Y[0][.] = s(0) + s(1) + s(2), Y[1][.] = s(1) - s(2) - s(3), where s(3) arrives negated (T phase).
  // One tile block per round.
  // Stores are issue-bound (one 4-byte store per lane and instruction costs as much as a 16-byte one), so the exchange
  // also transposes: the finishing wave reads (tile, 4 consecutive columns) per lane -- 16 contiguous bytes of red --
  // and writes dwordx4: 8 lanes cover the 128 B of one pixel's 32-column tile.
  float* red = reinterpret_cast<float*>(vp);         // [wave = xi row][ox][column tile][e][lane = 32 lk + column]
  const int oy = wave >> 1, ox = wave & 1;
  const bool partial = p.splitk > 1;
  const int ldo = partial ? p.N : (int)p.ldc;
  float* __restrict__ obase = (partial ? p.ws + ((long long)ks * p.M) * p.N : p.C) +
                              ((long long)img * p.H * p.W + (long long)(y0 + oy) * p.W + (x0 + ox)) * ldo;
  const float* __restrict__ rbase =
      (!partial && p.res) ? p.res + ((long long)img * p.H * p.W + (long long)(y0 + oy) * p.W + (x0 + ox)) * p.ldr : nullptr;
  // item it = 64 i + lane of a (tile block, column tile): columns 4 (it & 7) .. + 3 of accumulator element
  // e' = it >> 4 in lane half lk' = (it >> 3) & 1, i.e. tile (4 a + (e' >> 2), (e' & 3) + 4 lk') of the patch
  const int c4 = 4 * (lane & 7), lk2 = (lane >> 3) & 1, e_lo = lane >> 4;      // e' = 4 i + e_lo
  const int dx = 2 * (e_lo + 4 * lk2);                                          // pixel column offset in the patch
  const bool xok = x0 + ox + dx < p.W;
  const int lane_o = dx * ldo + c4, lane_r = dx * (int)p.ldr + c4;
  const float* red_rd = red + ((ox * 2) * 16 + e_lo) * 64 + lk2 * 32 + c4;      // + ((row * 4 + b) * 16 + 4 i) * 64
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    __syncthreads();            // vp / red is free (last M phase, or the previous round's reads, are over)
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
#pragma unroll
      for (int i = 0; i < 4; ++i) {                 // e' = 4 i + e_lo: tile row 4 a + i of the patch
        const float4 s0 = *reinterpret_cast<const float4*>(red_rd + (((oy + 0) * 4 + b) * 16 + 4 * i) * 64);
        const float4 s1 = *reinterpret_cast<const float4*>(red_rd + (((oy + 1) * 4 + b) * 16 + 4 * i) * 64);
        const float4 s2 = *reinterpret_cast<const float4*>(red_rd + (((oy + 2) * 4 + b) * 16 + 4 * i) * 64);
        float4 v;
        if (oy == 0) {
          v = make_float4(s0.x + s1.x + s2.x, s0.y + s1.y + s2.y, s0.z + s1.z + s2.z, s0.w + s1.w + s2.w);
        } else {
          v = make_float4(s0.x - s1.x + s2.x, s0.y - s1.y + s2.y, s0.z - s1.z + s2.z, s0.w - s1.w + s2.w);
        }
        const int dy = 8 * a + 2 * i;
        if (y0 + oy + dy >= p.H) continue;          // wave-uniform
        const bool ok = nok && xok;
        float* __restrict__ op = obase + (dy * p.W) * ldo + lane_o + (jn0 + b) * 32;
        if (partial) {
          if (ok) *reinterpret_cast<float4*>(op) = v;
        } else {
          v = make_float4(v.x * p.alpha + bv.x, v.y * p.alpha + bv.y, v.z * p.alpha + bv.z, v.w * p.alpha + bv.w);
          if (rbase && ok) {
            const float4 r = *reinterpret_cast<const float4*>(rbase + (dy * p.W) * (int)p.ldr + lane_r + (jn0 + b) * 32);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
          }
          if (p.accumulate && ok) {
            const float4 r = *reinterpret_cast<const float4*>(op);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
          }
          if (ok) *reinterpret_cast<float4*>(op) = v;
        }
      }
    }
  }
#if WN_ABL & 64
  {
    WN_T(t_end);
    prof[5] = t_end - t_loop;
    if (lane == 0) {
      for (int i = 0; i < 6; ++i) atomicAdd(&g_wn_prof[i], prof[i]);
      atomicAdd(&g_wn_prof[6], t_end - t_begin);
      atomicAdd(&g_wn_prof[7], 1ULL);
    }
  }
#endif
}

// OIHW fp32 -> Winograd-domain weights U = G g G^T, G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1], as np bf16 planes in
// MFMA-fragment order [plane][xi = 4 a + b][16-channel slab s][n/32 j][lane l][e]: n = 32 j + (l & 31),
// k = 16 s + 8 (l >> 5) + e.  forward: n = Cout, k = Cin; data-gradient: n = Cin, k = Cout, taps flipped.
// U is formed in double, rounded once to fp32, and that fp32 value is split exactly into the planes.
__global__ void pack_weight_wino_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout, int Cin,
                                        int np, int dgrad) {
  const int N = dgrad ? Cin : Cout;
  const int K = dgrad ? Cout : Cin;
  const int nt32 = (N + 31) / 32;
  const int ksteps = 2 * ((K + 31) / 32);
  const long long per_plane = 16LL * ksteps * nt32 * 512;
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
    float rr = (float)u;
    for (int q2 = 0; q2 < np; ++q2) {
      const __bf16 bb = (__bf16)rr;
      out[q2 * per_plane + i] = __builtin_bit_cast(unsigned short, bb);
      rr -= (float)bb;
    }
  }
}

#if WN_ABL & 64
}  // namespace
extern "C" int osm_debug_wino_prof(unsigned long long* out8, int reset) {   // instrumented builds only (tools/wino_phases.py)
  if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_wn_prof), sizeof(g_wn_prof)) != hipSuccess) return 1;
  if (reset) {
    const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wn_prof), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
namespace {
#endif

#endif   // !OSM_ACT_F16
