// EXPERIMENT (round 6, measurement builds only: -DOSM_WITH_WINO4): Winograd F(2x2,3x3) f16x3 with ONE wave per SIMD and a 96-column
// workgroup tile.  Included inside igemm.hip's anonymous namespace after conv3_wino8.inc.h.
//
// Why: the shipped 8-wave kernel spends half of the register file on the working sets of eight waves (8 x 128) and half on
// accumulators (16 xi x 64 tiles x 64 columns).  Four waves need 4 x 128 for their working sets, which leaves 4 x 384 accumulator
// registers = 16 xi x 64 tiles x 96 columns: every V fragment (the transform + split VALU, the LDS reads) then feeds 9 MFMAs instead
// of 6, and a wave owns a whole xi ROW, so the column sums of the output transform stay in registers (half the LDS exchange per output).
//   wave r      : xi row r (B^T d rows as in the 8-wave kernel), xi columns j = 0..3, both tile blocks, three column tiles:
//                 acc[j][tb][ct], 24 x 16 registers
//   unit (j, tb): V = t[ca] + sb t[cb] for tile block tb, split into two half planes, 9 MFMAs (3 column tiles x 3 plane pairs)
//   U           : 6 fragments per xi (3 column tiles x 2 planes), loaded one xi ahead into a second register set
// Plain case only: K % 16 == 0, no split-K, no fused statistics (MODE 1 epilogue: alpha, bias, residual, accumulate).
#ifndef W4_NCT
#define W4_NCT 3
#endif
#ifndef W4_PAIR
#define W4_PAIR 0        // v2: 1 = two xi per chunk (four accumulators in rotation)
#endif
#ifndef W4_ABL
#define W4_ABL 0          // 4: no epilogue
#endif
constexpr int W4_NJ = 6;                     // raw staging pieces per thread and slab (1296 pieces, 256 threads)

template <int NCT>
__global__ __launch_bounds__(256, 1) void conv3_wino4_kernel(const act_t* __restrict__ Aglob, const unsigned short* __restrict__ Uglob,
                                                              IGemmParams p) {
  // the two raw slabs (46 KB) during the slab loop, the exchange buffer of the epilogue after it: [row 4][ox 2][ct][e 16][lane 64]
  __shared__ __attribute__((aligned(16))) float smem[4 * 2 * NCT * 16 * 64 > 2 * 4 * WN_QP * 4 ? 4 * 2 * NCT * 16 * 64 : 2 * 4 * WN_QP * 4];
  float4* raw = reinterpret_cast<float4*>(smem);
  float* red = smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wr = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lk = lane >> 5;

  // tile mapping: ids enumerate (M-tile, column tile); the column tiles of one M-tile are adjacent (one XCD, same time)
  const int nt = p.mtiles * p.ntiles;
  const int bid = blockIdx.x;
  const int qq = nt >> 3, rr8 = nt & 7, xcd = bid & 7, idx8 = bid >> 3;
  const int id = (xcd < rr8 ? xcd * (qq + 1) : rr8 * (qq + 1) + (xcd - rr8) * qq) + idx8;
  const int tile_n = id % p.ntiles, tile_m = id / p.ntiles;
  const int tpx = (p.W + 15) >> 4, tpy = (p.H + 15) >> 4;
  const int tx = tile_m % tpx, ty = (tile_m / tpx) % tpy, img = tile_m / (tpx * tpy);
  const int x0 = tx * 16, y0 = ty * 16;
  const int nslab = p.ksteps;

  // raw staging: piece s = tid + 256 j -> channel quad tid & 3, halo pixel (tid >> 2) + 64 j
  const int q4 = tid & 3;
  const long long rowB = (long long)p.lda * ACT_B;
  const char* __restrict__ sbaseA = reinterpret_cast<const char*>(Aglob) + (long long)img * p.H * p.W * rowB;
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(sbaseA), 0, (int)min((long long)p.H * p.W * rowB, 0x7fffffffLL), 0x00020000);
  unsigned aoff[W4_NJ], woff[W4_NJ];
#pragma unroll
  for (int j = 0; j < W4_NJ; ++j) {
    const int pix = (tid >> 2) + 64 * j;
    const int r = pix / 18, col = pix - r * 18;
    const bool in = pix < 324;
    woff[j] = (unsigned)(q4 * WN_QP + (in ? r * WN_ROWP + (col & 1) * 10 + (col >> 1) : 17 * WN_ROWP + 19));
    const int y = y0 - 1 + r, x = x0 - 1 + col;
    const bool ok = in && y >= 0 && y < p.H && x >= 0 && x < p.W;
    aoff[j] = ok ? (unsigned)((long long)(y * p.W + x) * rowB) + (unsigned)(q4 * 4 * ACT_B) : 0x80000000u;
  }
  // U fragments: image [plane][xi][slab][n/32][lane][8]; this wave reads xi = 4 wr + j
  const int jn0 = NCT * tile_n;
  const unsigned u_lane = (unsigned)((jn0 * 64 + lane) * 16);
  const unsigned u_slab = (unsigned)p.nt32 * 1024u;
  const unsigned u_xi = u_slab * (unsigned)nslab;
  const unsigned u_plane = u_xi * 16u;
  unsigned u_ct[NCT];
#pragma unroll
  for (int b = 0; b < NCT; ++b) u_ct[b] = jn0 + b < p.nt32 ? 1024u * b : 0u;     // (column tiles past N re-read tile 0: their results are dropped)
  const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(Uglob)), 0, 0x7fffffff, 0x00020000);

  f32x16 acc[4][2][NCT];
  {
    const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
    f32x16 z16;
#pragma unroll
    for (int e = 0; e < 16; ++e) z16[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NCT; ++b) {
          uint4 z = z4;
          asm("" : "+v"(z.x));
          acc[j][a][b] = mma16h(z, z4, z16);
        }
  }
  // this wave's row of B^T d: t = x + sg y, (x, y) = input rows (0, 2) | (1, 2) | (2, 1) | (1, 3) of the 4 x 4 tile, sg = -1 | +1 | -1 | -1
  const int tyl = lr >> 3, txl = lr & 7;
  const int rx = wr == 0 ? 0 : (wr == 2 ? 2 : 1), ry = wr == 2 ? 1 : (wr == 3 ? 3 : 2);
  const float sg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(wr == 1 ? 0x3f800000 : (int)0xbf800000));
  const osm::floatx4_t* t_x = reinterpret_cast<const osm::floatx4_t*>(raw) + (2 * lk) * WN_QP + (2 * tyl + rx) * WN_ROWP + txl;
  const osm::floatx4_t* t_y = reinterpret_cast<const osm::floatx4_t*>(raw) + (2 * lk) * WN_QP + (2 * tyl + ry) * WN_ROWP + txl;

  float4 ra[W4_NJ];
#pragma unroll
  for (int j = 0; j < W4_NJ; ++j)
    ra[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, (int)aoff[j], 0, 0));
  // x 2^ex brings the largest |x| of the image to [2^11, 2^12)
  float xscale, oscale;
  {
    static_assert(OSM_MAXABS_PARTS == 1024, "four partial maxima per thread");
    const unsigned* xm = reinterpret_cast<const unsigned*>(p.xmax) + (long long)img * OSM_MAXABS_PARTS;
    unsigned mb = max(max(xm[tid], xm[tid + 256]), max(xm[tid + 512], xm[tid + 768]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, o, 64));
    unsigned* red_u = reinterpret_cast<unsigned*>(smem);
    if (lane == 0) red_u[wr] = mb;
    __syncthreads();
    mb = max(max(red_u[0], red_u[1]), max(red_u[2], red_u[3]));
    __syncthreads();
    const float mx = __uint_as_float(mb);
    int ex = 0;
    if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &ex); ex = min(12 - ex, 100); }
    xscale = mx == mx ? ldexpf(1.f, ex) : mx;
    oscale = ldexpf(1.f, -ex) / p.wscale[0];
    xscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, xscale)));
    oscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, oscale)));
  }
  uint4 uq[2][NCT][2];      // [set][column tile][plane]: the xi in use and the next one
  uint4 va[2][2];           // A fragments [parity][plane]
  float4 ts[2][2];          // the t column two neighbouring xi share, [tile block][channel quad]

#define W4_STORE_RAW(buf_, j_) \
  raw[(buf_) * (4 * WN_QP) + woff[j_]] = make_float4(ra[j_].x * xscale, ra[j_].y * xscale, ra[j_].z * xscale, ra[j_].w * xscale);
#define W4_LOAD_RAW(cc_, j_) \
  ra[j_] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, (int)aoff[j_], (cc_) * (16 * ACT_B), 0));
#define W4_LOAD_U(set_, cc_, j_)                                                                   \
  {                                                                                                \
    const unsigned so_ = (unsigned)(4 * wr + (j_)) * u_xi + (unsigned)(cc_) * u_slab;              \
    _Pragma("unroll") for (int b = 0; b < NCT; ++b)                                                \
      _Pragma("unroll") for (int q2 = 0; q2 < 2; ++q2)                                             \
        uq[set_][b][q2] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(         \
            ursrc, (int)u_lane, (int)(so_ + (unsigned)q2 * u_plane + u_ct[b]), 0));                \
  }
// V fragment of unit (xi column j_, tile block tb_) from the raw slab at slot offset bo_: V = t[ca] + sb t[cb]; keep_ 1: t[cb] -> ts[tb_];
// use_ 1: t[cb] from ts[tb_], 2: t[ca] from ts[tb_]
#define W4_BUILD(par_, ca_, cb_, sb_, tb_, bo_, keep_, use_)                                       \
  {                                                                                                \
    uint2 vh_[2][2];                                                                               \
    _Pragma("unroll") for (int hq = 0; hq < 2; ++hq) {                                             \
      const int oa_ = (bo_) + hq * WN_QP + 8 * (tb_) * WN_ROWP + ((ca_) & 1) * 10 + ((ca_) >> 1);  \
      const int ob_ = (bo_) + hq * WN_QP + 8 * (tb_) * WN_ROWP + ((cb_) & 1) * 10 + ((cb_) >> 1);  \
      float4 ta_, tb2_;                                                                            \
      if ((use_) == 2) ta_ = ts[tb_][hq];                                                          \
      else {                                                                                       \
        const osm::floatx4_t xa_ = t_x[oa_], ya_ = t_y[oa_];                                       \
        ta_ = make_float4(fmaf(sg, ya_[0], xa_[0]), fmaf(sg, ya_[1], xa_[1]), fmaf(sg, ya_[2], xa_[2]), fmaf(sg, ya_[3], xa_[3])); \
      }                                                                                            \
      if ((use_) == 1) tb2_ = ts[tb_][hq];                                                         \
      else {                                                                                       \
        const osm::floatx4_t xb_ = t_x[ob_], yb_ = t_y[ob_];                                       \
        tb2_ = make_float4(fmaf(sg, yb_[0], xb_[0]), fmaf(sg, yb_[1], xb_[1]), fmaf(sg, yb_[2], xb_[2]), fmaf(sg, yb_[3], xb_[3])); \
      }                                                                                            \
      if (keep_) ts[tb_][hq] = tb2_;                                                               \
      const float4 v_ = make_float4(ta_.x + (sb_) * tb2_.x, ta_.y + (sb_) * tb2_.y, ta_.z + (sb_) * tb2_.z, ta_.w + (sb_) * tb2_.w); \
      split_f16x2(v_, vh_[hq]);                                                                    \
    }                                                                                              \
    va[par_][0] = make_uint4(vh_[0][0].x, vh_[0][0].y, vh_[1][0].x, vh_[1][0].y);                  \
    va[par_][1] = make_uint4(vh_[0][1].x, vh_[0][1].y, vh_[1][1].x, vh_[1][1].y);                  \
  }
// the 9 MFMAs of unit (j_, tb_): plane pairs (1,0), (0,1), (0,0), round-robin over the column tiles (no two in a row on one accumulator)
#define W4_MMA(par_, set_, j_, tb_)                                                                \
  _Pragma("unroll") for (int b = 0; b < NCT; ++b) acc[j_][tb_][b] = mma16h(va[par_][1], uq[set_][b][0], acc[j_][tb_][b]); \
  _Pragma("unroll") for (int b = 0; b < NCT; ++b) acc[j_][tb_][b] = mma16h(va[par_][0], uq[set_][b][1], acc[j_][tb_][b]); \
  _Pragma("unroll") for (int b = 0; b < NCT; ++b) acc[j_][tb_][b] = mma16h(va[par_][0], uq[set_][b][0], acc[j_][tb_][b]);
#define W4_FENCE() asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);

  // xi column j: V = t[CA] + SB t[CB]:  j = 0: t0 - t2 (keeps t2);  1: t1 + t2 (t2 from ts);  2: t2 - t1 (keeps t1);  3: t1 - t3 (t1 from ts)
  if (nslab > 0) {
#pragma unroll
    for (int j = 0; j < W4_NJ; ++j) { W4_STORE_RAW(0, j) }
    const int k1 = min(1, nslab - 1);
#pragma unroll
    for (int j = 0; j < W4_NJ; ++j) { W4_LOAD_RAW(k1, j) }
    W4_LOAD_U(0, 0, 0)
    __syncthreads();
    W4_BUILD(0, 0, 2, -1.f, 0, 0, 1, 0)
    W4_FENCE()
    for (int c = 0; c < nslab; ++c) {
      const int c1 = min(c + 1, nslab - 1), c2 = min(c + 2, nslab - 1);
      const int bo = (c & 1) * (4 * WN_QP), bn = ((c + 1) & 1) * (4 * WN_QP);
      const int qb = (c + 1) & 1;
      // xi 0: units (0, tb 0) | (0, tb 1); U of xi 1
      W4_LOAD_U(1, c, 1)
      W4_STORE_RAW(qb, 0) W4_LOAD_RAW(c2, 0)
      W4_BUILD(1, 0, 2, -1.f, 1, bo, 1, 0) W4_MMA(0, 0, 0, 0)
      W4_FENCE()
      W4_STORE_RAW(qb, 1) W4_LOAD_RAW(c2, 1)
      W4_BUILD(0, 1, 2, 1.f, 0, bo, 0, 1) W4_MMA(1, 0, 0, 1)
      W4_FENCE()
      // xi 1; U of xi 2
      W4_LOAD_U(0, c, 2)
      W4_STORE_RAW(qb, 2) W4_LOAD_RAW(c2, 2)
      W4_BUILD(1, 1, 2, 1.f, 1, bo, 0, 1) W4_MMA(0, 1, 1, 0)
      W4_FENCE()
      W4_STORE_RAW(qb, 3) W4_LOAD_RAW(c2, 3)
      W4_BUILD(0, 2, 1, -1.f, 0, bo, 1, 0) W4_MMA(1, 1, 1, 1)
      W4_FENCE()
      // xi 2; U of xi 3
      W4_LOAD_U(1, c, 3)
      W4_STORE_RAW(qb, 4) W4_LOAD_RAW(c2, 4)
      W4_BUILD(1, 2, 1, -1.f, 1, bo, 1, 0) W4_MMA(0, 0, 2, 0)
      W4_FENCE()
      W4_STORE_RAW(qb, 5) W4_LOAD_RAW(c2, 5)
      W4_BUILD(0, 1, 3, -1.f, 0, bo, 0, 2) W4_MMA(1, 0, 2, 1)
      W4_FENCE()
      // xi 3; U of xi 0 of the next slab
      W4_LOAD_U(0, c1, 0)
      W4_BUILD(1, 1, 3, -1.f, 1, bo, 0, 2) W4_MMA(0, 1, 3, 0)
      W4_FENCE()
      __syncthreads();          // raw(c + 1) is complete in its buffer; nobody reads raw(c) any more
      W4_BUILD(0, 0, 2, -1.f, 0, bn, 1, 0) W4_MMA(1, 1, 3, 1)
      W4_FENCE()
    }
  }
#undef W4_STORE_RAW
#undef W4_LOAD_RAW
#undef W4_LOAD_U
#undef W4_BUILD
#undef W4_MMA
#undef W4_FENCE
  if ((W4_ABL & 4) && p.alpha != 12345.f) return;

  // ---- Y = A^T M A.  xi columns (in registers): s0 = M0 + M1 + M2, s1 = M1 - M2 - M3;  xi rows (through LDS):
  // Y[0][.] = s(0) + s(1) + s(2), Y[1][.] = s(1) - s(2) - s(3).  One tile block per round.
  // red: [row r][ox][ct][e][lane = 32 lk + column];  finishing wave f = (oy = f >> 1, ox = f & 1), all column tiles.
  const int oy = wr >> 1, ox = wr & 1;
  const long long pix0 = (long long)img * p.H * p.W + (long long)(y0 + oy) * p.W + (x0 + ox);
  act_t* __restrict__ obase = p.C + pix0 * p.ldc;
  const act_t* __restrict__ rbase = p.res ? p.res + pix0 * p.ldr : nullptr;
  const int c4 = 4 * (lane & 7), lk2 = (lane >> 3) & 1, e_lo = lane >> 4;
  const int dx = 2 * (e_lo + 4 * lk2);
  const bool xok = x0 + ox + dx < p.W;
  const float sg1 = oy == 0 ? 1.f : -1.f;       // rows oy+1, oy+2 enter with + + (oy = 0) | - - (oy = 1)
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NCT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float m0 = acc[0][a][b][e], m1 = acc[1][a][b][e], m2 = acc[2][a][b][e], m3 = acc[3][a][b][e];
        red[(((wr * 2 + 0) * NCT + b) * 16 + e) * 64 + lane] = (m0 + m1) + m2;
        red[(((wr * 2 + 1) * NCT + b) * 16 + e) * 64 + lane] = (m1 - m2) - m3;
      }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NCT; ++b) {
      const int n = (jn0 + b) * 32 + c4;
      const bool nok = n < p.N && jn0 + b < p.nt32;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias && nok) bv = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* rd = red + ((ox * NCT + b) * 16 + 4 * i + e_lo) * 64 + lk2 * 32 + c4;
        const float4 s0 = *reinterpret_cast<const float4*>(rd + (oy + 0) * (2 * NCT * 16 * 64));
        const float4 s1 = *reinterpret_cast<const float4*>(rd + (oy + 1) * (2 * NCT * 16 * 64));
        const float4 s2 = *reinterpret_cast<const float4*>(rd + (oy + 2) * (2 * NCT * 16 * 64));
        float4 v = make_float4(s0.x + sg1 * s1.x + sg1 * s2.x, s0.y + sg1 * s1.y + sg1 * s2.y, s0.z + sg1 * s1.z + sg1 * s2.z,
                               s0.w + sg1 * s1.w + sg1 * s2.w);
        const int dy = 8 * a + 2 * i;
        if (y0 + oy + dy >= p.H || !(nok && xok)) continue;
        const int po = dy * p.W + dx;
        v = make_float4(v.x * oscale * p.alpha + bv.x, v.y * oscale * p.alpha + bv.y, v.z * oscale * p.alpha + bv.z,
                        v.w * oscale * p.alpha + bv.w);
        act_t* __restrict__ op = obase + po * (int)p.ldc + n;
        if (rbase) { const float4 r4 = osm::ld4(rbase + po * (int)p.ldr + n); v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
        if (p.accumulate) { const float4 a4 = osm::ld4(op); v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w; }
        osm::st4(op, v);
      }
    }
  }
}

// ================================================================================================ v2
template <int NCT>
__global__ __launch_bounds__(256, 1) void conv3_wino4p_kernel(const act_t* __restrict__ Aglob, const unsigned short* __restrict__ Uglob,
                                                              IGemmParams p) {
  // the two raw slabs (46 KB) during the slab loop, the exchange buffer of the epilogue after it: [row 4][ox 2][ct][e 16][lane 64]
  __shared__ __attribute__((aligned(16))) float smem[4 * 2 * NCT * 16 * 64 > 2 * 4 * WN_QP * 4 ? 4 * 2 * NCT * 16 * 64 : 2 * 4 * WN_QP * 4];
  float4* raw = reinterpret_cast<float4*>(smem);
  float* red = smem;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wr = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lk = lane >> 5;

  // tile mapping: ids enumerate (M-tile, column tile); the column tiles of one M-tile are adjacent (one XCD, same time)
  const int nt = p.mtiles * p.ntiles;
  const int bid = blockIdx.x;
  const int qq = nt >> 3, rr8 = nt & 7, xcd = bid & 7, idx8 = bid >> 3;
  const int id = (xcd < rr8 ? xcd * (qq + 1) : rr8 * (qq + 1) + (xcd - rr8) * qq) + idx8;
  const int tile_n = id % p.ntiles, tile_m = id / p.ntiles;
  const int tpx = (p.W + 15) >> 4, tpy = (p.H + 15) >> 4;
  const int tx = tile_m % tpx, ty = (tile_m / tpx) % tpy, img = tile_m / (tpx * tpy);
  const int x0 = tx * 16, y0 = ty * 16;
  const int nslab = p.ksteps;

  // raw staging: piece s = tid + 256 j -> channel quad tid & 3, halo pixel (tid >> 2) + 64 j
  const int q4 = tid & 3;
  const long long rowB = (long long)p.lda * ACT_B;
  const char* __restrict__ sbaseA = reinterpret_cast<const char*>(Aglob) + (long long)img * p.H * p.W * rowB;
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(sbaseA), 0, (int)min((long long)p.H * p.W * rowB, 0x7fffffffLL), 0x00020000);
  unsigned aoff[W4_NJ], woff[W4_NJ];
#pragma unroll
  for (int j = 0; j < W4_NJ; ++j) {
    const int pix = (tid >> 2) + 64 * j;
    const int r = pix / 18, col = pix - r * 18;
    const bool in = pix < 324;
    woff[j] = (unsigned)(q4 * WN_QP + (in ? r * WN_ROWP + (col & 1) * 10 + (col >> 1) : 17 * WN_ROWP + 19));
    const int y = y0 - 1 + r, x = x0 - 1 + col;
    const bool ok = in && y >= 0 && y < p.H && x >= 0 && x < p.W;
    aoff[j] = ok ? (unsigned)((long long)(y * p.W + x) * rowB) + (unsigned)(q4 * 4 * ACT_B) : 0x80000000u;
  }
  // U fragments: image [plane][xi][slab][n/32][lane][8]; this wave reads xi = 4 wr + j
  const int jn0 = NCT * tile_n;
  const unsigned u_lane = (unsigned)((jn0 * 64 + lane) * 16);
  const unsigned u_slab = (unsigned)p.nt32 * 1024u;
  const unsigned u_xi = u_slab * (unsigned)nslab;
  const unsigned u_plane = u_xi * 16u;
  unsigned u_ct[NCT];
#pragma unroll
  for (int b = 0; b < NCT; ++b) u_ct[b] = jn0 + b < p.nt32 ? 1024u * b : 0u;     // (column tiles past N re-read tile 0: their results are dropped)
  const __amdgpu_buffer_rsrc_t ursrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<char*>(reinterpret_cast<const char*>(Uglob)), 0, 0x7fffffff, 0x00020000);

  f32x16 acc[4][2][NCT];
  {
    const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
    f32x16 z16;
#pragma unroll
    for (int e = 0; e < 16; ++e) z16[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NCT; ++b) {
          uint4 z = z4;
          asm("" : "+v"(z.x));
          acc[j][a][b] = mma16h(z, z4, z16);
        }
  }
  // this wave's row of B^T d: t = x + sg y, (x, y) = input rows (0, 2) | (1, 2) | (2, 1) | (1, 3) of the 4 x 4 tile, sg = -1 | +1 | -1 | -1
  const int tyl = lr >> 3, txl = lr & 7;
  const int rx = wr == 0 ? 0 : (wr == 2 ? 2 : 1), ry = wr == 2 ? 1 : (wr == 3 ? 3 : 2);
  const float sg = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(wr == 1 ? 0x3f800000 : (int)0xbf800000));
  const osm::floatx4_t* t_x = reinterpret_cast<const osm::floatx4_t*>(raw) + (2 * lk) * WN_QP + (2 * tyl + rx) * WN_ROWP + txl;
  const osm::floatx4_t* t_y = reinterpret_cast<const osm::floatx4_t*>(raw) + (2 * lk) * WN_QP + (2 * tyl + ry) * WN_ROWP + txl;

  float4 ra[W4_NJ];
#pragma unroll
  for (int j = 0; j < W4_NJ; ++j)
    ra[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, (int)aoff[j], 0, 0));
  // x 2^ex brings the largest |x| of the image to [2^11, 2^12)
  float xscale, oscale;
  {
    static_assert(OSM_MAXABS_PARTS == 1024, "four partial maxima per thread");
    const unsigned* xm = reinterpret_cast<const unsigned*>(p.xmax) + (long long)img * OSM_MAXABS_PARTS;
    unsigned mb = max(max(xm[tid], xm[tid + 256]), max(xm[tid + 512], xm[tid + 768]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, o, 64));
    unsigned* red_u = reinterpret_cast<unsigned*>(smem);
    if (lane == 0) red_u[wr] = mb;
    __syncthreads();
    mb = max(max(red_u[0], red_u[1]), max(red_u[2], red_u[3]));
    __syncthreads();
    const float mx = __uint_as_float(mb);
    int ex = 0;
    if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &ex); ex = min(12 - ex, 100); }
    xscale = mx == mx ? ldexpf(1.f, ex) : mx;
    oscale = ldexpf(1.f, -ex) / p.wscale[0];
    xscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, xscale)));
    oscale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, oscale)));
  }
  // ---- v2: software-pipelined at tile-block granularity, the whole xi ROW of a wave sharing its t columns.
  //   phase 0 of slab c: the 4 x 2 NCT x 3 MFMAs of tile block 0 (V fragments vf[0][j], built during the previous phase 1) while the
  //                      V fragments of tile block 1 are built from raw(c) and the staged pieces of slab c + 1 are stored; barrier;
  //   phase 1:           the MFMAs of tile block 1 (vf[1][j]) while vf[0][j] of slab c + 1 is built from raw(c + 1), every xi's U
  //                      fragments are re-requested for slab c + 1 right after their last MFMA, the pieces of slab c + 2 are requested.
  //   A build = per channel quad hq: 8 LDS reads (t columns 0..3: rows x, y) issued one xi EARLIER than their arithmetic
  //   (inline asm, waited for by hand), 16 FMAs (the four t), 16 adds (V0 = t0 - t2, V1 = t1 + t2, V2 = t2 - t1, V3 = t1 - t3),
  //   4 splits: 28 VALU and 4 LDS reads per V fragment (the 8-wave kernel: 36-40 and 6).
  uint4 uq[4][NCT][2];      // [xi column j][column tile][plane] of the slab in hand
  uint4 vf[2][4][2];        // A fragments [tile block][xi column j][plane]
  osm::floatx4_t rd[8];     // landing registers of one channel quad's reads: column c -> rd[2 c] (row x), rd[2 c + 1] (row y)
  const unsigned txa = (unsigned)(size_t)t_x, tya = (unsigned)(size_t)t_y;
#define W4P_RD(dst_, addr_, slot_) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "n"((slot_) * 16));
#define W4P_READS(hq_, buf_, tb_)                                                                   \
  _Pragma("unroll") for (int cc = 0; cc < 4; ++cc) {                                                \
    W4P_RD(rd[2 * cc], txa, ((buf_) * 4 + (hq_)) * WN_QP + 8 * (tb_) * WN_ROWP + (cc & 1) * 10 + (cc >> 1))     \
    W4P_RD(rd[2 * cc + 1], tya, ((buf_) * 4 + (hq_)) * WN_QP + 8 * (tb_) * WN_ROWP + (cc & 1) * 10 + (cc >> 1)) \
  }
#define W4P_WAIT() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rd[0]), "+v"(rd[1]), "+v"(rd[2]), "+v"(rd[3]), "+v"(rd[4]), "+v"(rd[5]), "+v"(rd[6]), "+v"(rd[7]));
  // arithmetic of channel quad hq_ of tile block tb_: the four t columns, the four V, split -> words 2 hq_, 2 hq_ + 1 of vf[tb_][j][plane]
  float4 tt[4];
#define W4P_PUT(hq_, tb_, j_, v_)                                                                   \
  {                                                                                                 \
    uint2 vh_[2];                                                                                   \
    split_f16x2(v_, vh_);                                                                           \
    if ((hq_) == 0) { vf[tb_][j_][0].x = vh_[0].x; vf[tb_][j_][0].y = vh_[0].y; vf[tb_][j_][1].x = vh_[1].x; vf[tb_][j_][1].y = vh_[1].y; } \
    else            { vf[tb_][j_][0].z = vh_[0].x; vf[tb_][j_][0].w = vh_[0].y; vf[tb_][j_][1].z = vh_[1].x; vf[tb_][j_][1].w = vh_[1].y; } \
  }
  // part A: the four t columns (the landing registers are free afterwards) and V0;  part B: V1, V2, V3 -- 26 / 30 VALU
#define W4P_MATH_A(hq_, tb_)                                                                        \
  {                                                                                                 \
    W4P_WAIT()                                                                                      \
    _Pragma("unroll") for (int cc = 0; cc < 4; ++cc)                                                \
      tt[cc] = make_float4(fmaf(sg, rd[2 * cc + 1][0], rd[2 * cc][0]), fmaf(sg, rd[2 * cc + 1][1], rd[2 * cc][1]), \
                           fmaf(sg, rd[2 * cc + 1][2], rd[2 * cc][2]), fmaf(sg, rd[2 * cc + 1][3], rd[2 * cc][3])); \
    W4P_PUT(hq_, tb_, 0, make_float4(tt[0].x - tt[2].x, tt[0].y - tt[2].y, tt[0].z - tt[2].z, tt[0].w - tt[2].w)) \
  }
#define W4P_MATH_B(hq_, tb_)                                                                        \
  {                                                                                                 \
    W4P_PUT(hq_, tb_, 1, make_float4(tt[1].x + tt[2].x, tt[1].y + tt[2].y, tt[1].z + tt[2].z, tt[1].w + tt[2].w)) \
    W4P_PUT(hq_, tb_, 2, make_float4(tt[2].x - tt[1].x, tt[2].y - tt[1].y, tt[2].z - tt[1].z, tt[2].w - tt[1].w)) \
    W4P_PUT(hq_, tb_, 3, make_float4(tt[1].x - tt[3].x, tt[1].y - tt[3].y, tt[1].z - tt[3].z, tt[1].w - tt[3].w)) \
  }
#define W4P_MATH(hq_, tb_) W4P_MATH_A(hq_, tb_) W4P_MATH_B(hq_, tb_)
#define W4P_MMA(j_, tb_)                                                                            \
  _Pragma("unroll") for (int b = 0; b < NCT; ++b) acc[j_][tb_][b] = mma16h(vf[tb_][j_][1], uq[j_][b][0], acc[j_][tb_][b]); \
  _Pragma("unroll") for (int b = 0; b < NCT; ++b) acc[j_][tb_][b] = mma16h(vf[tb_][j_][0], uq[j_][b][1], acc[j_][tb_][b]); \
  _Pragma("unroll") for (int b = 0; b < NCT; ++b) acc[j_][tb_][b] = mma16h(vf[tb_][j_][0], uq[j_][b][0], acc[j_][tb_][b]);
// two xi at a time: four independent accumulators in rotation (a dependent MFMA then follows its predecessor three MFMAs later)
#define W4P_MMA2(ja_, jb_, tb_)                                                                     \
  _Pragma("unroll") for (int b = 0; b < NCT; ++b) { acc[ja_][tb_][b] = mma16h(vf[tb_][ja_][1], uq[ja_][b][0], acc[ja_][tb_][b]);   \
                                                    acc[jb_][tb_][b] = mma16h(vf[tb_][jb_][1], uq[jb_][b][0], acc[jb_][tb_][b]); } \
  _Pragma("unroll") for (int b = 0; b < NCT; ++b) { acc[ja_][tb_][b] = mma16h(vf[tb_][ja_][0], uq[ja_][b][1], acc[ja_][tb_][b]);   \
                                                    acc[jb_][tb_][b] = mma16h(vf[tb_][jb_][0], uq[jb_][b][1], acc[jb_][tb_][b]); } \
  _Pragma("unroll") for (int b = 0; b < NCT; ++b) { acc[ja_][tb_][b] = mma16h(vf[tb_][ja_][0], uq[ja_][b][0], acc[ja_][tb_][b]);   \
                                                    acc[jb_][tb_][b] = mma16h(vf[tb_][jb_][0], uq[jb_][b][0], acc[jb_][tb_][b]); }
#define W4P_LOAD_U(cc_, j_)                                                                         \
  {                                                                                                 \
    const unsigned so_ = (unsigned)(4 * wr + (j_)) * u_xi + (unsigned)(cc_) * u_slab;               \
    _Pragma("unroll") for (int b = 0; b < NCT; ++b)                                                 \
      _Pragma("unroll") for (int q2 = 0; q2 < 2; ++q2)                                              \
        uq[j_][b][q2] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(            \
            ursrc, (int)u_lane, (int)(so_ + (unsigned)q2 * u_plane + u_ct[b]), 0));                 \
  }
#define W4P_STORE_RAW(buf_, j_) \
  raw[(buf_) * (4 * WN_QP) + woff[j_]] = make_float4(ra[j_].x * xscale, ra[j_].y * xscale, ra[j_].z * xscale, ra[j_].w * xscale);
#define W4P_LOAD_RAW(cc_, j_) \
  ra[j_] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, (int)aoff[j_], (cc_) * (16 * ACT_B), 0));
#define W4P_FENCE() asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0);
  if (nslab > 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { W4P_LOAD_U(0, j) }
#pragma unroll
    for (int j = 0; j < W4_NJ; ++j) { W4P_STORE_RAW(0, j) }
    const int k1 = min(1, nslab - 1);
#pragma unroll
    for (int j = 0; j < W4_NJ; ++j) { W4P_LOAD_RAW(k1, j) }
    __syncthreads();
    W4P_READS(0, 0, 0) W4P_MATH(0, 0)
    W4P_READS(1, 0, 0) W4P_MATH(1, 0)
    W4P_READS(0, 0, 1)
    W4P_FENCE()
    auto slab = [&](auto pc, const int c) __attribute__((always_inline)) {
      constexpr int P = decltype(pc)::value, Q = 1 - P;
      const int c1 = min(c + 1, nslab - 1), c2 = min(c + 2, nslab - 1);
      // eight chunks of 2 NCT x 3 MFMAs, each beside ~28 VALU of the builds:
      //   k0 A(hq0, tb1) then the reads of hq1 | k1 B(hq0, tb1) | k2 A(hq1, tb1) | barrier | k3 B(hq1, tb1), the reads of (hq0, tb0) of slab c + 1
      //   k4 A(hq0, tb0') then the reads of hq1 | k5 B(hq0, tb0') | k6 A(hq1, tb0') | k7 B(hq1, tb0'), the reads of (hq0, tb1) of slab c + 1
#if W4_PAIR
      W4P_STORE_RAW(Q, 0) W4P_STORE_RAW(Q, 1) W4P_STORE_RAW(Q, 2)
      W4P_MATH_A(0, 1) W4P_READS(1, P, 1) W4P_MATH_B(0, 1)
      W4P_MMA2(0, 1, 0)
      W4P_FENCE()
      W4P_STORE_RAW(Q, 3) W4P_STORE_RAW(Q, 4) W4P_STORE_RAW(Q, 5)
      W4P_MATH_A(1, 1)
      __syncthreads();
      W4P_READS(0, Q, 0)
      W4P_MATH_B(1, 1)
      W4P_LOAD_RAW(c2, 0) W4P_LOAD_RAW(c2, 1) W4P_LOAD_RAW(c2, 2)
      W4P_MMA2(2, 3, 0)
      W4P_FENCE()
      W4P_LOAD_RAW(c2, 3) W4P_LOAD_RAW(c2, 4) W4P_LOAD_RAW(c2, 5)
      W4P_MATH_A(0, 0) W4P_READS(1, Q, 0) W4P_MATH_B(0, 0)
      W4P_MMA2(0, 1, 1)
      W4P_FENCE()
      W4P_LOAD_U(c1, 0) W4P_LOAD_U(c1, 1)
      W4P_MATH_A(1, 0) W4P_READS(0, Q, 1) W4P_MATH_B(1, 0)
      W4P_MMA2(2, 3, 1)
      W4P_FENCE()
      W4P_LOAD_U(c1, 2) W4P_LOAD_U(c1, 3)
#else
      W4P_STORE_RAW(Q, 0) W4P_STORE_RAW(Q, 1)
      W4P_MATH_A(0, 1) W4P_READS(1, P, 1)
      W4P_MMA(0, 0)
      W4P_FENCE()
      W4P_STORE_RAW(Q, 2) W4P_STORE_RAW(Q, 3)
      W4P_MATH_B(0, 1)
      W4P_MMA(1, 0)
      W4P_FENCE()
      W4P_STORE_RAW(Q, 4) W4P_STORE_RAW(Q, 5)
      W4P_MATH_A(1, 1)
      W4P_MMA(2, 0)
      W4P_FENCE()
      __syncthreads();          // raw(c + 1) is complete in its buffer; every read of raw(c) has returned (the waits above)
      W4P_READS(0, Q, 0)
      W4P_MATH_B(1, 1)
      W4P_LOAD_RAW(c2, 0) W4P_LOAD_RAW(c2, 1)
      W4P_MMA(3, 0)
      W4P_FENCE()
      W4P_LOAD_RAW(c2, 2) W4P_LOAD_RAW(c2, 3)
      W4P_MATH_A(0, 0) W4P_READS(1, Q, 0)
      W4P_MMA(0, 1)
      W4P_FENCE()
      W4P_LOAD_U(c1, 0)
      W4P_LOAD_RAW(c2, 4) W4P_LOAD_RAW(c2, 5)
      W4P_MATH_B(0, 0)
      W4P_MMA(1, 1)
      W4P_FENCE()
      W4P_LOAD_U(c1, 1)
      W4P_MATH_A(1, 0)
      W4P_MMA(2, 1)
      W4P_FENCE()
      W4P_LOAD_U(c1, 2)
      W4P_READS(0, Q, 1)
      W4P_MATH_B(1, 0)
      W4P_MMA(3, 1)
      W4P_FENCE()
      W4P_LOAD_U(c1, 3)
#endif
    };
    for (int c = 0; c < nslab; c += 2) {
      slab(std::integral_constant<int, 0>{}, c);
      if (c + 1 >= nslab) break;
      slab(std::integral_constant<int, 1>{}, c + 1);
    }
    W4P_WAIT()
  }
#undef W4P_RD
#undef W4P_READS
#undef W4P_WAIT
#undef W4P_MATH
#undef W4P_MATH_A
#undef W4P_MATH_B
#undef W4P_PUT
#undef W4P_MMA
#undef W4P_MMA2
#undef W4P_LOAD_U
#undef W4P_STORE_RAW
#undef W4P_LOAD_RAW
#undef W4P_FENCE
  if ((W4_ABL & 4) && p.alpha != 12345.f) return;

  // ---- Y = A^T M A.  xi columns (in registers): s0 = M0 + M1 + M2, s1 = M1 - M2 - M3;  xi rows (through LDS):
  // Y[0][.] = s(0) + s(1) + s(2), Y[1][.] = s(1) - s(2) - s(3).  One tile block per round.
  // red: [row r][ox][ct][e][lane = 32 lk + column];  finishing wave f = (oy = f >> 1, ox = f & 1), all column tiles.
  const int oy = wr >> 1, ox = wr & 1;
  const long long pix0 = (long long)img * p.H * p.W + (long long)(y0 + oy) * p.W + (x0 + ox);
  act_t* __restrict__ obase = p.C + pix0 * p.ldc;
  const act_t* __restrict__ rbase = p.res ? p.res + pix0 * p.ldr : nullptr;
  const int c4 = 4 * (lane & 7), lk2 = (lane >> 3) & 1, e_lo = lane >> 4;
  const int dx = 2 * (e_lo + 4 * lk2);
  const bool xok = x0 + ox + dx < p.W;
  const float sg1 = oy == 0 ? 1.f : -1.f;       // rows oy+1, oy+2 enter with + + (oy = 0) | - - (oy = 1)
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NCT; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float m0 = acc[0][a][b][e], m1 = acc[1][a][b][e], m2 = acc[2][a][b][e], m3 = acc[3][a][b][e];
        red[(((wr * 2 + 0) * NCT + b) * 16 + e) * 64 + lane] = (m0 + m1) + m2;
        red[(((wr * 2 + 1) * NCT + b) * 16 + e) * 64 + lane] = (m1 - m2) - m3;
      }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NCT; ++b) {
      const int n = (jn0 + b) * 32 + c4;
      const bool nok = n < p.N && jn0 + b < p.nt32;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias && nok) bv = *reinterpret_cast<const float4*>(p.bias + n);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* rd = red + ((ox * NCT + b) * 16 + 4 * i + e_lo) * 64 + lk2 * 32 + c4;
        const float4 s0 = *reinterpret_cast<const float4*>(rd + (oy + 0) * (2 * NCT * 16 * 64));
        const float4 s1 = *reinterpret_cast<const float4*>(rd + (oy + 1) * (2 * NCT * 16 * 64));
        const float4 s2 = *reinterpret_cast<const float4*>(rd + (oy + 2) * (2 * NCT * 16 * 64));
        float4 v = make_float4(s0.x + sg1 * s1.x + sg1 * s2.x, s0.y + sg1 * s1.y + sg1 * s2.y, s0.z + sg1 * s1.z + sg1 * s2.z,
                               s0.w + sg1 * s1.w + sg1 * s2.w);
        const int dy = 8 * a + 2 * i;
        if (y0 + oy + dy >= p.H || !(nok && xok)) continue;
        const int po = dy * p.W + dx;
        v = make_float4(v.x * oscale * p.alpha + bv.x, v.y * oscale * p.alpha + bv.y, v.z * oscale * p.alpha + bv.z,
                        v.w * oscale * p.alpha + bv.w);
        act_t* __restrict__ op = obase + po * (int)p.ldc + n;
        if (rbase) { const float4 r4 = osm::ld4(rbase + po * (int)p.ldr + n); v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
        if (p.accumulate) { const float4 a4 = osm::ld4(op); v.x += a4.x; v.y += a4.y; v.z += a4.z; v.w += a4.w; }
        osm::st4(op, v);
      }
    }
  }
}
