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
//   SOFTWARE-PIPELINED single loop: every wave interleaves the MFMAs of chunk k (fragments from LDS buffer
//       k&1) with the split + LDS store of chunk k+1 (into buffer (k+1)&1) and the global prefetches of
//       chunks k+2 / k+3, one barrier per chunk.  (Measured with s_memtime stamps: separate store / MFMA
//       phases -- whether in lockstep or ping-ponged between two wave groups -- serialise on the in-order
//       waves: 48 % MFMA busy.  An MFMA leaves ~5 issue slots per 32-cycle gap; the ~140 VALU + 40 memory
//       instructions of the staging work fit in the 48 gaps of a chunk.)

// (split helpers, fragment types and mma16<NP>: mfma_split.h)
constexpr int ACT_B = (int)sizeof(act_t);   // bytes per stored activation (4, or 2 in the OSM_ACT_F16 family)
constexpr int S_ROWB = 80;              // bytes per staged A row (32 bf16 + 16 B pad)
constexpr int S_PLANE = 128 * S_ROWB;   // bytes per 128-row plane

// 32 B of zeros every masked staging load is redirected to (halo / ragged edge / channel tail).
__device__ uint4 g_zero_page[2];   // zero-initialised device global (never written)

// HP ("f16x3", NP = 2, 1x1 layers): the two planes are IEEE halves of x * 2^ex (2^ex from the per-image max |x| the caller
// supplies, IGemmParams::xmax, brings it to [2^13, 2^14)); the weight image holds halves of w * 2^ew behind a scale word; three
// fp16 MFMAs per product, the exact power-of-two rescale in the epilogue.  A tile must not straddle two images.
template <int TAPS, int NP, bool HP = false>
__global__ __launch_bounds__(256, 2) void igemm_bf16s_kernel(const act_t* __restrict__ Aglob,
                                                              const unsigned short* __restrict__ Bglob,
                                                              IGemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * NP * S_PLANE];   // two A stages
  unsigned char* As = smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  const int nt = p.mtiles * p.ntiles;
  const int bid = blockIdx.x;
  const int q = nt >> 3, r = nt & 7, xcd = bid & 7, idx = bid >> 3;
  const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  const int tile_n = id % p.ntiles, tile_m = id / p.ntiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const act_t* __restrict__ A = Aglob;
  const unsigned short* __restrict__ Bm = Bglob;

  const int ks = blockIdx.y;
  const int per = (p.nchunks + p.splitk - 1) / p.splitk;
  const int kc0 = ks * per;
  const int kc1 = min(p.nchunks, kc0 + per);

  float xscale = 1.f, oscale = 1.f;
  if (HP) {
    static_assert(!HP || NP == 2, "f16x3: two half planes");
    static_assert(OSM_MAXABS_PARTS == 1024, "four partial maxima per thread");
    const unsigned* xm = reinterpret_cast<const unsigned*>(p.xmax) + (long long)(m0 / (p.H * p.W)) * OSM_MAXABS_PARTS;
    unsigned mb = max(max(xm[tid], xm[tid + 256]), max(xm[tid + 512], xm[tid + 768]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mb = max(mb, (unsigned)__shfl_xor((int)mb, o, 64));
    unsigned* red_u = reinterpret_cast<unsigned*>(smem);
    if (lane == 0) red_u[wave] = mb;
    __syncthreads();
    mb = max(max(red_u[0], red_u[1]), max(red_u[2], red_u[3]));
    __syncthreads();              // the staging stores that follow reuse smem
    const float mx = __uint_as_float(mb);
    int ex = 0;
    if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &ex); ex = min(14 - ex, 100); }   // (denormal maxima: 2^ex stays finite)
    xscale = mx == mx ? ldexpf(1.f, ex) : mx;       // a NaN in the input poisons the output
    oscale = ldexpf(1.f, -ex) / p.wscale[0];
  }

  // ---- A staging coordinates: float4 column group cg of rows r0 + 32 i.
  // Addressing is "uniform 64-bit base (SGPR) + per-lane 32-bit byte offset": the base points (W+1) pixels
  // before the tile so that every tap offset is non-negative; per chunk only ONE scalar delta changes.
  // Masked lanes (halo / ragged edge / channel tail) read their own centre pixel (always valid memory)
  // and are zeroed after the load -- ~9 VALU per load instead of ~30 of 64-bit pointer arithmetic.
  const int cg = tid & 7, r0 = tid >> 3;
  const long long rowB = (long long)p.lda * ACT_B;                     // bytes per pixel row
  const long long biasB = (TAPS == 9) ? (long long)(p.W + 1) * rowB : 0;
  const char* __restrict__ sbaseA = reinterpret_cast<const char*>(A) + (long long)m0 * rowB - biasB;
  unsigned vcen[4], vsafe[4], amask[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + r0 + 32 * i;
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
    vsafe[i] = (unsigned)(biasB + (m < p.M ? (long long)(r0 + 32 * i) * rowB : 0));
    vcen[i] = vsafe[i] + (unsigned)(4 * ACT_B) * cg;
  }

  // ---- B fragment addressing: image [plane][tap][k16-step][n/32][lane][8]; wave-uniform base + lane offset
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int jn = (n0 >> 5) + wave_u;               // this wave's 32-column tile
  const bool b_ok = jn < p.nt32;
  const unsigned b_lane = b_ok ? (unsigned)((jn * 64 + lane) * 16) : (unsigned)(lane & 1) * 16u;
  const long long b_step = (long long)p.nt32 * 64 * 16;               // bytes per k16-step
  const long long b_tap = b_step * p.ksteps;                          // bytes per tap
  const long long b_plane = b_tap * TAPS;                             // bytes per plane
  const char* __restrict__ sbaseB0 = b_ok ? reinterpret_cast<const char*>(Bm) : reinterpret_cast<const char*>(g_zero_page);

  // Two register sets per operand: every global load is issued >= one full chunk before it is consumed,
  // and every prefetch is UNCONDITIONAL (tail chunks re-load the last chunk) so that the compiler's
  // vmcnt bookkeeping is static -- with conditional loads it has to assume the shortest path and ends
  // up waiting for the newest loads inside the MFMA stream.
  float4 raE[4], raO[4];
  unsigned okE = 0, okO = 0;                  // validity bits of the staged A registers
  uint4 bE00, bE01, bE02, bE10, bE11, bE12;   // [k16-step][plane] B fragments, even chunks
  uint4 bO00, bO01, bO02, bO10, bO11, bO12;   // odd chunks
  bE02 = bE12 = bO02 = bO12 = bE01 = bE11 = bO01 = bO11 = make_uint4(0u, 0u, 0u, 0u);

#define OSM_S_LOAD_A(ra_, ok_, kc_)                                                                  \
  {                                                                                                  \
    const int cc_ = (kc_) / TAPS;                                                                    \
    const int tap_ = (kc_) - cc_ * TAPS;                                                             \
    int sdelta_ = cc_ * (BK * ACT_B);                                                                \
    if (TAPS == 9) sdelta_ += (int)(((tap_ / 3 - 1) * p.W + (tap_ % 3 - 1)) * rowB);                 \
    const bool cok_ = cc_ * BK + 4 * cg < p.K;                                                       \
    ok_ = 0;                                                                                         \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                  \
      const bool o_ = cok_ && ((amask[i] >> tap_) & 1u);                                             \
      ok_ |= (o_ ? 1u : 0u) << i;                                                                    \
      const unsigned vo_ = o_ ? vcen[i] + (unsigned)sdelta_ : vsafe[i];                              \
      ra_[i] = osm::ld4(reinterpret_cast<const act_t*>(sbaseA + vo_));                               \
    }                                                                                                \
  }
#define OSM_S_LOAD_B(b_, kc_)                                                                        \
  {                                                                                                  \
    const int cc_ = (kc_) / TAPS;                                                                    \
    const int tap_ = (kc_) - cc_ * TAPS;                                                             \
    const char* sb_ = sbaseB0 + (b_ok ? tap_ * b_tap + (2 * cc_) * b_step : 0);                      \
    const long long st_ = b_ok ? b_step : 0;                                                         \
    const long long pl_ = b_ok ? b_plane : 0;                                                        \
    b_##00 = *reinterpret_cast<const uint4*>(sb_ + b_lane);                                          \
    b_##10 = *reinterpret_cast<const uint4*>(sb_ + st_ + b_lane);                                    \
    if (NP >= 2) {                                                                                   \
      b_##01 = *reinterpret_cast<const uint4*>(sb_ + pl_ + b_lane);                                  \
      b_##11 = *reinterpret_cast<const uint4*>(sb_ + pl_ + st_ + b_lane);                            \
    }                                                                                                \
    if (NP == 3) {                                                                                   \
      b_##02 = *reinterpret_cast<const uint4*>(sb_ + 2 * pl_ + b_lane);                              \
      b_##12 = *reinterpret_cast<const uint4*>(sb_ + 2 * pl_ + st_ + b_lane);                        \
    }                                                                                                \
  }
// fragment reads of one quarter (2 tiles x NP planes) from A stage `buf_`
#define OSM_S_READ(f_, buf_, st_, half_)                                                             \
  _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                      \
    _Pragma("unroll") for (int q2 = 0; q2 < NP; ++q2)                                                \
      f_[t][q2] = *reinterpret_cast<const uint4*>(a_rd + (buf_) * (NP * S_PLANE) + q2 * S_PLANE + \
                                                     (64 * (half_) + 32 * t) * S_ROWB + 32 * (st_));
#define OSM_S_MMA(f_, b_, st_, half_)                                                                \
  {                                                                                                  \
    uint4 bf[3];                                                                                     \
    bf[0] = (st_) ? b_##10 : b_##00;                                                                 \
    bf[1] = (st_) ? b_##11 : b_##01;                                                                 \
    bf[2] = (st_) ? b_##12 : b_##02;                                                                 \
    _Pragma("unroll") for (int pa = NP - 1; pa >= 0; --pa)                                           \
      _Pragma("unroll") for (int pb = NP - 1 - pa; pb >= 0; --pb) {                                  \
        acc[2 * (half_)] = HP ? mma16h(f_[0][pa], bf[pb], acc[2 * (half_)])                          \
                              : mma16<NP>(f_[0][pa], bf[pb], acc[2 * (half_)]);                      \
        acc[2 * (half_) + 1] = HP ? mma16h(f_[1][pa], bf[pb], acc[2 * (half_) + 1])                  \
                                  : mma16<NP>(f_[1][pa], bf[pb], acc[2 * (half_) + 1]);              \
      }                                                                                              \
  }
// split one staged float4 (rows r0 + 32 i) into NP planes of A stage `buf_`
#define OSM_S_SPLIT(ra_, ok_, i_, buf_)                                                              \
  {                                                                                                  \
    uint2 pl[NP];                                                                                    \
    if constexpr (HP) {                                                                              \
      const float4 v_ = sel4(((ok_) >> (i_)) & 1u, ra_[i_]);                                         \
      uint2 ph_[2];                                                                                  \
      split_f16x2(make_float4(v_.x * xscale, v_.y * xscale, v_.z * xscale, v_.w * xscale), ph_);     \
      pl[0] = ph_[0]; pl[NP - 1] = ph_[1];                                                           \
    } else {                                                                                         \
      split_planes<NP>(sel4(((ok_) >> (i_)) & 1u, ra_[i_]), pl);                                     \
    }                                                                                                \
    _Pragma("unroll") for (int q2 = 0; q2 < NP; ++q2)                                                \
      *reinterpret_cast<uint2*>(As + (buf_) * (NP * S_PLANE) + q2 * S_PLANE + (r0 + 32 * (i_)) * S_ROWB + \
                                8 * cg) = pl[q2];                                                    \
  }
// one chunk `it_`: MFMAs on stage it&1 with B set bc_, interleaved with the split of A(it+1) (register set
// ra_) into stage (it+1)&1; prefetch B(it+1) -> bn_ and, once ra_ is consumed, A(it+3) -> ra_.
#define OSM_S_CHUNK(ra_, ok_, bc_, bn_, it_)                                                              \
  {                                                                                                  \
    const int rb_ = (it_) & 1, wb_ = rb_ ^ 1;                                                        \
    uint4 fx[2][NP], fy[2][NP];                                                                      \
    OSM_S_LOAD_B(bn_, min(kc0 + (it_) + 1, kc1 - 1));                                                \
    OSM_S_READ(fx, rb_, 0, 0)                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    OSM_S_READ(fy, rb_, 0, 1)                                                                        \
    OSM_S_MMA(fx, bc_, 0, 0)                                                                         \
    OSM_S_SPLIT(ra_, ok_, 0, wb_)                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    OSM_S_READ(fx, rb_, 1, 0)                                                                        \
    OSM_S_MMA(fy, bc_, 0, 1)                                                                         \
    OSM_S_SPLIT(ra_, ok_, 1, wb_)                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    OSM_S_READ(fy, rb_, 1, 1)                                                                        \
    OSM_S_MMA(fx, bc_, 1, 0)                                                                         \
    OSM_S_SPLIT(ra_, ok_, 2, wb_)                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                               \
    OSM_S_MMA(fy, bc_, 1, 1)                                                                         \
    OSM_S_SPLIT(ra_, ok_, 3, wb_)                                                                         \
    OSM_S_LOAD_A(ra_, ok_, min(kc0 + (it_) + 3, kc1 - 1));                                                \
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
    OSM_S_LOAD_A(raE, okE, kc0);
    OSM_S_LOAD_B(bE, kc0);
    OSM_S_LOAD_A(raO, okO, min(kc0 + 1, kc1 - 1));
    // stage chunk 0 (not overlapped), then refill its register set two chunks ahead
#pragma unroll
    for (int i = 0; i < 4; ++i) OSM_S_SPLIT(raE, okE, i, 0)
    OSM_S_LOAD_A(raE, okE, min(kc0 + 2, kc1 - 1));
    __syncthreads();
    for (int it = 0; it < nk; it += 2) {
      OSM_S_CHUNK(raO, okO, bE, bO, it);
      if (it + 1 < nk) OSM_S_CHUNK(raE, okE, bO, bE, it + 1);
    }
  }
#undef OSM_S_LOAD_A
#undef OSM_S_LOAD_B
#undef OSM_S_CHUNK
#undef OSM_S_READ
#undef OSM_S_MMA
#undef OSM_S_SPLIT

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
  const bool partial = p.splitk > 1;
  float* Wb = p.ws + ((long long)ks * p.M) * p.N;       // fp32 partials [ks][M][N]
  // Stores are ISSUE-bound (a store instruction costs the same whether a lane writes 4 or 16 bytes, and the 64 four-byte
  // stores of this layout were two thirds of a 256-channel 1x1 convolution): each wave transposes its 128 x 32 tile
  // through its own slice of the (now idle) staging LDS, RP rows at a time, so that a lane owns 4 consecutive columns
  // of a row: 16 x (ds_read_b128 + one 16-byte store) instead of 64 scalar stores.  LDS executes a wave's operations in
  // order, so no barrier is needed around the wave-private slice.
  const auto al = [](const void* q, unsigned bytes) { return (reinterpret_cast<unsigned long long>(q) & (bytes - 1)) == 0; };
  const bool vec = (p.N & 3) == 0 &&
                   (partial ? al(p.ws, 16)
                            : ((p.ldc & 3) == 0 && al(p.C, 4 * ACT_B) && (!p.bias || al(p.bias, 16)) &&
                               (!p.res || ((p.ldr & 3) == 0 && al(p.res, 4 * ACT_B)))));
  if (vec) {
    constexpr int RP = NP == 1 ? 32 : 64;               // rows per pass (the fp16 family's staging LDS is 20 KB)
    float* tb = reinterpret_cast<float*>(smem) + wave * (RP * 32);
    const int c4 = 4 * (lane & 7), rr = lane >> 3;
    const int nb = n0 + 32 * wave + c4;
    const bool nok = nb < p.N;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!partial && p.bias && nok) bv = *reinterpret_cast<const float4*>(p.bias + nb);
#pragma unroll
    for (int pass = 0; pass < 128 / RP; ++pass) {
#pragma unroll
      for (int t = 0; t < RP / 32; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          tb[(32 * t + (e & 3) + 8 * (e >> 2) + 4 * lk) * 32 + lr] = acc[pass * (RP / 32) + t][e];
#pragma unroll
      for (int it = 0; it < RP / 8; ++it) {
        const int row = 8 * it + rr;
        float4 v = *reinterpret_cast<const float4*>(tb + row * 32 + c4);
        if (HP) { v.x *= oscale; v.y *= oscale; v.z *= oscale; v.w *= oscale; }
        const int m = m0 + pass * RP + row;
        if (m >= p.M || !nok) continue;
        if (partial) {
          *reinterpret_cast<float4*>(Wb + (long long)m * p.N + nb) = v;
        } else {
          act_t* cp = p.C + (long long)m * p.ldc + nb;
          v = make_float4(v.x * p.alpha + bv.x, v.y * p.alpha + bv.y, v.z * p.alpha + bv.z, v.w * p.alpha + bv.w);
          if (p.res) {
            const float4 r4 = osm::ld4(p.res + (long long)m * p.ldr + nb);
            v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
          }
          if (p.accumulate) {
            const float4 r4 = osm::ld4(cp);
            v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
          }
          osm::st4(cp, v);
        }
      }
    }
    return;
  }
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
      if (HP) v *= oscale;
      if (partial) {
        Wb[(long long)m * p.N + n] = v;
      } else {
        v = v * p.alpha + bv;
        if (p.res) v += osm::ld1(p.res + (long long)m * p.ldr + n);
        if (p.accumulate) v += osm::ld1(p.C + (long long)m * p.ldc + n);
        osm::st1(p.C + (long long)m * p.ldc + n, v);
      }
    }
  }
}

// OIHW fp32 -> NP bf16 planes in MFMA-fragment order
//   [plane][tap][k16-step s][n/32 j][lane l][e],  n = 32 j + (l & 31),  k = 16 s + 8 (l >> 5) + e
// forward: n = Cout, k = Cin ; data-gradient: n = Cin, k = Cout, taps flipped.  Out-of-range (n, k) are zero;
// the step count is even (2 per 32-wide chunk) so a chunk never reads past the image.
#ifndef OSM_ACT_F16
// f16x3 image, pass 1: max |w| as the bit pattern of a non-negative float (atomicMax on uint; pack time only)
__global__ void wmax_kernel(const float* __restrict__ w, unsigned* __restrict__ wmax, long long total) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(w[i]));
  m = osm::wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(wmax, __float_as_uint(m));
}
__global__ void pack_weight_bf16s_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout,
                                         int Cin, int k, int np, int dgrad) {
  const int N = dgrad ? Cin : Cout;
  const int K = dgrad ? Cout : Cin;
  const int nt32 = (N + 31) / 32;
  const int ksteps = 2 * ((K + 31) / 32);
  const long long per_plane = (long long)k * k * ksteps * nt32 * 512;
  // np = 4 (f16x3): 2^ew = the power of two that brings max |w| (left behind the planes by wmax_kernel, copied to word [1] by
  // wino_scale_word_kernel stage 0) to [2^13, 2^14); stage 1 of that kernel rewrites word [0] as the scale
  float wsc = 1.f;
  if (np == 4) {
    const float mx = __uint_as_float(*reinterpret_cast<const unsigned*>(out + 2 * per_plane + 2));
    int e2 = 0;
    if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &e2); e2 = min(14 - e2, 100); }
    wsc = ldexpf(1.f, e2);
  }
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
    if (np == 1) {      // wfmt 1: ONE plane of IEEE half (RNE) for the fp16-arithmetic family
      out[i] = __builtin_bit_cast(unsigned short, (_Float16)v);
      continue;
    }
    if (np == 4) {      // wfmt 4 (f16x3): two IEEE-half planes of w * 2^ew
      const float rs = v * wsc;
      const _Float16 h0 = (_Float16)rs;
      out[i] = __builtin_bit_cast(unsigned short, h0);
      out[per_plane + i] = __builtin_bit_cast(unsigned short, (_Float16)(rs - (float)h0));
      continue;
    }
    float rr = v;
    for (int qq = 0; qq < np; ++qq) {
      const __bf16 b = (__bf16)rr;
      out[qq * per_plane + i] = __builtin_bit_cast(unsigned short, b);
      rr -= (float)b;
    }
  }
}
#endif   // !OSM_ACT_F16
