// Cooperative single-read GroupNorm (round 4).  Included inside norm.hip's anonymous namespace.
//
// The chunked path reads every operand twice (statistics pass, apply pass: 1.6x / 1.9x the minimal HBM bytes, forward / backward)
// and takes three launches.  Here ONE launch does both: the tensor is split into `nwg` row chunks per image, every workgroup
// (512 threads, two waves per SIMD) loads its chunk -- x, and dy in the backward -- into REGISTERS with all loads in flight,
// reduces its 32 per-group partial sums, publishes them, collects everybody's, and applies from the registers.  Only
// 512 bytes per workgroup cross the chip.
//
// Cross-workgroup exchange without fences or one-address atomics (DESIGN "learned" 5, 28, 31: agent-scope fences are L2
// write-backs, same-address atomics serialise at ~28 ns each): a partial is published as ONE 8-byte relaxed agent-scope
// atomic store of (fp32 bits | tag << 32) into its own slot ws[b][g][w][2]; readers poll the slots with relaxed agent-scope
// loads until the tag is the one of this launch.  Every datum carries its own tag, so no ordering between different
// addresses is needed.  The tag of a launch is (tag found in the workgroup's OWN slot) + 1: a workspace is zero-initialised
// once and then always used with the same (B, nwg) -- one workspace per call site -- so all slots advance in lockstep and
// nothing has to be reset or handed over by the host (graph-replayable).
//
// Spin-waiting needs every workgroup of the grid resident: the host-side plan only picks grids that fit the device
// (occupancy x CUs).  That guarantee ends where several processes share a GPU (two spinning grids can starve each other), so
// the wait is BOUNDED: after `timeout` ticks of the 100 MHz wall clock a workgroup stops waiting, computes the partial sums
// that are still missing itself (streaming those chunks from HBM with the same per-thread arithmetic, hence the same bits),
// combines in the same order as the fast path, applies its chunk by re-reading it, and exits -- which frees its CU for
// whoever had not started.  Results are bit-identical whichever path a workgroup takes (tests force both).
constexpr int COOP_T = 512;       // threads per workgroup
constexpr int COOP_G = 32;        // groups (GroupNorm32)
constexpr int COOP_LANES = COOP_T / COOP_G;   // 16 gather lanes per group
constexpr int COOP_MAXW = 256;    // workgroups per image: 16 gather items per lane, in two batches of 8
constexpr int COOP_BATCH = 8;

struct CoopArgs {
  GNArgs a;
  unsigned long long* ws;   // [B][G][nwg][2]
  int nwg;                  // workgroups (row chunks) per image
  int rowT;                 // rows per sweep of the 512 threads = COOP_T / (C / 4)
  long long timeout;        // wall_clock64 ticks (100 MHz) a workgroup waits for the others before it helps itself
};

__device__ __forceinline__ unsigned long long coop_pack(float v, unsigned tag) {
  return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}
__device__ __forceinline__ unsigned long long coop_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void coop_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// per-thread constants of the channel vector a thread owns
struct CoopVec {
  float ga[4], be[4], sc[4], sh[4];
  float mean, rstd;     // forward statistics (MODE 1: input; MODE 0: filled after the exchange)
  int c, g;
  bool film;
};

// One thread's share of the partial sums of row chunk [p0, p0 + NV * rowT) of image b: rows p0 + tr + k * rowT, channel vector c.
// KEEP: the loaded vectors stay in xr (and, MODE 1, dxh replaces dy in dr) for the apply phase.  The arithmetic and its order
// are the same with and without KEEP: a helper that recomputes another workgroup's partial gets that workgroup's bits.
template <int MODE, int NV, bool KEEP>
__device__ __forceinline__ void coop_partial(const GNArgs& a, const CoopVec& cv, int b, int p0, int tr, int rowT, bool live,
                                             float4 (&xr)[NV], float4 (&dr)[MODE == 1 ? NV : 1], float& s1, float& s2,
                                             float& imax, unsigned& inan) {
  const long long row0 = (long long)b * a.HW;
  s1 = s2 = 0.f;
  float4 xl[NV], dl[MODE == 1 ? NV : 1];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int p = p0 + tr + k * rowT;
    const bool ok = live && p < a.HW;
    const long long row = row0 + (ok ? p : 0);
    const float4 t = osm::ld4(a.x + row * a.ldx + cv.c);
    xl[k] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == 1) {
      const float4 u = osm::ld4(a.dy + row * a.lddy + cv.c);
      dl[k] = ok ? u : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int p = p0 + tr + k * rowT;
    const bool ok = live && p < a.HW;
    const float xv[4] = {xl[k].x, xl[k].y, xl[k].z, xl[k].w};
    if (MODE == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s1 += xv[e];
        s2 = fmaf(xv[e], xv[e], s2);
        imax = fmaxf(imax, fabsf(xv[e]));
        if (xv[e] != xv[e]) inan = 0x7fc00000u;
      }
    } else {
      const float dv[4] = {dl[k].x, dl[k].y, dl[k].z, dl[k].w};
      float dxh[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float xh, z;
        gn_fwd_elem(xv[e], cv.mean, cv.rstd, cv.ga[e], cv.be[e], cv.film, cv.sc[e], cv.sh[e], xh, z);
        float dz = dv[e] * (a.silu ? osm::dsilu_f(z) : 1.0f);
        if (cv.film) dz *= (1.0f + cv.sc[e]);
        dxh[e] = ok ? dz * cv.ga[e] : 0.f;       // rows beyond the tensor contribute nothing (x = 0 would still give xh != 0)
        s1 += dxh[e];
        s2 = fmaf(dxh[e], xh, s2);
      }
      if (KEEP) dr[k] = make_float4(dxh[0], dxh[1], dxh[2], dxh[3]);
    }
    if (KEEP) xr[k] = xl[k];
  }
}

// fold the 512 per-thread sums into the 32 groups (fixed order); thread g < 32 returns group g's pair
__device__ __forceinline__ void coop_fold(float* red, int tid, int vpr, int rowT, int gs, float s1, float s2, float& t1, float& t2) {
  __syncthreads();          // red may still be read by a previous fold
  red[2 * tid] = s1;
  red[2 * tid + 1] = s2;
  __syncthreads();
  t1 = t2 = 0.f;
  if (tid < COOP_G) {
    const int v0 = tid * (gs >> 2), v1 = (tid + 1) * (gs >> 2);
    for (int v = v0; v < v1; ++v)
      for (int r = 0; r < rowT; ++r) {
        const int t = r * vpr + v;
        t1 += red[2 * t];
        t2 += red[2 * t + 1];
      }
  }
}

// the two statistics of a group from the fp64 sums (identical code on both paths)
template <int MODE>
__device__ __forceinline__ void coop_stats(double d1, double d2, double n, float eps, float& o0, float& o1) {
  if (MODE == 0) {
    const double mu = d1 / n;
    double var = d2 / n - mu * mu;
    if (var < 0.0) var = 0.0;
    o0 = (float)mu;
    o1 = (float)(1.0 / sqrt(var + (double)eps));
  } else {
    o0 = (float)(d1 / n);
    o1 = (float)(d2 / n);
  }
}

// butterfly over the 16 gather lanes of a group (lanes of one group are 16 consecutive lanes of a wave)
__device__ __forceinline__ void coop_lanes_sum(double& d1, double& d2) {
#pragma unroll
  for (int o = COOP_LANES / 2; o > 0; o >>= 1) {
    d1 += __shfl_xor(d1, o, 64);
    d2 += __shfl_xor(d2, o, 64);
  }
}

template <int MODE>
__device__ __forceinline__ void coop_apply_vec(const GNArgs& a, const CoopVec& cv, float q0, float q1, const float4& xq,
                                               const float4& dq, const float (&av)[4], bool has_add, float (&ov)[4]) {
  const float xv[4] = {xq.x, xq.y, xq.z, xq.w};
  if (MODE == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float xh, z;
      gn_fwd_elem(xv[e], q0, q1, cv.ga[e], cv.be[e], cv.film, cv.sc[e], cv.sh[e], xh, z);
      ov[e] = a.silu ? osm::silu_f(z) : z;
    }
  } else {
    const float dxv[4] = {dq.x, dq.y, dq.z, dq.w};      // dxh
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (xv[e] - cv.mean) * cv.rstd;
      float r = cv.rstd * (dxv[e] - q0 - xh * q1);
      if (has_add) r += av[e];
      ov[e] = r;
    }
  }
}

template <int MODE, int NV>
__global__ __launch_bounds__(COOP_T) void gn_coop_kernel(CoopArgs ca) {
  const GNArgs& a = ca.a;
  __shared__ float red[2 * COOP_T];
  __shared__ float sst[2 * COOP_G];
  __shared__ float pp[2 * COOP_G];
  __shared__ unsigned s_tag, s_slow, s_miss;
  const int tid = threadIdx.x;
  const int w = blockIdx.x, b = blockIdx.y;
  const int nwg = ca.nwg, rowT = ca.rowT;
  const int vpr = a.C >> 2;
  const int tc = tid % vpr, tr = tid / vpr;
  const bool live = tr < rowT;
  const int p0 = w * (NV * rowT);

  CoopVec cv;
  cv.c = 4 * tc;
  cv.g = cv.c / a.gs;
  cv.film = a.film != nullptr;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    cv.ga[e] = a.gamma[cv.c + e];
    cv.be[e] = a.beta[cv.c + e];
    cv.sc[e] = cv.film ? a.film[(long long)b * a.ldf + cv.c + e] : 0.f;
    cv.sh[e] = cv.film ? a.film[(long long)b * a.ldf + a.C + cv.c + e] : 0.f;
  }
  cv.mean = cv.rstd = 0.f;
  if (MODE == 1) {
    cv.mean = a.stats[(b * a.G + cv.g) * 2];
    cv.rstd = a.stats[(b * a.G + cv.g) * 2 + 1];
  }
  unsigned long long* const wsb = ca.ws + (long long)b * COOP_G * nwg * 2;
  if (tid == 0) {
    s_tag = (unsigned)(coop_load(wsb + (long long)w * 2) >> 32) + 1u;     // own slot of group 0: all slots move in lockstep
    s_slow = 0u;
  }

  // ---- phase 1: own chunk into registers, partial sums, publish
  float4 xr[NV], dr[MODE == 1 ? NV : 1];
  float s1, s2, imax = 0.f;
  unsigned inan = 0u;
  coop_partial<MODE, NV, true>(a, cv, b, p0, tr, rowT, live, xr, dr, s1, s2, imax, inan);
  float t1, t2;
  coop_fold(red, tid, vpr, rowT, a.gs, s1, s2, t1, t2);       // (two barriers inside: s_tag is visible after them)
  const unsigned tag = s_tag;
  if (tid < COOP_G) {
    unsigned long long* sl = wsb + ((long long)tid * nwg + w) * 2;
    coop_store(sl, coop_pack(t1, tag));
    coop_store(sl + 1, coop_pack(t2, tag));
  }

  // ---- phase 2: collect everybody's partials: thread (g, l) takes workgroups l, l + 16, ... of group g
  const int g2 = tid / COOP_LANES, l2 = tid % COOP_LANES;
  const unsigned long long* const gl = wsb + (long long)g2 * nwg * 2;
  double d1 = 0.0, d2 = 0.0;
  bool timed_out = false;
  const long long t_start = wall_clock64();
  for (int base = 0; base < nwg && !timed_out; base += COOP_LANES * COOP_BATCH) {
    float v1[COOP_BATCH], v2[COOP_BATCH];
    unsigned pend = 0u;
#pragma unroll
    for (int i = 0; i < COOP_BATCH; ++i)
      if (base + l2 + COOP_LANES * i < nwg) pend |= 1u << i;
    while (pend) {
      unsigned long long q1[COOP_BATCH], q2[COOP_BATCH];
#pragma unroll
      for (int i = 0; i < COOP_BATCH; ++i)
        if (pend >> i & 1u) {
          const unsigned long long* s = gl + (long long)(base + l2 + COOP_LANES * i) * 2;
          q1[i] = coop_load(s);
          q2[i] = coop_load(s + 1);
        }
#pragma unroll
      for (int i = 0; i < COOP_BATCH; ++i)
        if ((pend >> i & 1u) && (unsigned)(q1[i] >> 32) == tag && (unsigned)(q2[i] >> 32) == tag) {
          v1[i] = __uint_as_float((unsigned)q1[i]);
          v2[i] = __uint_as_float((unsigned)q2[i]);
          pend &= ~(1u << i);
        }
      if (pend) {
        if (wall_clock64() - t_start > ca.timeout) {
          timed_out = true;
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
    }
    if (!timed_out) {
#pragma unroll
      for (int i = 0; i < COOP_BATCH; ++i)
        if (base + l2 + COOP_LANES * i < nwg) {
          d1 += (double)v1[i];
          d2 += (double)v2[i];
        }
    }
  }
  if (timed_out) s_slow = 1u;
  __syncthreads();
  const bool slow = s_slow != 0u;

  if (!slow) {
    // ---- phase 3 (fast): statistics, apply from the registers
    coop_lanes_sum(d1, d2);
    if (l2 == 0) {
      float o0, o1;
      coop_stats<MODE>(d1, d2, a.n, a.eps, o0, o1);
      sst[2 * g2] = o0;
      sst[2 * g2 + 1] = o1;
      if (w == 0) {
        a.fin[(b * a.G + g2) * 2] = o0;
        a.fin[(b * a.G + g2) * 2 + 1] = o1;
      }
    }
    __syncthreads();
    const float q0 = sst[2 * cv.g], q1 = sst[2 * cv.g + 1];
    float omax = 0.f;
    unsigned onan = 0u;
    const long long row0 = (long long)b * a.HW;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int p = p0 + tr + k * rowT;
      if (!live || p >= a.HW) continue;
      float av[4] = {0.f, 0.f, 0.f, 0.f};
      bool has_add = false;
      if (MODE == 1) {
        if (a.addend) {
          const float4 t = osm::ld4(a.addend + (row0 + p) * a.ldadd + cv.c);
          av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w;
          has_add = true;
        }
        if (a.addend2) {
          const float4 t = osm::ld4(a.addend2 + (row0 + p) * a.ldadd2 + cv.c);
          av[0] += t.x; av[1] += t.y; av[2] += t.z; av[3] += t.w;
          has_add = true;
        }
      }
      float ov[4];
      coop_apply_vec<MODE>(a, cv, q0, q1, xr[k], dr[MODE == 1 ? k : 0], av, has_add, ov);
      if (a.maxabs) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sv = (float)(act_t)ov[e];
          omax = fmaxf(omax, fabsf(sv));
          if (sv != sv) onan = 0x7fc00000u;
        }
      }
      osm::st4(a.out + (row0 + p) * a.ldo + cv.c, make_float4(ov[0], ov[1], ov[2], ov[3]));
    }
    if (a.maxabs) gn_publish_max(a.maxabs, b, w, nwg, omax, onan);
    if (MODE == 0 && a.maxabs_in) {
      __syncthreads();
      gn_publish_max(a.maxabs_in, b, w, nwg, imax, inan);
    }
    return;
  }

  // ---- slow path (the others did not show up in time): every missing partial is recomputed here, in workgroup order, and
  // added into the same lane accumulators, in the same order, as the fast path adds them
  d1 = d2 = 0.0;
  for (int w2 = 0; w2 < nwg; ++w2) {
    unsigned long long q1 = 0ull, q2 = 0ull;
    if (tid == 0) s_miss = 0u;
    __syncthreads();
    if (tid < COOP_G) {
      const unsigned long long* s = wsb + ((long long)tid * nwg + w2) * 2;
      q1 = coop_load(s);
      q2 = coop_load(s + 1);
      if ((unsigned)(q1 >> 32) != tag || (unsigned)(q2 >> 32) != tag) s_miss = 1u;
    }
    __syncthreads();
    float u1 = __uint_as_float((unsigned)q1), u2 = __uint_as_float((unsigned)q2);
    if (s_miss != 0u) {           // uniform: recompute chunk w2 as its own workgroup would
      float4 xd[NV], dd[MODE == 1 ? NV : 1];
      float r1, r2, jm = 0.f;
      unsigned jn = 0u;
      coop_partial<MODE, NV, false>(a, cv, b, w2 * (NV * rowT), tr, rowT, live, xd, dd, r1, r2, jm, jn);
      coop_fold(red, tid, vpr, rowT, a.gs, r1, r2, u1, u2);
    }
    if (tid < COOP_G) {
      pp[2 * tid] = u1;
      pp[2 * tid + 1] = u2;
    }
    __syncthreads();
    if (l2 == (w2 % COOP_LANES)) {
      d1 += (double)pp[2 * g2];
      d2 += (double)pp[2 * g2 + 1];
    }
  }
  coop_lanes_sum(d1, d2);
  if (l2 == 0) {
    float o0, o1;
    coop_stats<MODE>(d1, d2, a.n, a.eps, o0, o1);
    sst[2 * g2] = o0;
    sst[2 * g2 + 1] = o1;
    if (w == 0) {
      a.fin[(b * a.G + g2) * 2] = o0;
      a.fin[(b * a.G + g2) * 2 + 1] = o1;
    }
  }
  __syncthreads();
  {
    const float q0 = sst[2 * cv.g], q1 = sst[2 * cv.g + 1];
    float omax = 0.f;
    unsigned onan = 0u;
    const long long row0 = (long long)b * a.HW;
    for (int k = 0; k < NV; ++k) {
      const int p = p0 + tr + k * rowT;
      if (!live || p >= a.HW) continue;
      const float4 xq = osm::ld4(a.x + (row0 + p) * a.ldx + cv.c);
      float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
      float av[4] = {0.f, 0.f, 0.f, 0.f};
      bool has_add = false;
      if (MODE == 1) {
        const float4 u = osm::ld4(a.dy + (row0 + p) * a.lddy + cv.c);
        const float xv[4] = {xq.x, xq.y, xq.z, xq.w}, dv[4] = {u.x, u.y, u.z, u.w};
        float dxh[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float xh, z;
          gn_fwd_elem(xv[e], cv.mean, cv.rstd, cv.ga[e], cv.be[e], cv.film, cv.sc[e], cv.sh[e], xh, z);
          float dz = dv[e] * (a.silu ? osm::dsilu_f(z) : 1.0f);
          if (cv.film) dz *= (1.0f + cv.sc[e]);
          dxh[e] = dz * cv.ga[e];
        }
        dq = make_float4(dxh[0], dxh[1], dxh[2], dxh[3]);
        if (a.addend) {
          const float4 t = osm::ld4(a.addend + (row0 + p) * a.ldadd + cv.c);
          av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w;
          has_add = true;
        }
        if (a.addend2) {
          const float4 t = osm::ld4(a.addend2 + (row0 + p) * a.ldadd2 + cv.c);
          av[0] += t.x; av[1] += t.y; av[2] += t.z; av[3] += t.w;
          has_add = true;
        }
      }
      float ov[4];
      coop_apply_vec<MODE>(a, cv, q0, q1, xq, dq, av, has_add, ov);
      if (a.maxabs) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sv = (float)(act_t)ov[e];
          omax = fmaxf(omax, fabsf(sv));
          if (sv != sv) onan = 0x7fc00000u;
        }
      }
      osm::st4(a.out + (row0 + p) * a.ldo + cv.c, make_float4(ov[0], ov[1], ov[2], ov[3]));
    }
    if (a.maxabs) gn_publish_max(a.maxabs, b, w, nwg, omax, onan);
    if (MODE == 0 && a.maxabs_in) {
      __syncthreads();
      gn_publish_max(a.maxabs_in, b, w, nwg, imax, inan);
    }
  }
}

// ---------------------------------------------------------------- host side: which grid, if any
struct CoopPlan {
  int nv = 0, nwg = 0, rowT = 0;
};

inline long long coop_env_ll(const char* name, long long dflt) {
  const char* e = std::getenv(name);
  return (e && *e) ? std::atoll(e) : dflt;
}

// knobs (environment defaults; osm_gn_coop_set changes them at run time -- tests, probes)
struct CoopConfig {
  // OFF by default: measured a net loss in the guided step (same-box A/B, tools/step_ab.py: +0.16 ... +0.42 ms of 19.1 at B = 1;
  // backward-only at 32 x 32: -0.05 ms): the exchange costs ~9 us of dependent memory round trips per launch, the two launches it
  // removes cost ~10 us, and one workgroup per CU streams its chunk in serial phases (load | wait | store) at a lower rate than the
  // chunked kernels' many small workgroups.  See DESIGN.md "learned 35".  OSM_GN_COOP=1 / osm_gn_coop_set("on", 1) enables it.
  long long on = coop_env_ll("OSM_GN_COOP", 0);
  long long kb_target = coop_env_ll("OSM_GN_COOP_KB", 32);        // KB of one fp32 operand per workgroup
  long long min_kb = coop_env_ll("OSM_GN_COOP_MIN_KB", 512);      // smaller images keep the one-launch kernels
  long long force = coop_env_ll("OSM_GN_COOP_FORCE", 0);          // ignore the residency limit (tests of the bounded wait)
  long long timeout_us = coop_env_ll("OSM_GN_COOP_TIMEOUT_US", 2000);
  long long modes = coop_env_ll("OSM_GN_COOP_MODES", 3);          // bit 0: forward, bit 1: backward
  long long max_kb = coop_env_ll("OSM_GN_COOP_MAX_KB", 1 << 30);  // larger images keep the chunked path
};
inline CoopConfig& coop_cfg() {
  static CoopConfig c;
  return c;
}

template <int MODE, int NV>
int coop_capacity() {       // workgroups of this instance the device holds at once
  static const int cap = [] {
    int dev = 0, ncu = 0, nb = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gn_coop_kernel<MODE, NV>, COOP_T, 0) != hipSuccess) return 0;
    return ncu * nb;
  }();
  return cap;
}

template <int MODE>
int coop_capacity_nv(int nv) {
  switch (nv) {
    case 1: return coop_capacity<MODE, 1>();
    case 2: return coop_capacity<MODE, 2>();
    case 4: return coop_capacity<MODE, 4>();
    case 8: return coop_capacity<MODE, 8>();
    case 16: return MODE == 0 ? coop_capacity<0, 16>() : 0;
  }
  return 0;
}

// mode 0: forward (x resident), 1: backward (x and dy resident).  Shapes only: the pointers' alignment is checked at the call.
inline CoopPlan coop_plan(int B, int HW, int C, int G, int mode) {
  CoopPlan pl;
  const CoopConfig& cf = coop_cfg();
  const long long on = cf.on, kb_target = cf.kb_target < 1 ? 1 : cf.kb_target, min_kb = cf.min_kb, force = cf.force;
  if (!on || G != COOP_G || C % (4 * G) != 0 || (C >> 2) > COOP_T || B < 1 || HW < 1) return pl;
  const long long img_bytes = (long long)HW * C * 4;
  if (img_bytes < min_kb * 1024 || img_bytes > cf.max_kb * 1024 || !(cf.modes >> mode & 1)) return pl;
  const int vpr = C >> 2, rowT = COOP_T / vpr;
  // (32 vectors forward / 16 backward would hold a 256^2 x 256 tensor, but those instances spill ~130 registers: the gather
  // phase and the address arithmetic need ~150 registers next to the resident data)
  const int nvmax = mode == 0 ? 16 : 8;
  long long want = img_bytes / (kb_target * 1024);
  want = want < 2 ? 2 : (want > COOP_MAXW ? COOP_MAXW : want);
  for (int nv = 1; nv <= nvmax; nv *= 2) {
    const int rpw = nv * rowT;
    const int nwg = (HW + rpw - 1) / rpw;
    if (nwg > want && nv < nvmax) continue;         // too many small workgroups: take more rows each
    const int cap = mode == 0 ? coop_capacity_nv<0>(nv) : coop_capacity_nv<1>(nv);
    if (nwg > COOP_MAXW || (!force && (long long)nwg * B > cap)) continue;
    pl.nv = nv;
    pl.nwg = nwg;
    pl.rowT = rowT;
    return pl;
  }
  return pl;
}

template <int MODE>
int coop_launch(GNArgs& a, void* ws, float* finalized, hipStream_t st, const char* who) {
  OSM_REQUIRE(ws != nullptr && (reinterpret_cast<uintptr_t>(ws) & 7) == 0, "%s: null / misaligned workspace", who);
  const CoopPlan pl = coop_plan(a.B, a.HW, a.C, a.G, MODE);
  OSM_REQUIRE(pl.nwg > 0, "%s: no cooperative plan for B=%d HW=%d C=%d G=%d (ask osm_gn_coop_plan first)", who, a.B, a.HW, a.C, a.G);
  a.gs = a.C / a.G;
  OSM_REQUIRE(use_vec4(a) && a.gs % 4 == 0, "%s: operands must be 4-element aligned", who);
  OSM_REQUIRE(!a.maxabs || pl.nwg <= OSM_MAXABS_PARTS, "%s: grid does not fit the maxabs slots", who);
  a.fin = finalized;
  a.n = (double)a.HW * a.gs;
  CoopArgs ca;
  ca.a = a;
  ca.ws = static_cast<unsigned long long*>(ws);
  ca.nwg = pl.nwg;
  ca.rowT = pl.rowT;
  ca.timeout = coop_cfg().timeout_us * 100;          // 100 MHz wall clock
  const dim3 grid(pl.nwg, a.B);
  switch (pl.nv) {
    case 1: hipLaunchKernelGGL((gn_coop_kernel<MODE, 1>), grid, dim3(COOP_T), 0, st, ca); break;
    case 2: hipLaunchKernelGGL((gn_coop_kernel<MODE, 2>), grid, dim3(COOP_T), 0, st, ca); break;
    case 4: hipLaunchKernelGGL((gn_coop_kernel<MODE, 4>), grid, dim3(COOP_T), 0, st, ca); break;
    case 8: hipLaunchKernelGGL((gn_coop_kernel<MODE, 8>), grid, dim3(COOP_T), 0, st, ca); break;
    case 16:
      if constexpr (MODE == 0) {
        hipLaunchKernelGGL((gn_coop_kernel<0, 16>), grid, dim3(COOP_T), 0, st, ca);
        break;
      }
      [[fallthrough]];
    default: return osm::fail(OSM_ERR_INVALID, "%s: no instance for %d vectors per thread", who, pl.nv);
  }
  return osm::check_launch("gn_coop_kernel");
}
