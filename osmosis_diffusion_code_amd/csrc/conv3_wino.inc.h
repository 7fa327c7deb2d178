// Winograd F(2x2, 3x3) form of the 3x3 convolution: shared constants (LDS staging layout) and the weight-image packing.
// The kernel is conv3_wino8.inc.h (round 3: 8 waves, two per SIMD); the one-wave-per-SIMD, hand-pipelined kernel of round 2
// that lived here reached the same time (both sit at the socket power limit) and was removed.
//
// Why Winograd: with several MFMAs per fp32 product the matrix work is what costs time and power; F(2x2, 3x3) computes a
// 2 x 2 output tile from a 4 x 4 input tile with 16 instead of 36 multiplications per (input channel, output channel):
//     V = B^T d B (input: adds only),  U = G g G^T (weights: precomputed),  M_xi = sum_c V_xi[c] U_xi[c][n],  Y = A^T M A
// i.e. 16 independent GEMMs [tiles x Cin] . [Cin x Cout]: 2.25x fewer MFMAs for the same result.  (cuDNN picks the same
// algorithm for these layers of the reference.)
//
// Staging layout (per 16-channel slab of a 16 x 16-pixel patch): the 18 x 18 halo as raw fp32 in LDS,
// [channel quad 4][row 18][column parity 2][column / 2: 10 slots, 9 used] of 16-byte slots: the 16 lanes of a ds_read_b128
// group (neighbouring tiles, pixel stride 2) read 16 consecutive slots -> conflict-free.
constexpr int WN_ROWP = 20;                  // 16-byte slots per staged row: [column parity 2][10 (9 used)]
constexpr int WN_QP = 18 * WN_ROWP + 1;      // slots per channel-quad plane (+1: the 4 quads of a pixel hit 4 bank groups)

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
    if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &e); e = min(14 - e, 100); }
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
  if (mx > 0.f && mx < 3.0e38f) { (void)frexpf(mx, &e); e = min(14 - e, 100); }
  word[0] = __float_as_uint(ldexpf(1.f, e));
}


#endif   // !OSM_ACT_F16
