/* osmosis_hip.h -- C ABI of libosmosis_hip.so (MI355X / gfx950 kernels for the Osmosis
 * guided-diffusion p_sample_loop hot path).
 *
 * The reference (osmosis-diffusion/osmosis-diffusion-code) is pure Python/PyTorch and has no
 * FFI of its own; its device work is stock ATen.  Each entry point below therefore cites the
 * reference Python call site whose ATen ops it replaces (paths relative to the reference repo).
 * The reference-side binding (ctypes) is shown in INTEGRATION.md and implemented in
 * osmosis_diffusion_code_amd/_lib.py.
 *
 * Conventions
 *   - every function returns 0 on success, a negative osm_status on failure and never throws;
 *     osm_last_error() returns a thread-local message for the last failure.
 *   - all pointers are DEVICE pointers (fp32 unless stated), owned by the caller; the library
 *     allocates nothing persistent.  `stream` is a hipStream_t (NULL = default stream).
 *   - activations are NHWC "matrix views": row m = pixel (b*H+h)*W+w, `ld*` = row stride in
 *     floats (>= channels, multiple of 4), so channel slices / concatenations are zero-copy.
 *   - entry points are re-entrant; no global mutable state.
 */
#ifndef OSMOSIS_HIP_H
#define OSMOSIS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef enum osm_status {
  OSM_OK = 0,
  OSM_ERR_INVALID = -1,   /* bad argument (shape / alignment / null pointer) */
  OSM_ERR_LAUNCH = -2,    /* HIP launch failure */
  OSM_ERR_UNSUPPORTED = -3
} osm_status;

int osm_version(void);                 /* (major<<16)|(minor<<8)|patch */
const char* osm_last_error(void);

/* ------------------------------------------------------------------ dense contraction
 * osm_conv2d_nhwc serves
 *   3x3 conv  (nn.py:22-32 conv_nd; unet.py:264,290,561,694)      ksize=3
 *   1x1 conv  (unet.py:301 skip_connection; unet.py:365,373 qkv/proj_out conv1d)  ksize=1
 * and, with the dgrad weight image, their input-gradients (autograd of the above); osm_gemm the batched attention GEMMs of the
 * unfused attention path (unet.py:428,432 einsum).  Which gfx950 kernel runs is decided by the WEIGHT IMAGE the caller hands
 * over (`wfmt`) and the shape (osm_conv_kernel_kind reports it):
 *   wfmt 0                    igemm_f32_kernel: exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), 128 x 128 x 32 LDS-staged tiles -- the
 *                             parity arithmetic ("f32"), 157 TFLOP/s roof;
 *   wfmt 2 / 3                fp32 operands split exactly into 2 / 3 bf16 planes, 3 / 6 v_mfma_f32_32x32x16_bf16 per product
 *                             ("bf16x3" / "bf16x6", the latter fp32-class): 3x3 layers by conv3_halo_bf16s_kernel (halo tile in
 *                             LDS, weight fragments straight from L2; 8 x 8 images by its PH = 8 instance), 1x1 layers and
 *                             ragged shapes by igemm_bf16s_kernel;
 *   wfmt 4                    "f16x3": both operands as two IEEE-half planes after power-of-two scaling (~22-bit operands), three
 *                             v_mfma_f32_32x32x16_f16 per product: 1x1 layers (H * W a multiple of 128) by igemm_bf16s_kernel<1,2,true>,
 *                             3x3 layers with H, W >= 8 by conv3_halo_bf16s_kernel<2,...,HP> (the 8 x 8 layers of the sampler);
 *   wfmt | OSM_WFMT_WINOGRAD  3x3 layers with H, W >= 16 by conv3_wino8_kernel: Winograd F(2x2, 3x3), 16 x 16 pixels x 64
 *                             output channels per workgroup, input transform in the registers that feed the MFMA, 16/36 of the
 *                             direct kernel's multiplies -- with wfmt 4 the default of the sampler (the dominant kernel of a step);
 *   wfmt 1                    the fp16-storage family (`_h` entry points, end of this file): one fp16 MFMA per product.
 * Split-K layers write fp32 partials to splitk_ws and are finished by splitk_reduce*_kernel (bias, residual, accumulate,
 * optional column sums) in the same call.
 */
typedef struct osm_conv_desc {
  const float* x;      /* [B*H*W][ldx] input, Cin channels used                       */
  const float* w;      /* packed weights [ksize*ksize][Cout][Cin] (osm_pack_conv_weight) */
  const float* bias;   /* [Cout] or NULL                                               */
  const float* res;    /* optional residual [B*H*W][ldr] added in the epilogue, or NULL */
  float* y;            /* [B*H*W][ldy] output                                          */
  float* splitk_ws;    /* workspace of splitk*B*H*W*Cout floats when splitk>1, else NULL */
  int B, H, W, Cin, Cout;
  int ksize;           /* 1 or 3; stride 1, zero padding ksize/2                       */
  int splitk;          /* >=1: number of K slices                                      */
  int accumulate;      /* !=0: y += result                                             */
  long long ldx, ldy, ldr;
  int wfmt;            /* weight image: 0 = fp32 [tap][Cout][Cin] (exact-f32 MFMA);
                          3 / 2 = split-bf16 planes from osm_pack_conv_weight_bf16s
                          (3 planes = "bf16x6", fp32-class accuracy; 2 planes = "bf16x3", ~2^-16);
                          1 = one fp16 plane (fp16 x fp16 -> fp32 MFMA): the fp16 family only, see the end of this file;
                          4 = "f16x3": two IEEE-half planes per operand, both operands scaled into the fp16 range by powers of
                          two (~22-bit operands, three fp16 MFMAs per product); needs x_maxabs, refuses gn_table.  3x3 layers:
                          4 | OSM_WFMT_WINOGRAD (osm_pack_conv_weight_winograd; H, W >= 16) or plain 4 (osm_pack_conv_weight_bf16s:
                          the direct halo-tile kernel, H, W >= 8); 1x1 layers: plain 4, H * W a multiple of 128 */
  const float* gn_table; /* optional fused input transform (3x3, split-bf16 formats, W >= 8, H >= 8 only):
                          x' = act(((x - mean_c) * rstd_c) * g_c + b_c) applied while staging, zero padding AFTER it
                          (= conv(SiLU(GroupNorm+FiLM(x)))).  [B][4][Cin] = mean | rstd | g | b rows from
                          osm_gn_prep; NULL = plain convolution */
  int gn_silu;           /* act = SiLU when != 0 */
  /* optional side output: per-column sums of the result, so that the GroupNorm that reads (or back-propagates
   * through) it needs no reduction pass of its own.  colsum: [B][chunks][2][Cout] floats, chunks =
   * osm_conv_stat_chunks(...) (0 = this layer's kernel cannot emit them).  stat_mode 1: (sum y, sum y^2);
   * stat_mode 2: y is d/d(act(xh g + b)), xh = (x - mean) rstd with x = stat_x [B*H*W][ld_sx] and the per-channel
   * rows mean | rstd | g | b in stat_table [B][4][Cout] (the table of osm_gn_prep / osm_gn_finalize_cols): sums of
   * dxh = y act'(z) g and dxh xh -- the two reductions of the GroupNorm backward.  Feed to osm_gn_finalize_cols. */
  float* colsum;
  int stat_mode, stat_silu;
  const float* stat_x;
  long long ld_sx;
  const float* stat_table;
  const float* x_maxabs; /* wfmt 4 only ("f16x3" images): 16-byte aligned [B][OSM_MAXABS_PARTS] partial max |x| of the input from osm_maxabs --
                          the kernel scales x into the fp16 range by a power of two and undoes it in its epilogue */
} osm_conv_desc;
int osm_conv2d_nhwc(const osm_conv_desc* d, void* stream);

/* OIHW [Cout][Cin][k][k] -> forward pack [k*k][Cout][Cin] and data-gradient pack
 * [k*k][Cin][Cout] (taps flipped, channels transposed).  Either output may be NULL. */
int osm_pack_conv_weight(const float* w_oihw, float* w_fwd, float* w_dgrad,
                         int Cout, int Cin, int ksize, void* stream);

typedef struct osm_gemm_desc {
  const float* A;      /* [M][lda], K contiguous                                       */
  const float* Bm;     /* b_kn==0: [N][ldb] (K contiguous);  b_kn==1: [K][ldb] (N contiguous) */
  const float* bias;   /* [N] or NULL                                                  */
  const float* res;    /* [M][ldr] or NULL                                             */
  float* C;            /* [M][ldc]                                                     */
  int M, N, K;
  int b_kn;
  int nb1, nb2;        /* two-level batch (e.g. heads, images); >=1                    */
  int accumulate;
  float alpha;         /* C = alpha*A*B (+bias +res)                                   */
  long long lda, ldb, ldc, ldr;
  long long sA1, sB1, sC1, sA2, sB2, sC2;   /* batch strides in floats (res uses sC*)  */
  int splitk;          /* <=1: none; else K is cut into `splitk` slices ...            */
  float* splitk_ws;    /* ... with fp32 partials in splitk*nb1*nb2*M*N floats, reduced deterministically */
} osm_gemm_desc;
/* Split-bf16 (wfmt 2, 3) / fp16 (wfmt 1: ONE plane of IEEE half, round-to-nearest-even) weight images: `wfmt` planes in MFMA-fragment order
 * [plane][tap][k16-step s][n/32 j][lane l][8]: n = 32j + (l&31), k = 16s + 8(l>>5) + e, zero padded, an even
 * number of k16 steps; forward n=Cout,k=Cin; data-gradient n=Cin,k=Cout (taps flipped).  Sizes in uint16
 * elements from osm_packed_weight_elems (wfmt 0 -> float elements of osm_pack_conv_weight).
 * wfmt 4 ("f16x3", ksize 1 only): two planes of IEEE halves of w * 2^ew (2^ew brings max |w| to [2^13, 2^14)), followed by
 * 16 bytes whose first float is 2^ew. */
long long osm_packed_weight_elems(int Cout, int Cin, int ksize, int wfmt, int dgrad);
int osm_pack_conv_weight_bf16s(const float* w_oihw, void* w_fwd, void* w_dgrad, int Cout, int Cin, int ksize,
                               int wfmt, void* stream);

/* Winograd F(2x2, 3x3) weight images (fp32 family; 3x3, stride 1): U = G g G^T of every (Cout, Cin) filter, formed
 * in double, rounded to fp32 and split into `wfmt` (2 / 3) bf16 planes, in MFMA-fragment order
 * [plane][xi = 16 transform positions][16-channel slab][n/32][lane][8].  A layer with osm_conv_winograd_ok(...) == 1
 * (H, W >= 16, Cin >= 16, Cout >= 64) may be run with desc.w = such an image and desc.wfmt = wfmt | OSM_WFMT_WINOGRAD:
 * 2.25x fewer MFMAs than the direct kernel, same fp32-class result (the transforms add roughly one more fp32 rounding
 * per operand).  gn_table, splitk, res / bias / accumulate and colsum (both stat modes; chunks = 16 x 16 patches per
 * image without split-K) work as in the direct kernel; osm_conv_stat_chunks / osm_conv_splitk take the flagged wfmt.
 * y, res, bias, stat_x, stat_table: 16-byte aligned, ld multiples of 4. */
#define OSM_WFMT_WINOGRAD 0x10
int osm_conv_winograd_ok(int H, int W, int Cin, int Cout, int ksize, int wfmt);
long long osm_winograd_weight_elems(int Cout, int Cin, int wfmt, int dgrad);
int osm_pack_conv_weight_winograd(const float* w_oihw, void* w_fwd, void* w_dgrad, int Cout, int Cin, int wfmt,
                                  void* stream);

int osm_gemm(const osm_gemm_desc* d, void* stream);
/* suggested split-K factor for a (M,N,K,taps) contraction with `nbatch` batches (1 = none) */
int osm_splitk_hint(int M, int N, int K, int taps, int nbatch);
/* the split-K factor osm_conv2d_nhwc(_h) should be given for a layer (pass it as desc.splitk with a workspace of
 * splitk*B*H*W*Cout floats): aims at ~one workgroup per CU of the kernel that layer runs. */
int osm_conv_splitk(int B, int H, int W, int Cin, int Cout, int ksize, int wfmt, int has_gn_table);
/* which kernel osm_conv2d_nhwc(_h) launches for a layer (measurement tooling: bench.py classifies its HIP-event times by
 * it): 0 exact-fp32 MFMA implicit GEMM, 1 split-plane implicit GEMM (1x1 and tap-chunked 3x3), 2 halo-tile 3x3 on 8 x 16
 * patches, 3 halo-tile 3x3 on 8 x 8 patches, 4 narrow-N halo-tile 3x3, 5 Winograd F(2x2, 3x3) */
int osm_conv_kernel_kind(int B, int H, int W, int Cin, int Cout, int ksize, int wfmt);
/* chunks per image of osm_conv_desc.colsum for a layer run with `splitk` (0: no column sums from that layer) */
int osm_conv_stat_chunks(int B, int H, int W, int Cin, int Cout, int ksize, int wfmt, int splitk, int has_gn_table);

/* ------------------------------------------------------------------ fused attention core (low resolutions)
 * QKVAttentionLegacy.forward / QKVAttention.forward (unet.py:416-433, 459-467) for T in {64, 256} tokens and
 * ch in {16, 32, 64}:  a = softmax(scale * q k^T) v  per (image, head), logits never leave the CU.
 * qkv is the [B*T][ldqkv] token matrix; head h has q / k / v at columns q_off / k_off / v_off + h*head_stride
 * (legacy order: offsets 0, ch, 2ch, stride 3ch; new order: 0, C, 2C, stride ch).  Forward writes `out`
 * [B*T][ldout] (head h at columns h*ch).  Backward takes d(out) in `dout` and writes dq | dk | dv into `dqkv`
 * (same column layout as qkv; every q/k/v column of the used heads is overwritten); `ws` is scratch of
 * 2*B*heads*T*T floats.  P is recomputed in the backward: nothing is kept from the forward. */
typedef struct osm_attn_desc {
  const float* qkv;
  long long ldqkv;
  int q_off, k_off, v_off, head_stride;
  int B, T, heads, ch;
  float scale;
  float* out;
  long long ldout;
  const float* dout;
  long long lddout;
  float* dqkv;
  long long lddqkv;
  float* ws;
  int arith;            /* osm_attn_flash_*: 0 = bf16x6 (fp32-class: three bf16 planes per operand, six MFMAs per product); 1 = ONE
                           IEEE-half plane per operand, one fp16 MFMA per product, fp32 accumulation and softmax -- the arithmetic of
                           the reference's use_fp16 attention (unet.py:426-433 on half tensors); 2 = "f16x3": every fp32 operand
                           tile scaled by a power of two (its max |.| is found in the kernel) and split into two IEEE-half terms,
                           three fp16 MFMAs per product, fp32 accumulation and softmax (fp32-class, ~22-bit operands: the
                           convolutions' default arithmetic).  Other entry points ignore it */
} osm_attn_desc;
/* Flash-style core on the matrix cores for 64-wide heads and T = 64 or a multiple of 128 (the 8x8 / 16x16 / 32x32 blocks):
 * logits / probabilities stay in registers, fp32 operands are split into 3 bf16 planes (6 MFMAs per product, fp32-class),
 * softmax in fp32.  Forward writes `out` and lse[B*heads][T] (max + log-sum of the scaled logits of each query row);
 * backward takes that lse and the forward output (for delta = rowsum(d(out) * out), scratch `delta` [B*heads][T]),
 * recomputes P and writes dq | dk | dv into dqkv. */
int osm_attn_flash_supported(int T, int ch);
int osm_attn_flash_fwd(const osm_attn_desc* d, float* lse, void* stream);
int osm_attn_flash_bwd(const osm_attn_desc* d, const float* out, long long ldout, const float* lse, float* delta,
                       void* stream);
int osm_attn_small_supported(int T, int ch);
int osm_attn_small_fwd(const osm_attn_desc* d, void* stream);
int osm_attn_small_bwd(const osm_attn_desc* d, void* stream);

/* ------------------------------------------------------------------ GroupNorm(32)+SiLU(+FiLM)
 * nn.py:17-19,93-100 GroupNorm32; unet.py:263,287 SiLU; unet.py:327-331 scale-shift.
 * stats: [B][G][2] = (mean, rstd).  part: workspace of B*nchunk*G*2 floats,
 * nchunk = osm_gn_nchunk(HW).  film: row b at film + b*ldfilm holds (scale[C] | shift[C]), or NULL. */
int osm_gn_nchunk(int HW);
int osm_gn_stats(const float* x, long long ldx, int B, int HW, int C, int G, float eps,
                 float* part, float* stats, void* stream);
int osm_gn_apply(const float* x, long long ldx, float* y, long long ldy, int B, int HW, int C, int G,
                 const float* stats, const float* gamma, const float* beta, const float* film,
                 long long ldfilm, int silu, float* maxabs_out, void* stream);
/* maxabs_out (here and in osm_gn_fwd / osm_gn_bwd / osm_gn_bwd_apply; NULL = off): the pass that writes the output also
 * leaves its per-image partial max |out| in the osm_maxabs format [B][OSM_MAXABS_PARTS] -- the f16x3 convolution that reads
 * the tensor next then needs no osm_maxabs pass.  Requires osm_gn_nchunk(HW) <= OSM_MAXABS_PARTS. */
/* stats + apply in one call (one launch for HW <= 256); `stats` is written (kept for the backward).
 * maxabs_in (NULL = off; fp32 family, HW > 256): the statistics pass also leaves the partial max |x| of the INPUT in the osm_maxabs
 * format -- for an f16x3 1x1 convolution that reads x itself (the skip connection of a ResBlock, unet.py:274-283). */
int osm_gn_fwd(const float* x, long long ldx, float* y, long long ldy, int B, int HW, int C, int G, float eps,
               float* part, float* stats, const float* gamma, const float* beta, const float* film,
               long long ldfilm, int silu, float* maxabs_out, float* maxabs_in, void* stream);
/* Statistics only + the per-channel table a convolution applies itself (osm_conv_desc.gn_table):
 * table [B][4][C] = mean | rstd | gamma*(1+scale) | beta*(1+scale)+shift.  `stats` is written as by osm_gn_stats. */
int osm_gn_prep(const float* x, long long ldx, int B, int HW, int C, int G, float eps, float* part, float* stats,
                const float* gamma, const float* beta, const float* film, long long ldfilm, float* table,
                float* maxabs_in /* as in osm_gn_fwd; NULL = off */, void* stream);
/* Statistics from column sums emitted by the producing convolution (osm_conv_desc.colsum, [B][nchunk][2][C]):
 * mode 0: stats[B][G][2] = (mean, rstd) of the tensor (+ the per-channel table [B][4][C] when table != NULL, as
 * osm_gn_prep);  mode 1: stats = (sum dxh / n, sum dxh xh / n), the two means of the GroupNorm backward (gstats). */
int osm_gn_finalize_cols(const float* colsum, int nchunk, int B, int HW, int C, int G, float eps, int mode,
                         float* stats, const float* gamma, const float* beta, const float* film, long long ldfilm,
                         float* table, void* stream);
/* the apply pass of osm_gn_bwd alone, with gstats given (osm_gn_finalize_cols mode 1) */
int osm_gn_bwd_apply(const float* x, long long ldx, const float* dy, long long lddy, float* dx, long long lddx,
                     const float* addend, long long ldadd, const float* addend2, long long ldadd2, int B, int HW, int C, int G,
                     const float* stats, const float* gstats, const float* gamma, const float* beta, const float* film,
                     long long ldfilm, int silu, float* maxabs_out, void* stream);
/* dx = dGN(dy) (+ addend) (+ addend2).  part: workspace as above.  An addend may alias dx (in-place accumulation: the
 * residual / concat gradients of the UNet are added here instead of in a pass of their own). */
int osm_gn_bwd(const float* x, long long ldx, const float* dy, long long lddy, float* dx, long long lddx,
               const float* addend, long long ldadd, const float* addend2, long long ldadd2, int B, int HW, int C, int G,
               const float* stats, const float* gamma, const float* beta, const float* film,
               long long ldfilm, int silu, float* part, float* gstats, float* maxabs_out, void* stream);
/* ------------------------------------------------------------------ resampling (unet.py:186, 215)
 * y[B][H/2][W/2][C] = scale * sum_{2x2} x   (avg-pool: scale=0.25; upsample-backward: scale=1)
 * y[B][2H][2W][C]   = scale * x[h/2][w/2]   (nearest-upsample: scale=1; avg-pool-backward: 0.25) */
int osm_pool2x2(const float* x, long long ldx, float* y, long long ldy, int B, int H, int W, int C,
                float scale, void* stream);
int osm_upsample2x(const float* x, long long ldx, float* y, long long ldy, int B, int H, int W, int C,
                   float scale, void* stream);
/* Round 5 -- pieces of the UNet variants no shipped Osmosis config uses (reference unet.py:160-219 Upsample / Downsample WITH a
 * convolution, i.e. resblock_updown=False; :329-332 additive conditioning, i.e. use_scale_shift_norm=False; :729-731 class embedding):
 *   osm_stride2_pick    y[b][i][j][:] = x[b][2i][2j][:]   (x: H x W -> y: H/2 x W/2): a stride-2 3x3 convolution = osm_conv2d_nhwc at
 *                       stride 1, every other pixel kept;
 *   osm_stride2_place   y[b][2i][2j][:] = x[b][i][j][:], 0 elsewhere  (x: H/2 x W/2 -> y: H x W; H, W = the OUTPUT's): its adjoint;
 *   osm_add_rowvec      y[b][p][c] += v[b][c]  (y: [B * HW][ldy] activations, v: [B][ldv] fp32): h + emb_out, emb + label_emb[y]. */
int osm_stride2_pick(const float* x, long long ldx, float* y, long long ldy, int B, int H, int W, int C, void* stream);
int osm_stride2_place(const float* x, long long ldx, float* y, long long ldy, int B, int H, int W, int C, void* stream);
int osm_add_rowvec(float* y, long long ldy, const float* v, long long ldv, int B, long long HW, int C, void* stream);
/* two tensors of one shape through ONE launch (up != 0: osm_upsample2x, else osm_pool2x2; 4-element vectors only): an up / down
 * ResBlock resamples both its input and its normalised input (unet.py:263-270), its backward both gradients */
int osm_resample_pair(int up, const float* x1, long long ldx1, float* y1, long long ldy1, const float* x2, long long ldx2,
                      float* y2, long long ldy2, int B, int H, int W, int C, float scale, void* stream);

/* ------------------------------------------------------------------ attention pieces
 * unet.py:431 softmax over the last dim (rows of length T); P and optionally P^T are written.
 * bwd: dS = P*(dP - rowsum(dP*P)); dS and optionally dS^T written.  nmat = B*heads matrices. */
int osm_softmax_rows(const float* S, float* P, float* PT, int nmat, int T, void* stream);
int osm_softmax_rows_bwd(const float* P, const float* dP, float* dS, float* dST, int nmat, int T,
                         void* stream);

/* ------------------------------------------------------------------ embeddings (nn.py:103-121, unet.py:550-554, 278-284)
 * temb: out[B][dim] = [cos(t*f) | sin(t*f)].  t: device floats [B].
 * linear: y[B][N] = act(x)[B][K] @ W[N][K]^T + b, act = SiLU if silu_in. silu_out applies SiLU to y. */
int osm_timestep_embedding(const float* t, float* out, int B, int dim, float max_period, void* stream);
int osm_linear(const float* x, const float* W, const float* b, float* y, int B, int K, int N,
               int silu_in, int silu_out, void* stream);

/* ------------------------------------------------------------------ layout / elementwise */
int osm_nchw_to_nhwc(const float* x, float* y, long long ldy, int B, int C, int HW, void* stream);
int osm_nhwc_to_nchw(const float* x, long long ldx, float* y, int B, int C, int HW, void* stream);
/* per-image max |x| of [B][rows_per_img][C] (row stride ldx) as OSM_MAXABS_PARTS partial maxima per image:
 * out[B][OSM_MAXABS_PARTS], every entry rewritten on every call (no clearing, no atomics); the consumer folds them.  Feeds
 * osm_conv_desc::x_maxabs.  No reference counterpart: the f16x3 arithmetic needs the operand range (nn.py:22-32 conv_nd
 * has fp32's exponent range). */
#define OSM_MAXABS_PARTS 1024
int osm_maxabs_parts(void);
int osm_maxabs(const float* x, long long ldx, int B, long long rows_per_img, int C, float* out, void* stream);
int osm_copy2d(const float* x, long long ldx, float* y, long long ldy, long long M, int C,
               int accumulate, void* stream);

/* ------------------------------------------------------------------ sampler step (NCHW [B,4,H,W])
 * coef: device float[8] = {sqrt_recip_ac, sqrt_recipm1_ac, post_coef1, post_coef2,
 *                          min_log(posterior_log_variance_clipped), max_log(log beta), noise_on, t}
 * posterior (posterior_mean_variance.py:127-136, 246-258; gaussian_diffusion.py:345-365):
 *   x0 = c0*x - c1*eps ; mean = c2*x0 + c3*x ; logvar = f*c5 + (1-f)*c4, f=(v+1)/2 */
int osm_posterior(const float* model_out /*[B,8,HW]*/, const float* x /*[B,4,HW]*/, const float* coef,
                  float* x0, float* mean, float* logvar, int B, int HW, void* stream);
/* The same step for every registered processor pair (posterior_mean_variance.py:53-136 mean, :171-258 variance; osm_posterior is
 * (0, 0)).  The row keeps ONE meaning for all mean kinds: c0 = d x0/d x, c1 = -d x0/d out (what osm_posterior_bwd and the
 * update kernels read), c2 / c3 = posterior_mean_coef1 / 2, coef[4] / coef[5] = the variance processor's row.
 *   mean_kind 0 'epsilon'    : x0 = c0*x - c1*out, mean = c2*x0 + c3*x       (c0 = sqrt_recip_ac, c1 = sqrt_recipm1_ac)
 *             1 'start_x'    : x0 = out,           mean = c2*x0 + c3*x       (row: c0 = 0, c1 = -1)
 *             2 'previous_x' : x0 = c0*x - c1*out, mean = out                (row: c0 = -coef2/coef1, c1 = -1/coef1)
 *   var_kind  0 'learned_range': logvar = f*coef[5] + (1-f)*coef[4], f = (v+1)/2 ; 1 'fixed_small' / 'fixed_large': logvar = coef[4]
 *             (log posterior_variance[t] / log(append(posterior_variance[1], betas[1:]))[t]) ; 2 'learned': logvar = v
 * (v = model_out[:, 4:8]: the network of the path always has 2 C output channels, gaussian_diffusion.py:349-350).
 * clip_denoised != 0 (process_xstart, posterior_mean_variance.py:43-50; `clip_denoised: True` of configs/rgb_guidance_sample_config.yaml):
 * x0 = clamp(prediction, -1, 1), the mean is formed from the clamped x0, and the unclamped prediction is written to x0_raw [B,4,HW]
 * (required then) for osm_clamp_bwd. */
int osm_posterior_typed(const float* model_out, const float* x, const float* coef, int mean_kind, int var_kind,
                        int clip_denoised, float* x0_raw, float* x0, float* mean, float* logvar, int B, int HW, void* stream);
/* Backward of clamp(x_raw, lo, hi) applied to a gradient in place: g[i] = 0 where x_raw[i] is outside [lo, hi] (bounds pass, NaN does
 * not: ATen's clamp_backward).  With clip_denoised the guidance gradient d loss/d x0 is masked by it before osm_posterior_bwd and the
 * update kernels read it. */
int osm_clamp_bwd(float* g, const float* x_raw, float lo, float hi, long long n, void* stream);

/* physical forward model + guidance loss (measurements.py:138-151,251-264,363-376;
 * condition_methods.py:109-144; losses.py:29-83; utils.py:544-566,674-700) */
typedef struct osm_phys_desc {
  int kind;            /* 0 underwater_physical_revised, 1 underwater_physical, 2 haze_physical,
                          3 identity: I = x0[:, 0:3] -- the 'noise' / 'rgb_guidance' operators of the rgb-guidance ('ps') path
                          (measurements.py:41-66, condition_methods.py:35-41: loss = ||y - x0[:, 0:3]||); loss_type 0, no
                          weight, no auxiliary losses, no parameters (osm_phys_finalize with do_update = 0 only) */
  int depth_type;      /* 0 original, 1 gamma, 2 move                                  */
  float dval[3];       /* depth params (gamma: v0,v1,v2 ; move: v0)                    */
  int weight_type;     /* 0 none, 1 depth                                              */
  int wdepth_type;     /* depth_type of the loss weight function                       */
  float wval[3];
  int loss_type;       /* 0 norm, 1 mse                                                */
  float gamma_avrg;    /* aux avrg_loss coefficient (0 = off)                          */
  float gamma_val;     /* aux val_loss coefficient (0 = off)                           */
  float eta[3];        /* step size (lr) for phi_a(phi_ab), phi_b, phi_inf             */
  int B, HW;
  int optimizer;       /* 0: plain gradient descent ('sgd' / 'GD', measurements.py:157-177); otherwise the torch.optim class of
                          utils.py:494-524 with ITS DEFAULT hyper-parameters (the reference passes none) and lr = eta per group:
                          1 Adam, 2 AdamW (weight_decay 0.01), 3 Adamax, 4 RMSprop, 5 Adagrad, 6 Adadelta, 7 ASGD, 8 Rprop.
                          ('sparseadam' and 'lbfgs' raise in the reference itself: dense gradients / step() without a closure) */
} osm_phys_desc;
/* phi: device float [B][9] = phi_a[3] | phi_b[3] | phi_inf[3]  (kinds 1,2 use phi_a as phi_ab;
 * haze uses phi_a[0] only).  part: workspace B*nblk*16 floats, nblk = osm_phys_nblk(HW).
 * red: device float [B][16]: 0 sum r^2, 1..9 raw phi-gradient sums, 10..12 sum rgb_c, 13 val-loss sum.
 * loss_out: [B] data-term loss per image (the reference's sep_loss). */
int osm_phys_nblk(int HW);
int osm_phys_reduce(const osm_phys_desc* d, const float* x0, const float* y, const float* phi,
                    float* part, void* stream);
/* opt_state: [B][20] floats of optimizer state per image (Adam / AdamW: exp_avg[9] | exp_avg_sq[9] | step; the other optimizers:
 * see csrc/guidance.hip), zero-initialised by the caller, required when d->optimizer != 0 and do_update != 0; NULL otherwise. */
int osm_phys_finalize(const osm_phys_desc* d, const float* part, float* red, float* phi, int do_update,
                      float* loss_out, float* opt_state, void* stream);
/* g[B,4,HW] = d(total loss)/d(x0) for the current phi and the reductions in `red`. */
int osm_phys_grad(const osm_phys_desc* d, const float* x0, const float* y, const float* phi,
                  const float* red, float* g, void* stream);

/* The inner phi loop of one guided step in one call (measurements.py:266-303 `optimize`, condition_methods.py:109-144):
 * n_inner x { osm_phys_reduce; osm_phys_finalize with the phi step }, the loss (loss_out [B]) and g = dL/dx0 taken at the phi of the
 * last iteration, which is stepped afterwards; freeze_phi != 0 (n_inner must be 1): loss and g only, phi untouched. */
int osm_phys_optimize(const osm_phys_desc* d, const float* x0, const float* y, float* phi, float* part, float* red,
                      float* loss_out, float* g, int n_inner, int freeze_phi, float* opt_state, void* stream);

/* d_out[B,8,HW]: channels 0..3 = -c1*g, 4..7 = 0   (chain rule through x0 = c0*x - c1*eps) */
int osm_posterior_bwd(const float* g, const float* coef, float* d_out, int B, int HW, void* stream);
/* condition_methods.py:211-224 + gaussian_diffusion.py:266-268:
 *   grad = c0*g + dx_unet ; x_t = mean - scale[c]*clamp(grad,+-clip) ; x_next = x_t + exp(.5*logvar)*noise*noise_on
 * grad_out (optional) receives the unclipped gradient. clip < 0 disables clipping; clip >= 0 clamps to +-clip
 * (clip = 0 zeroes the guidance term, as torch.clamp(g, -0, 0) does). */
int osm_guide_update(const float* mean, const float* logvar, const float* g, const float* dx_unet,
                     const float* noise, const float* coef, const float* scale4, float clip,
                     float* x_next, float* grad_out, int B, int HW, void* stream);
/* osm_guide_update with the step noise drawn INSIDE the kernel (gaussian_diffusion.py:266-268 `noise = torch.randn_like(img)`):
 * Philox-4x32-10, key = seed, counter = (element / 4, img0 + b * img_stride, *step, stream id) -> the four normals (two Box-Muller
 * pairs) of four consecutive elements of image b.  An image's noise depends on (seed, its index, the step) only: not on the batch
 * it travels in, not on chunking, not on the grid; img_stride = 0 gives every image of the batch the SAME noise (what separately
 * seeded batch-1 runs draw).  `step` is the device-side step counter osm_fetch_coefs moves (read, not
 * moved, here); the counter word is *step + step_offset (+1 when the fetch of this step has already moved it by -1).  noise_out (optional, [B,4,HW]) receives the draws (tests, traces).  H*W % 4 == 0; tensors 16-byte aligned.
 * NOT the bit stream of ATen's Philox (offsets / Box-Muller pairing differ): parity runs inject noise through osm_guide_update. */
int osm_guide_update_rng(const float* mean, const float* logvar, const float* g, const float* dx_unet, const float* coef,
                         const float* scale4, float clip, float* x_next, float* grad_out, float* noise_out, int B, int HW,
                         unsigned long long seed, const int* step, int step_offset, int img0, int img_stride, void* stream);
/* out[b][0..n) ~ N(0,1) from the same generator (image b: counter word 1 = img0 + b * img_stride; word 2 = *step_dev if given, else
 * step_const): what osm_guide_update_rng draws for (seed, image, step) when n = 4*H*W.  out 16-byte aligned; B > 1 needs n % 4 == 0. */
int osm_randn(float* out, int B, long long n, unsigned long long seed, const int* step_dev, int step_const, int img0,
              int img_stride, void* stream);
/* out[4 q .. 4 q + 3] = Philox-4x32-10(counter = (q, c1, c2, c3), key = (k0, k1)) for q < n4: the raw generator, checked in the
 * tests against the Random123 known-answer vectors. */
int osm_philox_raw(unsigned* out, long long n4, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, void* stream);
/* DDIM step + guidance of the rgb-guidance path (gaussian_diffusion.py:505-535 `DDIM.p_sample`, condition_methods.py:247-251):
 *   eps = (r0*x - x0)/r1 ; sigma = eta*sqrt((1-abp)/(1-ab))*sqrt(1-ab/abp) ; grad = c0*g + dx_unet
 *   x_next = x0*sqrt(abp) + sqrt(1-abp-sigma^2)*eps + noise_on*sigma*noise - scale[c]*clamp(grad,+-clip)
 * coef: the posterior row of the step (only c0 = d x0/d x is read); dcoef: device float[8] = {alpha_bar, alpha_bar_prev, eta,
 * noise_on, r0 = sqrt_recip_alphas_cumprod, r1 = sqrt_recipm1_alphas_cumprod, -, t}: predict_eps_from_x_start (:533-536) reads the
 * sampler's own tables, whatever the mean processor (for 'epsilon' r0 = c0, r1 = c1).
 * g / dx_unet / noise / grad_out optional; x_next may alias x. */
int osm_ddim_update(const float* x0, const float* x, const float* g, const float* dx_unet, const float* noise,
                    const float* coef, const float* dcoef, const float* scale4, float clip, float* x_next, float* grad_out,
                    int B, int HW, void* stream);
/* unconditional ancestral step of the RGBD prior sampler (osmosis_utils/diffusion.py:94-122), NCHW:
 *   eps = model_out[:, :C]; x_next = c_a (x - c_b eps) + c_s z; x0 = c_r x - c_m eps (x0, z optional)
 * coef: device float[8] = {c_a, c_b, c_s, c_r, c_m, -, -, t}.  x_next may alias x (in-place update). */
int osm_ancestral_step(const float* model_out, const float* x, const float* z, const float* coef,
                       float* x_next, float* x0, int B, int C, int Cout, int HW, void* stream);
/* coef_out[8] = table[clamp(*step, 0, n_rows-1)][8]; t_out[b] = coef_out[7]; then *step += delta
 * (graph-replayable; B <= 256) */
int osm_fetch_coefs(const float* table, int n_rows, int* step, int delta, float* coef_out, float* t_out, int B,
                    void* stream);

/* ================================================================== fp16-storage family (`use_fp16: True`)
 * The reference's fp16 mode (guided_diffusion/unet.py:544,697-703,733; fp16_util.py:13-20 convert_module_to_f16)
 * keeps activations and conv weights in IEEE half, accumulates in fp32 (ATen/cuDNN) and runs GroupNorm32 in
 * fp32 (nn.py:17-19).  The entry points below are the SAME operations on activations stored as half
 * (osm_half_t = the 16 bits of an IEEE binary16): every `float*` that addresses NHWC activations / their
 * gradients becomes `osm_half_t*`, `ld*` count halfs (multiples of 4), statistics / gamma / beta / FiLM /
 * bias / split-K partials stay fp32, arithmetic and accumulation are fp32.  osm_conv2d_nhwc_h takes wfmt 1
 * weight images (osm_pack_conv_weight_bf16s(..., wfmt = 1)) and multiplies fp16 x fp16 into fp32 on
 * v_mfma_f32_32x32x16_f16; its descriptor has the layout of osm_conv_desc.  Everything without a `_h` twin
 * (embeddings, attention core, sampler step) is fp32 in both modes; osm_half_to_f32 / osm_f32_to_half bridge. */
typedef unsigned short osm_half_t;
typedef struct osm_conv_desc_h {
  const osm_half_t* x;
  const void* w;        /* wfmt 1 image */
  const float* bias;
  const osm_half_t* res;
  osm_half_t* y;
  float* splitk_ws;
  int B, H, W, Cin, Cout;
  int ksize, splitk, accumulate;
  long long ldx, ldy, ldr;
  int wfmt;             /* 1 */
  const float* gn_table;
  int gn_silu;
  float* colsum;
  int stat_mode, stat_silu;
  const osm_half_t* stat_x;
  long long ld_sx;
  const float* stat_table;
  const float* x_maxabs;  /* unused in this family (keeps the layout of osm_conv_desc) */
} osm_conv_desc_h;
int osm_conv2d_nhwc_h(const osm_conv_desc_h* d, void* stream);
int osm_gn_stats_h(const osm_half_t* x, long long ldx, int B, int HW, int C, int G, float eps,
                   float* part, float* stats, void* stream);
int osm_gn_apply_h(const osm_half_t* x, long long ldx, osm_half_t* y, long long ldy, int B, int HW, int C, int G,
                   const float* stats, const float* gamma, const float* beta, const float* film,
                   long long ldfilm, int silu, float* maxabs_out /* must be NULL */, void* stream);
int osm_gn_fwd_h(const osm_half_t* x, long long ldx, osm_half_t* y, long long ldy, int B, int HW, int C, int G, float eps,
                 float* part, float* stats, const float* gamma, const float* beta, const float* film,
                 long long ldfilm, int silu, float* maxabs_out /* must be NULL */, float* maxabs_in /* must be NULL */, void* stream);
int osm_gn_prep_h(const osm_half_t* x, long long ldx, int B, int HW, int C, int G, float eps, float* part, float* stats,
                  const float* gamma, const float* beta, const float* film, long long ldfilm, float* table,
                  float* maxabs_in /* must be NULL */, void* stream);
int osm_gn_bwd_h(const osm_half_t* x, long long ldx, const osm_half_t* dy, long long lddy, osm_half_t* dx, long long lddx,
                 const osm_half_t* addend, long long ldadd, const osm_half_t* addend2, long long ldadd2, int B, int HW, int C, int G,
                 const float* stats, const float* gamma, const float* beta, const float* film,
                 long long ldfilm, int silu, float* part, float* gstats, float* maxabs_out /* must be NULL */, void* stream);
int osm_gn_bwd_apply_h(const osm_half_t* x, long long ldx, const osm_half_t* dy, long long lddy, osm_half_t* dx, long long lddx,
                       const osm_half_t* addend, long long ldadd, const osm_half_t* addend2, long long ldadd2, int B, int HW, int C, int G,
                       const float* stats, const float* gstats, const float* gamma, const float* beta, const float* film,
                       long long ldfilm, int silu, float* maxabs_out /* must be NULL */, void* stream);
int osm_pool2x2_h(const osm_half_t* x, long long ldx, osm_half_t* y, long long ldy, int B, int H, int W, int C,
                  float scale, void* stream);
int osm_upsample2x_h(const osm_half_t* x, long long ldx, osm_half_t* y, long long ldy, int B, int H, int W, int C,
                     float scale, void* stream);
int osm_stride2_pick_h(const osm_half_t* x, long long ldx, osm_half_t* y, long long ldy, int B, int H, int W, int C, void* stream);
int osm_stride2_place_h(const osm_half_t* x, long long ldx, osm_half_t* y, long long ldy, int B, int H, int W, int C, void* stream);
int osm_add_rowvec_h(osm_half_t* y, long long ldy, const float* v, long long ldv, int B, long long HW, int C, void* stream);
int osm_resample_pair_h(int up, const osm_half_t* x1, long long ldx1, osm_half_t* y1, long long ldy1, const osm_half_t* x2,
                        long long ldx2, osm_half_t* y2, long long ldy2, int B, int H, int W, int C, float scale, void* stream);
int osm_nchw_to_nhwc_h(const float* x, osm_half_t* y, long long ldy, int B, int C, int HW, void* stream);   /* fp32 NCHW -> half NHWC */
int osm_nhwc_to_nchw_h(const osm_half_t* x, long long ldx, float* y, int B, int C, int HW, void* stream);   /* half NHWC -> fp32 NCHW */
int osm_copy2d_h(const osm_half_t* x, long long ldx, osm_half_t* y, long long ldy, long long M, int C,
                 int accumulate, void* stream);
/* [M][C] strided conversions (C, ld multiples of 4) between the two storage types */
int osm_half_to_f32(const osm_half_t* x, long long ldx, float* y, long long ldy, long long M, int C, void* stream);
int osm_f32_to_half(const float* x, long long ldx, osm_half_t* y, long long ldy, long long M, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OSMOSIS_HIP_H */
