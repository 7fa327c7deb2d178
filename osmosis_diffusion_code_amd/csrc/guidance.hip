// Per-step sampler math outside the UNet (NCHW [B,4,H,W] like the reference's tensors):
//   posterior mean / x0 / log-variance            posterior_mean_variance.py:127-136, 246-258
//   physical image-formation model + guidance loss  measurements.py:138-151, 251-264, 363-376;
//                                                   condition_methods.py:109-144; losses.py:29-83
//   analytic gradient of that loss w.r.t. x0 and phi, on-device SGD on phi (measurements.py:266-303)
//   guidance update + ancestral noise               condition_methods.py:211-224; gaussian_diffusion.py:266-268
// All HBM/latency bound and tiny (a few MB per step); the point of doing them here is that the
// 20-iteration phi loop runs with no host synchronisation (the reference syncs 4+20 times a step).
// Batch semantics: every reduction is PER IMAGE (equal to B=1 reference runs image by image,
// SURVEY.md F1/F2).
#include "osm_common.h"

namespace {

constexpr int PPB = 1024;  // pixels per reduce workgroup
constexpr int NRED = 16;

__device__ __forceinline__ float conv_depth(float D, int type, const float* v, float& dd) {
  if (type == 1) {  // gamma: ((D + v0) * v1) ^ v2
    const float base = (D + v[0]) * v[1];
    if (v[2] == 1.0f) {
      dd = v[1];
      return base;
    }
    const float pw = powf(base, v[2]);
    dd = v[1] * v[2] * powf(base, v[2] - 1.0f);
    return pw;
  }
  if (type == 2) {  // move
    dd = 1.0f;
    return D + v[0];
  }
  dd = 0.5f;  // original
  return 0.5f * (D + 1.0f);
}

struct Pix {
  float rgb[3], D, y[3];
  float d, dd, w;
  float Ea[3], Eb[3], J[3], r[3];
};

__device__ __forceinline__ void eval_pixel(const osm_phys_desc& ds, const float* __restrict__ x0,
                                           const float* __restrict__ y, const float* __restrict__ phi,
                                           int b, int p, Pix& q) {
  const long long base = (long long)b * 4 * ds.HW + p;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    q.rgb[c] = x0[base + (long long)c * ds.HW];
    q.y[c] = y[(long long)b * 3 * ds.HW + (long long)c * ds.HW + p];
  }
  q.D = x0[base + 3LL * ds.HW];
  if (ds.kind == 3) {   // identity forward model of the rgb-guidance ('ps') path: I = x0[:, 0:3], unweighted (condition_methods.py:35-41)
    q.d = 0.f; q.dd = 0.f; q.w = 1.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      q.Ea[c] = q.Eb[c] = 1.f;
      q.J[c] = 0.5f * (q.rgb[c] + 1.0f);
      q.r[c] = q.y[c] - q.rgb[c];
    }
    return;
  }
  q.d = conv_depth(q.D, ds.depth_type, ds.dval, q.dd);
  float wdd;
  q.w = ds.weight_type == 1 ? conv_depth(q.D, ds.wdepth_type, ds.wval, wdd) : 1.0f;
  const float* ph = phi + b * 9;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float pa = ds.kind == 2 ? ph[0] : ph[c];
    const float pb = ds.kind == 0 ? ph[3 + c] : pa;
    const float pinf = ph[6 + c];
    q.Ea[c] = expf(-pa * q.d);
    q.Eb[c] = expf(-pb * q.d);
    q.J[c] = 0.5f * (q.rgb[c] + 1.0f);
    const float I = q.J[c] * q.Ea[c] + pinf * (1.0f - q.Eb[c]);
    q.r[c] = (q.y[c] - (2.0f * I - 1.0f)) * q.w;
  }
}

__global__ __launch_bounds__(256) void phys_reduce_kernel(osm_phys_desc ds, const float* __restrict__ x0,
                                                           const float* __restrict__ y,
                                                           const float* __restrict__ phi,
                                                           float* __restrict__ part, int nblk) {
  __shared__ float red[4][NRED];
  const int b = blockIdx.y, blk = blockIdx.x;
  float s[NRED];
#pragma unroll
  for (int k = 0; k < NRED; ++k) s[k] = 0.f;
  const int pend = min(ds.HW, (blk + 1) * PPB);
  const float* ph = phi + b * 9;
  for (int p = blk * PPB + threadIdx.x; p < pend; p += 256) {
    Pix q;
    eval_pixel(ds, x0, y, phi, b, p, q);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float pinf = ph[6 + c];
      const float k2 = -2.0f * q.w * q.r[c];                 // r_c * d r_c / d I_c
      s[0] += q.r[c] * q.r[c];
      s[1 + c] += k2 * (-q.d * q.J[c] * q.Ea[c]);            // d I / d phi_a
      s[4 + c] += k2 * (pinf * q.d * q.Eb[c]);               // d I / d phi_b
      s[7 + c] += k2 * (1.0f - q.Eb[c]);                     // d I / d phi_inf
      s[10 + c] += q.rgb[c];
      const float e = fmaxf(fabsf(q.rgb[c]) - 0.7f, 0.0f);
      s[13] += e * e;
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NRED; ++k) {
    const float t = osm::wave_sum(s[k]);
    if (lane == 0) red[wv][k] = t;
  }
  __syncthreads();
  if (threadIdx.x < NRED) {
    const int k = threadIdx.x;
    part[((long long)b * nblk + blk) * NRED + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
  }
}

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__global__ void phys_finalize_kernel(osm_phys_desc ds, const float* __restrict__ part, float* __restrict__ red,
                                     float* __restrict__ phi, int do_update, float* __restrict__ loss_out,
                                     float* __restrict__ opt_state, int nblk) {
  __shared__ double tot[NRED];
  const int b = blockIdx.x;
  {   // one wave: component lane >> 2, four lanes share its nblk partials (fixed order: deterministic), fp64, two shuffle folds
    //  (a single lane per component walked 64 dependent loads: 10.6 us per launch, 21 launches per step)
    static_assert(NRED * 4 == 64, "one wave covers the components");
    const int comp = threadIdx.x >> 2, sub = threadIdx.x & 3;
    double a = 0.0;
#pragma unroll 4
    for (int k = sub; k < nblk; k += 4) a += (double)part[((long long)b * nblk + k) * NRED + comp];
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    if (sub == 0) {
      tot[comp] = a;
      red[b * NRED + comp] = (float)a;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double n = 3.0 * (double)ds.HW;
    double L, gscale;
    if (ds.loss_type == 0) {
      L = sqrt(tot[0]);
      gscale = 1.0 / L;
    } else {
      L = tot[0] / n;
      gscale = 2.0 / n;
    }
    if (loss_out) loss_out[b] = (float)L;
    if (do_update) {
      float* ph = phi + b * 9;
      // gradient of every live parameter (as the reference's autograd leaves have it), then the optimizer step
      float g[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float lr[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      int live = 0;                      // bit i: phi[i] is a parameter of this operator
      if (ds.kind == 0) {
        for (int c = 0; c < 3; ++c) {
          g[c] = (float)(tot[1 + c] * gscale); lr[c] = ds.eta[0];
          g[3 + c] = (float)(tot[4 + c] * gscale); lr[3 + c] = ds.eta[1];
        }
        live = 0x3f;
      } else if (ds.kind == 1) {
        for (int c = 0; c < 3; ++c) { g[c] = (float)((tot[1 + c] + tot[4 + c]) * gscale); lr[c] = ds.eta[0]; }
        live = 0x7;
      } else {
        double gs = 0.0;
        for (int c = 0; c < 3; ++c) gs += tot[1 + c] + tot[4 + c];
        g[0] = (float)(gs * gscale); lr[0] = ds.eta[0];
        live = 0x1;
      }
      for (int c = 0; c < 3; ++c) { g[6 + c] = (float)(tot[7 + c] * gscale); lr[6 + c] = ds.eta[2]; }
      live |= 0x1c0;
      // torch.optim with its DEFAULT hyper-parameters (utils.py:494-524 passes none), single-tensor path, fp32 state, one parameter
      // group per phi with lr = eta (measurements.py:132-136, 244-249).  st: [B][20] floats, layout per optimizer below.  A parameter
      // with learn_flag False has no gradient: the optimizer skips it (its state stays untouched).
      float* st = opt_state ? opt_state + b * 20 : nullptr;
      if (ds.optimizer == 1 || ds.optimizer == 2) {          // Adam | AdamW (weight_decay 0.01, decoupled): exp_avg[9] | exp_avg_sq[9] | step
        const float step = st[18] + 1.f;
        st[18] = step;
        const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
        const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
        const float bc2_sqrt = (float)sqrt(bc2);
        for (int i = 0; i < 9; ++i) {
          if (!((live >> i) & 1) || lr[i] == 0.f) continue;
          if (ds.optimizer == 2) ph[i] *= (float)(1.0 - (double)lr[i] * 0.01);      // param.mul_(1 - lr * weight_decay)
          const float m = st[i] + (g[i] - st[i]) * (float)(1.0 - b1);             // exp_avg.lerp_(grad, 1 - beta1)
          const float v = st[9 + i] * (float)b2 + (float)(1.0 - b2) * g[i] * g[i];  // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
          st[i] = m; st[9 + i] = v;
          const float denom = sqrtf(v) / bc2_sqrt + (float)eps;
          ph[i] += (float)(-(double)lr[i] / bc1) * (m / denom);                   // param.addcdiv_(exp_avg, denom, value=-step_size)
        }
      } else if (ds.optimizer == 3) {   // Adamax (betas 0.9 / 0.999, eps 1e-8): exp_avg[9] | exp_inf[9] | step
        const float step = st[18] + 1.f;
        st[18] = step;
        const double bc = 1.0 - pow(0.9, (double)step);
        for (int i = 0; i < 9; ++i) {
          if (!((live >> i) & 1) || lr[i] == 0.f) continue;
          const float m = st[i] + (g[i] - st[i]) * (float)(1.0 - 0.9);            // exp_avg.lerp_(grad, 1 - beta1)
          const float u = fmaxf(st[9 + i] * (float)0.999, fabsf(g[i]) + (float)1e-8);   // max(exp_inf * beta2, |grad| + eps)
          st[i] = m; st[9 + i] = u;
          ph[i] += (float)(-(double)lr[i] / bc) * (m / u);                         // param.addcdiv_(exp_avg, exp_inf, value=-clr)
        }
      } else if (ds.optimizer == 4) {   // RMSprop (alpha 0.99, eps 1e-8, no momentum, not centered): square_avg[9]
        for (int i = 0; i < 9; ++i) {
          if (!((live >> i) & 1) || lr[i] == 0.f) continue;
          const float sq = st[i] * (float)0.99 + (float)(1.0 - 0.99) * g[i] * g[i];
          st[i] = sq;
          ph[i] += -lr[i] * (g[i] / (sqrtf(sq) + (float)1e-8));
        }
      } else if (ds.optimizer == 5) {   // Adagrad (lr_decay 0, initial accumulator 0, eps 1e-10): sum[9]
        for (int i = 0; i < 9; ++i) {
          if (!((live >> i) & 1) || lr[i] == 0.f) continue;
          const float sm = st[i] + g[i] * g[i];
          st[i] = sm;
          ph[i] += -lr[i] * (g[i] / (sqrtf(sm) + (float)1e-10));
        }
      } else if (ds.optimizer == 6) {   // Adadelta (rho 0.9, eps 1e-6; lr = eta): square_avg[9] | acc_delta[9]
        for (int i = 0; i < 9; ++i) {
          if (!((live >> i) & 1) || lr[i] == 0.f) continue;
          const float sq = st[i] * (float)0.9 + (float)(1.0 - 0.9) * g[i] * g[i];
          const float sd = sqrtf(sq + (float)1e-6);
          const float dl = sqrtf(st[9 + i] + (float)1e-6) / sd * g[i];
          st[i] = sq;
          st[9 + i] = st[9 + i] * (float)0.9 + (float)(1.0 - 0.9) * dl * dl;
          ph[i] += -lr[i] * dl;
        }
      } else if (ds.optimizer == 7) {   // ASGD (lambd 1e-4, alpha 0.75, t0 1e6): eta[9] | - | step (19); eta starts at lr (0 = not yet set)
        const float step = st[19] + 1.f;
        st[19] = step;
        for (int i = 0; i < 9; ++i) {
          if (!((live >> i) & 1) || lr[i] == 0.f) continue;
          const float eta = step == 1.f ? lr[i] : st[i];
          ph[i] *= (float)(1.0 - 1e-4 * (double)eta);                             // param.mul_(1 - lambd * eta)
          ph[i] += -eta * g[i];                                                    // param.add_(grad, alpha=-eta)
          st[i] = (float)((double)lr[i] / pow(1.0 + 1e-4 * (double)lr[i] * (double)step, 0.75));
        }                                                                          // (the averaged iterate `ax` is not what the operator reads)
      } else if (ds.optimizer == 8) {   // Rprop (etas 0.5 / 1.2, step sizes 1e-6 .. 50): prev[9] | step_size[9] | step
        const float step = st[18] + 1.f;
        st[18] = step;
        for (int i = 0; i < 9; ++i) {
          if (!((live >> i) & 1) || lr[i] == 0.f) continue;
          float ss = step == 1.f ? lr[i] : st[9 + i];
          const float pr = g[i] * st[i];
          float gi = g[i];
          ss *= pr > 0.f ? 1.2f : (pr < 0.f ? 0.5f : 1.f);
          ss = fminf(fmaxf(ss, 1e-6f), 50.f);
          if (pr < 0.f) gi = 0.f;
          ph[i] += -(sgn(gi) * ss);
          st[i] = gi; st[9 + i] = ss;
        }
      } else {
        for (int i = 0; i < 9; ++i)
          if ((live >> i) & 1) ph[i] -= lr[i] * g[i];
      }
      if (ds.kind == 2) ph[1] = ph[2] = ph[0];
    }
  }
}

__global__ __launch_bounds__(256) void phys_grad_kernel(osm_phys_desc ds, const float* __restrict__ x0,
                                                         const float* __restrict__ y,
                                                         const float* __restrict__ phi,
                                                         const float* __restrict__ red, float* __restrict__ g) {
  const int b = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= ds.HW) return;
  Pix q;
  eval_pixel(ds, x0, y, phi, b, p, q);
  const float* rd = red + b * NRED;
  const float n = 3.0f * (float)ds.HW;
  const float gscale = ds.loss_type == 0 ? 1.0f / sqrtf(rd[0]) : 2.0f / n;
  const float* ph = phi + b * 9;
  float gD = 0.f;
  const long long base = (long long)b * 4 * ds.HW + p;
  if (ds.kind == 3) {   // d ||y - x0[:, 0:3]|| / d x0 = -(y - x0) / ||.|| on the colour channels, nothing on depth
#pragma unroll
    for (int c = 0; c < 3; ++c) g[base + (long long)c * ds.HW] = -(q.r[c] * gscale);
    g[base + 3LL * ds.HW] = 0.f;
    return;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float pa = ds.kind == 2 ? ph[0] : ph[c];
    const float pb = ds.kind == 0 ? ph[3 + c] : pa;
    const float pinf = ph[6 + c];
    const float dLdI = -2.0f * q.w * (q.r[c] * gscale);
    float grgb = dLdI * 0.5f * q.Ea[c];
    if (ds.gamma_avrg != 0.f) grgb += ds.gamma_avrg * sgn(rd[10 + c]) / (float)ds.HW;
    if (ds.gamma_val != 0.f) {
      const float e = fmaxf(fabsf(q.rgb[c]) - 0.7f, 0.0f);
      grgb += ds.gamma_val * 2.0f * e * sgn(q.rgb[c]) / n;
    }
    g[base + (long long)c * ds.HW] = grgb;
    gD += dLdI * (-pa * q.J[c] * q.Ea[c] + pinf * pb * q.Eb[c]) * q.dd;
  }
  g[base + 3LL * ds.HW] = gD;
}

// ---------------------------------------------------------------- posterior
// MK: mean processor (0 epsilon / start_x / previous_x through ONE form, x0 = c0 x - c1 out: the row holds
//     c0 = d x0 / d x and c1 = -d x0 / d out, which is also what posterior_bwd_kernel and the update kernels read;
//     1 start_x, x0 = out exactly; 2 previous_x, mean = out exactly).  VK: variance processor (0 learned_range,
//     1 fixed_small / fixed_large: the row's value, 2 learned: the network's second half).
// x0_raw != nullptr: clip_denoised (process_xstart, posterior_mean_variance.py:43-50): the unclamped prediction goes to x0_raw
// (clamp_bwd_kernel masks the guidance gradient with it), x0 = clamp(x0_raw, -1, 1) and the mean is formed from the clamped x0.
template <int MK, int VK>
__global__ __launch_bounds__(256) void posterior_kernel(const float* __restrict__ mo, const float* __restrict__ x,
                                                         const float* __restrict__ coef, float* __restrict__ x0_raw,
                                                         float* __restrict__ x0, float* __restrict__ mean,
                                                         float* __restrict__ logvar, int B, int HW) {
  const long long total = (long long)B * 4 * HW;
  const float c0 = coef[0], c1 = coef[1], c2 = coef[2], c3 = coef[3], mn = coef[4], mxl = coef[5];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / (4LL * HW);
    const long long rem = i - b * 4LL * HW;
    const float eps = mo[b * 8LL * HW + rem];
    const float xv = x[i];
    float xs = MK == 1 ? eps : c0 * xv - c1 * eps;
    if (x0_raw) {
      x0_raw[i] = xs;
      xs = (xs != xs) ? xs : fminf(fmaxf(xs, -1.0f), 1.0f);          // torch.clamp keeps NaN
    }
    x0[i] = xs;
    mean[i] = MK == 2 ? eps : c2 * xs + c3 * xv;
    if (VK == 1) {
      logvar[i] = mn;
    } else {
      const float v = mo[b * 8LL * HW + 4LL * HW + rem];
      if (VK == 2) {
        logvar[i] = v;
      } else {
        const float frac = (v + 1.0f) / 2.0f;
        logvar[i] = frac * mxl + (1.0f - frac) * mn;
      }
    }
  }
}

// backward of x.clamp(lo, hi) (ATen clamp_backward: the gradient passes where lo <= x <= hi, bounds included; NaN -> 0)
__global__ __launch_bounds__(256) void clamp_bwd_kernel(float* __restrict__ g, const float* __restrict__ x_raw, float lo, float hi,
                                                         long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = x_raw[i];
    if (!(v >= lo && v <= hi)) g[i] = 0.f;
  }
}

__global__ __launch_bounds__(256) void posterior_bwd_kernel(const float* __restrict__ g,
                                                             const float* __restrict__ coef,
                                                             float* __restrict__ d_out, int B, int HW) {
  const long long total = (long long)B * 8 * HW;
  const float c1 = coef[1];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / (8LL * HW);
    const long long rem = i - b * 8LL * HW;
    d_out[i] = rem < 4LL * HW ? -c1 * g[b * 4LL * HW + rem] : 0.f;
  }
}

__global__ __launch_bounds__(256) void guide_update_kernel(const float* __restrict__ mean,
                                                            const float* __restrict__ logvar,
                                                            const float* __restrict__ g,
                                                            const float* __restrict__ dxu,
                                                            const float* __restrict__ noise,
                                                            const float* __restrict__ coef,
                                                            const float* __restrict__ scale4, float clip,
                                                            float* __restrict__ x_next,
                                                            float* __restrict__ grad_out, int B, int HW) {
  const long long total = (long long)B * 4 * HW;
  const float c0 = coef[0], noise_on = coef[6];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % 4);
    float grad = 0.f;
    if (g) grad = c0 * g[i] + (dxu ? dxu[i] : 0.f);
    if (grad_out) grad_out[i] = grad;
    float gc = grad;
    if (clip >= 0.f) gc = (grad != grad) ? grad : fminf(fmaxf(grad, -clip), clip);   // torch.clamp keeps NaN (fminf/fmaxf would drop it)
    float xt = mean[i] - (g ? scale4[c] * gc : 0.f);
    if (noise_on != 0.f && noise) xt += expf(0.5f * logvar[i]) * noise[i];
    x_next[i] = xt;
  }
}

// unconditional ancestral step of the RGBD prior sampler (reference osmosis_utils/diffusion.py:94-122):
//   eps = model_out[:, :C] ; x_next = c_a (x - c_b eps) + c_s z ; x0 = c_r x - c_m eps
// coef = {c_a, c_b, c_s, c_r, c_m, -, -, t}
__global__ __launch_bounds__(256) void ancestral_step_kernel(const float* __restrict__ mo,
                                                              const float* x,   // may alias x_next (in place)
                                                              const float* __restrict__ z,
                                                              const float* __restrict__ coef,
                                                              float* x_next, float* __restrict__ x0,
                                                              int B, int C, int Cout, int HW) {
  const long long total = (long long)B * C * HW;
  const float ca = coef[0], cb = coef[1], cs = coef[2], cr = coef[3], cm = coef[4];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / ((long long)C * HW);
    const long long rem = i - b * (long long)C * HW;
    const float eps = mo[b * (long long)Cout * HW + rem];
    const float xv = x[i];
    if (x0) x0[i] = cr * xv - cm * eps;
    float v = ca * (xv - cb * eps);
    if (z) v += cs * z[i];
    x_next[i] = v;
  }
}


// ---------------------------------------------------------------- step noise inside the library (gaussian_diffusion.py:266-268)
// Philox-4x32-10 (Salmon et al., SC'11; the Random123 constants), counter = (element quad, image, step, stream id), key = the
// 64-bit seed: one counter gives the four normals of four consecutive elements of one image (two Box-Muller pairs), so the
// noise of an image depends on (seed, image index, step) only -- not on the batch it is processed in, its chunking or the grid.
struct Philox4 { unsigned v[4]; };
__device__ __forceinline__ Philox4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{{c0, c1, c2, c3}};
}
// (0, 1): 24 random bits, centred in their 2^-24 cell
__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * 5.9604644775390625e-8f + 2.98023223876953125e-8f; }
__device__ __forceinline__ void normal4(const Philox4& r, float z[4]) {
  const float r0 = sqrtf(-2.0f * logf(u01(r.v[0]))), r1 = sqrtf(-2.0f * logf(u01(r.v[2])));
  float s0, c0, s1, c1;
  sincosf(6.283185307179586f * u01(r.v[1]), &s0, &c0);
  sincosf(6.283185307179586f * u01(r.v[3]), &s1, &c1);
  z[0] = r0 * c0; z[1] = r0 * s0; z[2] = r1 * c1; z[3] = r1 * s1;
}
constexpr unsigned OSM_RNG_STREAM_STEP_NOISE = 0x6f736d31u;   // "osm1": the per-step noise of a chain

__global__ __launch_bounds__(256) void philox_raw_kernel(unsigned* __restrict__ out, long long n4, unsigned c1, unsigned c2,
                                                          unsigned c3, unsigned k0, unsigned k1) {
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long long)gridDim.x * blockDim.x) {
    const Philox4 r = philox4x32_10((unsigned)q, c1, c2, c3, k0, k1);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[4 * q + e] = r.v[e];
  }
}

// out[b][0..n) ~ N(0, 1): image b uses counter word 1 = img0 + b, word 2 = the step (from the device counter when given)
__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, int B, long long n, unsigned k0, unsigned k1,
                                                     const int* __restrict__ step_dev, int step_const, int img0, int img_stride) {
  const unsigned step = (unsigned)(step_dev ? *step_dev : step_const);
  const long long nq = (n + 3) >> 2;
  const long long total = (long long)B * nq;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / nq, q = i - b * nq;
    float z[4];
    normal4(philox4x32_10((unsigned)q, (unsigned)(img0 + (int)b * img_stride), step, OSM_RNG_STREAM_STEP_NOISE, k0, k1), z);
    float* o = out + b * n + 4 * q;
    if (4 * q + 3 < n) *reinterpret_cast<float4*>(o) = make_float4(z[0], z[1], z[2], z[3]);    // (n % 4 == 0 and out 16-byte aligned: checked by the host)
    else for (int e = 0; e < 4 && 4 * q + e < n; ++e) o[e] = z[e];
  }
}

// guide_update with the noise drawn in the kernel: four consecutive elements of one image per thread (4 HW % 4 == 0)
__global__ __launch_bounds__(256) void guide_update_rng_kernel(const float* __restrict__ mean, const float* __restrict__ logvar,
                                                                const float* __restrict__ g, const float* __restrict__ dxu,
                                                                const float* __restrict__ coef, const float* __restrict__ scale4,
                                                                float clip, float* __restrict__ x_next, float* __restrict__ grad_out,
                                                                float* __restrict__ noise_out, int B, int HW, unsigned k0, unsigned k1,
                                                                const int* __restrict__ step_dev, int step_offset, int img0,
                                                                int img_stride) {
  const long long nq = (long long)HW;             // 4 HW elements per image = HW quads
  const long long total = (long long)B * nq;
  const float c0 = coef[0], noise_on = coef[6];
  const unsigned step = (unsigned)(*step_dev + step_offset);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / nq, q = i - b * nq;
    const long long e0 = b * 4LL * HW + 4 * q;
    const int c = (int)((4 * q) / HW);            // HW % 4 == 0: a quad never straddles two channels
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (noise_on != 0.f) normal4(philox4x32_10((unsigned)q, (unsigned)(img0 + (int)b * img_stride), step, OSM_RNG_STREAM_STEP_NOISE, k0, k1), z);
    const float4 m = *reinterpret_cast<const float4*>(mean + e0);
    const float4 lv = *reinterpret_cast<const float4*>(logvar + e0);
    float4 gv = make_float4(0.f, 0.f, 0.f, 0.f), du = gv;
    if (g) gv = *reinterpret_cast<const float4*>(g + e0);
    if (g && dxu) du = *reinterpret_cast<const float4*>(dxu + e0);
    const float mm[4] = {m.x, m.y, m.z, m.w}, ll[4] = {lv.x, lv.y, lv.z, lv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w},
                dd[4] = {du.x, du.y, du.z, du.w};
    float xo[4], go[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float grad = g ? c0 * gg[e] + dd[e] : 0.f;
      go[e] = grad;
      float gc = grad;
      if (clip >= 0.f) gc = (grad != grad) ? grad : fminf(fmaxf(grad, -clip), clip);
      float xt = mm[e] - (g ? scale4[c] * gc : 0.f);
      if (noise_on != 0.f) xt += expf(0.5f * ll[e]) * z[e];
      xo[e] = xt;
    }
    *reinterpret_cast<float4*>(x_next + e0) = make_float4(xo[0], xo[1], xo[2], xo[3]);
    if (grad_out) *reinterpret_cast<float4*>(grad_out + e0) = make_float4(go[0], go[1], go[2], go[3]);
    if (noise_out) *reinterpret_cast<float4*>(noise_out + e0) = make_float4(z[0], z[1], z[2], z[3]);
  }
}

// DDIM step + guidance (gaussian_diffusion.py:505-535, condition_methods.py:247-251), in the reference's operation order:
//   eps = (r0 x - x0) / r1 ; sigma = eta sqrt((1 - abp) / (1 - ab)) sqrt(1 - ab / abp) ;
//   x_next = x0 sqrt(abp) + sqrt(1 - abp - sigma^2) eps + [t != 0] sigma noise - scale[c] clamp(grad) ; grad = c0 g + dx_unet
// coef = the posterior row (c0 = d x0 / d x of the mean processor), dcoef = {alpha_bar, alpha_bar_prev, eta, noise_on,
// r0 = sqrt_recip_ac, r1 = sqrt_recipm1_ac}: predict_eps_from_x_start (:533-536) uses the SAMPLER's tables whatever the mean
// processor is (for `epsilon` r0 = c0 and r1 = c1).
// x_next may alias x (each element is read, then written, by one thread).
__global__ __launch_bounds__(256) void ddim_update_kernel(const float* __restrict__ x0, const float* x, const float* __restrict__ g,
                                                           const float* __restrict__ dxu, const float* __restrict__ noise,
                                                           const float* __restrict__ coef, const float* __restrict__ dcoef,
                                                           const float* __restrict__ scale4, float clip, float* x_next,
                                                           float* __restrict__ grad_out, int B, int HW) {
  const long long total = (long long)B * 4 * HW;
  const float c0 = coef[0];
  const float ab = dcoef[0], abp = dcoef[1], eta = dcoef[2], noise_on = dcoef[3], r0 = dcoef[4], r1 = dcoef[5];
  const float sigma = eta * sqrtf((1.0f - abp) / (1.0f - ab)) * sqrtf(1.0f - ab / abp);
  const float sa = sqrtf(abp), sb = sqrtf(1.0f - abp - sigma * sigma);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)((i / HW) % 4);
    const float xs = x0[i];
    const float eps = (r0 * x[i] - xs) / r1;
    float xt = xs * sa + sb * eps;
    if (noise_on != 0.f && noise) xt += sigma * noise[i];
    float grad = 0.f;
    if (g) grad = c0 * g[i] + (dxu ? dxu[i] : 0.f);
    if (grad_out) grad_out[i] = grad;
    float gc = grad;
    if (clip >= 0.f) gc = (grad != grad) ? grad : fminf(fmaxf(grad, -clip), clip);
    x_next[i] = xt - (g ? scale4[c] * gc : 0.f);
  }
}

__global__ void fetch_coefs_kernel(const float* __restrict__ table, int* __restrict__ step, int delta,
                                   float* __restrict__ coef_out, float* __restrict__ t_out, int B, int n_rows) {
  const int s = min(max(*step, 0), n_rows - 1);   // a counter that ran off the table re-reads its last row
  if (threadIdx.x < 8) coef_out[threadIdx.x] = table[s * 8 + threadIdx.x];
  if (threadIdx.x < B) t_out[threadIdx.x] = table[s * 8 + 7];
  __syncthreads();
  if (threadIdx.x == 0) *step = s + delta;
}

inline int grid_for(long long total) {
  long long b = (total + 255) / 256;
  if (b > 2048) b = 2048;
  return (int)(b < 1 ? 1 : b);
}

int check_desc(const osm_phys_desc* d, const char* who) {
  OSM_REQUIRE(d, "%s: null descriptor", who);
  OSM_REQUIRE(d->kind >= 0 && d->kind <= 3, "%s: unknown operator kind %d", who, d->kind);
  OSM_REQUIRE(d->kind != 3 || (d->loss_type == 0 && d->weight_type == 0 && d->gamma_avrg == 0.f && d->gamma_val == 0.f),
              "%s: the identity operator (kind 3) is the plain norm loss: no weight, no auxiliary losses", who);
  OSM_REQUIRE(d->depth_type >= 0 && d->depth_type <= 2, "%s: unknown depth_type %d", who, d->depth_type);
  OSM_REQUIRE(d->loss_type == 0 || d->loss_type == 1, "%s: unknown loss_type %d", who, d->loss_type);
  OSM_REQUIRE(d->B > 0 && d->HW > 0, "%s: bad shape", who);
  return OSM_OK;
}

}  // namespace

extern "C" int osm_phys_nblk(int HW) { return (HW + PPB - 1) / PPB; }

extern "C" int osm_phys_reduce(const osm_phys_desc* d, const float* x0, const float* y, const float* phi,
                               float* part, void* stream) {
  int rc = check_desc(d, "osm_phys_reduce");
  if (rc) return rc;
  OSM_REQUIRE(x0 && y && phi && part, "osm_phys_reduce: null pointer");
  const int nblk = osm_phys_nblk(d->HW);
  hipLaunchKernelGGL(phys_reduce_kernel, dim3(nblk, d->B), dim3(256), 0, (hipStream_t)stream, *d, x0, y, phi,
                     part, nblk);
  return osm::check_launch("phys_reduce_kernel");
}

extern "C" int osm_phys_finalize(const osm_phys_desc* d, const float* part, float* red, float* phi,
                                 int do_update, float* loss_out, float* opt_state, void* stream) {
  int rc = check_desc(d, "osm_phys_finalize");
  if (rc) return rc;
  OSM_REQUIRE(part && red && phi, "osm_phys_finalize: null pointer");
  OSM_REQUIRE(d->optimizer >= 0 && d->optimizer <= 8, "osm_phys_finalize: optimizer must be 0 (sgd / GD) .. 8 (see osm_phys_desc)");
  OSM_REQUIRE(!(d->optimizer != 0 && do_update) || opt_state, "osm_phys_finalize: a stateful optimizer needs opt_state [B][20]");
  OSM_REQUIRE(!(d->kind == 3 && do_update), "osm_phys_finalize: the identity operator (kind 3) has no parameters to step");
  hipLaunchKernelGGL(phys_finalize_kernel, dim3(d->B), dim3(64), 0, (hipStream_t)stream, *d, part, red, phi,
                     do_update, loss_out, opt_state, osm_phys_nblk(d->HW));
  return osm::check_launch("phys_finalize_kernel");
}

extern "C" int osm_phys_grad(const osm_phys_desc* d, const float* x0, const float* y, const float* phi,
                             const float* red, float* g, void* stream) {
  int rc = check_desc(d, "osm_phys_grad");
  if (rc) return rc;
  OSM_REQUIRE(x0 && y && phi && red && g, "osm_phys_grad: null pointer");
  hipLaunchKernelGGL(phys_grad_kernel, dim3((d->HW + 255) / 256, d->B), dim3(256), 0, (hipStream_t)stream, *d,
                     x0, y, phi, red, g);
  return osm::check_launch("phys_grad_kernel");
}

// The inner phi optimisation + dL/dx0 of one guided step, enqueued by ONE call (measurements.py:266-303, condition_methods.py:109-144):
//   n_inner x { reduce; finalize (+ phi step) }, with the loss and the x0-gradient taken at the phi of the LAST iteration, which is
//   stepped afterwards (unless freeze_phi: then n_inner = 1 and phi stays).  The same launches as the three entry points above in the
//   same order -- from C the 2 n_inner + 2 launches cost the host ~2 us each instead of a Python call each (measured: the GPU idled
//   0.24 ms per step between them).
extern "C" int osm_phys_optimize(const osm_phys_desc* d, const float* x0, const float* y, float* phi, float* part, float* red,
                                 float* loss_out, float* g, int n_inner, int freeze_phi, float* opt_state, void* stream) {
  int rc = check_desc(d, "osm_phys_optimize");
  if (rc) return rc;
  OSM_REQUIRE(x0 && y && phi && part && red && loss_out && g, "osm_phys_optimize: null pointer");
  OSM_REQUIRE(n_inner >= 1, "osm_phys_optimize: n_inner must be >= 1");
  OSM_REQUIRE(!(freeze_phi && n_inner != 1), "osm_phys_optimize: freeze_phi goes with n_inner = 1");
  for (int it = 0; it < n_inner; ++it) {
    if ((rc = osm_phys_reduce(d, x0, y, phi, part, stream))) return rc;
    if (it == n_inner - 1) {
      if ((rc = osm_phys_finalize(d, part, red, phi, 0, loss_out, nullptr, stream))) return rc;
      if ((rc = osm_phys_grad(d, x0, y, phi, red, g, stream))) return rc;
      if (!freeze_phi && (rc = osm_phys_finalize(d, part, red, phi, 1, nullptr, opt_state, stream))) return rc;
    } else if ((rc = osm_phys_finalize(d, part, red, phi, 1, loss_out, opt_state, stream))) {
      return rc;
    }
  }
  return OSM_OK;
}

extern "C" int osm_posterior_typed(const float* model_out, const float* x, const float* coef, int mean_kind, int var_kind,
                                   int clip_denoised, float* x0_raw, float* x0, float* mean, float* logvar, int B, int HW,
                                   void* stream) {
  OSM_REQUIRE(model_out && x && coef && x0 && mean && logvar && B > 0 && HW > 0, "osm_posterior_typed: bad argument");
  OSM_REQUIRE(!clip_denoised || x0_raw, "osm_posterior_typed: clip_denoised needs x0_raw (the unclamped prediction, read by osm_clamp_bwd)");
  if (!clip_denoised) x0_raw = nullptr;
  OSM_REQUIRE(mean_kind >= 0 && mean_kind <= 2, "osm_posterior_typed: mean_kind must be 0 (epsilon), 1 (start_x) or 2 (previous_x)");
  OSM_REQUIRE(var_kind >= 0 && var_kind <= 2, "osm_posterior_typed: var_kind must be 0 (learned_range), 1 (fixed) or 2 (learned)");
  const dim3 grid(grid_for((long long)B * 4 * HW)), block(256);
  hipStream_t st = (hipStream_t)stream;
#define OSM_POST(MK, VK) \
  hipLaunchKernelGGL((posterior_kernel<MK, VK>), grid, block, 0, st, model_out, x, coef, x0_raw, x0, mean, logvar, B, HW)
  switch (mean_kind * 3 + var_kind) {
    case 0: OSM_POST(0, 0); break;
    case 1: OSM_POST(0, 1); break;
    case 2: OSM_POST(0, 2); break;
    case 3: OSM_POST(1, 0); break;
    case 4: OSM_POST(1, 1); break;
    case 5: OSM_POST(1, 2); break;
    case 6: OSM_POST(2, 0); break;
    case 7: OSM_POST(2, 1); break;
    default: OSM_POST(2, 2); break;
  }
#undef OSM_POST
  return osm::check_launch("posterior_kernel");
}

extern "C" int osm_posterior(const float* model_out, const float* x, const float* coef, float* x0, float* mean,
                             float* logvar, int B, int HW, void* stream) {
  return osm_posterior_typed(model_out, x, coef, 0, 0, 0, nullptr, x0, mean, logvar, B, HW, stream);
}

extern "C" int osm_clamp_bwd(float* g, const float* x_raw, float lo, float hi, long long n, void* stream) {
  OSM_REQUIRE(g && x_raw && n > 0 && lo <= hi, "osm_clamp_bwd: bad argument");
  hipLaunchKernelGGL(clamp_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, x_raw, lo, hi, n);
  return osm::check_launch("clamp_bwd_kernel");
}

extern "C" int osm_posterior_bwd(const float* g, const float* coef, float* d_out, int B, int HW, void* stream) {
  OSM_REQUIRE(g && coef && d_out && B > 0 && HW > 0, "osm_posterior_bwd: bad argument");
  hipLaunchKernelGGL(posterior_bwd_kernel, dim3(grid_for((long long)B * 8 * HW)), dim3(256), 0,
                     (hipStream_t)stream, g, coef, d_out, B, HW);
  return osm::check_launch("posterior_bwd_kernel");
}

extern "C" int osm_guide_update(const float* mean, const float* logvar, const float* g, const float* dx_unet,
                                const float* noise, const float* coef, const float* scale4, float clip,
                                float* x_next, float* grad_out, int B, int HW, void* stream) {
  OSM_REQUIRE(mean && logvar && coef && x_next && B > 0 && HW > 0, "osm_guide_update: bad argument");
  OSM_REQUIRE(!g || scale4, "osm_guide_update: guidance needs the per-channel scale");
  hipLaunchKernelGGL(guide_update_kernel, dim3(grid_for((long long)B * 4 * HW)), dim3(256), 0,
                     (hipStream_t)stream, mean, logvar, g, dx_unet, noise, coef, scale4, clip, x_next, grad_out, B,
                     HW);
  return osm::check_launch("guide_update_kernel");
}


extern "C" int osm_guide_update_rng(const float* mean, const float* logvar, const float* g, const float* dx_unet,
                                    const float* coef, const float* scale4, float clip, float* x_next, float* grad_out,
                                    float* noise_out, int B, int HW, unsigned long long seed, const int* step, int step_offset,
                                    int img0, int img_stride, void* stream) {
  OSM_REQUIRE(mean && logvar && coef && x_next && step && B > 0 && HW > 0, "osm_guide_update_rng: bad argument");
  OSM_REQUIRE(!g || scale4, "osm_guide_update_rng: guidance needs the per-channel scale");
  OSM_REQUIRE(HW % 4 == 0, "osm_guide_update_rng: H*W must be a multiple of 4 (one Philox counter per four elements)");
  OSM_REQUIRE(((reinterpret_cast<size_t>(mean) | reinterpret_cast<size_t>(logvar) | reinterpret_cast<size_t>(g) |
                reinterpret_cast<size_t>(dx_unet) | reinterpret_cast<size_t>(x_next) | reinterpret_cast<size_t>(grad_out) |
                reinterpret_cast<size_t>(noise_out)) & 15) == 0, "osm_guide_update_rng: tensors must be 16-byte aligned");
  hipLaunchKernelGGL(guide_update_rng_kernel, dim3(grid_for((long long)B * HW)), dim3(256), 0, (hipStream_t)stream, mean, logvar,
                     g, dx_unet, coef, scale4, clip, x_next, grad_out, noise_out, B, HW, (unsigned)(seed & 0xffffffffull),
                     (unsigned)(seed >> 32), step, step_offset, img0, img_stride);
  return osm::check_launch("guide_update_rng_kernel");
}

extern "C" int osm_randn(float* out, int B, long long n, unsigned long long seed, const int* step_dev, int step_const, int img0,
                         int img_stride, void* stream) {
  OSM_REQUIRE(out && B > 0 && n > 0, "osm_randn: bad argument");
  OSM_REQUIRE((reinterpret_cast<size_t>(out) & 15) == 0, "osm_randn: out must be 16-byte aligned");
  OSM_REQUIRE(n % 4 == 0 || B == 1, "osm_randn: a batch needs n %% 4 == 0 (every image's row starts 16-byte aligned)");
  hipLaunchKernelGGL(randn_kernel, dim3(grid_for((long long)B * ((n + 3) / 4))), dim3(256), 0, (hipStream_t)stream, out, B, n,
                     (unsigned)(seed & 0xffffffffull), (unsigned)(seed >> 32), step_dev, step_const, img0, img_stride);
  return osm::check_launch("randn_kernel");
}

extern "C" int osm_philox_raw(unsigned* out, long long n4, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                              void* stream) {
  OSM_REQUIRE(out && n4 > 0, "osm_philox_raw: bad argument");
  hipLaunchKernelGGL(philox_raw_kernel, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, out, n4, c1, c2, c3, k0, k1);
  return osm::check_launch("philox_raw_kernel");
}

extern "C" int osm_ddim_update(const float* x0, const float* x, const float* g, const float* dx_unet, const float* noise,
                               const float* coef, const float* dcoef, const float* scale4, float clip, float* x_next,
                               float* grad_out, int B, int HW, void* stream) {
  OSM_REQUIRE(x0 && x && coef && dcoef && x_next && B > 0 && HW > 0, "osm_ddim_update: bad argument");
  OSM_REQUIRE(!g || scale4, "osm_ddim_update: guidance needs the per-channel scale");
  hipLaunchKernelGGL(ddim_update_kernel, dim3(grid_for((long long)B * 4 * HW)), dim3(256), 0, (hipStream_t)stream, x0, x, g,
                     dx_unet, noise, coef, dcoef, scale4, clip, x_next, grad_out, B, HW);
  return osm::check_launch("ddim_update_kernel");
}

extern "C" int osm_fetch_coefs(const float* table, int n_rows, int* step, int delta, float* coef_out, float* t_out,
                               int B, void* stream) {
  OSM_REQUIRE(table && step && coef_out && t_out && B > 0 && B <= 256 && n_rows > 0, "osm_fetch_coefs: bad argument");
  hipLaunchKernelGGL(fetch_coefs_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, table, step, delta, coef_out,
                     t_out, B, n_rows);
  return osm::check_launch("fetch_coefs_kernel");
}

extern "C" int osm_ancestral_step(const float* model_out, const float* x, const float* z, const float* coef,
                                  float* x_next, float* x0, int B, int C, int Cout, int HW, void* stream) {
  OSM_REQUIRE(model_out && x && coef && x_next && B > 0 && C > 0 && Cout >= C && HW > 0, "osm_ancestral_step: bad argument");
  hipLaunchKernelGGL(ancestral_step_kernel, dim3(grid_for((long long)B * C * HW)), dim3(256), 0, (hipStream_t)stream,
                     model_out, x, z, coef, x_next, x0, B, C, Cout, HW);
  return osm::check_launch("ancestral_step_kernel");
}
