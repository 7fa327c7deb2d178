"""Samplers -- registry surface of the reference's guided_diffusion/gaussian_diffusion.py
(register_sampler/get_sampler/create_sampler :19-62, GaussianDiffusion :65-370, space_timesteps
:373-426, SpacedDiffusion :429-474, DDPM :492-502, DDIM :505-535, get_named_beta_schedule :542-566).

`p_sample_loop` keeps the reference signature and return values.  For the Osmosis configuration
(pretrain_model == 'osmosis', our UNetModel, the 'osmosis' conditioning method with
gradient_x_prev, any registered mean / variance processor pair, clip_denoised or not) every step runs as
ONE device-resident sequence with no host synchronisation:

    fetch_coefs -> UNet forward plan -> osm_posterior_typed -> [osm_phys_reduce/finalize x n_iter ->
    osm_phys_grad] -> [osm_clamp_bwd] -> osm_posterior_bwd -> UNet data-gradient plan -> osm_guide_update

(the reference performs 4 + n_iter device->host copies per step: gaussian_diffusion.py:216,276,288,
condition_methods.py:130,224).  Per-timestep coefficients live in a device table indexed by a
device-side step counter, so the whole step can be replayed from a hipGraph.  The rgb-guidance
configuration (`rgb_guidance=True`: DDPM / DDIM `p_sample` + 'ps' conditioning on an identity operator,
gaussian noiser) runs through the same loop: osm_phys_* with the identity operator (kind 3) for
||y - x0[:, 0:3]|| and its gradient, osm_guide_update(_rng) or osm_ddim_update for the step.  The per-step
noise is drawn inside the update kernel (Philox-4x32-10) unless `noise="aten"` asks for torch's stream.
Any other combination (third-party conditioners / operators / processors, dynamic_threshold) falls back to a
generic loop that follows the reference control flow on top of the HIP UNet operator through torch.autograd.
"""
import math

import numpy as np
import torch

from .. import ops
from ..osmosis_utils import utils as utilso
from .posterior_mean_variance import get_mean_processor, get_var_processor

__SAMPLER__ = {}


def register_sampler(name: str):
    def wrapper(cls):
        if __SAMPLER__.get(name, None):
            raise NameError(f"Name {name} is already registered!")
        __SAMPLER__[name] = cls
        return cls
    return wrapper


def get_sampler(name: str):
    if __SAMPLER__.get(name, None) is None:
        raise NameError(f"Name {name} is not defined!")
    return __SAMPLER__[name]


def create_sampler(sampler, steps, noise_schedule, model_mean_type, model_var_type, dynamic_threshold,
                   clip_denoised, rescale_timesteps, timestep_respacing="", **kwargs):
    cls = get_sampler(name=sampler)
    betas = get_named_beta_schedule(noise_schedule, steps)
    if not timestep_respacing:
        timestep_respacing = [steps]
    return cls(use_timesteps=space_timesteps(steps, timestep_respacing), betas=betas,
               model_mean_type=model_mean_type, model_var_type=model_var_type,
               dynamic_threshold=dynamic_threshold, clip_denoised=clip_denoised,
               rescale_timesteps=rescale_timesteps, annealing_time=kwargs.get("annealing_time", False))


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps):
    T = num_diffusion_timesteps
    if schedule_name == "linear":
        k = 1000 / T
        return np.linspace(k * 0.0001, k * 0.02, T, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(T, lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    T = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / T) / alpha_bar(i / T), max_beta) for i in range(T)])


def space_timesteps(num_timesteps, section_counts):
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == want:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    elif isinstance(section_counts, int):
        section_counts = [section_counts]
    base, extra = divmod(num_timesteps, len(section_counts))
    start, picked = 0, []
    for i, count in enumerate(section_counts):
        size = base + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        pos = 0.0
        for _ in range(count):
            picked.append(start + round(pos))   # Python round (banker's), as the reference
            pos += stride
        start += size
    return set(picked)


def extract_and_expand(array, time, target):
    a = torch.from_numpy(np.asarray(array)).to(target.device)[time].float()
    while a.ndim < target.ndim:
        a = a.unsqueeze(-1)
    return a.expand_as(target)


class GaussianDiffusion:
    def __init__(self, betas, model_mean_type, model_var_type, dynamic_threshold, clip_denoised,
                 rescale_timesteps, **kwargs):
        betas = np.array(betas, dtype=np.float64)
        self.betas = betas
        assert betas.ndim == 1, "betas must be 1-D"
        assert (0 < betas).all() and (betas <= 1).all(), "betas must be in (0..1]"
        self.num_timesteps = int(betas.shape[0])
        self.rescale_timesteps = rescale_timesteps
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        self.mean_processor = get_mean_processor(model_mean_type, betas=betas, dynamic_threshold=dynamic_threshold,
                                                 clip_denoised=clip_denoised)
        self.var_processor = get_var_processor(model_var_type, betas=betas)
        self._fast = None

    # ------------------------------------------------------------------ reference tensor API
    def q_mean_variance(self, x_start, t):
        return (extract_and_expand(self.sqrt_alphas_cumprod, t, x_start) * x_start,
                extract_and_expand(1.0 - self.alphas_cumprod, t, x_start),
                extract_and_expand(self.log_one_minus_alphas_cumprod, t, x_start))

    def q_sample(self, x_start, t):
        noise = torch.randn_like(x_start)
        return (extract_and_expand(self.sqrt_alphas_cumprod, t, x_start) * x_start
                + extract_and_expand(self.sqrt_one_minus_alphas_cumprod, t, x_start) * noise)

    def q_posterior_mean_variance(self, x_start, x_t, t):
        assert x_start.shape == x_t.shape
        mean = (extract_and_expand(self.posterior_mean_coef1, t, x_start) * x_start
                + extract_and_expand(self.posterior_mean_coef2, t, x_t) * x_t)
        return (mean, extract_and_expand(self.posterior_variance, t, x_t),
                extract_and_expand(self.posterior_log_variance_clipped, t, x_t))

    def _scale_timesteps(self, t):
        if self.rescale_timesteps:
            return t.float() * (1000.0 / self.num_timesteps)
        return t

    def _model_timesteps(self, idx: int) -> float:
        """What the network is fed for loop index idx (base class: no respacing)."""
        return float(idx) * (1000.0 / self.num_timesteps) if self.rescale_timesteps else float(idx)

    def p_mean_variance(self, model, x, t):
        model_output = model(x, self._scale_timesteps(t))
        if model_output.shape[1] == 2 * x.shape[1]:
            model_output, model_var_values = torch.split(model_output, x.shape[1], dim=1)
        else:
            model_var_values = model_output
        mean, x0 = self.mean_processor.get_mean_and_xstart(x, t, model_output)
        var, logvar = self.var_processor.get_variance(model_var_values, t)
        assert mean.shape == logvar.shape == x0.shape == x.shape
        return {"mean": mean, "variance": var, "log_variance": logvar, "pred_xstart": x0}

    def p_sample(self, model, x, t):
        raise NotImplementedError

    # ------------------------------------------------------------------ fused (HIP) loop
    def _fast_path_ok(self, model, cond_fn, pretrain_model, rgb_guidance, sample_pattern):
        """The conditioner whose step the fused loop implements, or None (-> `_generic_loop`).  Two configurations:
        the Osmosis one (pretrain_model 'osmosis', 'osmosis' conditioning with gradient_x_prev, a physical operator) and the
        rgb-guidance one (`rgb_guidance=True`: `DDPM.p_sample` / `DDIM.p_sample` + 'ps' conditioning on an identity operator
        with the gaussian noiser; gaussian_diffusion.py:231-232,296-302, condition_methods.py:234-251)."""
        from .condition_methods import PosteriorSampling, PosteriorSamplingOsmosis
        from .unet import UNetModel
        cond = getattr(cond_fn, "__self__", None)
        if not isinstance(model, UNetModel) or model.in_channels != 4:
            return None
        if self.mean_processor.hip_kernel != "osm_posterior" or self.var_processor.hip_kernel != "osm_posterior":
            return None
        if self.mean_processor.dynamic_threshold:            # (a batch-wide quantile: no shipped config; `_generic_loop`)
            return None
        if sample_pattern is not None and sample_pattern.get("pattern") not in (None, "original"):
            if sample_pattern.get("local_M", 1) != 1:
                return None
        if pretrain_model != "osmosis" and not rgb_guidance:
            return None       # the reference's mean-only step (:234-236: p_mean_variance, sample = mean, no p_sample): `_generic_loop`
        if rgb_guidance:
            # the 'ps' step: only the two registered step rules, un-overridden, and a chain that is guided at every index (the
            # reference calls the conditioner at every step of this branch: an unguided index raises in its autograd.grad)
            if type(cond) is not PosteriorSampling or not cond.hip_ok():
                return None
            if getattr(type(self), "p_sample", None) not in (DDPM.p_sample, DDIM.p_sample):
                return None
            if sample_pattern is not None and not all(self._guidance_flag(sample_pattern, i) for i in (0, self.num_timesteps - 1)):
                return None
            if sample_pattern is not None and utilso.set_alternate_length(sample_pattern, 0, self.num_timesteps) != 1:
                return None
            return cond
        if not isinstance(cond, PosteriorSamplingOsmosis):
            return None
        if not cond.gradient_x_prev or not cond.hip_ok():       # (a third-party operator / auxiliary loss: autograd conditioning)
            return None
        return cond

    def coef_table(self) -> np.ndarray:
        """[T][8] fp32: c0,c1,c2,c3 (mean processor's `kernel_coefs`: c0 = d x0/d x, c1 = -d x0/d out, posterior_mean_coef1 / 2),
        the variance processor's two (learned_range: min_log, max_log; fixed_*: log variance, -), noise_on, t_model."""
        T = self.num_timesteps
        tab = np.zeros((T, 8), dtype=np.float32)
        for i in range(T):
            tab[i, 0:4] = self.mean_processor.kernel_coefs(i)
            tab[i, 4:6] = self.var_processor.kernel_coefs(i)
            tab[i, 6] = 0.0 if i == 0 else 1.0
            tab[i, 7] = self._model_timesteps(i)
        return tab

    def ddim_table(self, eta: float = 0.0) -> np.ndarray:
        """[T][8] fp32 rows of `DDIM.p_sample` (gaussian_diffusion.py:505-528) for osm_ddim_update:
        alpha_bar, alpha_bar_prev, eta, noise_on, sqrt_recip_ac, sqrt_recipm1_ac (predict_eps_from_x_start :533-536: the sampler's
        own tables, whatever the mean processor), -, t_model."""
        T = self.num_timesteps
        tab = np.zeros((T, 8), dtype=np.float32)
        tab[:, 0] = self.alphas_cumprod
        tab[:, 1] = self.alphas_cumprod_prev
        tab[:, 2] = eta
        tab[1:, 3] = 1.0
        tab[:, 4] = self.sqrt_recip_alphas_cumprod
        tab[:, 5] = self.sqrt_recipm1_alphas_cumprod
        tab[:, 7] = [self._model_timesteps(i) for i in range(T)]
        return tab

    @staticmethod
    def chunk_sizes(B: int, cap: int):
        """How a batch of B independent images is walked when at most `cap` fit the device at once: the fewest chunks n such
        that either all are equal (B % n == 0, ONE engine of B / n images) or the two sizes ceil(B / n) and floor(B / n)
        together stay within the cap (an engine owns its activation buffers, and a ragged walk keeps two engines).
        B = 37, cap = 32 -> [13, 12, 12] (rounds 1-2 fell back to 37 chunks of one image)."""
        cap = max(1, int(cap))
        n = -(-B // cap)
        while True:
            hi, lo = -(-B // n), B // n
            if B % n == 0:
                return [hi] * n
            if hi + lo <= cap:
                return [hi] * (B % n) + [lo] * (n - B % n)
            n += 1

    def _guidance_flag(self, sample_pattern, idx):
        if sample_pattern is None or sample_pattern["pattern"] == "original" or sample_pattern["pattern"] is None:
            return True
        T = self.num_timesteps
        return sample_pattern["start_guidance"] * T >= idx >= sample_pattern["stop_guidance"] * T

    def _fused_loop(self, model, cond, x_start, measurement, sample_pattern, kwargs, record=False, record_every=150):
        """One device-resident step per index, for the Osmosis configuration and for the rgb-guidance ('ps') one (`_fast_path_ok`).
        Per-step noise (gaussian_diffusion.py:266-268 / :497 / :522): by default drawn INSIDE osm_guide_update_rng from the
        library's Philox-4x32-10 stream (seed: `noise_seed=`, else one draw per chain from the device's torch generator, so
        `torch.manual_seed` still fixes the chain); `noise="aten"` (or OSM_STEP_NOISE=aten) draws it with torch on the device in the reference's
        call order instead (the reference's own realisation for the same seed); `noise_fn=` injects it (parity runs)."""
        import os
        from .condition_methods import PosteriorSampling
        ps = isinstance(cond, PosteriorSampling)
        ddim = ps and getattr(type(self), "p_sample", None) is DDIM.p_sample
        dev = x_start.device
        B, C, H, W = x_start.shape
        HW = H * W
        T = self.num_timesteps
        # the loop drives the engine's plans directly and calls the network as model(x, t) (gaussian_diffusion.py:243): a
        # class-conditional network has no labels here -- the reference's forward asserts (unet.py:713-716); so does this path,
        # instead of replaying the plans with zero / stale label rows
        assert getattr(model, "num_classes", None) is None, "must specify y if and only if the model is class-conditional"
        # optional sub-range of the chain (benchmarks / resumed chains): idx = first .. last, descending
        first, last = kwargs.get("index_range", (T - 1, 0))
        if not (0 <= last <= first <= T - 1):
            raise ValueError(f"index_range must satisfy 0 <= last <= first <= {T - 1}, got ({first}, {last})")
        # Images are independent chains: a batch whose kept activations would not fit the device
        # (~8 GB per 256 x 256 image in fp32) is walked in chunks per step; per-image state (x_t, phi, losses) stays in
        # [B]-sized tensors, the chunks are contiguous row blocks.
        sizes = self.chunk_sizes(B, model.images_in_flight(B, H, W))
        chunks, c0 = [], 0
        for sz in sizes:
            chunks.append((c0, c0 + sz))
            c0 += sz
        engs = {}
        for sz in sorted(set(sizes), reverse=True):       # at most two sizes; the second call keeps the first engine alive
            engs[sz] = model.engine(sz, H, W, keep=tuple(engs.values()))
        eng = engs[sizes[0]]
        f32 = dict(device=dev, dtype=torch.float32)
        table = torch.from_numpy(self.coef_table()).to(dev)
        step = torch.tensor([first], device=dev, dtype=torch.int32)
        coef = torch.zeros(8, **f32)
        dtable = torch.from_numpy(self.ddim_table(float(kwargs.get("eta", 0.0)))).to(dev) if ddim else None
        dcoef = torch.zeros(8, **f32) if ddim else None
        x0, mean, logvar = (torch.empty(B, 4, H, W, **f32) for _ in range(3))
        # `clip_denoised: True` (configs/rgb_guidance_sample_config.yaml; posterior_mean_variance.py:43-50): x0 is clamped inside
        # osm_posterior_typed, the unclamped prediction is kept for the clamp's backward (osm_clamp_bwd masks d loss / d x0)
        x0_raw = torch.empty(B, 4, H, W, **f32) if self.mean_processor.clip_denoised else None
        g = torch.empty(B, 4, H, W, **f32)
        loss_all = torch.zeros(B, **f32)
        scale4 = cond.scale4(dev)
        clip = -1.0 if ps else cond.clip_value
        y = measurement.detach().to(dev, torch.float32).contiguous()
        phi = None if ps else cond.operator.phi
        single = len(chunks) == 1
        x_state = eng.x_in if single else torch.empty(B, 4, H, W, **f32)
        x_state.copy_(x_start.detach())
        noise_fn = kwargs.get("noise_fn", None)           # (k, shape) -> tensor : injected noise (parity runs)
        trace = kwargs.get("trace", None)                 # list collecting per-step tensors (tests)
        records = kwargs.get("record_out", [] if record else None)   # (idx, pred_xstart cpu) snapshots
        shared = bool(kwargs.get("shared_noise", False)) and B > 1   # every image receives the SAME noise (separately seeded batch-1 runs)
        source = "fn" if noise_fn is not None else str(kwargs.get("noise", os.environ.get("OSM_STEP_NOISE", "library"))).lower()
        if source not in ("fn", "library", "aten"):
            raise ValueError(f"noise must be 'library' or 'aten', got {source!r}")
        if source == "library" and (HW % 4 != 0):
            source = "aten"                               # (one Philox counter covers four consecutive elements of an image)
        # q_sample's unused draw (reference :241) and, for 'ps', p_sample's draw before it: only meaningful on torch's generator
        draw_measurement_noise = kwargs.get("reference_rng_order", source == "aten") and source == "aten"
        lib_rng = source == "library" and not ddim        # (DDIM at eta = 0 adds no noise; osm_ddim_update takes a tensor for eta > 0)
        seed = 0
        if source == "library":
            seed = kwargs.get("noise_seed")
            if seed is None:      # one draw per CHAIN from the device's generator (what torch.manual_seed seeds, and only device ops consume)
                seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64, device=dev).item())
        img_base, img_stride = int(kwargs.get("image_index0", 0)), 0 if shared else 1
        noise = None if lib_rng else torch.zeros(B, 4, H, W, **f32)
        noise1 = torch.zeros(1, 4, H, W, **f32) if (shared and source == "aten") else None
        noise_used = torch.empty(B, 4, H, W, **f32) if (lib_rng and trace is not None) else None
        have_loss = False
        for k, idx in enumerate(range(first, last - 1, -1)):
            guided = True if ps else self._guidance_flag(sample_pattern, idx)
            freeze = False if ps else utilso.is_freeze_phi(sample_pattern, idx, T)
            if source == "fn":
                noise.copy_(noise_fn(k, noise.shape))
            elif source == "aten":
                if ps:                                    # DDPM / DDIM.p_sample draw first (:497, :522), then q_sample (:241)
                    noise1.normal_() if noise1 is not None else noise.normal_()
                if draw_measurement_noise:
                    torch.randn_like(y[:1] if noise1 is not None else y)
                if not ps:
                    noise1.normal_() if noise1 is not None else noise.normal_()
                if noise1 is not None:
                    noise.copy_(noise1.expand_as(noise))
            elif not lib_rng:                             # library stream, DDIM: as a tensor (used only for eta > 0)
                ops.randn(noise, B, 4 * HW, seed, step=step, img0=img_base, img_stride=img_stride)   # (before the fetch: counter = idx)
            # every engine's timestep vector is filled from the SAME step counter; the counter moves once, in the last fetch,
            # which always goes through the first engine (stream order: the delta-0 fetches read it before it moves)
            for e2 in engs.values():
                if e2 is not eng:
                    ops.fetch_coefs(table, step, 0, coef, e2.t_dev, e2.B)
            if ddim:
                ops.fetch_coefs(dtable, step, 0, dcoef, eng.t_dev, eng.B)
            ops.fetch_coefs(table, step, -1, coef, eng.t_dev, eng.B)
            if trace is not None:
                rec = {"x_in": x_state.clone()}
                grad_all = torch.empty_like(g) if guided else None
                model_out = torch.empty(B, eng.out.shape[1], H, W, **f32)
            for c0, c1 in chunks:
                ce, Bc = engs[c1 - c0], c1 - c0            # this chunk's engine (`eng` stays the first one)
                if not single:
                    ce.x_in.copy_(x_state[c0:c1])
                ce.run_forward()
                ops.posterior(ce.out, ce.x_in, coef, x0[c0:c1], mean[c0:c1], logvar[c0:c1], Bc, HW,
                              self.mean_processor.kernel_kind, self.var_processor.kernel_kind,
                              None if x0_raw is None else x0_raw[c0:c1])
                if trace is not None:
                    model_out[c0:c1].copy_(ce.out)
                gg = dxu = grad_out = None
                if guided:
                    if ps:
                        cond.loss_grad_x0(x0[c0:c1], y[c0:c1], g_out=g[c0:c1], loss_out=loss_all[c0:c1])
                    else:
                        cond.loss_grad_x0(x0[c0:c1], y[c0:c1], freeze_phi=freeze, g_out=g[c0:c1], phi=phi[c0:c1],
                                          loss_out=loss_all[c0:c1])
                    have_loss = True
                    if x0_raw is not None:
                        ops.clamp_bwd(g[c0:c1], x0_raw[c0:c1])
                    ops.posterior_bwd(g[c0:c1], coef, ce.d_out, Bc, HW)
                    ce.run_backward()
                    gg, dxu = g[c0:c1], ce.dx
                    grad_out = grad_all[c0:c1] if trace is not None else None
                nz = None if noise is None else noise[c0:c1]
                sc, cl = (scale4, clip) if guided else (None, -1.0)
                if ddim:
                    ops.ddim_update(x0[c0:c1], ce.x_in, gg, dxu, nz, coef, dcoef, sc, cl, x_state[c0:c1], grad_out, Bc, HW)
                elif lib_rng:
                    ops.guide_update_rng(mean[c0:c1], logvar[c0:c1], gg, dxu, coef, sc, cl, x_state[c0:c1], grad_out,
                                         None if noise_used is None else noise_used[c0:c1], Bc, HW, seed, step, step_offset=1,
                                         img0=img_base + (0 if shared else c0), img_stride=img_stride)   # (+1: the fetch moved the counter)
                else:
                    ops.guide_update(mean[c0:c1], logvar[c0:c1], gg, dxu, nz, coef, sc, cl, x_state[c0:c1], grad_out, Bc, HW)
            if trace is not None:
                rec.update(x0=x0.clone(), mean=mean.clone(), x_out=x_state.clone(), model_out=model_out,
                           loss=loss_all.clone() if have_loss else None, phi=None if phi is None else phi.clone())
                if noise_used is not None:
                    rec["noise"] = noise_used.clone()
                if guided:
                    rec["grad"] = grad_all
                trace.append(rec)
            # snapshots of pred_xstart during the chain (reference :308-327): same steps, one D2H copy each
            if records is not None and ((idx % record_every == 0) or idx == 0 or idx == 999):
                records.append((idx, x0.detach().cpu()))
        img = x_state.clone()
        if record and records:
            self._save_process_grid(records, kwargs.get("save_grids_path"), kwargs.get("original_file_name"))
        if ps:                                            # the rgb-guidance branch returns the sample only (:339-340)
            return img
        variables = cond.operator.optimize(freeze_phi=True)
        loss_np = loss_all.detach().cpu().numpy() if have_loss else None
        return img, variables, loss_np, x0.detach().cpu()

    @staticmethod
    def _save_process_grid(records, save_grids_path, original_file_name):
        """`<name>_process.png` of the reference (:308-333): clipped RGB of image 0 on the first row, the
        percentile-normalised viridis depth on the second, one column per recorded step (2-pixel padding, as
        torchvision.utils.make_grid lays it out)."""
        if save_grids_path is None:
            return None
        import os

        from PIL import Image
        rgb = [torch.clamp(0.5 * (x0[0, 0:3] + 1), 0, 1) for _, x0 in records]
        dep = [utilso.depth_tensor_to_color_image(
            utilso.min_max_norm_range_percentile(x0[:, 3], percent_low=0.05, percent_high=0.99)) for _, x0 in records]
        dep = [d.reshape(3, *d.shape[-2:]) for d in dep]
        # torchvision.utils.make_grid(rgb + depth, nrow=len(rgb)) -> tvtf.to_pil_image (:330-333): the stack is float64 (the viridis
        # depth), the background 0, and to_pil_image TRUNCATES (pic.mul(255).byte())
        from .. import sampling
        arr = sampling._to_pil_u8(sampling.make_grid(rgb + dep, nrow=len(rgb)))
        path = os.path.join(save_grids_path, f"{original_file_name}_process.png")
        Image.fromarray(arr).save(path)
        return path

    # ------------------------------------------------------------------ public loop
    def p_sample_loop(self, model, x_start, measurement, measurement_cond_fn, record, save_root,
                      pretrain_model=None, image_idx=None, record_every=150, rgb_guidance=False,
                      sample_pattern=None, **kwargs):
        from .posterior_mean_variance import PreviousXMeanProcessor
        if isinstance(self.mean_processor, PreviousXMeanProcessor) and not (
                rgb_guidance and getattr(type(self), "p_sample", None) is DDIM.p_sample):
            # Error behaviour of the reference: `previous_x` hands the network's split output on AS the mean
            # (posterior_mean_variance.py:68-72), and every step rule except DDIM.p_sample then adds to it in place
            # (gaussian_diffusion.py:268 / :499, condition_methods.py:223 / :249) -- autograd refuses that on a multi-output view.
            raise RuntimeError("Output 0 of SplitBackward0 is a view and is being modified inplace: the 'previous_x' mean processor "
                               "only runs with the DDIM sampler on the rgb-guidance branch (as in the reference)")
        cond = self._fast_path_ok(model, measurement_cond_fn, pretrain_model, rgb_guidance, sample_pattern)
        if cond is not None:
            return self._fused_loop(model, cond, x_start, measurement, sample_pattern, kwargs, record=record,
                                    record_every=record_every)
        return self._generic_loop(model, x_start, measurement, measurement_cond_fn, pretrain_model,
                                  rgb_guidance, sample_pattern, kwargs, record=record, record_every=record_every)

    def _generic_loop(self, model, x_start, measurement, cond_fn, pretrain_model, rgb_guidance, sample_pattern,
                      kwargs, record=False, record_every=150):
        """Third-party conditioners / operators / processors the fused loop has no kernels for: the reference's control flow
        (gaussian_diffusion.py:213-340) over `p_mean_variance` / `p_sample` with torch.autograd through the HIP UNet operator.
        `record=True` snapshots pred_xstart at the reference's indices (:308-327) here as well."""
        img = x_start
        device = x_start.device
        T = self.num_timesteps
        loss = variable_dict = out = None
        records = kwargs.get("record_out", [] if record else None)
        osmosis = pretrain_model == "osmosis" and not rgb_guidance
        for idx in range(T - 1, -1, -1):
            time = torch.tensor([idx] * img.shape[0], device=device)
            guided = self._guidance_flag(sample_pattern, idx) if sample_pattern is not None else True
            alt = utilso.set_alternate_length(sample_pattern, idx, T) if sample_pattern is not None else 1
            for _ in range(alt):
                img = img.detach().requires_grad_(bool(guided))
                if rgb_guidance:
                    out = self.p_sample(x=img, t=time, model=model)
                else:
                    out = self.p_mean_variance(model=model, x=img, t=time)
                    out["sample"] = out["mean"]
                noisy_measurement = self.q_sample(measurement, t=time)
                if not osmosis:
                    img, loss = cond_fn(x_t=out["sample"], measurement=measurement, noisy_measurement=noisy_measurement,
                                        x_prev=img, x_0_hat=out["pred_xstart"])
                    img = img.detach()
                    continue
                if guided:
                    img, loss, variable_dict, _grads, _aux = cond_fn(
                        x_t=out["sample"], measurement=measurement, noisy_measurement=noisy_measurement, x_prev=img,
                        x_0_hat=out["pred_xstart"], freeze_phi=utilso.is_freeze_phi(sample_pattern, idx, T),
                        time_index=float(idx) / T)
                else:
                    img = out["sample"]
                noise = torch.randn_like(img)
                img = img.detach()
                if idx != 0:
                    img = img + torch.exp(0.5 * out["log_variance"].detach()) * noise
            if records is not None and ((idx % record_every == 0) or idx == 0 or idx == 999):
                records.append((idx, out["pred_xstart"].detach().cpu()))
        if record and records:
            self._save_process_grid(records, kwargs.get("save_grids_path"), kwargs.get("original_file_name"))
        if osmosis:
            return img, variable_dict, loss, out["pred_xstart"].detach().cpu()
        return img


class SpacedDiffusion(GaussianDiffusion):
    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.timestep_map = []
        self.original_num_steps = len(kwargs["betas"])
        base_ac = np.cumprod(1.0 - np.array(kwargs["betas"], dtype=np.float64), axis=0)
        last, new_betas = 1.0, []
        for i, ac in enumerate(base_ac):
            if i in self.use_timesteps:
                new_betas.append(1 - ac / last)
                last = ac
                self.timestep_map.append(i)
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)

    def _model_timesteps(self, idx: int) -> float:
        t = float(self.timestep_map[idx])
        return t * (1000.0 / self.original_num_steps) if self.rescale_timesteps else t

    def p_mean_variance(self, model, *args, **kwargs):
        return super().p_mean_variance(self._wrap_model(model), *args, **kwargs)

    def _wrap_model(self, model):
        if isinstance(model, _WrappedModel):
            return model
        return _WrappedModel(model, self.timestep_map, self.rescale_timesteps, self.original_num_steps)

    def _scale_timesteps(self, t):
        return t   # done by the wrapped model


class _WrappedModel:
    def __init__(self, model, timestep_map, rescale_timesteps, original_num_steps):
        self.model = model
        self.timestep_map = timestep_map
        self.rescale_timesteps = rescale_timesteps
        self.original_num_steps = original_num_steps

    def __call__(self, x, ts, **kwargs):
        map_tensor = torch.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
        new_ts = map_tensor[ts]
        if self.rescale_timesteps:
            new_ts = new_ts.float() * (1000.0 / self.original_num_steps)
        return self.model(x, new_ts, **kwargs)


@register_sampler(name="ddpm")
class DDPM(SpacedDiffusion):
    """Ancestral step (reference :492-502).  In the fused loop it IS osm_guide_update / osm_guide_update_rng (the row's
    `noise_on` switches the noise off at index 0); `p_sample` is the same row applied to torch tensors, for conditioners the
    fused loop has no kernels for (`_generic_loop`, autograd through `pred_xstart`)."""

    def p_sample(self, model, x, t):
        out = self.p_mean_variance(model, x, t)
        z = torch.randn_like(x)                            # drawn at every index (the reference's RNG order), used when noise_on
        noise_on = int(t[0]) != 0
        sample = out["mean"] + torch.exp(0.5 * out["log_variance"]) * z if noise_on else out["mean"]
        return {"sample": sample, "pred_xstart": out["pred_xstart"]}


@register_sampler(name="ddim")
class DDIM(SpacedDiffusion):
    """DDIM step (reference :505-535, Song et al. eq. 12).  In the fused loop it IS osm_ddim_update fed by `ddim_table`;
    `p_sample` applies the same fp32 row scalars to torch tensors (`_generic_loop`)."""

    def step_scalars(self, idx: int, eta: float = 0.0):
        """fp32 scalars of index idx in the kernel's operation order: (sqrt_recip_ac, sqrt_recipm1_ac, sqrt(ab_prev),
        sqrt(1 - ab_prev - sigma^2), sigma)."""
        f = np.float32
        ab, abp = f(self.alphas_cumprod[idx]), f(self.alphas_cumprod_prev[idx])
        sigma = f(eta) * np.sqrt((f(1) - abp) / (f(1) - ab)) * np.sqrt(f(1) - ab / abp)
        return (f(self.sqrt_recip_alphas_cumprod[idx]), f(self.sqrt_recipm1_alphas_cumprod[idx]), np.sqrt(abp),
                np.sqrt(f(1) - abp - sigma * sigma), sigma)

    def p_sample(self, model, x, t, eta=0.0):
        out = self.p_mean_variance(model, x, t)
        idx = int(t[0])
        c0, c1, sa, sb, sigma = (float(v) for v in self.step_scalars(idx, eta))
        x0 = out["pred_xstart"]
        z = torch.randn_like(x)                            # drawn at every index (the reference's RNG order)
        sample = x0 * sa + sb * ((c0 * x - x0) / c1)
        if idx != 0:
            sample = sample + sigma * z
        return {"sample": sample, "pred_xstart": x0}

    def predict_eps_from_x_start(self, x_t, t, pred_xstart):
        c0, c1 = (float(v) for v in self.step_scalars(int(t[0]))[:2])
        return (c0 * x_t - pred_xstart) / c1
