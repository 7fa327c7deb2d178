"""Unconditional RGBD-prior sampler -- surface of the reference's osmosis_utils/diffusion.py
(`GaussianDiffusion(T, schedule)` :19-47, `sample` :48-58, `inverse` :59-130), SURVEY.md section 8(f) N1.

`inverse` is the plain ancestral DDPM chain used by RGBD_prior_sampling.py: no guidance, no autograd,
fixed-small variance (the learned-variance channels of the UNet are ignored), 1-based float timesteps,
and `steps` TRUNCATES the chain (t = start_t .. start_t-steps+1) rather than respacing it.
On our UNetModel every step is: fetch_coefs -> UNet forward plan -> osm_ancestral_step, device-resident.
"""
import math

import numpy as np
import torch

from .. import ops
from . import utils as utilso


class GaussianDiffusion:
    def __init__(self, T, schedule):
        self.T = T
        if schedule == "linear":
            self.beta = np.linspace(1e-4, 2e-2, T)
        elif schedule == "cosine":
            f = lambda t: np.cos(math.pi * 0.5 * (t / T + 0.008) / 1.008) ** 2  # noqa: E731
            ab = f(np.arange(0, T + 1, 1)) / f(0)
            self.beta = np.clip(1 - (ab[1:] / ab[:-1]), None, 0.999)
        else:
            raise NotImplementedError(f"unknown schedule: {schedule}")
        self.betabar = np.cumprod(self.beta)
        self.alpha = 1 - self.beta
        self.alphabar = np.cumprod(self.alpha)

    def sample(self, x0, t):
        dims = (x0.shape[0],) + tuple(1 for _ in x0.shape[1:])
        atbar = torch.from_numpy(self.alphabar[t - 1]).view(dims).to(x0.device)
        eps = torch.randn_like(x0)
        return torch.sqrt(atbar) * x0 + torch.sqrt(1 - atbar) * eps, eps

    def step_coefs(self, t: int):
        """fp32 coefficients of the update at 1-based timestep t: (c_a, c_b, c_s, c_r, c_m)."""
        at, atbar = self.alpha[t - 1], self.alphabar[t - 1]
        bt = self.beta[t - 1] * (1 - self.alphabar[t - 2]) / (1 - atbar) if t > 1 else 0.0
        return (np.float32(1 / np.sqrt(at)), np.float32((1 - at) / np.sqrt(1 - atbar)), np.float32(np.sqrt(bt)),
                np.float32(1 / np.sqrt(atbar)), np.float32(np.sqrt(1 - atbar) / np.sqrt(atbar)))

    def inverse(self, net, shape=(1, 64, 64), image_channels=3, steps=None, x=None, start_t=None, device="cpu", **kwargs):
        from ..guided_diffusion.unet import UNetModel
        if not isinstance(net, UNetModel):
            raise NotImplementedError("inverse() drives the HIP UNetModel (there is no eager fallback)")
        # net(x, t) without labels (osmosis_utils/diffusion.py:104 of the reference): a class-conditional network asserts there
        assert net.num_classes is None, "must specify y if and only if the model is class-conditional"
        noise_fn = kwargs.get("noise_fn", None)          # (k, shape) -> tensor : injected noise (parity runs)
        dev = torch.device(device)
        if x is None:
            x = torch.randn((1,) + tuple(shape), device=dev)
        start_t = self.T if start_t is None else start_t
        steps = self.T if steps is None else steps
        if not (1 <= start_t <= self.T and 1 <= steps <= start_t):
            raise ValueError(f"inverse(): need 1 <= start_t - steps + 1 <= start_t <= T = {self.T} "
                             f"(got start_t = {start_t}, steps = {steps})")
        B, C, H, W = x.shape
        HW = H * W
        eng = net.engine(B, H, W)
        ts = list(range(start_t, start_t - steps, -1))
        tab = np.zeros((len(ts), 8), dtype=np.float32)
        for k, t in enumerate(ts):
            tab[k, 0:5] = self.step_coefs(t)
            tab[k, 7] = float(t)
        table = torch.from_numpy(tab).to(dev)
        step = torch.zeros(1, device=dev, dtype=torch.int32)
        coef = torch.zeros(8, device=dev)
        z = torch.zeros(B, C, H, W, device=dev)
        x0 = torch.empty(B, C, H, W, device=dev)
        eng.x_in.copy_(x)
        # `record_process` (reference :63-71, :106-128): at every t with t % record_every == 0, and at t = 1, keep x_t, the clipped RGB
        # of the predicted x_0 and its colour-mapped depth; written at the end as ONE grid, `<save_path>/image_<idx>_process.png`
        record = bool(kwargs.get("record_process", False)) and kwargs.get("save_path", None) is not None
        record_every = int(kwargs.get("record_every", 200))
        xt_list, rgb_list, depth_list = [], [], []

        def views(x0_cpu):
            x0v = 0.5 * (x0_cpu.squeeze() + 1)
            rgb = torch.clamp(x0v[0:3], 0, 1)
            dep = None
            if image_channels == 4:
                dep = utilso.depth_tensor_to_color_image(
                    utilso.min_max_norm_range_percentile(x0v[3].unsqueeze(0), percent_low=0.05, percent_high=0.99))
            return rgb, dep
        for k, t in enumerate(ts):
            rec = record and ((t % record_every == 0) or t == 1)
            if rec:
                xt_list.append(torch.clamp(0.5 * (eng.x_in.detach().cpu().squeeze() + 1), 0, 1)[0:3])
            if t > 1:
                if noise_fn is not None:
                    z.copy_(noise_fn(k, z.shape))
                else:
                    z.normal_()
            else:
                z.zero_()
            ops.fetch_coefs(table, step, 1, coef, eng.t_dev, B)
            eng.run_forward()
            ops.ancestral_step(eng.out, eng.x_in, z, coef, eng.x_in, x0, B, C, eng.cout, HW)
            if rec:
                rgb, dep = views(x0.detach().cpu())
                rgb_list.append(rgb)
                if dep is not None:
                    depth_list.append(dep)
        out = eng.x_in.clone()
        if record:
            import os

            from PIL import Image

            from .. import sampling
            grid = sampling.make_grid(xt_list + rgb_list + depth_list, nrow=len(xt_list), pad_value=1.0)
            Image.fromarray(sampling._to_pil_u8(grid)).save(os.path.join(kwargs["save_path"], f"image_{kwargs.get('image_idx', 0)}_process.png"))
        # the reference returns the LAST recorded views (t = 1 is always recorded) -- the clipped RGB of the predicted x_0 and the
        # viridis image of its percentile-normalised depth -- and raises UnboundLocalError when it did not record; they are
        # returned here either way
        x_start_rgb, x_depth = views(x0[0:1].detach().cpu())
        return out, [x_start_rgb, x_depth]
