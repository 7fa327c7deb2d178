"""CPU restatement of the reference's schedules / posterior math / guidance / sampling loop
(TEST INFRASTRUCTURE, see oracle/__init__.py).  torch-CPU fp32 for tensors, numpy float64 for
the per-timestep tables, exactly like the reference (tables are cast to fp32 AFTER indexing,
gaussian_diffusion.py:593-597).

Follows (reference file:line):
  * get_named_beta_schedule          guided_diffusion/gaussian_diffusion.py:542-566
  * space_timesteps                  guided_diffusion/gaussian_diffusion.py:373-426
  * GaussianDiffusion.__init__       guided_diffusion/gaussian_diffusion.py:66-121
  * SpacedDiffusion.__init__         guided_diffusion/gaussian_diffusion.py:437-451
  * _WrappedModel.__call__           guided_diffusion/gaussian_diffusion.py:484-489
  * EpsilonXMeanProcessor            guided_diffusion/posterior_mean_variance.py:104-136
  * LearnedRangeVarianceProcessor    guided_diffusion/posterior_mean_variance.py:227-258
  * p_mean_variance / p_sample_loop  guided_diffusion/gaussian_diffusion.py:345-365, 179-340
  * physical operators               guided_diffusion/measurements.py:138-151, 251-264, 363-376
  * PosteriorSamplingOsmosis         guided_diffusion/condition_methods.py:109-231
  * aux losses                       osmosis_utils/losses.py:29-83
  * convert_depth / set_loss_weight / is_freeze_phi   osmosis_utils/utils.py:544-566, 674-700, 571-590
"""
import math
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch


# ----------------------------------------------------------------------------- schedules
def named_beta_schedule(name: str, T: int) -> np.ndarray:
    if name == "linear":
        s = 1000 / T
        return np.linspace(s * 0.0001, s * 0.02, T, dtype=np.float64)
    if name == "cosine":
        f = lambda u: math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2  # noqa: E731
        return np.array([min(1 - f((i + 1) / T) / f(i / T), 0.999) for i in range(T)])
    raise NotImplementedError(f"unknown beta schedule: {name}")


def space_timesteps(T: int, section_counts) -> set:
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for i in range(1, T):
                if len(range(0, T, i)) == want:
                    return set(range(0, T, i))
            raise ValueError(f"cannot create exactly {T} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    elif isinstance(section_counts, int):
        section_counts = [section_counts]
    size_per, extra = divmod(T, len(section_counts))
    start, out = 0, []
    for i, cnt in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError(f"cannot divide section of {size} steps into {cnt}")
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            out.append(start + round(cur))     # Python banker's rounding, as the reference
            cur += stride
        start += size
    return set(out)


class Tables:
    """All per-timestep float64 tables of a (respaced) diffusion."""

    def __init__(self, base_betas: np.ndarray, use_timesteps: Optional[Sequence[int]] = None):
        base_betas = np.asarray(base_betas, dtype=np.float64)
        self.original_num_steps = len(base_betas)
        if use_timesteps is None:
            use_timesteps = range(len(base_betas))
        use = set(use_timesteps)
        ac = np.cumprod(1.0 - base_betas)
        last, nb, tmap = 1.0, [], []
        for i, a in enumerate(ac):
            if i in use:
                nb.append(1 - a / last)
                last = a
                tmap.append(i)
        betas = np.array(nb, dtype=np.float64)
        self.timestep_map = tmap
        self.betas = betas
        self.num_timesteps = len(betas)
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(
            np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        self.log_betas = np.log(betas)


def make_tables(steps: int, noise_schedule: str, timestep_respacing="") -> Tables:
    """create_sampler's table construction (gaussian_diffusion.py:38-62)."""
    betas = named_beta_schedule(noise_schedule, steps)
    if not timestep_respacing:
        timestep_respacing = [steps]
    return Tables(betas, space_timesteps(steps, timestep_respacing))


def _coef(arr: np.ndarray, t: int) -> float:
    """float64 table -> fp32 scalar (extract_and_expand casts with .float())."""
    return float(np.float32(arr[t]))


# ----------------------------------------------------------------------------- posterior
def process_xstart(x0: torch.Tensor, clip_denoised: bool = False, dynamic_threshold: bool = False) -> torch.Tensor:
    """pmv.py:43-50; `dynamic_thresholding(x, s=0.98)` = util/img_utils.py:8-15: x times the 0.98-quantile of |x| over the whole
    tensor, clipped to [-1, 1]."""
    if dynamic_threshold:
        x0 = torch.clip(x0 * torch.quantile(x0.abs(), 0.98), -1.0, 1.0)
    if clip_denoised:
        x0 = x0.clamp(-1, 1)
    return x0


def p_mean_variance(tb: Tables, model_out: torch.Tensor, x: torch.Tensor, t: int, mean_type: str = "epsilon",
                    var_type: str = "learned_range", clip_denoised: bool = False,
                    dynamic_threshold: bool = False) -> Dict[str, torch.Tensor]:
    """gd.py:345-365 over the registered processors of posterior_mean_variance.py: mean `epsilon` (:104-136, the one every
    shipped config names), `start_x` (:75-101: the network predicts x_0), `previous_x` (:53-72: the network predicts the mean,
    x_0 solved from it); variance `learned_range` (:225-258), `fixed_small` (:171-187), `fixed_large` (:190-212), `learned`
    (:215-222).  The network of the path has 2 C output channels, so the split of gd.py:349-350 always happens."""
    C = x.shape[1]
    out, v = torch.split(model_out, C, dim=1)
    pc1, pc2 = tb.posterior_mean_coef1, tb.posterior_mean_coef2
    if mean_type == "epsilon":
        x0 = process_xstart(_coef(tb.sqrt_recip_alphas_cumprod, t) * x - _coef(tb.sqrt_recipm1_alphas_cumprod, t) * out,
                            clip_denoised, dynamic_threshold)
        mean = _coef(pc1, t) * x0 + _coef(pc2, t) * x
    elif mean_type == "start_x":
        x0 = process_xstart(out, clip_denoised, dynamic_threshold)
        mean = _coef(pc1, t) * x0 + _coef(pc2, t) * x
    elif mean_type == "previous_x":
        mean = out
        x0 = process_xstart(_coef(1.0 / pc1, t) * out - _coef(pc2 / pc1, t) * x, clip_denoised, dynamic_threshold)
    else:
        raise NameError(f"Name {mean_type} is not defined.")
    if var_type == "learned_range":
        frac = (v + 1.0) / 2.0
        logvar = frac * _coef(tb.log_betas, t) + (1 - frac) * _coef(tb.posterior_log_variance_clipped, t)
    elif var_type in ("fixed_small", "fixed_large"):
        with np.errstate(divide="ignore"):                  # fixed_small: log 0 at index 0, where no noise is added
            table = np.log(tb.posterior_variance if var_type == "fixed_small"
                           else np.append(tb.posterior_variance[1], tb.betas[1:]))
        logvar = torch.full_like(x, _coef(table, t))
    elif var_type == "learned":
        logvar = v
    else:
        raise NameError(f"Name {var_type} is not defined.")
    return {"mean": mean, "log_variance": logvar, "variance": torch.exp(logvar), "pred_xstart": x0}


# ----------------------------------------------------------------------------- physics
def convert_depth(depth, depth_type, value):
    if depth_type == "move":
        return depth + value
    if depth_type == "gamma":
        return torch.pow((depth + value[0]) * value[1], value[2])
    if depth_type is None or depth_type == "original":
        return 0.5 * (depth + 1.0)
    raise NotImplementedError


def parse_value(v):
    if isinstance(v, (float, int)):
        return float(v)
    if isinstance(v, str):
        return np.array([float(s) for s in v.split(",")], dtype=float)
    return v


class PhysOperator:
    """The three learnable image-formation models.  kind in
    {'underwater_physical_revised', 'underwater_physical', 'haze_physical'}.
    phi tensors are [B,3,1,1] ([B,1,1,1] for haze phi_ab), plain SGD with lr=eta."""

    def __init__(self, kind: str, batch_size: int = 1, depth_type=None, value=None, **kw):
        self.kind = kind
        self.depth_type = depth_type
        self.value = parse_value(value)

        def vec(s, n=None):
            a = torch.tensor([float(u) for u in str(s).split(",")], dtype=torch.float32)
            return a.repeat(batch_size, 1)[..., None, None].contiguous()

        if kind == "underwater_physical_revised":
            self.names = ["phi_a", "phi_b", "phi_inf"]
        elif kind in ("underwater_physical", "haze_physical"):
            self.names = ["phi_ab", "phi_inf"]
        else:
            raise NameError(f"Name {kind} is not defined.")
        self.phi = {n: vec(kw[n]) for n in self.names}
        self.eta = {n: (float(kw.get(n + "_eta", 1e-5)) if kw.get(n + "_learn_flag", True) else 0.0)
                    for n in self.names}
        # measurements.py:244-249 / utils.py:494-524: `optimizer: <name>` = the torch.optim class with ITS defaults over one parameter
        # group per phi (lr = eta); None / "GD" / "sgd" with default momentum 0 = plain phi -= eta * grad (measurements.py:272-280)
        name = (kw.get("optimizer") or "").lower()
        classes = {"adam": torch.optim.Adam, "adamw": torch.optim.AdamW, "adamax": torch.optim.Adamax, "rmsprop": torch.optim.RMSprop,
                   "adagrad": torch.optim.Adagrad, "adadelta": torch.optim.Adadelta, "asgd": torch.optim.ASGD, "rprop": torch.optim.Rprop}
        if name in ("", "gd", "sgd"):
            self.optimizer = None
        elif name in classes:
            self.optimizer = classes[name]([{"params": self.phi[n], "lr": self.eta[n]} for n in self.names])
        else:
            raise ValueError(f"Optimizer '{name}' is not supported.")

    def forward(self, data):
        rgb01 = 0.5 * (data[:, 0:-1] + 1)
        d = convert_depth(data[:, -1:].clone(), self.depth_type, self.value)
        if self.kind == "underwater_physical_revised":
            pa, pb = self.phi["phi_a"], self.phi["phi_b"]
        else:
            pa = pb = self.phi["phi_ab"]
        return rgb01 * torch.exp(-pa * d) + self.phi["phi_inf"] * (1 - torch.exp(-pb * d))

    def set_requires_grad(self, flag: bool):
        for p in self.phi.values():
            p.requires_grad_(flag)

    def sgd_step(self):
        """measurements.py:266-303 `optimize`: the GD branch, or optimizer.step() + zero_grad()."""
        if self.optimizer is not None:
            self.optimizer.step()
            self.optimizer.zero_grad()
            return
        with torch.no_grad():
            for n, p in self.phi.items():
                if p.grad is not None:
                    p.add_(p.grad, alpha=-self.eta[n])
                    p.grad = None

    def variables(self):
        return {n: p.detach().clone() for n, p in self.phi.items()}


def aux_loss(x0, gammas: Optional[Dict[str, float]]):
    """losses.py:29-83: avrg_loss = sum |mean_hw rgb| ; val_loss = mean(relu(|rgb|-0.7)^2)."""
    if not gammas:
        return None
    total = 0
    for name, g in gammas.items():
        rgb = x0[:, 0:3]
        if name == "avrg_loss":
            cur = torch.sum(torch.abs(torch.mean(rgb, dim=(2, 3))))
        elif name == "val_loss":
            cur = (torch.maximum(rgb.abs() - 0.7, torch.zeros_like(rgb)) ** 2).mean()
        else:
            raise NameError(f"Name {name} is not defined.")
        total = total + torch.tensor(float(g)) * cur
    return total


class OsmosisGuidance:
    """PosteriorSamplingOsmosis (condition_methods.py:61-231), gradient_x_prev path."""

    def __init__(self, operator: PhysOperator, scale="7,7,7,0.9", gradient_clip="True,0.005",
                 loss_function="norm", loss_weight="depth", weight_function="gamma,1.4,1.4,1",
                 n_iter=20, aux: Optional[Dict[str, float]] = None, gradient_x_prev=True):
        self.op = operator
        try:
            self.scale = torch.tensor([float(scale)])
        except ValueError:
            self.scale = torch.tensor([float(s) for s in scale.split(",")])
        parts = gradient_clip.split(",")
        self.clip = float(parts[1]) if parts[0].strip().lower() in ("yes", "true", "t", "y", "1") else None
        self.loss_function = loss_function
        self.loss_weight = loss_weight
        self.weight_function = weight_function
        self.n_iter = n_iter
        self.aux = aux
        assert gradient_x_prev, "oracle restates the gradient_x_prev=True path (all osmosis configs)"

    def _weight(self, x0):
        if self.loss_weight in (None, "none"):
            return 1
        if self.loss_weight != "depth":
            raise NotImplementedError
        parts = self.weight_function.split(",")
        val = np.asarray(parts[1:]).astype(float)
        val = val.item() if val.shape[0] == 1 else val
        return convert_depth(x0.detach()[:, 3:4], parts[0], val)

    def loss(self, x0, y):
        I = self.op.forward(x0)
        diff = (y - (2 * I - 1)) * self._weight(x0)
        if self.loss_function == "norm":
            loss = torch.linalg.norm(diff)
            sep = torch.norm(diff.detach(), p=2, dim=[1, 2, 3]).numpy()
        elif self.loss_function == "mse":
            mse = (diff ** 2).mean(dim=(1, 2, 3))
            loss = mse.sum()
            sep = mse.detach().numpy()
        else:
            raise NotImplementedError
        return sep, loss

    def conditioning(self, x_prev, x_t, x_0_hat, y, freeze_phi: bool):
        """x_prev requires grad and x_0_hat depends on it.  Mutates x_t in place."""
        self.op.set_requires_grad(not freeze_phi)
        n_inner = 1 if freeze_phi else self.n_iter
        for it in range(n_inner):
            sep, loss = self.loss(x_0_hat, y)
            a = aux_loss(x_0_hat, self.aux)
            total = loss if a is None else loss + a
            phis = [] if freeze_phi else list(self.op.phi.values())
            if it == n_inner - 1:
                total.backward(inputs=[x_prev] + phis)
            else:
                total.backward(inputs=phis, retain_graph=True)
            if not freeze_phi:
                self.op.sgd_step()
        with torch.no_grad():
            g = x_prev.grad
            gc = torch.clamp(g, -self.clip, self.clip) if self.clip is not None else g
            x_t -= self.scale[None, :, None, None] * gc
        return x_t, sep, self.op.variables(), g.detach().clone()


def is_freeze_phi(pattern: Optional[dict], idx: int, T: int) -> bool:
    if pattern is None or pattern["pattern"] == "original":
        return False
    if idx > pattern["start_guidance"] * T or idx < pattern["stop_guidance"] * T:
        return True
    if idx > pattern["update_start"] * T or idx < pattern["update_end"] * T:
        return True
    return False


def p_sample_loop(model: Callable, tb: Tables, x_T: torch.Tensor, y: torch.Tensor,
                  guidance: OsmosisGuidance, pattern: Optional[dict],
                  noises: List[torch.Tensor], trace: Optional[list] = None, mean_type: str = "epsilon",
                  var_type: str = "learned_range", clip_denoised: bool = False):
    """gaussian_diffusion.py:179-340 (osmosis branch, alternate_len=1, guidance always on).
    `model(x, t_mapped_float_tensor)` -> [B,8,H,W].  `noises[k]` = the randn_like(img) drawn at
    loop iteration k (the reference's unused randn_like(measurement) draw is not modelled here;
    the golden generator records only the used draw)."""
    img = x_T.clone()
    T = tb.num_timesteps
    loss = variables = x0 = None
    for k, idx in enumerate(range(T - 1, -1, -1)):
        img = img.detach().requires_grad_(True)
        t_model = torch.tensor([tb.timestep_map[idx]] * img.shape[0])
        out = p_mean_variance(tb, model(img, t_model), img, idx, mean_type, var_type, clip_denoised)
        freeze = is_freeze_phi(pattern, idx, T)
        x_t, loss, variables, grad = guidance.conditioning(img, out["mean"], out["pred_xstart"], y, freeze)
        x0 = out["pred_xstart"].detach()
        new = x_t.detach()
        if idx != 0:
            new = new + torch.exp(0.5 * out["log_variance"].detach()) * noises[k]
        if trace is not None:
            trace.append({"x_in": img.detach().clone(), "x0": x0.clone(), "grad": grad,
                          "loss": np.array(loss), "x_out": new.clone(),
                          "phi": {n: v.clone() for n, v in variables.items()}})
        img = new
    return img, variables, loss, x0
