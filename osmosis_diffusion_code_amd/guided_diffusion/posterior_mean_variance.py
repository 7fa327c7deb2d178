"""Mean / variance processors -- registry surface of the reference's
guided_diffusion/posterior_mean_variance.py (register/get_mean_processor :15-28,
register/get_var_processor :146-159).

Every registered pair runs in the fused osm_posterior_typed kernel (the pair every shipped config uses -- 'epsilon'
(:104-136) + 'learned_range' (:227-258) -- is kinds (0, 0) = osm_posterior): the sampler asks the two processors for their
`kernel_kind` and their float64 table rows (`kernel_coefs`) and never calls the tensor methods on the hot path.  The tensor
methods are kept (plain torch elementwise ops on whatever device the tensors live on) so third-party code written against the
reference API -- and `_generic_loop` -- keeps working.  Tables are float64 and cast to fp32 after indexing, as in the
reference (:265-269).
"""
from abc import ABC, abstractmethod

import numpy as np
import torch

__MODEL_MEAN_PROCESSOR__ = {}
__MODEL_VAR_PROCESSOR__ = {}


def register_mean_processor(name: str):
    def wrapper(cls):
        if __MODEL_MEAN_PROCESSOR__.get(name, None):
            raise NameError(f"Name {name} is already registerd.")
        __MODEL_MEAN_PROCESSOR__[name] = cls
        return cls
    return wrapper


def get_mean_processor(name: str, **kwargs):
    if __MODEL_MEAN_PROCESSOR__.get(name, None) is None:
        raise NameError(f"Name {name} is not defined.")
    return __MODEL_MEAN_PROCESSOR__[name](**kwargs)


def register_var_processor(name: str):
    def wrapper(cls):
        if __MODEL_VAR_PROCESSOR__.get(name, None):
            raise NameError(f"Name {name} is already registerd.")
        __MODEL_VAR_PROCESSOR__[name] = cls
        return cls
    return wrapper


def get_var_processor(name: str, **kwargs):
    if __MODEL_VAR_PROCESSOR__.get(name, None) is None:
        raise NameError(f"Name {name} is not defined.")
    return __MODEL_VAR_PROCESSOR__[name](**kwargs)


def extract_and_expand(array, time, target):
    a = torch.from_numpy(np.asarray(array)).to(target.device)[time].float()
    while a.ndim < target.ndim:
        a = a.unsqueeze(-1)
    return a.expand_as(target)


def _posterior_tables(betas):
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    coef1 = betas * np.sqrt(ac_prev) / (1.0 - ac)
    coef2 = (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)
    var = betas * (1.0 - ac_prev) / (1.0 - ac)
    return ac, ac_prev, coef1, coef2, var


class MeanProcessor(ABC):
    hip_kernel = None      # name of the fused kernel family that implements this processor, if any
    kernel_kind = 0        # osm_posterior_typed's mean_kind

    @abstractmethod
    def __init__(self, betas, dynamic_threshold, clip_denoised):
        self.dynamic_threshold = dynamic_threshold
        self.clip_denoised = clip_denoised
        _, _, self.posterior_mean_coef1, self.posterior_mean_coef2, _ = _posterior_tables(betas)

    @abstractmethod
    def get_mean_and_xstart(self, x, t, model_output):
        pass

    def process_xstart(self, x):
        if self.dynamic_threshold:
            # the reference's `dynamic_thresholding(x, s=0.98)` (util/img_utils.py:8-15): the tensor TIMES the 0.98-quantile of
            # |x| over all of its elements (the batch included), then clipped to [-1, 1] -- not Imagen's per-image clip-and-divide
            x = torch.clip(x * torch.quantile(x.abs(), 0.98), -1.0, 1.0)
        if self.clip_denoised:
            x = x.clamp(-1, 1)
        return x

    def q_posterior_mean(self, x_start, x_t, t):
        assert x_start.shape == x_t.shape
        return (extract_and_expand(self.posterior_mean_coef1, t, x_start) * x_start
                + extract_and_expand(self.posterior_mean_coef2, t, x_t) * x_t)


@register_mean_processor(name="previous_x")
class PreviousXMeanProcessor(MeanProcessor):
    """The network predicts the posterior mean; x_0 is solved from it (reference :53-72).  The reference returns the network's
    split output AS the mean, and every step rule that adds to the mean in place (the Osmosis branch gaussian_diffusion.py:268,
    condition_methods.py:223; DDPM.p_sample :499) raises autograd's "view ... modified inplace" there: only DDIM.p_sample runs
    with it (`GaussianDiffusion.p_sample_loop` raises the same RuntimeError for the other two)."""
    hip_kernel = "osm_posterior"
    kernel_kind = 2

    def __init__(self, betas, dynamic_threshold, clip_denoised):
        super().__init__(betas, dynamic_threshold, clip_denoised)

    def kernel_coefs(self, t: int):
        """Row of osm_posterior_typed: c0 = d x0/d x = -coef2/coef1, c1 = -d x0/d out = -1/coef1 (fp32-rounded like
        extract_and_expand's .float(), then negated: exact), c2 / c3 unused (mean = out)."""
        return (-np.float32((self.posterior_mean_coef2 / self.posterior_mean_coef1)[t]), -np.float32((1.0 / self.posterior_mean_coef1)[t]),
                np.float32(self.posterior_mean_coef1[t]), np.float32(self.posterior_mean_coef2[t]))

    def predict_xstart(self, x_t, t, x_prev):
        c1 = extract_and_expand(1.0 / self.posterior_mean_coef1, t, x_t)
        c2 = extract_and_expand(self.posterior_mean_coef2 / self.posterior_mean_coef1, t, x_t)
        return c1 * x_prev - c2 * x_t

    def get_mean_and_xstart(self, x, t, model_output):
        return model_output, self.process_xstart(self.predict_xstart(x, t, model_output))


@register_mean_processor(name="start_x")
class StartXMeanProcessor(MeanProcessor):
    hip_kernel = "osm_posterior"
    kernel_kind = 1

    def __init__(self, betas, dynamic_threshold, clip_denoised):
        super().__init__(betas, dynamic_threshold, clip_denoised)

    def kernel_coefs(self, t: int):
        """Row of osm_posterior_typed: x0 = out, so d x0/d x = 0 and -d x0/d out = -1; c2 / c3 = the posterior mean's."""
        return (np.float32(0.0), np.float32(-1.0), np.float32(self.posterior_mean_coef1[t]), np.float32(self.posterior_mean_coef2[t]))

    def get_mean_and_xstart(self, x, t, model_output):
        x0 = self.process_xstart(model_output)
        return self.q_posterior_mean(x_start=x0, x_t=x, t=t), x0


@register_mean_processor(name="epsilon")
class EpsilonXMeanProcessor(MeanProcessor):
    hip_kernel = "osm_posterior"

    def __init__(self, betas, dynamic_threshold, clip_denoised):
        super().__init__(betas, dynamic_threshold, clip_denoised)
        ac = np.cumprod(1.0 - betas, axis=0)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / ac)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / ac - 1)

    def kernel_coefs(self, t: int):
        """(c0, c1, c2, c3) of osm_posterior for timestep index t, fp32-rounded like .float()."""
        return (np.float32(self.sqrt_recip_alphas_cumprod[t]), np.float32(self.sqrt_recipm1_alphas_cumprod[t]),
                np.float32(self.posterior_mean_coef1[t]), np.float32(self.posterior_mean_coef2[t]))

    def predict_xstart(self, x_t, t, eps):
        return (extract_and_expand(self.sqrt_recip_alphas_cumprod, t, x_t) * x_t
                - extract_and_expand(self.sqrt_recipm1_alphas_cumprod, t, eps) * eps)

    def get_mean_and_xstart(self, x, t, model_output):
        x0 = self.process_xstart(self.predict_xstart(x, t, model_output))
        return self.q_posterior_mean(x0, x, t), x0


class VarianceProcessor(ABC):
    hip_kernel = None
    kernel_kind = 0        # osm_posterior_typed's var_kind

    @abstractmethod
    def __init__(self, betas):
        pass

    @abstractmethod
    def get_variance(self, x, t):
        pass


@register_var_processor(name="fixed_small")
class FixedSmallVarianceProcessor(VarianceProcessor):
    hip_kernel = "osm_posterior"
    kernel_kind = 1

    def __init__(self, betas):
        self.posterior_variance = _posterior_tables(betas)[4]

    def kernel_coefs(self, t: int):
        """(log variance of the step, unused): log 0 = -inf at index 0, as in the reference (:182), where no noise is added."""
        with np.errstate(divide="ignore"):
            return np.float32(np.log(self.posterior_variance)[t]), np.float32(0.0)

    def get_variance(self, x, t):
        v = self.posterior_variance
        return extract_and_expand(v, t, x), extract_and_expand(np.log(v), t, x)


@register_var_processor(name="fixed_large")
class FixedLargeVarianceProcessor(VarianceProcessor):
    hip_kernel = "osm_posterior"
    kernel_kind = 1

    def __init__(self, betas):
        self.betas = betas
        self.posterior_variance = _posterior_tables(betas)[4]

    def kernel_coefs(self, t: int):
        return np.float32(np.log(np.append(self.posterior_variance[1], self.betas[1:]))[t]), np.float32(0.0)

    def get_variance(self, x, t):
        v = np.append(self.posterior_variance[1], self.betas[1:])
        return extract_and_expand(v, t, x), extract_and_expand(np.log(v), t, x)


@register_var_processor(name="learned")
class LearnedVarianceProcessor(VarianceProcessor):
    hip_kernel = "osm_posterior"
    kernel_kind = 2

    def __init__(self, betas):
        pass

    def kernel_coefs(self, t: int):
        return np.float32(0.0), np.float32(0.0)

    def get_variance(self, x, t):
        return torch.exp(x), x


@register_var_processor(name="learned_range")
class LearnedRangeVarianceProcessor(VarianceProcessor):
    hip_kernel = "osm_posterior"

    def __init__(self, betas):
        self.betas = betas
        pv = _posterior_tables(betas)[4]
        self.posterior_log_variance_clipped = np.log(np.append(pv[1], pv[1:]))

    def kernel_coefs(self, t: int):
        """(min_log, max_log) of osm_posterior for timestep index t."""
        return np.float32(self.posterior_log_variance_clipped[t]), np.float32(np.log(self.betas)[t])

    def get_variance(self, x, t):
        min_log = extract_and_expand(self.posterior_log_variance_clipped, t, x)
        max_log = extract_and_expand(np.log(self.betas), t, x)
        frac = (x + 1.0) / 2.0
        logvar = frac * max_log + (1 - frac) * min_log
        return torch.exp(logvar), logvar
