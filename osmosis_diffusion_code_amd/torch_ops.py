"""`torch.library` registration of the hot path (north_star: "driven from Python through PyTorch-ROCm
custom ops"): the UNet forward / input-gradient plans and the guidance-step kernels as dispatcher-visible operators in the
`osmosis::` namespace -- schema, fake-tensor (meta) implementations and autograd registration -- with the C ABI
(include/osmosis_hip.h, called through ctypes in ops.py / engine.py) underneath.  Nothing here computes: every operator
enqueues hand-written gfx950 kernels on the current HIP stream; there is no CPU implementation (the operators are
registered for the "cuda" device type only, which is HIP on ROCm).

    osmosis::unet_fwd(x, t, engine) -> out                  UNetModel.forward (reference unet.py:713-742); differentiable
                                                           w.r.t. x (condition_methods.py:188-191 back-propagates through it)
    osmosis::unet_bwd_data(grad_out, engine) -> dx          the recorded data-gradient plan of the same engine
    osmosis::posterior(model_out, x, coef) -> (pred_xstart, mean, log_variance)     gaussian_diffusion.py:349-376 +
                                                           posterior_mean_variance.py (every registered processor pair)
    osmosis::posterior_clip(...) -> (pred_xstart clamped, mean, log_variance, raw prediction), osmosis::clamp_bwd(g, raw) -> masked g
                                                           (`clip_denoised: True` of the shipped rgb-guidance config)
    osmosis::posterior_bwd(g, coef) -> d_model_out         d(pred_xstart)/d(model_out)^T g  (the chain rule into the UNet)
    osmosis::guide_update(mean, log_variance, g, dx_unet, noise, coef, scale4, clip) -> (x_next, grad)
                                                           condition_methods.py:186-221 update rule + the noise add of :262-271
    osmosis::phys_loss_grad(x0, y, phi, icfg, fcfg, n_inner, freeze_phi) -> (loss, grad_x0, phi_new)
                                                           measurements.py forward models + the inner phi optimisation
                                                           (cm.py:141-184), functional (phi is returned, not updated in place)

`engine` is an integer handle (`engine_handle(eng)`) because operator schemas carry tensors and scalars only; the handle
table holds weak references, so an engine dies with its model.  An engine keeps the activations of its LAST forward pass:
the autograd node of `unet_fwd` remembers which pass it belongs to (the engine's pass counter, a Python-side attribute, not an
operator output -- outputs are functions of the inputs alone, so fake-tensor tracing and `torch.library.opcheck` see pure
operators) and raises instead of differentiating through activations a later forward has overwritten."""
import itertools
import weakref
from typing import List, Tuple

import torch

from . import ops
from ._lib import OsmosisHipError, PhysDesc

_ENGINES = weakref.WeakValueDictionary()
_next_handle = itertools.count(1)


def engine_handle(eng) -> int:
    """A handle that is never reused (id(eng) can be, once the engine is collected -- and tickets restart at 0)."""
    h = getattr(eng, "_osm_handle", None)
    if h is None:
        h = eng._osm_handle = next(_next_handle)
    _ENGINES[h] = eng
    return h


def _engine(handle: int):
    eng = _ENGINES.get(int(handle))
    if eng is None:
        raise OsmosisHipError(f"osmosis:: operator called with a stale engine handle ({handle}): the engine was released")
    return eng


# ----------------------------------------------------------------------------------------------------------------- UNet
@torch.library.custom_op("osmosis::unet_fwd", mutates_args=(), device_types="cuda")
def unet_fwd(x: torch.Tensor, t: torch.Tensor, engine: int) -> torch.Tensor:
    eng = _engine(engine)
    if tuple(x.shape) != tuple(eng.x_in.shape):
        raise OsmosisHipError(f"osmosis::unet_fwd: engine is planned for x {tuple(eng.x_in.shape)}, got {tuple(x.shape)}")
    return eng.forward(x, t, need_grad=True).clone()


@unet_fwd.register_fake
def _unet_fwd_fake(x, t, engine):
    eng = _engine(engine)
    return x.new_empty((x.shape[0], eng.cout, x.shape[2], x.shape[3]))


@torch.library.custom_op("osmosis::unet_bwd_data", mutates_args=(), device_types="cuda")
def unet_bwd_data(grad_out: torch.Tensor, engine: int) -> torch.Tensor:
    """dL/dx of the engine's LAST forward pass (its activations are what the engine holds)."""
    return _engine(engine).backward(grad_out.contiguous()).clone()


@unet_bwd_data.register_fake
def _unet_bwd_fake(grad_out, engine):
    eng = _engine(engine)
    return grad_out.new_empty((grad_out.shape[0], eng.cin, grad_out.shape[2], grad_out.shape[3]))


def _unet_setup(ctx, inputs, output):
    ctx.engine = inputs[2]
    ctx.engine_obj = _engine(inputs[2])           # strong reference: the activations backward needs live in this engine
    ctx.ticket = ctx.engine_obj.ticket            # the forward pass this node belongs to (eager mode: checked below)


def _unet_backward(ctx, grad_out):
    if _engine(ctx.engine) is not ctx.engine_obj or ctx.engine_obj.ticket != ctx.ticket:
        raise RuntimeError("UNet activations were overwritten by a later forward before backward ran")
    return torch.ops.osmosis.unet_bwd_data(grad_out, ctx.engine), None, None


unet_fwd.register_autograd(_unet_backward, setup_context=_unet_setup)


# ------------------------------------------------------------------------------------------------------------ sampler step
def _chw(x):
    if x.dim() != 4 or x.dtype != torch.float32 or not x.is_contiguous():
        raise OsmosisHipError("osmosis:: guidance operators take contiguous fp32 [B,C,H,W] tensors")
    return x.shape[0], x.shape[2] * x.shape[3]


@torch.library.custom_op("osmosis::posterior", mutates_args=(), device_types="cuda")
def posterior(model_out: torch.Tensor, x: torch.Tensor, coef: torch.Tensor, mean_kind: int = 0,
              var_kind: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """coef: the 8 fp32 coefficients of the step (GaussianDiffusion.coef_table row, fetched by osm_fetch_coefs); mean_kind / var_kind:
    the processors' `kernel_kind` (osm_posterior_typed; 0, 0 = epsilon / learned_range)."""
    B, HW = _chw(x)
    x0, mean, logvar = (torch.empty_like(x) for _ in range(3))
    ops.posterior(model_out.contiguous(), x, coef, x0, mean, logvar, B, HW, mean_kind, var_kind)
    return x0, mean, logvar


@posterior.register_fake
def _posterior_fake(model_out, x, coef, mean_kind=0, var_kind=0):
    return torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)


@torch.library.custom_op("osmosis::posterior_clip", mutates_args=(), device_types="cuda")
def posterior_clip(model_out: torch.Tensor, x: torch.Tensor, coef: torch.Tensor, mean_kind: int = 0,
                   var_kind: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """osmosis::posterior with `clip_denoised` (process_xstart, posterior_mean_variance.py:43-50; configs/rgb_guidance_sample_config.yaml):
    (pred_xstart clamped to [-1, 1], mean formed from it, log_variance, the unclamped prediction for osmosis::clamp_bwd)."""
    B, HW = _chw(x)
    x0, mean, logvar, raw = (torch.empty_like(x) for _ in range(4))
    ops.posterior(model_out.contiguous(), x, coef, x0, mean, logvar, B, HW, mean_kind, var_kind, x0_raw=raw)
    return x0, mean, logvar, raw


@posterior_clip.register_fake
def _posterior_clip_fake(model_out, x, coef, mean_kind=0, var_kind=0):
    return torch.empty_like(x), torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)


@torch.library.custom_op("osmosis::clamp_bwd", mutates_args=(), device_types="cuda")
def clamp_bwd(g: torch.Tensor, x_raw: torch.Tensor, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    """g where lo <= x_raw <= hi, 0 elsewhere (NaN included): the backward of x_raw.clamp(lo, hi) on a gradient (osm_clamp_bwd)."""
    out = g.detach().clone().contiguous()
    ops.clamp_bwd(out, x_raw.contiguous(), lo, hi)
    return out


@clamp_bwd.register_fake
def _clamp_bwd_fake(g, x_raw, lo=-1.0, hi=1.0):
    return torch.empty_like(g)


@torch.library.custom_op("osmosis::posterior_bwd", mutates_args=(), device_types="cuda")
def posterior_bwd(g: torch.Tensor, coef: torch.Tensor) -> torch.Tensor:
    B, HW = _chw(g)
    d_out = torch.zeros(B, 2 * g.shape[1], g.shape[2], g.shape[3], device=g.device, dtype=torch.float32)
    ops.posterior_bwd(g, coef, d_out, B, HW)
    return d_out


@posterior_bwd.register_fake
def _posterior_bwd_fake(g, coef):
    return g.new_empty((g.shape[0], 2 * g.shape[1], g.shape[2], g.shape[3]))


@torch.library.custom_op("osmosis::guide_update", mutates_args=(), device_types="cuda")
def guide_update(mean: torch.Tensor, log_variance: torch.Tensor, g: torch.Tensor, dx_unet: torch.Tensor, noise: torch.Tensor,
                 coef: torch.Tensor, scale4: torch.Tensor, clip: float) -> Tuple[torch.Tensor, torch.Tensor]:
    B, HW = _chw(mean)
    x_next, grad = torch.empty_like(mean), torch.empty_like(mean)
    ops.guide_update(mean, log_variance, g, dx_unet, noise, coef, scale4, clip, x_next, grad, B, HW)
    return x_next, grad


@guide_update.register_fake
def _guide_update_fake(mean, log_variance, g, dx_unet, noise, coef, scale4, clip):
    return torch.empty_like(mean), torch.empty_like(mean)


@torch.library.custom_op("osmosis::guide_update_rng", mutates_args=(), device_types="cuda")
def guide_update_rng(mean: torch.Tensor, log_variance: torch.Tensor, g: torch.Tensor, dx_unet: torch.Tensor, coef: torch.Tensor,
                     scale4: torch.Tensor, clip: float, seed: int, step: torch.Tensor, step_offset: int, img0: int,
                     img_stride: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """osmosis::guide_update with the step noise drawn in the kernel (Philox-4x32-10: key = seed, counter = (element / 4,
    img0 + b * img_stride, step[0] + step_offset)); a pure function of its inputs.  Returns (x_next, grad, the noise drawn)."""
    B, HW = _chw(mean)
    x_next, grad, noise = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
    ops.guide_update_rng(mean, log_variance, g, dx_unet, coef, scale4, clip, x_next, grad, noise, B, HW, seed, step,
                         step_offset=step_offset, img0=img0, img_stride=img_stride)
    return x_next, grad, noise


@guide_update_rng.register_fake
def _guide_update_rng_fake(mean, log_variance, g, dx_unet, coef, scale4, clip, seed, step, step_offset, img0, img_stride):
    return torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)


@torch.library.custom_op("osmosis::ddim_update", mutates_args=(), device_types="cuda")
def ddim_update(x0: torch.Tensor, x: torch.Tensor, g: torch.Tensor, dx_unet: torch.Tensor, noise: torch.Tensor, coef: torch.Tensor,
                dcoef: torch.Tensor, scale4: torch.Tensor, clip: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """DDIM step + guidance (osm_ddim_update): coef = the posterior row, dcoef = row of GaussianDiffusion.ddim_table()."""
    B, HW = _chw(x0)
    x_next, grad = torch.empty_like(x0), torch.empty_like(x0)
    ops.ddim_update(x0, x, g, dx_unet, noise, coef, dcoef, scale4, clip, x_next, grad, B, HW)
    return x_next, grad


@ddim_update.register_fake
def _ddim_update_fake(x0, x, g, dx_unet, noise, coef, dcoef, scale4, clip):
    return torch.empty_like(x0), torch.empty_like(x0)


PHYS_ICFG = ("kind", "depth_type", "weight_type", "wdepth_type", "loss_type", "optimizer")
PHYS_FCFG = ("dval0", "dval1", "dval2", "wval0", "wval1", "wval2", "gamma_avrg", "gamma_val", "eta0", "eta1", "eta2")


def phys_config(desc: PhysDesc) -> Tuple[List[int], List[float]]:
    """The (int list, float list) form of a physics descriptor (what ConditioningMethod._prepare fills) for phys_loss_grad."""
    return ([int(getattr(desc, k)) for k in PHYS_ICFG],
            [float(v) for v in (*desc.dval, *desc.wval, desc.gamma_avrg, desc.gamma_val, *desc.eta)])


@torch.library.custom_op("osmosis::phys_loss_grad", mutates_args=(), device_types="cuda")
def phys_loss_grad(x0: torch.Tensor, y: torch.Tensor, phi: torch.Tensor, icfg: List[int], fcfg: List[float], n_inner: int,
                   freeze_phi: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """n_inner x (reduce, finalize + phi step) of the physical model's loss, then loss and dL/dx0 at the last phi and its step
    (osm_phys_optimize).  SGD / GD only here (Adam carries optimizer state across steps: use the conditioning method)."""
    B, HW = _chw(x0)
    if len(icfg) != len(PHYS_ICFG) or len(fcfg) != len(PHYS_FCFG):
        raise OsmosisHipError("osmosis::phys_loss_grad: icfg / fcfg must come from torch_ops.phys_config(desc)")
    d = PhysDesc()
    for k, v in zip(PHYS_ICFG, icfg):
        setattr(d, k, int(v))
    if d.optimizer != 0:
        raise OsmosisHipError("osmosis::phys_loss_grad is functional: optimizer state (adam, ...) lives with the conditioning method")
    if d.kind == 3 and not freeze_phi:
        raise OsmosisHipError("osmosis::phys_loss_grad: the identity operator (kind 3) has no parameters: pass freeze_phi=True")
    for i in range(3):
        d.dval[i], d.wval[i], d.eta[i] = fcfg[i], fcfg[3 + i], fcfg[8 + i]
    d.gamma_avrg, d.gamma_val = fcfg[6], fcfg[7]
    d.B, d.HW = B, HW
    dev = x0.device
    phi_new = phi.detach().clone().contiguous()
    part = torch.empty(B * ops.phys_nblk(HW) * 16, device=dev, dtype=torch.float32)
    red = torch.zeros(B * 16, device=dev, dtype=torch.float32)
    loss = torch.zeros(B, device=dev, dtype=torch.float32)
    g = torch.empty_like(x0)
    ops.phys_optimize(d, x0, y.contiguous(), phi_new, part, red, loss, g, 1 if freeze_phi else n_inner, freeze_phi)
    return loss, g, phi_new


@phys_loss_grad.register_fake
def _phys_loss_grad_fake(x0, y, phi, icfg, fcfg, n_inner, freeze_phi):
    return x0.new_empty((x0.shape[0],)), torch.empty_like(x0), torch.empty_like(phi)


OPS = ("unet_fwd", "unet_bwd_data", "posterior", "posterior_clip", "clamp_bwd", "posterior_bwd", "guide_update", "guide_update_rng", "ddim_update", "phys_loss_grad")
