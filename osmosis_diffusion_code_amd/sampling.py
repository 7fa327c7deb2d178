"""Per-image driver contract of the reference's `osmosis_sampling.py` main loop (:117-345), as functions
(SURVEY.md section 8 rows a23 and N2): what the reference does around `p_sample_loop` for every image.

    restore_image(model, ref_img, cfg)      one image (or one independent batch) through the guided sampler
    postprocess(...)                        the outputs the reference saves / logs for that image
    restore_images(model, images, cfg, ...) a list of images, sharded images[rank::world] (section 8e)

`cfg` is the parsed YAML of the reference (`configs/osmosis_sample_config.yaml`): keys `measurement`
{operator, noise}, `conditioning` {method, params}, `diffusion`, `sample_pattern`, `aux_loss`, `unet_model`,
`manual_seed`, `degamma_input`, `rgb_guidance`.  File output (PNG grids, logger) stays with the caller.
"""
import numpy as np
import torch

from .guided_diffusion.condition_methods import get_conditioning_method
from .guided_diffusion.gaussian_diffusion import create_sampler
from .guided_diffusion.measurements import get_noise, get_operator
from .osmosis_utils import utils as utilso
from .sharding import shard_indices


def global_iterations(sample_pattern):
    """osmosis_sampling.py:182-188."""
    if sample_pattern["pattern"] == "original":
        return 1
    if sample_pattern["pattern"] == "pcgs":
        return sample_pattern["global_N"]
    raise ValueError(f"Unrecognized sample pattern: {sample_pattern['pattern']}")


def degamma(y):
    """osmosis_sampling.py:173-175 (haze configs): [-1,1] image -> linear light, back to [-1,1]."""
    return 2 * torch.pow(0.5 * (y + 1), 2.2) - 1


def postprocess(out_xstart, variable_dict, ref_img, operator_cfg, loss=None):
    """Outputs of one restored image (osmosis_sampling.py:199-300), all CPU tensors.

    out_xstart [B,4,H,W] (the final pred_xstart -- the reference saves THAT, not the final x_t), ref_img
    [B,3,H,W] in [-1,1]; like the reference only image 0 of the batch is post-processed."""
    out_xstart = out_xstart.detach().cpu()
    ref = ref_img.detach().cpu()
    ref_img_01 = 0.5 * (ref[0] + 1)
    sample_rgb = out_xstart[0, 0:-1, :, :]
    depth = out_xstart[0, -1, :, :].unsqueeze(0)
    rgb01 = 0.5 * (sample_rgb + 1)
    out = {
        "rgb": sample_rgb,
        "rgb_01": rgb01,
        "rgb_01_clip": torch.clamp(rgb01, min=0, max=1),
        "depth": depth,
        "depth_mm": utilso.min_max_norm_range(depth[0].unsqueeze(0)),
        "depth_pmm": utilso.min_max_norm_range_percentile(depth, vmin=0, vmax=1, percent_low=0.03,
                                                          percent_high=0.99, is_uint8=False),
    }
    depth_calc = utilso.convert_depth(depth.repeat(3, 1, 1), depth_type=operator_cfg["depth_type"],
                                      value=operator_cfg["value"])
    name = operator_cfg["name"]
    ones = torch.ones_like(sample_rgb)
    phi_inf = variable_dict["phi_inf"].cpu().squeeze(0) * ones
    if "underwater_physical_revised" in name:
        phi_a = variable_dict["phi_a"].cpu().squeeze(0) * ones
        phi_b = variable_dict["phi_b"].cpu().squeeze(0) * ones
    elif "haze" in name or "underwater_physical" in name:
        phi_a = phi_b = variable_dict["phi_ab"].cpu().squeeze(0) * ones
    else:
        raise NotImplementedError("Operator can be for 'underwater' or 'haze' ")
    backscatter = phi_inf * (1 - torch.exp(-phi_b * depth_calc))
    attenuation = torch.exp(-phi_a * depth_calc)
    forward_pred = rgb01 * attenuation + backscatter
    degraded = 2 * forward_pred - 1
    out.update(
        depth_calc=depth_calc, backscatter=backscatter, attenuation=attenuation,
        forward_predicted=forward_pred, degraded=degraded,
        norm_loss_final=float(np.round(torch.linalg.norm(degraded - ref).numpy(), decimals=3)),
        rgb_recon=torch.exp(phi_a * depth_calc) * (ref_img_01 - backscatter),   # "clean" image from phi and the input
        phi={k: v.detach().cpu() for k, v in variable_dict.items()},
        loss=None if loss is None else np.asarray(loss),
    )
    return out


def depth_color(post):
    """viridis rendering of the percentile-normalised depth (what the reference writes to depth_pmm_color)."""
    return utilso.depth_tensor_to_color_image(post["depth_pmm"])


def restore_image(model, ref_img, cfg, device=None, image_idx=0, x_scale=1.0, **loop_kwargs):
    """One image through the reference's per-image sequence: fresh operator / noiser / conditioning method /
    sampler (:142-155), y = noiser(ref) (+ degamma), manual_seed + x_T ~ N(0, I) per global iteration
    (:191-196), guided p_sample_loop, post-processing.  Returns a list with one dict per global iteration."""
    device = device if device is not None else ref_img.device
    measure, cond_cfg = cfg["measurement"], cfg["conditioning"]
    op_cfg = dict(measure["operator"])
    op_cfg["batch_size"] = ref_img.shape[0]
    operator = get_operator(device=device, **op_cfg)
    noiser = get_noise(**measure["noise"])
    cond = get_conditioning_method(cond_cfg["method"], operator, noiser, **cond_cfg["params"],
                                   **cfg["sample_pattern"], **cfg["aux_loss"])
    sampler = create_sampler(**cfg["diffusion"])
    ref_img = ref_img.to(device)
    y_n = noiser(ref_img)
    if cfg.get("degamma_input", False):
        y_n = degamma(y_n)
    pretrain = cfg["unet_model"]["pretrain_model"]
    shape = list(ref_img.shape)
    shape[1] = 4 if pretrain == "osmosis" else shape[1]
    results = []
    for global_ii in range(global_iterations(cfg["sample_pattern"])):
        torch.manual_seed(cfg.get("manual_seed", 0))
        x_start = torch.randn(shape, device=device)
        if x_scale != 1.0:          # sub-chains started at a low timestep (tools/full_chain.py --last)
            x_start = x_start * x_scale
        sample, variable_dict, loss, out_xstart = sampler.p_sample_loop(
            model=model, x_start=x_start, measurement=y_n, measurement_cond_fn=cond.conditioning,
            record=False, save_root=None, pretrain_model=pretrain, image_idx=image_idx,
            rgb_guidance=cfg.get("rgb_guidance", False), sample_pattern=cfg["sample_pattern"],
            global_iteration=global_ii, **loop_kwargs)
        post = postprocess(out_xstart, variable_dict, ref_img, measure["operator"], loss)
        post.update(sample=sample.detach().cpu(), pred_xstart=out_xstart, measurement=y_n.detach().cpu())
        results.append(post)
    return results


def restore_images(model, images, cfg, rank=0, world=1, device=None, gt_rgb=None):
    """images[rank::world] (no collective on the path).  Returns {image index: result dict of the
    last global iteration}; when `gt_rgb` (list of [3,H,W] in [0,1]) is given each result carries `psnr`."""
    out = {}
    for i in shard_indices(len(images), rank, world):
        res = restore_image(model, images[i], cfg, device=device, image_idx=i)[-1]
        if gt_rgb is not None:
            res["psnr"] = float(utilso.psnr(res["rgb_01_clip"], gt_rgb[i]))
        out[i] = res
    return out
