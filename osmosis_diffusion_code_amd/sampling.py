"""Per-image driver contract of the reference's `osmosis_sampling.py` main loop (:117-345), as functions
(SURVEY.md section 8 rows a23 and N2): what the reference does around `p_sample_loop` for every image.

    restore_image(model, ref_img, cfg)      one image (or one independent batch) through the guided sampler
    postprocess(...)                        the outputs the reference saves / logs for that image
    restore_images(model, images, cfg, ...) a list of images, sharded images[rank::world] (section 8e)

`cfg` is the parsed YAML of the reference (`configs/osmosis_sample_config.yaml`): keys `measurement`
{operator, noise}, `conditioning` {method, params}, `diffusion`, `sample_pattern`, `aux_loss`, `unet_model`,
`manual_seed`, `degamma_input`, `rgb_guidance`.

    load_config(path)                       the YAML file -> that dictionary (osmosis_utils/utils.py:357-360,466-476)
    save_outputs(post, out_dir, name, ...)  the files the reference writes per image (osmosis_sampling.py:319-353)
"""
import os

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


def load_config(path):
    """The reference's `arguments_from_file` (osmosis_utils/utils.py:466-476 -> load_yaml :357-360): the YAML file as a plain
    dictionary (the reference copies the same keys onto an argparse.Namespace; `restore_image(s)` reads them by key).
    yaml.FullLoader like the reference, so `1e-5` stays the STRING the operators parse themselves and `32, 16, 8` a string."""
    import yaml
    with open(path) as f:
        cfg = yaml.load(f, Loader=yaml.FullLoader)
    if not isinstance(cfg, dict):
        raise ValueError(f"{path}: expected a mapping at the top level of the configuration file")
    return cfg


def _to_pil_u8(t):
    """torchvision.transforms.functional.to_pil_image on a float tensor [C,H,W] (0.14.1, functional.py:257-340 as the
    reference uses it at osmosis_sampling.py:321-337): `pic.mul(255).byte()` -- TRUNCATION, not rounding -- then HWC;
    one channel -> mode 'L', three -> 'RGB'."""
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.dim() != 3 or t.shape[0] not in (1, 3):
        raise ValueError(f"pic should be 2/3 dimensional with 1 or 3 channels. Got {tuple(t.shape)}")
    if t.is_floating_point():
        t = t.mul(255).byte()
    arr = t.permute(1, 2, 0).contiguous().numpy()
    return arr[:, :, 0] if arr.shape[2] == 1 else arr


def make_grid(tensors, nrow=8, padding=2, pad_value=0.0):
    """torchvision.utils.make_grid (0.14.1) for a list of equally sized [C,H,W] tensors, normalize=False: the list is stacked
    (dtype promotion as torch.stack does it: the viridis depth is float64), single-channel images are repeated to three, tile k
    sits at row k // xmaps, column k % xmaps of a (pad_value)-filled canvas with `padding` pixels before every tile and after
    the last one."""
    dt = tensors[0].dtype
    for t in tensors[1:]:
        dt = torch.promote_types(dt, t.dtype)
    x = torch.stack([t.to(dt) for t in tensors], 0)
    if x.shape[1] == 1:
        x = x.repeat(1, 3, 1, 1)
    if x.shape[0] == 1:              # make_grid returns the single image itself (no border)
        return x[0]
    n = x.shape[0]
    xmaps = min(nrow, n)
    ymaps = (n + xmaps - 1) // xmaps
    h, w = x.shape[2] + padding, x.shape[3] + padding
    grid = x.new_full((x.shape[1], h * ymaps + padding, w * xmaps + padding), pad_value)
    for k in range(n):
        r, c = divmod(k, xmaps)
        grid[:, r * h + padding: r * h + padding + x.shape[2], c * w + padding: c * w + padding + x.shape[3]] = x[k]
    return grid


def output_images(post, ref_img, gt_rgb_01=None, gt_depth_01=None):
    """The uint8 arrays of the five images the reference writes for one restored image (osmosis_sampling.py:319-353):
    `input` = the reference image in [0,1], `rgb` = the clipped restoration, `depth_color` = viridis of the percentile-normalised
    depth, `depth_raw` = the min-max-normalised depth (one channel), `grid` = make_grid([input, rgb, depth_color] (+ [zeros,
    gt rgb, gt depth colour] when a ground truth exists: :341-344), nrow=3, pad_value=1.) through the reference's
    clip_image(scale=False, move=False, is_uint8=True) (clamp, truncate)."""
    ref01 = 0.5 * (ref_img.detach().cpu()[0] + 1)
    col = depth_color(post)
    tiles = [ref01, post["rgb_01_clip"], col]
    if gt_rgb_01 is not None:
        tiles += [torch.zeros_like(post["rgb_01"]), gt_rgb_01, utilso.depth_tensor_to_color_image(gt_depth_01)]
    grid = make_grid(tiles, nrow=3, pad_value=1.0)
    grid = (grid * 255).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous().numpy()
    return {"input": _to_pil_u8(ref01), "rgb": _to_pil_u8(post["rgb_01_clip"]), "depth_color": _to_pil_u8(col),
            "depth_raw": _to_pil_u8(post["depth_mm"]), "grid": grid}


def save_outputs(post, ref_img, out_dir, name, global_ii=0, save_singles=True, save_grids=True, gt_rgb_01=None,
                 gt_depth_01=None, rgb_guidance=None):
    """Writes what the reference writes for one image (osmosis_sampling.py:84-104 directory layout, :319-353 files):
    `<out_dir>/single_images/{input,rgb,depth_color,depth_raw}/<name>.png` and `<out_dir>/grid_results/<name>_g<ii>_grid.png`.
    Returns {kind: path}.  (`<name>_process.png` is written by the sampler itself when `record` is on.)
    A result of the rgb-guidance branch (`restore_image` with `rgb_guidance: True`: the dict carries `sample`, no phi; override
    with `rgb_guidance=`) is written as the reference's second branch writes it (:382-401): the same four single images -- the
    min-max depth as a three-channel PNG, it is `depth.repeat(3, 1, 1)` there -- and the grid as `<name>.png`."""
    if rgb_guidance is None:
        rgb_guidance = "sample" in post and "phi" not in post
    from PIL import Image
    imgs = output_images(post, ref_img, gt_rgb_01, gt_depth_01)
    paths = {}
    if save_singles:
        for kind in ("input", "rgb", "depth_color", "depth_raw"):
            d = os.path.join(out_dir, "single_images", kind)
            os.makedirs(d, exist_ok=True)
            paths[kind] = os.path.join(d, f"{name}.png")
            Image.fromarray(imgs[kind], mode="L" if imgs[kind].ndim == 2 else "RGB").save(paths[kind])
    if save_grids:
        d = os.path.join(out_dir, "grid_results")
        os.makedirs(d, exist_ok=True)
        paths["grid"] = os.path.join(d, f"{name}.png" if rgb_guidance else f"{name}_g{global_ii}_grid.png")
        Image.fromarray(imgs["grid"], mode="RGB").save(paths["grid"])
    return paths


def depth_color(post):
    """viridis rendering of the percentile-normalised depth (what the reference writes to depth_pmm_color)."""
    return utilso.depth_tensor_to_color_image(post["depth_pmm"])


def rgb_guidance_result(sample, measurement):
    """What the reference driver derives from the sample of an rgb-guidance / non-osmosis chain (osmosis_sampling.py:366-380; image 0
    of `sample`): RGB and depth split, the clipped [0, 1] RGB, the min-max and the percentile-normalised three-channel depth.  No phi,
    no recomposition.  CPU tensors."""
    depth3 = sample[0, -1].repeat(3, 1, 1)
    return {"sample": sample, "rgb": sample[0, 0:-1], "rgb_01_clip": torch.clamp(0.5 * (sample[0, 0:-1] + 1), 0, 1),
            "depth_mm": utilso.min_max_norm_range(depth3, vmin=0, vmax=1, is_uint8=False),
            "depth_pmm": utilso.min_max_norm_range_percentile(depth3, percent_low=0.05, percent_high=0.99),
            "measurement": measurement}


def restore_image(model, ref_img, cfg, device=None, image_idx=0, x_scale=1.0, same_seed_per_image=False,
                  postprocess_batch=True, **loop_kwargs):
    """One image through the reference's per-image sequence: fresh operator / noiser / conditioning method /
    sampler (:142-155), y = noiser(ref) (+ degamma), manual_seed + x_T ~ N(0, I) per global iteration
    (:191-196), guided p_sample_loop, post-processing.  Returns a list with one dict per global iteration.

    `ref_img` may carry B > 1 images: one batch of independent chains.  With `same_seed_per_image` every image of
    the batch starts from the SAME x_T and receives the SAME per-step noise -- exactly what B separate calls (each
    re-seeded with `manual_seed`, as the reference driver does per image) would draw -- so an image's result does not
    depend on how images are grouped into batches or spread over ranks.  `postprocess_batch=False` skips the
    reference's image-0-only post-processing (a caller that post-processes every image of the batch itself)."""
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
        if same_seed_per_image and shape[0] > 1:
            x_start = torch.randn([1] + shape[1:], device=device).repeat(shape[0], 1, 1, 1)
            loop_kwargs = dict(loop_kwargs, shared_noise=True)
        else:
            x_start = torch.randn(shape, device=device)
        if x_scale != 1.0:          # sub-chains started at a low timestep (tools/full_chain.py --last)
            x_start = x_start * x_scale
        rgb_guidance = cfg.get("rgb_guidance", False)
        ret = sampler.p_sample_loop(
            model=model, x_start=x_start, measurement=y_n, measurement_cond_fn=cond.conditioning,
            record=cfg.get("record_process", False) and loop_kwargs.get("save_grids_path") is not None, save_root=None,
            pretrain_model=pretrain, image_idx=image_idx, record_every=cfg.get("record_every", 150),
            rgb_guidance=rgb_guidance, sample_pattern=cfg["sample_pattern"], global_iteration=global_ii, **loop_kwargs)
        if rgb_guidance or pretrain != "osmosis":
            # the rgb-guidance / non-osmosis chain returns the sample only (gaussian_diffusion.py:340); the
            # reference driver splits it into RGB and depth (osmosis_sampling.py:366-380): no phi, no recomposition
            results.append(rgb_guidance_result(ret.detach().cpu(), y_n.detach().cpu()))
            continue
        sample, variable_dict, loss, out_xstart = ret
        if postprocess_batch:
            post = postprocess(out_xstart, variable_dict, ref_img, measure["operator"], loss)
        else:
            post = dict(phi={k: v.detach().cpu() for k, v in variable_dict.items()},
                        loss=None if loss is None else np.asarray(loss))
        post.update(sample=sample.detach().cpu(), pred_xstart=out_xstart, measurement=y_n.detach().cpu())
        results.append(post)
    return results


def postprocess_each(out_xstart, variable_dict, ref_img, operator_cfg, loss=None):
    """`postprocess` for every image of a batch (the reference only ever has one)."""
    outs = []
    for b in range(out_xstart.shape[0]):
        vd = {k: v[b:b + 1] for k, v in variable_dict.items()}
        outs.append(postprocess(out_xstart[b:b + 1], vd, ref_img[b:b + 1], operator_cfg,
                                None if loss is None else np.asarray(loss)[b:b + 1]))
    return outs


def restore_images(model, images, cfg, rank=0, world=1, device=None, gt_rgb=None, batch_size=1, **loop_kwargs):
    """images[rank::world] (no collective on the path).  Returns {image index: result dict of the
    last global iteration}; when `gt_rgb` (list of [3,H,W] in [0,1]) is given each result carries `psnr`.

    `batch_size` > 1 carries that many of this rank's images per pass (BASELINE config 4: 8 images per GPU): they
    are independent chains with per-image phi, per-image reductions and -- like the reference's per-image
    `manual_seed` -- the same x_T / noise stream each, so image i's result is the one a batch-1 run gives."""
    out = {}
    mine = shard_indices(len(images), rank, world)
    for k in range(0, len(mine), max(1, batch_size)):
        idxs = mine[k:k + max(1, batch_size)]
        if len(idxs) == 1:
            res = [restore_image(model, images[idxs[0]], cfg, device=device, image_idx=idxs[0], **loop_kwargs)[-1]]
        else:
            ref = torch.cat([images[i] for i in idxs], 0)
            full = restore_image(model, ref, cfg, device=device, image_idx=idxs[0], same_seed_per_image=True,
                                 postprocess_batch=False, **loop_kwargs)[-1]
            if "pred_xstart" in full:
                res = postprocess_each(full["pred_xstart"], full["phi"], ref, cfg["measurement"]["operator"], full["loss"])
                for b, r in enumerate(res):
                    r.update(sample=full["sample"][b:b + 1], pred_xstart=full["pred_xstart"][b:b + 1],
                             measurement=full["measurement"][b:b + 1])
            else:
                res = [rgb_guidance_result(full["sample"][b:b + 1], full["measurement"][b:b + 1]) for b in range(len(idxs))]
        for i, r in zip(idxs, res):
            if gt_rgb is not None:
                r["psnr"] = float(utilso.psnr(r["rgb_01_clip"], gt_rgb[i]))
            out[i] = r
    return out
