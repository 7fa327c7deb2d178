#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (/root/reference, Python) on CPU.

Runs only in the build container (the reference does not travel to the GPU box).  It puts
/root/reference and three import-time stand-ins (torchvision / cv2 / natsort, which the
reference imports but never touches on the numeric path) on sys.path, drives the reference's own
classes, and writes small .npz fixtures to tests/golden/.  Fixtures are data only (inputs and
expected outputs); no reference source is stored.

    python oracle/tools/gen_golden.py            # rewrites tests/golden/*.npz

Sections (SURVEY.md section 8c recipe):
  spacing.json    (round 6) space_timesteps over every argument form it accepts / refuses
  schedules.npz   float64 tables of create_sampler for T' in {1000, 250, 10}
  blocks.npz      GroupNorm32 / timestep_embedding / ResBlock{plain,skip,up,down} /
                  AttentionBlock{legacy,new}: inputs, params, outputs, input-gradients
  tiny_unet.npz   create_model(tiny) with oracle.unet_ref.seeded_state_dict weights: y, dL/dx
  loop_<op>.npz   10-step guided p_sample_loop for each of the 3 physical operators with the
                  exact noise tensors the reference drew, per-step traces
  prior_inverse.npz   unconditional RGBD-prior sampler (osmosis_utils/diffusion.py)
  loop_optimizers.npz (round 6) the guided loop with each torch optimizer of utils.get_optimizer on phi: losses, phi, final image
  loop_processors.npz (round 6) the Osmosis loop and the rgb-guidance chains with the previous_x / start_x mean processors and the
                      fixed_small / fixed_large / learned variance processors
  loop_clip.npz       (round 6) `clip_denoised: True` (the shipped rgb-guidance config's setting): rgb-guidance chains, the Osmosis loop,
                      process_xstart with dynamic thresholding
  loop_record.npz     (round 6) the `<name>_process.png` grid of p_sample_loop(record=True) as a uint8 array
  loop_ps.npz         rgb-guidance chains (`ps` conditioning) through DDPM.p_sample and DDIM.p_sample
  postprocess.npz     depth normalisation / colour map / convert_depth helpers of osmosis_utils/utils.py
  unet_variants.npz   (round 5) tiny UNets with conv up / down-sampling layers, additive conditioning, class conditioning
  outputs.npz         (round 5) the five per-image output files of osmosis_sampling.py:319-353 as uint8 arrays
  configs.json        (round 5) configs/*.yaml as parsed by the reference's load_yaml
  full_unet.npz       (round 4) the full 552.8 M-parameter architecture through the real reference at 256 x 256 (every 4th pixel of y and
                      of the input gradient + norms, two timesteps)
  full_step.npz       (round 4) two guided steps of the real reference's loop with the full network (t = 299, then 0): subsampled traces
  fp16_reference.npz  (round 4) the reference with convert_to_fp16() applied, on CPU: tiny UNet forward / input gradient and the
                      10-step guided loop in fp16, next to the fp32 reference on the same inputs
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, "ref_stubs"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from guided_diffusion import unet as R_unet  # noqa: E402
from guided_diffusion import nn as R_nn  # noqa: E402
from guided_diffusion import gaussian_diffusion as R_gd  # noqa: E402
from guided_diffusion.measurements import get_operator, get_noise  # noqa: E402
from guided_diffusion.condition_methods import get_conditioning_method  # noqa: E402

from oracle.unet_ref import UNetConfig, seeded_state_dict  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
torch.set_num_threads(8)

TINY_KW = dict(image_size=256, num_channels=32, num_res_blocks=1, channel_mult="1,2,2",
               attention_resolutions="128,64", num_head_channels=16, num_heads=4,
               learn_sigma=True, use_scale_shift_norm=True, resblock_updown=True,
               pretrain_model="osmosis")


def npy(t):
    return t.detach().cpu().numpy()


def randomize_(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.ndim == 1:
                v = torch.randn(p.shape, generator=g)
                if "norm" in name or name.endswith("layers.0.weight") or name.endswith("layers.0.bias"):
                    p.copy_(1.0 + 0.1 * v if name.endswith("weight") else 0.05 * v)
                else:
                    p.copy_(0.05 * v)
            else:
                fan = int(np.prod(p.shape[1:]))
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 / np.sqrt(fan)))


def to_pil_u8(pic):
    """torchvision 0.14.1 to_pil_image on a float tensor (absent from this image): pic.mul(255).byte() -- truncation -- then HWC."""
    if pic.dim() == 2:
        pic = pic.unsqueeze(0)
    a = pic.mul(255).byte().permute(1, 2, 0).numpy()
    return a[:, :, 0] if a.shape[2] == 1 else a


def tv_make_grid(lst, nrow=8, pad_value=0.0, padding=2):
    """torchvision 0.14.1 utils.make_grid for a list of equally sized [C,H,W] tensors, normalize=False."""
    t = torch.stack(lst, dim=0)
    nmaps = t.size(0)
    xmaps = min(nrow, nmaps)
    ymaps = int(np.ceil(float(nmaps) / xmaps))
    height, width = int(t.size(2) + padding), int(t.size(3) + padding)
    grid = t.new_full((t.size(1), height * ymaps + padding, width * xmaps + padding), pad_value)
    k = 0
    for yy in range(ymaps):
        for xx in range(xmaps):
            if k >= nmaps:
                break
            grid.narrow(1, yy * height + padding, height - padding).narrow(2, xx * width + padding, width - padding).copy_(t[k])
            k += 1
    return grid


def gen_schedules():
    out = {}
    for tag, kw in {"T1000": dict(steps=1000, timestep_respacing=1000),
                    "T250": dict(steps=1000, timestep_respacing="250"),
                    "T10": dict(steps=1000, timestep_respacing=[10])}.items():
        s = R_gd.create_sampler(sampler="ddpm", noise_schedule="linear", model_mean_type="epsilon",
                                model_var_type="learned_range", dynamic_threshold=False,
                                clip_denoised=False, rescale_timesteps=False, **kw)
        out[f"{tag}.betas"] = s.betas
        out[f"{tag}.timestep_map"] = np.array(s.timestep_map, dtype=np.int64)
        out[f"{tag}.alphas_cumprod"] = s.alphas_cumprod
        out[f"{tag}.sqrt_recip_alphas_cumprod"] = s.mean_processor.sqrt_recip_alphas_cumprod
        out[f"{tag}.sqrt_recipm1_alphas_cumprod"] = s.mean_processor.sqrt_recipm1_alphas_cumprod
        out[f"{tag}.posterior_mean_coef1"] = s.mean_processor.posterior_mean_coef1
        out[f"{tag}.posterior_mean_coef2"] = s.mean_processor.posterior_mean_coef2
        out[f"{tag}.posterior_log_variance_clipped"] = s.var_processor.posterior_log_variance_clipped
        out[f"{tag}.log_betas"] = np.log(s.var_processor.betas)
    out["cosine50.betas"] = R_gd.get_named_beta_schedule("cosine", 50)
    out["space_1000_10_15_20"] = np.array(sorted(R_gd.space_timesteps(300, [10, 15, 20])), dtype=np.int64)
    out["space_ddim25"] = np.array(sorted(R_gd.space_timesteps(1000, "ddim25")), dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "schedules.npz"), **out)


def gen_spacing():
    """(round 6) `space_timesteps` (gaussian_diffusion.py:373-426) over the argument forms it accepts -- an int, a list, comma-separated
    section counts, the "ddimN" strides -- and the ones it refuses (ValueError), as JSON: {"<num_timesteps>|<spec repr>": sorted steps |
    "ValueError"}.  Python's banker's round() decides where the fractional strides land (:422)."""
    import json
    cases = [(1000, 1000), (1000, 250), (1000, "250"), (1000, "100"), (1000, "10,20,30"), (1000, [10, 15, 20]), (300, "10,15,20"),
             (1000, "ddim25"), (1000, "ddim50"), (1000, "ddim1000"), (1000, "ddim30"), (1000, "ddim7"), (100, "7"), (100, "3,3,3"),
             (97, "5,11"), (1000, "333,333,334"), (1000, "1"), (1000, "1,1,1"), (10, "11"), (1000, "500,600"), (50, "ddim13"),
             (1000, "13,0,27"), (7, [7]), (1000, "999")]
    out = {}
    for n, spec in cases:
        try:
            out[f"{n}|{spec!r}"] = sorted(int(v) for v in R_gd.space_timesteps(n, spec))
        except ValueError:
            out[f"{n}|{spec!r}"] = "ValueError"
    with open(os.path.join(OUT, "spacing.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("spacing cases", len(out), "refused", sum(v == "ValueError" for v in out.values()))


def gen_blocks():
    out = {}
    g = torch.Generator().manual_seed(11)

    # timestep embedding
    t = torch.tensor([0.0, 1.0, 37.0, 999.0])
    out["temb.t"] = npy(t)
    out["temb.out64"] = npy(R_nn.timestep_embedding(t, 64))
    out["temb.out256"] = npy(R_nn.timestep_embedding(t, 256))

    # GroupNorm32
    gn = R_nn.normalization(64)
    randomize_(gn, 21)
    with torch.no_grad():
        gn.weight.copy_(1 + 0.1 * torch.randn(64, generator=g))
        gn.bias.copy_(0.05 * torch.randn(64, generator=g))
    x = (torch.randn(2, 64, 8, 8, generator=g) * 1.5 + 0.3).requires_grad_(True)
    y = gn(x)
    w = torch.randn(y.shape, generator=g)
    (gx,) = torch.autograd.grad((y * w).sum(), x)
    out.update({"gn.x": npy(x), "gn.weight": npy(gn.weight), "gn.bias": npy(gn.bias),
                "gn.y": npy(y), "gn.dy": npy(w), "gn.dx": npy(gx)})

    # ResBlocks
    emb_ch = 128
    variants = {"res_plain": dict(channels=64, out_channels=64),
                "res_skip": dict(channels=96, out_channels=64),
                "res_up": dict(channels=64, out_channels=64, up=True),
                "res_down": dict(channels=64, out_channels=64, down=True)}
    for i, (tag, kw) in enumerate(variants.items()):
        blk = R_unet.ResBlock(emb_channels=emb_ch, dropout=0.0, use_scale_shift_norm=True, **kw)
        randomize_(blk, 100 + i)
        blk.eval()
        x = torch.randn(2, kw["channels"], 8, 8, generator=g).requires_grad_(True)
        emb = torch.randn(2, emb_ch, generator=g)
        y = blk(x, emb)
        w = torch.randn(y.shape, generator=g)
        (gx,) = torch.autograd.grad((y * w).sum(), x)
        out.update({f"{tag}.x": npy(x), f"{tag}.emb": npy(emb), f"{tag}.y": npy(y),
                    f"{tag}.dy": npy(w), f"{tag}.dx": npy(gx)})
        for k, v in blk.state_dict().items():
            out[f"{tag}.sd.{k}"] = npy(v)

    # Attention blocks (always checkpointed in the reference: exercises CheckpointFunction too)
    for tag, new in {"attn_legacy": False, "attn_new": True}.items():
        blk = R_unet.AttentionBlock(64, num_head_channels=16, use_new_attention_order=new)
        randomize_(blk, 200 + int(new))
        x = torch.randn(2, 64, 8, 8, generator=g).requires_grad_(True)
        y = blk(x)
        w = torch.randn(y.shape, generator=g)
        (gx,) = torch.autograd.grad((y * w).sum(), x)
        out.update({f"{tag}.x": npy(x), f"{tag}.y": npy(y), f"{tag}.dy": npy(w), f"{tag}.dx": npy(gx)})
        for k, v in blk.state_dict().items():
            out[f"{tag}.sd.{k}"] = npy(v)
    np.savez_compressed(os.path.join(OUT, "blocks.npz"), **out)


def tiny_model():
    m = R_unet.create_model(**TINY_KW)
    cfg = UNetConfig.from_create_model_kwargs(**TINY_KW)
    sd = seeded_state_dict(cfg, seed=1234)
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.eval(), cfg, sd


def gen_tiny_unet():
    m, cfg, sd = tiny_model()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 32, 32, generator=g).requires_grad_(True)
    t = torch.tensor([3, 250])
    y = m(x, t)
    (gx,) = torch.autograd.grad((y[:, :4] ** 2).sum(), x)
    chk = float(sum(v.double().abs().sum() for v in sd.values()))
    np.savez_compressed(os.path.join(OUT, "tiny_unet.npz"), x=npy(x), t=npy(t), y=npy(y), dx=npy(gx),
                        weight_abs_sum=np.array(chk), n_params=np.array(sum(v.numel() for v in sd.values())))


OPERATORS = {
    "underwater_physical_revised": dict(
        operator=dict(name="underwater_physical_revised", optimizer="sgd", depth_type="gamma", value="1.4,1.4,1",
                      phi_a="1.1,0.95,0.95", phi_a_eta="1e-5", phi_a_learn_flag=True,
                      phi_b="0.95, 0.8, 0.8", phi_b_eta="1e-5", phi_b_learn_flag=True,
                      phi_inf="0.14, 0.29, 0.49", phi_inf_eta="1e-5", phi_inf_learn_flag=True),
        cond=dict(loss_function="norm", loss_weight="depth", weight_function="gamma,1.4,1.4,1",
                  scale="7,7,7,0.9", gradient_x_prev=True, gradient_clip="True,0.005"),
        aux=dict(aux_loss={"avrg_loss": 0.5, "val_loss": 20})),
    "underwater_physical": dict(
        operator=dict(name="underwater_physical", optimizer="sgd", depth_type="original", value="1.4,1.4,1",
                      phi_ab="1.1,0.95,0.95", phi_ab_eta="1e-5", phi_ab_learn_flag=True,
                      phi_inf="0.2,0.4,0.7", phi_inf_eta="1e-5", phi_inf_learn_flag=True),
        cond=dict(loss_function="norm", loss_weight="depth", weight_function="gamma,1.4,1.4,1",
                  scale="4,4,4,1", gradient_x_prev=True, gradient_clip="True,0.001"),
        aux=dict(aux_loss={"val_loss": 40})),
    "haze_physical": dict(
        operator=dict(name="haze_physical", optimizer="sgd", depth_type="gamma", value="1.4,1.4,1",
                      phi_ab="1.0", phi_ab_eta="1e-5", phi_ab_learn_flag=True,
                      phi_inf="0.14, 0.29, 0.49", phi_inf_eta="1e-5", phi_inf_learn_flag=True),
        cond=dict(loss_function="norm", loss_weight="depth", weight_function="gamma,1.4,1.4,1",
                  scale="7,7,7,0.9", gradient_x_prev=True, gradient_clip="True,0.005"),
        aux=dict(aux_loss={"avrg_loss": 0.5, "val_loss": 20})),
}
PATTERN = dict(pattern="pcgs", update_start=0.7, update_end=0, global_N=1, local_M=1, s_start=1, s_end=0,
               n_iter=20, start_guidance=1, stop_guidance=0)


def _loop_trace(m, spec, mean_type="epsilon", var_type="learned_range", perturb=0.0, clip_denoised=False, record_kw=None):
    """10-step guided p_sample_loop of the reference (model m) for one operator spec; every randn_like draw logged."""
    operator = get_operator(device=torch.device("cpu"), batch_size=1, **spec["operator"])
    noiser = get_noise(name="clean")
    cond = get_conditioning_method("osmosis", operator, noiser, **spec["cond"], **PATTERN, **spec["aux"])
    sampler = R_gd.get_sampler("ddpm")(use_timesteps=range(0, 100, 10),
                                       betas=R_gd.get_named_beta_schedule("linear", 1000),
                                       model_mean_type=mean_type, model_var_type=var_type,
                                       dynamic_threshold=False, clip_denoised=clip_denoised,
                                       rescale_timesteps=False)
    x_T = 0.5 * torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(0))
    if perturb:                                            # sensitivity probes (gen_processors)
        x_T = x_T + perturb * torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(77))
    y = torch.rand(1, 3, 32, 32, generator=torch.Generator().manual_seed(7)) * 1.6 - 0.8

    trace = []
    orig_cond = cond.conditioning

    def traced(**kw):
        rec = {"x_in": kw["x_prev"].detach().clone(), "x0": kw["x_0_hat"].detach().clone(),
               "mean": kw["x_t"].detach().clone()}
        ret = orig_cond(**kw)
        rec["x_guided"] = ret[0].detach().clone()
        rec["loss"] = np.array(ret[1], dtype=np.float32)
        rec["phi"] = {k: v.detach().clone() for k, v in ret[2].items()}
        rec["grad"] = ret[3].clone()
        trace.append(rec)
        return ret

    draws = []
    orig_randn_like = torch.randn_like

    def logged_randn_like(t, **kw):
        r = orig_randn_like(t, **kw)
        draws.append(r.clone())
        return r

    torch.manual_seed(0)
    torch.randn_like = logged_randn_like
    try:
        img, variables, loss, x0 = sampler.p_sample_loop(
            model=m, x_start=x_T.clone().requires_grad_(), measurement=y, measurement_cond_fn=traced,
            record=record_kw is not None, save_root=None, pretrain_model="osmosis", rgb_guidance=False,
            sample_pattern=PATTERN, **(record_kw or {}))
    finally:
        torch.randn_like = orig_randn_like
    # draws alternate: randn_like(measurement) [unused], randn_like(img) [used]  
    assert len(draws) == 2 * len(trace)
    used = [d for d in draws if d.shape[1] == 4]
    out = {"x_T": npy(x_T), "y": npy(y), "final_img": npy(img), "final_x0": npy(x0),
           "final_loss": np.array(loss, dtype=np.float32),
           "noise": np.stack([npy(n) for n in used]),
           "timestep_map": np.array(sampler.timestep_map, dtype=np.int64)}
    for k, v in variables.items():
        out[f"final.{k}"] = npy(v)
    for key in ("x_in", "x0", "mean", "x_guided", "grad"):
        out[f"trace.{key}"] = np.stack([npy(r[key]) for r in trace])
    out["trace.loss"] = np.stack([r["loss"] for r in trace])
    for k in trace[0]["phi"]:
        out[f"trace.{k}"] = np.stack([npy(r["phi"][k]) for r in trace])
    return out, loss, variables


def gen_loops():
    m, cfg, sd = tiny_model()
    for opname, spec in OPERATORS.items():
        out, loss, variables = _loop_trace(m, spec)
        np.savez_compressed(os.path.join(OUT, f"loop_{opname}.npz"), **out)
        print(opname, "final loss", loss, {k: npy(v).ravel().round(4) for k, v in variables.items()})


# eta per optimizer: large enough that phi visibly moves in 10 guided steps x 20 inner iterations, small enough to stay physical
OPTIMIZER_ETAS = {"adam": "2e-4", "adamw": "2e-4", "adamax": "2e-4", "rmsprop": "1e-4", "adagrad": "2e-3", "adadelta": "0.5",
                  "asgd": "1e-4", "rprop": "1e-5"}


def gen_optimizers():
    """(round 6) The REAL reference's guided loop with every torch optimizer of utils.get_optimizer that can step phi
    (utils.py:494-524; measurements.py:244-249 builds it with one parameter group per phi, lr = eta): per-step loss and phi,
    final image.  x_T, y and the noise are those of loop_underwater_physical_revised.npz (same seeds, same draw order)."""
    m, cfg, sd = tiny_model()
    base = OPERATORS["underwater_physical_revised"]
    out = {}
    for name, eta in OPTIMIZER_ETAS.items():
        spec = dict(base, operator=dict(base["operator"], optimizer=name, phi_a_eta=eta, phi_b_eta=eta, phi_inf_eta=eta))
        tr, loss, variables = _loop_trace(m, spec)
        tr2, _, _ = _loop_trace(m, spec, perturb=1e-6)       # the reference's own sensitivity: x_T + 1e-6 N(0,1) (adadelta: 50x the others)
        out[f"{name}.phi_drift_1e-6"] = np.array(max(float(np.abs(tr2[f"trace.{k}"] - tr[f"trace.{k}"]).max()) for k in ("phi_a", "phi_b", "phi_inf")))
        out[f"{name}.img_drift_1e-6"] = np.array(float(np.abs(tr2["final_img"] - tr["final_img"]).max()))
        if "x_T" not in out:
            out.update({"x_T": tr["x_T"], "y": tr["y"], "noise": tr["noise"]})
        else:
            assert np.array_equal(out["noise"], tr["noise"]) and np.array_equal(out["x_T"], tr["x_T"])
        out[f"{name}.eta"] = np.array(float(eta))
        out[f"{name}.final_img"] = tr["final_img"]
        out[f"{name}.final_x0"] = tr["final_x0"]
        out[f"{name}.loss"] = tr["trace.loss"]
        for k in ("phi_a", "phi_b", "phi_inf"):
            out[f"{name}.{k}"] = tr[f"trace.{k}"]
        print(name, "final loss", loss, {k: npy(v).ravel().round(4) for k, v in variables.items()}, "phi drift", float(out[f"{name}.phi_drift_1e-6"]))
    np.savez_compressed(os.path.join(OUT, "loop_optimizers.npz"), **out)


def gen_fp16():
    """The reference in fp16 -- `create_model(use_fp16=True)` PLUS the `convert_to_fp16()` call its driver forgets (SURVEY F3;
    unet.py:697-703: input / middle / output blocks to half, time_embed and out stay fp32; GroupNorm32 computes in fp32,
    attention soft-maxes in fp32) -- on CPU: tiny seeded UNet forward + input gradient, and the 10-step guided loop of the
    revised underwater operator.  Next to each: the fp32 reference on the same inputs, so that a test can say "the HIP fp16
    family is as close to the reference's fp16 as the reference's fp16 is to its fp32"."""
    m32, cfg, sd = tiny_model()
    m16 = R_unet.create_model(**dict(TINY_KW, use_fp16=True))
    m16.load_state_dict(sd, strict=True)
    m16.eval()
    m16.convert_to_fp16()
    assert m16.input_blocks[0][0].weight.dtype == torch.float16 and m16.out[2].weight.dtype == torch.float32
    g = torch.Generator().manual_seed(0)
    x = 0.7 * torch.randn(2, 4, 32, 32, generator=g)
    t = torch.tensor([37.0, 5.0])
    w = torch.randn(2, 8, 32, 32, generator=g)
    out = {"x": npy(x), "t": npy(t), "w": npy(w)}
    for tag, m in (("16", m16), ("32", m32)):
        xr = x.clone().requires_grad_(True)
        y = m(xr, t)
        assert y.dtype == torch.float32
        (dx,) = torch.autograd.grad((y * w).sum(), xr)
        out["y" + tag], out["dx" + tag] = npy(y), npy(dx)
    print("tiny UNet: reference fp16 vs reference fp32: y", float(np.abs(out["y16"] - out["y32"]).max()), "of", float(np.abs(out["y32"]).max()),
          " dx", float(np.abs(out["dx16"] - out["dx32"]).max()), "of", float(np.abs(out["dx32"]).max()))
    # a second, wider model whose attention blocks have 64-wide heads at T = 1024 and 256 (the shapes of the 552.8 M-parameter net's
    # attention; the tiny model's 16-wide heads take another kernel in the HIP build): 1 x 4 x 64 x 64
    MID_KW = dict(TINY_KW, num_channels=64, num_head_channels=64)
    cfgm = UNetConfig.from_create_model_kwargs(**MID_KW)
    sdm = seeded_state_dict(cfgm, seed=77)
    xm = 0.7 * torch.randn(1, 4, 64, 64, generator=g)
    tm = torch.tensor([300.0])
    wm = torch.randn(1, 8, 64, 64, generator=g)
    out.update({"mid.x": npy(xm), "mid.t": npy(tm), "mid.w": npy(wm)})
    for tag, half in (("16", True), ("32", False)):
        mm = R_unet.create_model(**dict(MID_KW, use_fp16=half))
        mm.load_state_dict(sdm, strict=True)
        mm.eval()
        if half:
            mm.convert_to_fp16()
        xr = xm.clone().requires_grad_(True)
        y = mm(xr, tm)
        (dx,) = torch.autograd.grad((y * wm).sum(), xr)
        out["mid.y" + tag], out["mid.dx" + tag] = npy(y), npy(dx)
    print("mid UNet (64-wide heads): reference fp16 vs fp32: y", float(np.abs(out["mid.y16"] - out["mid.y32"]).max()), "of",
          float(np.abs(out["mid.y32"]).max()), " dx", float(np.abs(out["mid.dx16"] - out["mid.dx32"]).max()), "of",
          float(np.abs(out["mid.dx32"]).max()))
    tr, loss, variables = _loop_trace(m16, OPERATORS["underwater_physical_revised"])
    for k, v in tr.items():
        out["loop." + k] = v
    print("fp16 loop final loss", loss, {k: npy(v).ravel().round(4) for k, v in variables.items()})
    np.savez_compressed(os.path.join(OUT, "fp16_reference.npz"), **out)


FULL_KW = dict(image_size=256, num_channels=256, num_res_blocks=2, channel_mult="", learn_sigma=True, class_cond=False,
               use_checkpoint=False, attention_resolutions="32, 16, 8", num_heads=4, num_head_channels=64, num_heads_upsample=-1,
               use_scale_shift_norm=True, dropout=0.0, resblock_updown=True, use_fp16=False, use_new_attention_order=False,
               model_path="", pretrain_model="osmosis")


def gen_full_unet():
    """The REAL architecture (osmosis_sample_config.yaml: 552.8 M parameters, channel_mult (1,1,2,2,4,4) chosen by create_model for
    image_size 256, attention at 32 / 16 / 8 with 64-wide heads) through the real reference on CPU, seeded weights
    (oracle.unet_ref.seeded_state_dict(cfg, 1234)), x = 0.7 randn(seed 0) at 1 x 4 x 256 x 256, two timesteps.  The fixture keeps
    every 4th pixel of y and of d(sum(y w))/dx plus their norms: small, and enough to pin a full-size forward / backward (one wrong
    layer changes every pixel)."""
    cfg = UNetConfig.from_create_model_kwargs(**FULL_KW)
    sd = seeded_state_dict(cfg, seed=1234)
    m = R_unet.create_model(**FULL_KW)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    m.eval()
    assert sum(p.numel() for p in m.parameters()) == 552_821_000
    g = torch.Generator().manual_seed(0)
    x = 0.7 * torch.randn(1, 4, 256, 256, generator=g)
    w = torch.randn(1, 8, 256, 256, generator=g)
    out = {"seed": np.array(0), "x_scale": np.array(0.7), "stride": np.array(4), "x_abs_sum": np.array(float(x.double().abs().sum()))}
    for t in (37.0, 999.0):
        xr = x.clone().requires_grad_(True)
        y = m(xr, torch.tensor([t]))
        (dx,) = torch.autograd.grad((y * w).sum(), xr)
        tag = f"t{int(t)}"
        out[tag + ".y_sub"] = npy(y)[:, :, ::4, ::4]
        out[tag + ".dx_sub"] = npy(dx)[:, :, ::4, ::4]
        out[tag + ".y_l2"] = np.array(float(y.double().pow(2).sum().sqrt()))
        out[tag + ".dx_l2"] = np.array(float(dx.double().pow(2).sum().sqrt()))
        out[tag + ".y_max"] = np.array(float(y.abs().max()))
        out[tag + ".dx_max"] = np.array(float(dx.abs().max()))
        print(tag, "y max", out[tag + ".y_max"], "dx max", out[tag + ".dx_max"])
    np.savez_compressed(os.path.join(OUT, "full_unet.npz"), **out)


def gen_full_step():
    """TWO guided steps of the real reference's p_sample_loop with the FULL 552.8 M-parameter network at 256 x 256: the sampler is
    built on use_timesteps = {0, 299}, so the first step runs the network at t = 299 (where bench.py's timed window starts; phi
    is updated: 20 inner iterations) and the second at t = 0; revised underwater operator, x_T = 0.5 randn(seed 0) (bounded, so
    that the seeded network's pred_xstart stays inside the physical model's range), y ~ U(-0.8, 0.8) (seed 7).  The noise is NOT
    stored: it is the reference's own torch.randn_like draws after torch.manual_seed(0) (measurement-shaped, then image-shaped,
    per step), which a test reproduces with the same calls.  Kept: every 4th pixel of the per-step tensors, losses, phi."""
    cfg = UNetConfig.from_create_model_kwargs(**FULL_KW)
    sd = seeded_state_dict(cfg, seed=1234)
    m = R_unet.create_model(**FULL_KW)
    m.load_state_dict(sd, strict=True)
    m.eval()
    # the headline operator at stride 4; the operators of BASELINE configs 3 and 5 at stride 8 (smaller fixtures)
    for opname, st, fname in (("underwater_physical_revised", 4, "full_step.npz"), ("underwater_physical", 8, "full_step_underwater_physical.npz"),
                              ("haze_physical", 8, "full_step_haze_physical.npz")):
        _gen_full_step_one(m, opname, st, fname)


def _gen_full_step_one(m, opname, st, fname):
    spec = OPERATORS[opname]
    operator = get_operator(device=torch.device("cpu"), batch_size=1, **spec["operator"])
    cond = get_conditioning_method("osmosis", operator, get_noise(name="clean"), **spec["cond"], **PATTERN, **spec["aux"])
    sampler = R_gd.get_sampler("ddpm")(use_timesteps=(0, 299), betas=R_gd.get_named_beta_schedule("linear", 1000),
                                       model_mean_type="epsilon", model_var_type="learned_range", dynamic_threshold=False,
                                       clip_denoised=False, rescale_timesteps=False)
    assert list(sampler.timestep_map) == [0, 299]
    x_T = 0.5 * torch.randn(1, 4, 256, 256, generator=torch.Generator().manual_seed(0))
    y = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(7)) * 1.6 - 0.8
    trace = []
    orig_cond = cond.conditioning

    def traced(**kw):
        rec = {"x_in": kw["x_prev"].detach().clone(), "x0": kw["x_0_hat"].detach().clone(), "mean": kw["x_t"].detach().clone()}
        ret = orig_cond(**kw)
        rec["x_guided"] = ret[0].detach().clone()
        rec["loss"] = np.array(ret[1], dtype=np.float32)
        rec["phi"] = {k: v.detach().clone() for k, v in ret[2].items()}
        rec["grad"] = ret[3].clone()
        trace.append(rec)
        return ret

    torch.manual_seed(0)
    img, variables, loss, x0 = sampler.p_sample_loop(
        model=m, x_start=x_T.clone().requires_grad_(), measurement=y, measurement_cond_fn=traced, record=False, save_root=None,
        pretrain_model="osmosis", rgb_guidance=False, sample_pattern=PATTERN)
    assert len(trace) == 2
    out = {"stride": np.array(st), "final_img_sub": npy(img)[:, :, ::st, ::st], "final_x0_sub": npy(x0)[:, :, ::st, ::st],
           "final_loss": np.array(loss, dtype=np.float32), "timestep_map": np.array(sampler.timestep_map, dtype=np.int64),
           "x_T_abs_sum": np.array(float(x_T.double().abs().sum())), "y_abs_sum": np.array(float(y.double().abs().sum()))}
    for k, v in variables.items():
        out[f"final.{k}"] = npy(v)
    for key in ("x_in", "x0", "mean", "x_guided", "grad"):
        out[f"trace.{key}_sub"] = np.stack([npy(r[key])[:, :, ::st, ::st] for r in trace])
        out[f"trace.{key}_max"] = np.array([float(r[key].abs().max()) for r in trace])
    out["trace.loss"] = np.stack([r["loss"] for r in trace])
    for k in trace[0]["phi"]:
        out[f"trace.{k}"] = np.stack([npy(r["phi"][k]) for r in trace])
    print(opname, "full-size guided steps: losses", out["trace.loss"].ravel(), "max |x0|", out["trace.x0_max"], "max |grad|", out["trace.grad_max"],
          {k: npy(v).ravel().round(4) for k, v in variables.items()})
    assert np.isfinite(out["final_img_sub"]).all() and np.isfinite(out["trace.loss"]).all()
    np.savez_compressed(os.path.join(OUT, fname), **out)


def gen_prior():
    """Unconditional RGBD-prior sampler (osmosis_utils/diffusion.py:59-130): last 6 steps of the
    1000-step chain (t = 6..1) on the tiny seeded UNet.  The reference only defines its return values
    when it records, so recording is on (record_every = 2: snapshots at t = 6, 4, 2, 1); its two torchvision calls (make_grid,
    to_pil_image: absent from this image) are served by the 0.14.1 restatements above, and the process grid it "saves"
    (`image_<idx>_process.png`, :124-128) is captured as a uint8 array."""
    import tempfile
    from osmosis_utils import diffusion as R_diff
    m, cfg, sd = tiny_model()
    saved = {}

    class _Img:
        def __init__(self, arr):
            self.arr = arr

        def save(self, path, *a, **k):
            saved[os.path.basename(path)] = self.arr
    # torchvision is absent: its two calls are served by the restatements above (as gen_outputs does)
    R_diff.make_grid = lambda lst, nrow=8, pad_value=0.0, **k: tv_make_grid(lst, nrow=nrow, pad_value=pad_value)
    R_diff.tvtf.to_pil_image = lambda pic, *a, **k: _Img(to_pil_u8(pic))
    diff = R_diff.GaussianDiffusion(T=1000, schedule="linear")
    x_T = 0.3 * torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(3))
    draws, xs = [], []
    orig = torch.randn_like

    def logged(t, **kw):
        r = orig(t, **kw)
        draws.append(r.clone())
        return r

    def net(x, t):
        xs.append(x.detach().clone())
        return m(x, t)

    torch.manual_seed(0)
    torch.randn_like = logged
    try:
        x, (rgb, depth_color) = diff.inverse(net=net, shape=(4, 32, 32), image_channels=4, steps=6, x=x_T.clone(),
                                             start_t=6, device="cpu", record_process=True, record_every=2,
                                             save_path=tempfile.mkdtemp(), image_idx=7)
    finally:
        torch.randn_like = orig
    assert len(draws) == 5 and len(xs) == 6
    np.savez_compressed(os.path.join(OUT, "prior_inverse.npz"), x_T=npy(x_T), noise=np.stack([npy(d) for d in draws]),
                        x_steps=np.stack([npy(v) for v in xs]), x_final=npy(x), x_start_rgb=npy(rgb),
                        x_depth_color=npy(depth_color), process_png=saved["image_7_process.png"],
                        beta=diff.beta, alphabar=diff.alphabar,
                        cosine_beta=R_diff.GaussianDiffusion(T=50, schedule="cosine").beta)


def _ps_chain(m, name, mean_type="epsilon", var_type="learned_range", perturb=0.0, x_ins=None, clip_denoised=False,
              scale="0.6,0.5,0.4,0.0", sigma=0.05):
    """One rgb-guidance chain of the reference: sampler `name` ('ddpm' | 'ddim'), the given processors; every randn_like logged.
    perturb: amplitude of a seeded N(0,1) perturbation of x_T (sensitivity probes); x_ins: list receiving every step's input."""
    operator = get_operator(name="rgb_guidance", device=torch.device("cpu"), batch_size=1)
    noiser = get_noise(name="gaussian", sigma=sigma)
    cond = get_conditioning_method("ps", operator, noiser, scale=scale)
    sampler = R_gd.get_sampler(name)(use_timesteps=range(0, 100, 10),
                                     betas=R_gd.get_named_beta_schedule("linear", 1000),
                                     model_mean_type=mean_type, model_var_type=var_type,
                                     dynamic_threshold=False, clip_denoised=clip_denoised, rescale_timesteps=False)
    x_T = 0.5 * torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(2))
    if perturb:
        x_T = x_T + perturb * torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(77))
    y = torch.rand(1, 3, 32, 32, generator=torch.Generator().manual_seed(9)) * 1.6 - 0.8
    draws, losses = [], []
    orig = torch.randn_like

    def logged(t, **kw):
        r = orig(t, **kw)
        draws.append(r.clone())
        return r

    orig_cond = cond.conditioning

    def traced(**kw):
        if x_ins is not None:
            x_ins.append(kw["x_prev"].detach().clone())
        ret = orig_cond(**kw)
        losses.append(float(ret[1]))
        return ret

    torch.manual_seed(0)
    torch.randn_like = logged
    try:
        with np.errstate(divide="ignore"):                # fixed_small takes log(0) at index 0 (unused there)
            img = sampler.p_sample_loop(model=m, x_start=x_T.clone().requires_grad_(), measurement=y,
                                        measurement_cond_fn=traced, record=False, save_root=None,
                                        pretrain_model="osmosis", rgb_guidance=True, sample_pattern=PATTERN)
    finally:
        torch.randn_like = orig
    return x_T, y, img, losses, draws


def gen_ps():
    """rgb-guidance path: `ps` conditioning (condition_methods.py:234-251) + `rgb_guidance` operator
    (measurements.py:80-96) + gaussian noiser, through DDPM.p_sample / DDIM.p_sample (gaussian_diffusion.py:494-535),
    10 low-t steps on the tiny seeded UNet.  All randn_like draws are logged in call order (p_sample's noise, then
    q_sample's) so that a replay reproduces the chain exactly."""
    m, cfg, sd = tiny_model()
    out = {}
    for name in ("ddpm", "ddim"):
        x_T, y, img, losses, draws = _ps_chain(m, name)
        out[f"{name}.x_T"], out[f"{name}.y"], out[f"{name}.final_img"] = npy(x_T), npy(y), npy(img)
        out[f"{name}.loss"] = np.array(losses, dtype=np.float32)
        out[f"{name}.draw_is_x"] = np.array([d.shape[1] == 4 for d in draws])
        out[f"{name}.draws_x"] = np.stack([npy(d) for d in draws if d.shape[1] == 4])
        out[f"{name}.draws_y"] = np.stack([npy(d) for d in draws if d.shape[1] == 3])
        print(name, "draws", len(draws), "final loss", losses[-1])
    np.savez_compressed(os.path.join(OUT, "loop_ps.npz"), **out)


# (mean processor, variance processor) pairs that together cover every registered class other than the shipped pair
PROCESSOR_PAIRS = [("start_x", "fixed_small"), ("epsilon", "fixed_large"), ("epsilon", "learned"), ("start_x", "learned")]
PS_PROCESSOR_CHAINS = [("ddpm", "start_x", "fixed_large"), ("ddpm", "epsilon", "fixed_small"), ("ddpm", "start_x", "learned"),
                       ("ddim", "previous_x", "fixed_small"), ("ddim", "start_x", "learned_range")]


def gen_processors():
    """(round 6) The REAL reference's chains with the mean / variance processors no shipped config names
    (posterior_mean_variance.py:53-101 previous_x / start_x, :171-222 fixed_small / fixed_large / learned):
    the 10-step guided Osmosis loop (revised underwater operator; x_T, y, noise = those of
    loop_underwater_physical_revised.npz: same seeds, same draw order) and the rgb-guidance chains through DDPM.p_sample /
    DDIM.p_sample (x_T, y, draws = those of loop_ps.npz).  Per chain: per-step losses and pred_xstart of the first / last step,
    final image, final pred_xstart, final phi; `<tag>.drift_1e-6` = how far the REFERENCE's own final image moves when x_T is
    perturbed by 1e-6 N(0,1) (start_x feeds the network its own output, previous_x divides by posterior_mean_coef1: both amplify);
    chains whose drift exceeds the 1e-3 bar also carry every step's input (`<tag>.x_in`) for teacher-forced comparison.
    `previous_x` returns the network's split output AS the mean (:68-72), and both the Osmosis branch (gaussian_diffusion.py:268
    `img += ...`, condition_methods.py:223 `x_t -= ...`) and DDPM.p_sample (:499 `sample += ...`) then modify that view in place:
    autograd raises RuntimeError there.  Only DDIM.p_sample (a fresh tensor, :524-530) runs with it; the generator asserts the
    raises and records `previous_x.raises` so that the tests can hold the build to the same behaviour."""
    m, cfg, sd = tiny_model()
    base = dict(np.load(os.path.join(OUT, "loop_underwater_physical_revised.npz")))
    ps = dict(np.load(os.path.join(OUT, "loop_ps.npz")))
    out = {}
    for mt, vt in PROCESSOR_PAIRS:
        with np.errstate(divide="ignore"):
            tr, loss, variables = _loop_trace(m, OPERATORS["underwater_physical_revised"], mt, vt)
        assert np.array_equal(tr["noise"], base["noise"]) and np.array_equal(tr["x_T"], base["x_T"]) and np.array_equal(tr["y"], base["y"])
        tag = f"osmosis.{mt}.{vt}"
        with np.errstate(divide="ignore"):
            tr2, _, _ = _loop_trace(m, OPERATORS["underwater_physical_revised"], mt, vt, perturb=1e-6)
        out[f"{tag}.drift_1e-6"] = np.array(np.abs(tr2["final_img"] - tr["final_img"]).max())
        if out[f"{tag}.drift_1e-6"] > 1e-3:
            out[f"{tag}.x_in"] = tr["trace.x_in"]
        out[f"{tag}.final_img"], out[f"{tag}.final_x0"], out[f"{tag}.loss"] = tr["final_img"], tr["final_x0"], tr["trace.loss"]
        out[f"{tag}.x0_first"], out[f"{tag}.mean_first"] = tr["trace.x0"][0], tr["trace.mean"][0]
        for k in ("phi_a", "phi_b", "phi_inf"):
            out[f"{tag}.{k}"] = tr[f"final.{k}"]
        print(tag, "final loss", loss, "max |img|", float(np.abs(tr["final_img"]).max()), "drift", float(out[f"{tag}.drift_1e-6"]))
    raised = []
    for what, run in (("osmosis", lambda: _loop_trace(m, OPERATORS["underwater_physical_revised"], "previous_x", "learned_range")),
                      ("ps.ddpm", lambda: _ps_chain(m, "ddpm", "previous_x", "learned_range"))):
        try:
            run()
        except RuntimeError as e:
            assert "modified inplace" in str(e), e
            raised.append(what)
    assert raised == ["osmosis", "ps.ddpm"], raised
    out["previous_x.raises"] = np.array(raised)
    for name, mt, vt in PS_PROCESSOR_CHAINS:
        x_ins = []
        x_T, y, img, losses, draws = _ps_chain(m, name, mt, vt, x_ins=x_ins)
        img2 = _ps_chain(m, name, mt, vt, perturb=1e-6)[2]
        assert np.array_equal(npy(x_T), ps[f"{name}.x_T"]) and np.array_equal(npy(y), ps[f"{name}.y"])
        assert np.array_equal(np.stack([npy(d) for d in draws if d.shape[1] == 4]), ps[f"{name}.draws_x"])
        tag = f"ps.{name}.{mt}.{vt}"
        out[f"{tag}.final_img"], out[f"{tag}.loss"] = npy(img), np.array(losses, dtype=np.float32)
        out[f"{tag}.drift_1e-6"] = np.array(float((img2 - img).abs().max()))
        if out[f"{tag}.drift_1e-6"] > 1e-3:
            out[f"{tag}.x_in"] = np.stack([npy(x) for x in x_ins])
        print(tag, "final loss", losses[-1], "max |img|", float(img.abs().max()), "drift", float(out[f"{tag}.drift_1e-6"]))
    np.savez_compressed(os.path.join(OUT, "loop_processors.npz"), **out)


def gen_clip():
    """(round 6) `clip_denoised: True` -- what configs/rgb_guidance_sample_config.yaml ships (ddpm, ps, scale 3,3,3,0.1, gaussian noiser
    with sigma 0) -- through the REAL reference: the rgb-guidance chains (DDPM.p_sample with the shipped conditioning values, and
    DDIM.p_sample) and the Osmosis loop, 10 low-t steps on the tiny seeded UNet (x_T, y, draws of loop_ps.npz / noise of
    loop_underwater_physical_revised.npz), with the fraction of clamped pred_xstart elements per chain; plus `process_xstart` itself
    (posterior_mean_variance.py:43-50) with dynamic_threshold (util/img_utils.py:8-15) and both switches on a seeded tensor."""
    from guided_diffusion.posterior_mean_variance import get_mean_processor
    m, cfg, sd = tiny_model()
    base = dict(np.load(os.path.join(OUT, "loop_underwater_physical_revised.npz")))
    ps = dict(np.load(os.path.join(OUT, "loop_ps.npz")))
    out = {}
    for name, kw in (("ddpm", dict(scale="3,3,3,0.1", sigma=0)), ("ddim", dict())):
        x_T, y, img, losses, draws = _ps_chain(m, name, clip_denoised=True, **kw)
        img2 = _ps_chain(m, name, clip_denoised=True, perturb=1e-6, **kw)[2]
        assert np.array_equal(npy(x_T), ps[f"{name}.x_T"]) and np.array_equal(npy(y), ps[f"{name}.y"])
        assert np.array_equal(np.stack([npy(d) for d in draws if d.shape[1] == 4]), ps[f"{name}.draws_x"])
        tag = f"ps.{name}"
        out[f"{tag}.final_img"], out[f"{tag}.loss"] = npy(img), np.array(losses, dtype=np.float32)
        out[f"{tag}.drift_1e-6"] = np.array(float((img2 - img).abs().max()))
        print(tag, "final loss", losses[-1], "max |img|", float(img.abs().max()), "drift", float(out[f"{tag}.drift_1e-6"]))
    tr, loss, variables = _loop_trace(m, OPERATORS["underwater_physical_revised"], clip_denoised=True)
    tr2, _, _ = _loop_trace(m, OPERATORS["underwater_physical_revised"], clip_denoised=True, perturb=1e-6)
    assert np.array_equal(tr["noise"], base["noise"]) and np.array_equal(tr["x_T"], base["x_T"])
    out["osmosis.final_img"], out["osmosis.final_x0"], out["osmosis.loss"] = tr["final_img"], tr["final_x0"], tr["trace.loss"]
    out["osmosis.x0"], out["osmosis.grad"] = tr["trace.x0"], tr["trace.grad"]
    out["osmosis.drift_1e-6"] = np.array(np.abs(tr2["final_img"] - tr["final_img"]).max())
    out["osmosis.clamped_fraction"] = np.array(float((np.abs(tr["trace.x0"]) == 1.0).mean()))
    for k in ("phi_a", "phi_b", "phi_inf"):
        out[f"osmosis.{k}"] = tr[f"final.{k}"]
    print("osmosis clip: final loss", loss, "clamped fraction", float(out["osmosis.clamped_fraction"]), "drift", float(out["osmosis.drift_1e-6"]))
    # process_xstart on a seeded tensor, through the mean processor's own method
    betas = R_gd.get_named_beta_schedule("linear", 1000)
    x = 1.3 * torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(21))
    out["px.x"] = npy(x)
    for dyn, clip in ((True, False), (True, True), (False, True)):
        proc = get_mean_processor("epsilon", betas=betas, dynamic_threshold=dyn, clip_denoised=clip)
        out[f"px.dyn{int(dyn)}.clip{int(clip)}"] = npy(proc.process_xstart(x.clone()))
    np.savez_compressed(os.path.join(OUT, "loop_clip.npz"), **out)


def gen_record():
    """(round 6) `record=True` of p_sample_loop (gaussian_diffusion.py:308-333): the 10-step guided loop of
    loop_underwater_physical_revised.npz again with record_every = 3 -- snapshots of pred_xstart at idx 9, 6, 3, 0 -- and the
    `<name>_process.png` grid the reference hands to to_pil_image, captured as a uint8 array (torchvision's make_grid / to_pil_image:
    the 0.14.1 restatements above)."""
    import tempfile
    m, cfg, sd = tiny_model()
    base = dict(np.load(os.path.join(OUT, "loop_underwater_physical_revised.npz")))
    saved = {}

    class _Img:
        def __init__(self, arr):
            self.arr = arr

        def save(self, path, *a, **k):
            saved[os.path.basename(path)] = self.arr
    keep = (R_gd.make_grid, R_gd.tvtf.to_pil_image)
    R_gd.make_grid = lambda lst, nrow=8, pad_value=0.0, **k: tv_make_grid(lst, nrow=nrow, pad_value=pad_value)
    R_gd.tvtf.to_pil_image = lambda pic, *a, **k: _Img(to_pil_u8(pic))
    try:
        tr, loss, variables = _loop_trace(m, OPERATORS["underwater_physical_revised"],
                                          record_kw=dict(record_every=3, save_grids_path=tempfile.mkdtemp(), original_file_name="frame"))
    finally:
        R_gd.make_grid, R_gd.tvtf.to_pil_image = keep
    assert np.array_equal(tr["final_img"], base["final_img"])          # recording does not touch the chain
    np.savez_compressed(os.path.join(OUT, "loop_record.npz"), process_png=saved["frame_process.png"])
    print("process grid", saved["frame_process.png"].shape)


def gen_postprocess():
    """Output post-processing helpers of osmosis_utils/utils.py (min_max_norm_range :46-74,
    min_max_norm_range_percentile :77-114, depth_tensor_to_color_image :748-763, convert_depth :544-566)
    on seeded depth maps: what osmosis_sampling.py:208-223 applies to the final pred_xstart."""
    from osmosis_utils import utils as R_u
    g = torch.Generator().manual_seed(21)
    d3 = torch.randn(1, 24, 20, generator=g) * 0.7 + 0.1          # [1,H,W] depth channel
    d4 = torch.randn(1, 3, 12, 10, generator=g)                   # [B=1,C,H,W]
    const = torch.full((1, 6, 5), 0.25)
    out = {"d3": npy(d3), "d4": npy(d4), "const": npy(const),
           "mm_d3": npy(R_u.min_max_norm_range(d3)),
           "mm_d3_range": npy(R_u.min_max_norm_range(d3, vmin=-1, vmax=3)),
           "mm_d3_u8": npy(R_u.min_max_norm_range(d3, is_uint8=True)),
           "mm_d4": npy(R_u.min_max_norm_range(d4)),
           "mm_const": npy(R_u.min_max_norm_range(const)),
           "pmm_d3": npy(R_u.min_max_norm_range_percentile(d3, vmin=0, vmax=1, percent_low=0.03, percent_high=0.99,
                                                           is_uint8=False)),
           "pmm_d3_u8": npy(R_u.min_max_norm_range_percentile(d3, percent_low=0.1, percent_high=0.9, is_uint8=True)),
           "pmm_const": npy(R_u.min_max_norm_range_percentile(const, percent_low=0.03, percent_high=0.99))}
    pmm = R_u.min_max_norm_range_percentile(d3, vmin=0, vmax=1, percent_low=0.03, percent_high=0.99, is_uint8=False)
    out["color_pmm_d3"] = npy(R_u.depth_tensor_to_color_image(pmm))
    rep = d3.repeat(3, 1, 1)
    out["cd_gamma"] = npy(R_u.convert_depth(rep, depth_type="gamma", value="1.4,1.4,1"))
    out["cd_original"] = npy(R_u.convert_depth(rep, depth_type="original", value="1.4,1.4,1"))
    out["cd_move"] = npy(R_u.convert_depth(rep, depth_type="move", value=2.0))
    np.savez_compressed(os.path.join(OUT, "postprocess.npz"), **out)


VARIANTS = {
    # (a) the guided-diffusion defaults the Osmosis configs switch off: Upsample / Downsample layers with convolutions instead of
    #     up / down ResBlocks, additive instead of scale-shift conditioning
    "conv_updown_additive": dict(TINY_KW, resblock_updown=False, use_scale_shift_norm=False),
    # (b) conv resampling with scale-shift conditioning, class-conditional, a non-zero dropout rate (inference: identity)
    "conv_updown_classcond": dict(TINY_KW, resblock_updown=False, class_cond=True, dropout=0.1),
    # (c) additive conditioning with up / down ResBlocks
    "resblock_updown_additive": dict(TINY_KW, use_scale_shift_norm=False),
}


def gen_unet_variants():
    """UNet variants NO shipped Osmosis config uses (VERDICT r04 "missing" 5): the real reference's create_model with
    resblock_updown=False (Upsample / Downsample with 3x3 convolutions, unet.py:160-219), use_scale_shift_norm=False (:329-332),
    class_cond=True (label_emb, :556-557, 729-731) and dropout > 0 in eval mode, tiny sizes, oracle.unet_ref.seeded_state_dict weights
    (loaded strictly: the oracle's key / shape list is checked against the reference module on the way): y and d(sum(y w))/dx."""
    out = {}
    for tag, kw in VARIANTS.items():
        cfg = UNetConfig.from_create_model_kwargs(**kw)
        sd = seeded_state_dict(cfg, seed=4321)
        m = R_unet.create_model(**kw)
        res = m.load_state_dict(sd, strict=True)
        assert not res.missing_keys and not res.unexpected_keys
        m.eval()
        g = torch.Generator().manual_seed(5)
        x = (0.8 * torch.randn(2, 4, 32, 32, generator=g)).requires_grad_(True)
        t = torch.tensor([3.0, 640.0])
        w = torch.randn(2, 8, 32, 32, generator=g)
        ykw = {}
        if kw.get("class_cond"):
            ykw["y"] = torch.tensor([7, 993])
            out[f"{tag}.labels"] = npy(ykw["y"])
        y = m(x, t, **ykw)
        (dx,) = torch.autograd.grad((y * w).sum(), x)
        out.update({f"{tag}.x": npy(x), f"{tag}.t": npy(t), f"{tag}.w": npy(w), f"{tag}.y": npy(y), f"{tag}.dx": npy(dx),
                    f"{tag}.n_params": np.array(sum(p.numel() for p in m.parameters()))})
        print(tag, "params", int(out[f"{tag}.n_params"]), "y max", float(y.abs().max()), "dx max", float(dx.abs().max()))
    np.savez_compressed(os.path.join(OUT, "unet_variants.npz"), **out)


def gen_outputs():
    """The five per-image files of osmosis_sampling.py:319-353 as uint8 arrays, for a seeded final pred_xstart / phi / input of
    a revised-underwater run: the tensors are formed with the reference's OWN helpers (osmosis_utils/utils.py) in the order the
    driver applies them (:207-223), the grid goes through the reference's clip_image (:347-349).  torchvision is absent from
    this image, so its two layout / conversion steps are restated here from its 0.14.1 behaviour: to_pil_image on a float
    tensor = pic.mul(255).byte() (truncation); make_grid(list, nrow=3, pad_value=1.) = stack, 2-pixel padding, row-major tiles.
    A second case carries the ground-truth row of the simulation config (:341-344)."""
    from osmosis_utils import utils as R_u

    g = torch.Generator().manual_seed(77)
    H, W = 20, 28
    out_xstart = torch.randn(1, 4, H, W, generator=g) * 0.8        # leaves [-1, 1]: the clips matter
    ref_img = torch.rand(1, 3, H, W, generator=g) * 2 - 1
    gt_rgb = torch.rand(3, H, W, generator=g)
    gt_depth = torch.rand(1, H, W, generator=g)
    ref_img_01 = 0.5 * (ref_img[0] + 1)
    sample_rgb = out_xstart[0, 0:-1, :, :]
    sample_depth_tmp = out_xstart[0, -1, :, :].unsqueeze(0)
    sample_rgb_01 = 0.5 * (sample_rgb + 1)
    sample_rgb_01_clip = torch.clamp(sample_rgb_01, min=0, max=1)
    sample_depth_mm = R_u.min_max_norm_range(sample_depth_tmp[0].unsqueeze(0))
    pmm = R_u.min_max_norm_range_percentile(sample_depth_tmp, vmin=0, vmax=1, percent_low=0.03, percent_high=0.99, is_uint8=False)
    color = R_u.depth_tensor_to_color_image(pmm)
    out = {"out_xstart": npy(out_xstart), "ref_img": npy(ref_img), "gt_rgb_01": npy(gt_rgb), "gt_depth_01": npy(gt_depth),
           "input": to_pil_u8(ref_img_01), "rgb": to_pil_u8(sample_rgb_01_clip), "depth_color": to_pil_u8(color),
           "depth_raw": to_pil_u8(sample_depth_mm)}
    grid = tv_make_grid([ref_img_01, sample_rgb_01_clip, color], nrow=3, pad_value=1.)
    out["grid"] = R_u.clip_image(grid, scale=False, move=False, is_uint8=True).permute(1, 2, 0).numpy()
    gt_color = R_u.depth_tensor_to_color_image(gt_depth)
    grid6 = tv_make_grid([ref_img_01, sample_rgb_01_clip, color, torch.zeros_like(sample_rgb_01), gt_rgb, gt_color], nrow=3, pad_value=1.)
    out["grid_gt"] = R_u.clip_image(grid6, scale=False, move=False, is_uint8=True).permute(1, 2, 0).numpy()
    np.savez_compressed(os.path.join(OUT, "outputs.npz"), **out)


def gen_configs():
    """The reference's shipped YAML configurations parsed by ITS loader (osmosis_utils/utils.py:357-360, yaml.FullLoader), as JSON:
    pins tests/baseline_configs.py (hand transcriptions) and sampling.load_config (values only; no YAML text is stored)."""
    import json
    from osmosis_utils import utils as R_u
    cfgs = {}
    for f in sorted(os.listdir("/root/reference/configs")):
        if f.endswith(".yaml"):
            cfgs[f] = R_u.load_yaml(os.path.join("/root/reference/configs", f))
    with open(os.path.join(OUT, "configs.json"), "w") as fh:
        json.dump(cfgs, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1:                 # regenerate selected sections only: gen_golden.py postprocess prior ...
        for name in sys.argv[1:]:
            globals()["gen_" + name]()
        sys.exit(0)
    gen_schedules()
    gen_spacing()
    gen_blocks()
    gen_tiny_unet()
    gen_unet_variants()
    gen_loops()
    gen_prior()
    gen_postprocess()
    gen_outputs()
    gen_configs()
    gen_ps()
    gen_optimizers()
    gen_processors()
    gen_clip()
    gen_record()
    gen_fp16()
    gen_full_unet()
    gen_full_step()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
