"""Pins the CPU oracle (oracle/) to golden vectors captured from the real reference
(oracle/tools/gen_golden.py).  CPU-only."""
import os

import numpy as np
import pytest
import torch

from oracle import diffusion_ref as D
from oracle import unet_ref as U

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TINY_KW = dict(image_size=256, num_channels=32, num_res_blocks=1, channel_mult="1,2,2",
               attention_resolutions="128,64", num_head_channels=16, num_heads=4,
               learn_sigma=True, use_scale_shift_norm=True, resblock_updown=True,
               pretrain_model="osmosis")


def load(name):
    return dict(np.load(os.path.join(GOLD, name)))


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("tag,kw", [("T1000", dict(timestep_respacing=1000)),
                                     ("T250", dict(timestep_respacing="250")),
                                     ("T10", dict(timestep_respacing=[10]))])
def test_schedule_tables_bit_exact(tag, kw):
    g = load("schedules.npz")
    tb = D.make_tables(1000, "linear", **kw)
    assert np.array_equal(tb.betas, g[f"{tag}.betas"])
    assert np.array_equal(np.array(tb.timestep_map), g[f"{tag}.timestep_map"])
    for name in ("alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                 "posterior_mean_coef1", "posterior_mean_coef2", "posterior_log_variance_clipped",
                 "log_betas"):
        assert np.array_equal(getattr(tb, name), g[f"{tag}.{name}"]), name


def test_schedule_misc():
    g = load("schedules.npz")
    assert np.array_equal(D.named_beta_schedule("cosine", 50), g["cosine50.betas"])
    assert sorted(D.space_timesteps(300, [10, 15, 20])) == list(g["space_1000_10_15_20"])
    assert sorted(D.space_timesteps(1000, "ddim25")) == list(g["space_ddim25"])
    with pytest.raises(NotImplementedError):
        D.named_beta_schedule("nope", 10)
    with pytest.raises(ValueError):
        D.space_timesteps(10, [20])


def test_space_timesteps_every_argument_form():
    """`space_timesteps` (gaussian_diffusion.py:373-426) of the package AND of the oracle vs the reference over the argument forms it
    accepts (int, list, comma-separated section counts, "ddimN") and refuses (tests/golden/spacing.json: 24 cases, 3 ValueErrors)."""
    import ast
    import json
    from osmosis_diffusion_code_amd.guided_diffusion.gaussian_diffusion import space_timesteps
    with open(os.path.join(GOLD, "spacing.json")) as f:
        cases = json.load(f)
    assert len(cases) == 24 and sum(v == "ValueError" for v in cases.values()) == 3
    for key, want in cases.items():
        n, spec = key.split("|", 1)
        n, spec = int(n), ast.literal_eval(spec)
        for fn in (space_timesteps, D.space_timesteps):
            if want == "ValueError":
                with pytest.raises(ValueError):
                    fn(n, spec)
            else:
                assert sorted(int(v) for v in fn(n, spec)) == want, (key, fn.__module__)


def test_timestep_embedding():
    g = load("blocks.npz")
    t = T(g["temb.t"])
    assert torch.equal(U.timestep_embedding(t, 64), T(g["temb.out64"]))
    assert torch.equal(U.timestep_embedding(t, 256), T(g["temb.out256"]))


def test_group_norm():
    g = load("blocks.npz")
    x = T(g["gn.x"]).requires_grad_(True)
    y = U.group_norm32(x, T(g["gn.weight"]), T(g["gn.bias"]))
    (dx,) = torch.autograd.grad((y * T(g["gn.dy"])).sum(), x)
    assert torch.allclose(y, T(g["gn.y"]), atol=1e-6)
    assert torch.allclose(dx, T(g["gn.dx"]), atol=1e-6)


@pytest.mark.parametrize("tag,kw", [("res_plain", {}), ("res_skip", {}), ("res_up", dict(up=True)),
                                     ("res_down", dict(down=True))])
def test_res_block(tag, kw):
    g = load("blocks.npz")
    sd = {k[len(tag) + 4:]: T(v) for k, v in g.items() if k.startswith(tag + ".sd.")}
    x = T(g[f"{tag}.x"]).requires_grad_(True)
    y = U.res_block(sd, "", x, T(g[f"{tag}.emb"]), **kw)
    (dx,) = torch.autograd.grad((y * T(g[f"{tag}.dy"])).sum(), x)
    assert torch.allclose(y, T(g[f"{tag}.y"]), atol=2e-6)
    assert torch.allclose(dx, T(g[f"{tag}.dx"]), atol=2e-6)


@pytest.mark.parametrize("tag,new", [("attn_legacy", False), ("attn_new", True)])
def test_attention_block(tag, new):
    g = load("blocks.npz")
    sd = {k[len(tag) + 4:]: T(v) for k, v in g.items() if k.startswith(tag + ".sd.")}
    x = T(g[f"{tag}.x"]).requires_grad_(True)
    y = U.attention_block(sd, "", x, 4, new)
    (dx,) = torch.autograd.grad((y * T(g[f"{tag}.dy"])).sum(), x)
    assert torch.allclose(y, T(g[f"{tag}.y"]), atol=2e-6)
    assert torch.allclose(dx, T(g[f"{tag}.dx"]), atol=2e-6)


def tiny():
    cfg = U.UNetConfig.from_create_model_kwargs(**TINY_KW)
    return cfg, U.seeded_state_dict(cfg, 1234)


def test_tiny_unet_forward_and_input_grad():
    g = load("tiny_unet.npz")
    cfg, sd = tiny()
    assert sum(v.numel() for v in sd.values()) == int(g["n_params"])
    chk = float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(chk - float(g["weight_abs_sum"])) < 1e-6 * chk
    x = T(g["x"]).requires_grad_(True)
    y = U.unet_forward(sd, cfg, x, T(g["t"]))
    (dx,) = torch.autograd.grad((y[:, :4] ** 2).sum(), x)
    assert torch.allclose(y, T(g["y"]), atol=1e-5), (y - T(g["y"])).abs().max()
    assert torch.allclose(dx, T(g["dx"]), atol=1e-5 * float(T(g["dx"]).abs().max()) + 1e-6)


OPS = {
    "underwater_physical_revised": (
        dict(depth_type="gamma", value="1.4,1.4,1", phi_a="1.1,0.95,0.95", phi_b="0.95, 0.8, 0.8",
             phi_inf="0.14, 0.29, 0.49"),
        dict(scale="7,7,7,0.9", gradient_clip="True,0.005", aux={"avrg_loss": 0.5, "val_loss": 20})),
    "underwater_physical": (
        dict(depth_type="original", value="1.4,1.4,1", phi_ab="1.1,0.95,0.95", phi_inf="0.2,0.4,0.7"),
        dict(scale="4,4,4,1", gradient_clip="True,0.001", aux={"val_loss": 40})),
    "haze_physical": (
        dict(depth_type="gamma", value="1.4,1.4,1", phi_ab="1.0", phi_inf="0.14, 0.29, 0.49"),
        dict(scale="7,7,7,0.9", gradient_clip="True,0.005", aux={"avrg_loss": 0.5, "val_loss": 20})),
}
PATTERN = dict(pattern="pcgs", update_start=0.7, update_end=0, start_guidance=1, stop_guidance=0)


@pytest.mark.parametrize("opname", list(OPS))
def test_guided_loop_10_steps(opname):
    g = load(f"loop_{opname}.npz")
    cfg, sd = tiny()
    okw, ckw = OPS[opname]
    op = D.PhysOperator(opname, batch_size=1, **okw)
    guide = D.OsmosisGuidance(op, n_iter=20, **ckw)
    tb = D.Tables(D.named_beta_schedule("linear", 1000), range(0, 100, 10))
    assert tb.timestep_map == list(g["timestep_map"])
    trace = []
    model = lambda x, t: U.unet_forward(sd, cfg, x, t)  # noqa: E731
    img, variables, loss, x0 = D.p_sample_loop(model, tb, T(g["x_T"]), T(g["y"]), guide, PATTERN,
                                               [T(n) for n in g["noise"]], trace)
    # teacher-free 10-step run: same torch kernels as the reference -> tight agreement
    for k, rec in enumerate(trace):
        assert torch.allclose(rec["x0"], T(g["trace.x0"][k]), atol=5e-5), (k, "x0")
        assert np.allclose(rec["loss"], g["trace.loss"][k], rtol=1e-5), (k, "loss")
    assert torch.allclose(img, T(g["final_img"]), atol=5e-5)
    assert torch.allclose(x0, T(g["final_x0"]), atol=5e-5)
    for n, v in variables.items():
        assert torch.allclose(v, T(g[f"final.{n}"]), atol=1e-6), n


@pytest.mark.parametrize("optimizer", ["adamw", "rmsprop", "asgd", "rprop"])
def test_guided_loop_with_a_torch_optimizer_on_phi(optimizer):
    """Round 6: the oracle's phi step with `optimizer: <name>` (measurements.py:244-249, utils.py:494-524) against the REAL reference's
    guided loop with that optimizer (loop_optimizers.npz): the checker of the device-side optimizer steps is itself pinned."""
    g = load("loop_optimizers.npz")
    cfg, sd = tiny()
    okw, ckw = OPS["underwater_physical_revised"]
    eta = float(g[f"{optimizer}.eta"])
    op = D.PhysOperator("underwater_physical_revised", batch_size=1, optimizer=optimizer, phi_a_eta=eta, phi_b_eta=eta, phi_inf_eta=eta,
                        **okw)
    guide = D.OsmosisGuidance(op, n_iter=20, **ckw)
    tb = D.Tables(D.named_beta_schedule("linear", 1000), range(0, 100, 10))
    trace = []
    img, variables, loss, x0 = D.p_sample_loop(lambda x, t: U.unet_forward(sd, cfg, x, t), tb, T(g["x_T"]), T(g["y"]), guide, PATTERN,
                                               [T(n) for n in g["noise"]], trace)
    for k, rec in enumerate(trace):
        assert np.allclose(rec["loss"], g[f"{optimizer}.loss"][k], rtol=1e-5), (k, "loss")
        for n in ("phi_a", "phi_b", "phi_inf"):
            assert torch.allclose(rec["phi"][n], T(g[f"{optimizer}.{n}"][k]), atol=2e-6), (k, n)
    assert torch.allclose(img, T(g[f"{optimizer}.final_img"]), atol=5e-5)
    assert float((variables["phi_inf"] - T(g[f"{optimizer}.phi_inf"][0])).abs().max()) > 5e-3      # phi really moved


@pytest.mark.parametrize("mean_type,var_type", [("start_x", "fixed_small"), ("epsilon", "fixed_large"), ("epsilon", "learned"),
                                                ("start_x", "learned")])
def test_guided_loop_with_the_other_processors(mean_type, var_type):
    """Round 6: the mean / variance processors no shipped config names (posterior_mean_variance.py:53-101, :171-222) in the oracle's
    `p_mean_variance`, against the REAL reference's guided loop built with them (loop_processors.npz; x_T, y and the noise are those
    of loop_underwater_physical_revised.npz)."""
    g, base = load("loop_processors.npz"), load("loop_underwater_physical_revised.npz")
    cfg, sd = tiny()
    okw, ckw = OPS["underwater_physical_revised"]
    guide = D.OsmosisGuidance(D.PhysOperator("underwater_physical_revised", batch_size=1, **okw), n_iter=20, **ckw)
    tb = D.Tables(D.named_beta_schedule("linear", 1000), range(0, 100, 10))
    trace = []
    img, variables, loss, x0 = D.p_sample_loop(lambda x, t: U.unet_forward(sd, cfg, x, t), tb, T(base["x_T"]), T(base["y"]), guide,
                                               PATTERN, [T(n) for n in base["noise"]], trace, mean_type=mean_type, var_type=var_type)
    tag = f"osmosis.{mean_type}.{var_type}"
    scale = max(1.0, float(np.abs(g[f"{tag}.final_img"]).max()))
    assert torch.allclose(trace[0]["x0"], T(g[f"{tag}.x0_first"]), atol=5e-6)
    for k, rec in enumerate(trace):
        assert np.allclose(rec["loss"], g[f"{tag}.loss"][k], rtol=2e-5), (k, "loss", rec["loss"], g[f"{tag}.loss"][k])
    assert torch.allclose(img, T(g[f"{tag}.final_img"]), atol=5e-5 * scale)
    assert torch.allclose(x0, T(g[f"{tag}.final_x0"]), atol=5e-5 * scale)
    for n, v in variables.items():
        assert torch.allclose(v, T(g[f"{tag}.{n}"]), atol=1e-6), n


def test_guided_loop_with_clip_denoised():
    """Round 6: `clip_denoised: True` (configs/rgb_guidance_sample_config.yaml's setting; process_xstart, posterior_mean_variance.py:43-50)
    in the oracle's Osmosis loop vs the REAL reference (loop_clip.npz): 6.4 % of the pred_xstart elements sit on the clamp."""
    g, base = load("loop_clip.npz"), load("loop_underwater_physical_revised.npz")
    cfg, sd = tiny()
    okw, ckw = OPS["underwater_physical_revised"]
    guide = D.OsmosisGuidance(D.PhysOperator("underwater_physical_revised", batch_size=1, **okw), n_iter=20, **ckw)
    tb = D.Tables(D.named_beta_schedule("linear", 1000), range(0, 100, 10))
    trace = []
    img, variables, loss, x0 = D.p_sample_loop(lambda x, t: U.unet_forward(sd, cfg, x, t), tb, T(base["x_T"]), T(base["y"]), guide,
                                               PATTERN, [T(n) for n in base["noise"]], trace, clip_denoised=True)
    assert float(g["osmosis.clamped_fraction"]) > 0.03
    for k, rec in enumerate(trace):
        assert torch.allclose(rec["x0"], T(g["osmosis.x0"][k]), atol=5e-5), (k, "x0")
        assert torch.allclose(rec["grad"], T(g["osmosis.grad"][k]), atol=5e-5 * float(np.abs(g["osmosis.grad"][k]).max())), (k, "grad")
        assert np.allclose(rec["loss"], g["osmosis.loss"][k], rtol=1e-5), (k, "loss")
    assert float(torch.stack([r["x0"] for r in trace]).abs().max()) == 1.0
    assert torch.allclose(img, T(g["osmosis.final_img"]), atol=5e-5)


@pytest.mark.parametrize("dyn,clip", [(True, False), (True, True), (False, True)])
def test_process_xstart_matches_the_reference(dyn, clip):
    """process_xstart (posterior_mean_variance.py:43-50) of the oracle AND of the package's mean processors vs the reference's on a
    seeded tensor: its `dynamic_thresholding` (util/img_utils.py:8-15) multiplies by the 0.98-quantile of |x| over the whole tensor
    and clips -- the package restated Imagen's clip-and-divide until round 6."""
    from osmosis_diffusion_code_amd.guided_diffusion.posterior_mean_variance import get_mean_processor
    g = load("loop_clip.npz")
    want = T(g[f"px.dyn{int(dyn)}.clip{int(clip)}"])
    assert torch.equal(D.process_xstart(T(g["px.x"]), clip, dyn), want)
    for name in ("epsilon", "start_x", "previous_x"):
        proc = get_mean_processor(name, betas=D.named_beta_schedule("linear", 1000), dynamic_threshold=dyn, clip_denoised=clip)
        assert torch.equal(proc.process_xstart(T(g["px.x"])), want), name


def test_fp16_reference_fixture_is_consistent():
    """tests/golden/fp16_reference.npz (round 4: the real reference with convert_to_fp16() applied): its fp32 half is the same
    network on the same inputs -- the oracle reproduces it -- and its fp16 half differs from it by the half-precision amount."""
    from oracle import unet_ref as U
    g = dict(np.load(os.path.join(GOLD, "fp16_reference.npz")))
    kw = dict(image_size=256, num_channels=32, num_res_blocks=1, channel_mult="1,2,2", attention_resolutions="128,64",
              num_head_channels=16, num_heads=4, learn_sigma=True, use_scale_shift_norm=True, resblock_updown=True,
              pretrain_model="osmosis")
    cfg = U.UNetConfig.from_create_model_kwargs(**kw)
    sd = U.seeded_state_dict(cfg, 1234)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = U.unet_forward(sd, cfg, x, torch.from_numpy(g["t"]))
    (dx,) = torch.autograd.grad((y * torch.from_numpy(g["w"])).sum(), x)
    assert float((y.detach() - torch.from_numpy(g["y32"])).abs().max()) < 2e-5
    assert float((dx - torch.from_numpy(g["dx32"])).abs().max()) < 2e-5 * float(np.abs(g["dx32"]).max()) + 1e-6
    gap = float(np.abs(g["y16"] - g["y32"]).max()) / float(np.abs(g["y32"]).max())
    assert 5e-4 < gap < 1e-2            # really fp16 (not a silently un-converted model), and no worse than fp16
    assert g["loop.trace.x0"].shape == (10, 1, 4, 32, 32) and np.isfinite(g["loop.final_img"]).all()


def test_full_size_oracle_vs_reference_golden():
    """The oracle's restatement at the FULL architecture (552.8 M parameters, channel_mult picked for image_size 256, attention at
    32 / 16 / 8 with 64-wide heads) against tests/golden/full_unet.npz, which the real reference produced on CPU
    (oracle/tools/gen_golden.py full_unet): every 4th pixel of y and of the input gradient, and their norms, at t = 37 and 999.
    Until round 4 the full-size oracle was pinned only through the tiny configurations."""
    g = dict(np.load(os.path.join(GOLD, "full_unet.npz")))
    kw = dict(image_size=256, num_channels=256, num_res_blocks=2, channel_mult="", learn_sigma=True, class_cond=False,
              use_checkpoint=False, attention_resolutions="32, 16, 8", num_heads=4, num_head_channels=64, num_heads_upsample=-1,
              use_scale_shift_norm=True, dropout=0.0, resblock_updown=True, use_fp16=False, use_new_attention_order=False,
              model_path="", pretrain_model="osmosis")
    cfg = U.UNetConfig.from_create_model_kwargs(**kw)
    sd = U.seeded_state_dict(cfg, 1234)
    gen = torch.Generator().manual_seed(int(g["seed"]))
    x = float(g["x_scale"]) * torch.randn(1, 4, 256, 256, generator=gen)
    w = torch.randn(1, 8, 256, 256, generator=gen)
    assert abs(float(x.double().abs().sum()) - float(g["x_abs_sum"])) < 1e-6 * float(g["x_abs_sum"])
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    st = int(g["stride"])
    for t in (37, 999):
        xr = x.clone().requires_grad_(True)
        y = U.unet_forward(sd, cfg, xr, torch.tensor([float(t)]))
        (dx,) = torch.autograd.grad((y * w).sum(), xr)
        tag = f"t{t}"
        ey = float((y.detach()[:, :, ::st, ::st] - torch.from_numpy(g[tag + ".y_sub"])).abs().max())
        ed = float((dx[:, :, ::st, ::st] - torch.from_numpy(g[tag + ".dx_sub"])).abs().max())
        assert ey < 2e-5 * float(g[tag + ".y_max"]) and ed < 2e-5 * float(g[tag + ".dx_max"]), (t, ey, ed)
        assert abs(float(y.double().pow(2).sum().sqrt()) - float(g[tag + ".y_l2"])) < 1e-5 * float(g[tag + ".y_l2"])
        assert abs(float(dx.double().pow(2).sum().sqrt()) - float(g[tag + ".dx_l2"])) < 1e-5 * float(g[tag + ".dx_l2"])


def _reference_noise_draws(n_steps, y_shape, x_shape, seed=0):
    """The reference's per-step draws after torch.manual_seed(seed): randn_like(measurement) (unused), then
    randn_like(img)."""
    torch.manual_seed(seed)
    out = []
    for _ in range(n_steps):
        torch.randn_like(torch.empty(y_shape))
        out.append(torch.randn_like(torch.empty(x_shape)))
    return out


@pytest.mark.parametrize("fname,cfg_name", [("full_step.npz", "SAMPLE"), ("full_step_underwater_physical.npz", "SIMULATION"),
                                            ("full_step_haze_physical.npz", "HAZE")])
def test_full_size_guided_steps_oracle_vs_reference_golden(fname, cfg_name):
    """Two guided steps of the loop with the FULL network (t = 299 with the 20-iteration phi update, then t = 0) -- the oracle's
    p_sample_loop against tests/golden/full_step*.npz, which the real reference's p_sample_loop produced on CPU, for the operator /
    guidance settings of BASELINE configs 2 (revised underwater), 3 (underwater, original depth, val_loss 40) and 5 (haze)."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    import baseline_configs as BC
    c = getattr(BC, cfg_name)
    g = dict(np.load(os.path.join(GOLD, fname)))
    kw = dict(BC.UNET)
    cfg = U.UNetConfig.from_create_model_kwargs(**kw)
    sd = U.seeded_state_dict(cfg, 1234)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    x_T = 0.5 * torch.randn(1, 4, 256, 256, generator=torch.Generator().manual_seed(0))
    y = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(7)) * 1.6 - 0.8
    assert abs(float(x_T.double().abs().sum()) - float(g["x_T_abs_sum"])) < 1e-6 * float(g["x_T_abs_sum"])
    noises = _reference_noise_draws(2, y.shape, x_T.shape)
    tb = D.Tables(D.named_beta_schedule("linear", 1000), (0, 299))
    opc = dict(c["measurement"]["operator"])
    name = opc.pop("name")
    opc.pop("optimizer", None)
    rop = D.PhysOperator(name, batch_size=1, **opc)
    p = c["conditioning"]["params"]
    rg = D.OsmosisGuidance(rop, n_iter=20, scale=p["scale"], gradient_clip=p["gradient_clip"], loss_function=p["loss_function"],
                           loss_weight=p["loss_weight"], weight_function=p["weight_function"], aux=c["aux_loss"]["aux_loss"])
    trace = []
    img, variables, loss, x0 = D.p_sample_loop(lambda x, t: U.unet_forward(sd, cfg, x, t), tb, x_T, y, rg, c["sample_pattern"],
                                               noises, trace)
    st = int(g["stride"])
    assert float((img[:, :, ::st, ::st] - torch.from_numpy(g["final_img_sub"])).abs().max()) < 1e-4
    assert float((x0[:, :, ::st, ::st] - torch.from_numpy(g["final_x0_sub"])).abs().max()) < 1e-4
    assert np.allclose(np.asarray(loss).ravel(), g["final_loss"].ravel(), rtol=1e-4)
    for k, v in variables.items():
        assert np.allclose(v.detach().numpy().ravel(), g["final." + k].ravel(), atol=5e-6), k


VARIANT_KW = {
    "conv_updown_additive": dict(TINY_KW, resblock_updown=False, use_scale_shift_norm=False),
    "conv_updown_classcond": dict(TINY_KW, resblock_updown=False, class_cond=True, dropout=0.1),
    "resblock_updown_additive": dict(TINY_KW, use_scale_shift_norm=False),
}


@pytest.mark.parametrize("tag", sorted(VARIANT_KW))
def test_unet_variants_no_shipped_config_uses(tag):
    """Round 5: Upsample / Downsample layers with convolutions (resblock_updown=False), additive conditioning
    (use_scale_shift_norm=False), class conditioning -- the oracle vs the real reference (unet_variants.npz)."""
    g = load("unet_variants.npz")
    cfg = U.UNetConfig.from_create_model_kwargs(**VARIANT_KW[tag])
    sd = U.seeded_state_dict(cfg, 4321)
    assert sum(v.numel() for v in sd.values()) == int(g[f"{tag}.n_params"])
    x = T(g[f"{tag}.x"]).requires_grad_(True)
    y = U.unet_forward(sd, cfg, x, T(g[f"{tag}.t"]), y=T(g[f"{tag}.labels"]) if f"{tag}.labels" in g else None)
    (dx,) = torch.autograd.grad((y * T(g[f"{tag}.w"])).sum(), x)
    assert torch.allclose(y, T(g[f"{tag}.y"]), atol=5e-6), float((y - T(g[f"{tag}.y"])).abs().max())
    assert torch.allclose(dx, T(g[f"{tag}.dx"]), atol=2e-5), float((dx - T(g[f"{tag}.dx"])).abs().max())
