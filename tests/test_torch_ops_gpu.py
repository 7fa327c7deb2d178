"""The `torch.library` boundary (SURVEY.md section 8b / VERDICT r04 item 6): the osmosis:: operators are registered with schema, fake
implementations and autograd, `torch.library.opcheck` passes on them, `model(x, t)` goes through `osmosis::unet_fwd` and is
differentiable w.r.t. x (reference condition_methods.py:188-191), and the functional guidance operators give the numbers of the
in-place C-ABI calls the fused sampler loop makes."""
import numpy as np
import pytest
import torch

from oracle import unet_ref as U

DEV = "cuda:0"
TINY_KW = dict(image_size=256, num_channels=32, num_res_blocks=1, channel_mult="1,2,2", attention_resolutions="128,64",
               num_head_channels=16, num_heads=4, learn_sigma=True, use_scale_shift_norm=True, resblock_updown=True,
               pretrain_model="osmosis")


def test_operators_are_registered_with_schemas():
    """CPU: importing the package registers the eight operators (no GPU, no library call needed for that)."""
    from osmosis_diffusion_code_amd import torch_ops
    want = {"unet_fwd": "osmosis::unet_fwd(Tensor x, Tensor t, SymInt engine) -> Tensor",
            "unet_bwd_data": "osmosis::unet_bwd_data(Tensor grad_out, SymInt engine) -> Tensor",
            "posterior": "osmosis::posterior(Tensor model_out, Tensor x, Tensor coef, SymInt mean_kind=0, SymInt var_kind=0) -> (Tensor, Tensor, Tensor)",
            "posterior_bwd": "osmosis::posterior_bwd(Tensor g, Tensor coef) -> Tensor"}
    assert set(torch_ops.OPS) >= set(want)
    for name in torch_ops.OPS:
        op = getattr(torch.ops.osmosis, name).default
        if name in want:
            assert str(op._schema) == want[name]
    # no CPU kernel exists: the product path fails loudly instead of falling back
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.osmosis.posterior(torch.zeros(1, 8, 4, 4), torch.zeros(1, 4, 4, 4), torch.zeros(8))
    # fake tensors flow through (shape inference without a device)
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        x0, mean, lv = torch.ops.osmosis.posterior(torch.empty(2, 8, 16, 16, device="cuda"), torch.empty(2, 4, 16, 16, device="cuda"),
                                                   torch.empty(8, device="cuda"))
        assert x0.shape == mean.shape == lv.shape == (2, 4, 16, 16)
        d = torch.ops.osmosis.posterior_bwd(torch.empty(2, 4, 16, 16, device="cuda"), torch.empty(8, device="cuda"))
        assert d.shape == (2, 8, 16, 16)


def _tiny():
    from osmosis_diffusion_code_amd.guided_diffusion import unet
    cfg = U.UNetConfig.from_create_model_kwargs(**TINY_KW)
    sd = U.seeded_state_dict(cfg, 1234)
    m = unet.create_model(**TINY_KW)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval(), cfg, sd


@pytest.mark.gpu
def test_unet_operator_opcheck_and_autograd():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from osmosis_diffusion_code_amd import torch_ops
    m, cfg, sd = _tiny()
    m.conv_mode = "f32"
    g = torch.Generator().manual_seed(3)
    x = (0.6 * torch.randn(1, 4, 32, 32, generator=g)).to(DEV)
    t = torch.tensor([17.0], device=DEV)
    w = torch.randn(1, 8, 32, 32, generator=g).to(DEV)
    # model(x, t) IS the operator, and differentiable w.r.t. x
    xr = x.clone().requires_grad_(True)
    y = m(xr, t)
    assert y.requires_grad and y.grad_fn is not None
    (dx,) = torch.autograd.grad((y * w).sum(), xr)
    xc = x.cpu().clone().requires_grad_(True)
    yc = U.unet_forward(sd, cfg, xc, t.cpu())
    (dxc,) = torch.autograd.grad((yc * w.cpu()).sum(), xc)
    assert float((y.detach().cpu() - yc.detach()).abs().max()) < 2e-5
    assert float((dx.cpu() - dxc).abs().max()) < 2e-5 * max(1.0, float(dxc.abs().max()))
    eng = m.engine(1, 32, 32)
    h = torch_ops.engine_handle(eng)
    out = torch.ops.osmosis.unet_fwd(x, t, h)
    assert torch.equal(out, y.detach())
    # a second forward invalidates the first pass's ticket: differentiating through overwritten activations raises
    y1 = m(xr, t)
    m(x, t)
    with pytest.raises(RuntimeError, match="overwritten"):
        torch.autograd.grad((y1 * w).sum(), xr)
    # opcheck: schema, fake tensor, autograd registration, AOT dispatch (static and dynamic)
    torch.library.opcheck(torch.ops.osmosis.unet_fwd.default, (x.clone().requires_grad_(True), t, h))
    out = torch.ops.osmosis.unet_fwd(x, t, h)
    torch.library.opcheck(torch.ops.osmosis.unet_bwd_data.default, (w, h))
    with pytest.raises(Exception):
        torch.ops.osmosis.unet_fwd(x, t, 12345)          # stale / unknown handle


@pytest.mark.gpu
def test_guidance_operators_match_the_in_place_calls_and_opcheck():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from osmosis_diffusion_code_amd import ops, torch_ops
    from osmosis_diffusion_code_amd.guided_diffusion import condition_methods as CM
    from osmosis_diffusion_code_amd.guided_diffusion import gaussian_diffusion as gd
    from osmosis_diffusion_code_amd.guided_diffusion import measurements as M
    B, H, W = 2, 32, 32
    g = torch.Generator().manual_seed(9)
    model_out = torch.randn(B, 8, H, W, generator=g).to(DEV)
    x = (0.5 * torch.randn(B, 4, H, W, generator=g)).to(DEV)
    y = (torch.rand(B, 3, H, W, generator=g) * 1.6 - 0.8).to(DEV)
    noise = torch.randn(B, 4, H, W, generator=g).to(DEV)
    sampler = gd.create_sampler(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type="epsilon",
                                model_var_type="learned_range", dynamic_threshold=False, clip_denoised=False,
                                rescale_timesteps=False, timestep_respacing=1000)
    coef = torch.from_numpy(sampler.coef_table()[120].copy()).to(DEV)
    # posterior
    x0, mean, lv = torch.ops.osmosis.posterior(model_out, x, coef)
    r = [torch.empty_like(x) for _ in range(3)]
    ops.posterior(model_out, x, coef, r[0], r[1], r[2], B, H * W)
    assert torch.equal(x0, r[0]) and torch.equal(mean, r[1]) and torch.equal(lv, r[2])
    torch.library.opcheck(torch.ops.osmosis.posterior.default, (model_out, x, coef))
    # physics loss + gradient (functional) vs the conditioning method's in-place path
    op = M.get_operator("underwater_physical_revised", device=DEV, batch_size=B, optimizer="sgd", depth_type="gamma", value="1.4,1.4,1",
                        phi_a="1.1,0.95,0.95", phi_b="0.95, 0.8, 0.8", phi_inf="0.14, 0.29, 0.49")
    cond = CM.get_conditioning_method("osmosis", op, M.get_noise("clean"), loss_function="norm", loss_weight="depth",
                                      weight_function="gamma,1.4,1.4,1", scale="7,7,7,0.9", gradient_x_prev=True,
                                      gradient_clip="True,0.005", pattern="pcgs", update_start=0.7, update_end=0, global_N=1,
                                      local_M=1, s_start=1, s_end=0, n_iter=20, start_guidance=1, stop_guidance=0,
                                      aux_loss={"avrg_loss": 0.5, "val_loss": 20})
    x0s = (0.4 * x0).contiguous()
    phi0 = op.phi.clone()
    st = cond._prepare(B, H * W, torch.device(DEV))
    icfg, fcfg = torch_ops.phys_config(st["desc"])
    loss, gx0, phi_new = torch.ops.osmosis.phys_loss_grad(x0s, y, phi0, icfg, fcfg, 20, False)
    g_ref, loss_ref = cond.loss_grad_x0(x0s, y, freeze_phi=False)
    assert torch.equal(gx0, g_ref) and torch.equal(loss, loss_ref) and torch.equal(phi_new, op.phi)
    assert not torch.equal(phi_new, phi0)                   # phi moved (20 SGD steps), the input tensor did not
    torch.library.opcheck(torch.ops.osmosis.phys_loss_grad.default, (x0s, y, phi0, icfg, fcfg, 20, False))
    # chain rule into the UNet + update rule
    d_out = torch.ops.osmosis.posterior_bwd(gx0, coef)
    d_ref = torch.zeros(B, 8, H, W, device=DEV)
    ops.posterior_bwd(gx0, coef, d_ref, B, H * W)
    assert torch.equal(d_out, d_ref)
    torch.library.opcheck(torch.ops.osmosis.posterior_bwd.default, (gx0, coef))
    dx_unet = torch.randn(B, 4, H, W, generator=g).to(DEV)
    scale4 = cond.scale4(torch.device(DEV))
    x_next, grad = torch.ops.osmosis.guide_update(mean, lv, gx0, dx_unet, noise, coef, scale4, float(cond.clip_value))
    xr, gr = torch.empty_like(x), torch.empty_like(x)
    ops.guide_update(mean, lv, gx0, dx_unet, noise, coef, scale4, cond.clip_value, xr, gr, B, H * W)
    assert torch.equal(x_next, xr) and torch.equal(grad, gr)
    torch.library.opcheck(torch.ops.osmosis.guide_update.default, (mean, lv, gx0, dx_unet, noise, coef, scale4, float(cond.clip_value)))
    assert np.isfinite(float(loss.sum()))
    # round 6: the update with the noise drawn in the kernel, the DDIM step, the identity operator of the rgb-guidance path
    step = torch.tensor([120], device=DEV, dtype=torch.int32)
    xn2, gr2, nz2 = torch.ops.osmosis.guide_update_rng(mean, lv, gx0, dx_unet, coef, scale4, float(cond.clip_value), 77, step, 0, 0, 1)
    xr2, _ = torch.ops.osmosis.guide_update(mean, lv, gx0, dx_unet, nz2, coef, scale4, float(cond.clip_value))
    assert torch.equal(gr2, grad) and torch.allclose(xn2, xr2, atol=1e-6) and abs(float(nz2.double().var()) - 1.0) < 0.1
    torch.library.opcheck(torch.ops.osmosis.guide_update_rng.default,
                          (mean, lv, gx0, dx_unet, coef, scale4, float(cond.clip_value), 77, step, 0, 0, 1))
    ddim = gd.get_sampler("ddim")(use_timesteps=range(0, 1000, 4), betas=gd.get_named_beta_schedule("linear", 1000),
                                  model_mean_type="epsilon", model_var_type="learned_range", dynamic_threshold=False,
                                  clip_denoised=False, rescale_timesteps=False)
    c2 = torch.from_numpy(ddim.coef_table()[120].copy()).to(DEV)
    d2 = torch.from_numpy(ddim.ddim_table(0.3)[120].copy()).to(DEV)
    xd, gd_ = torch.ops.osmosis.ddim_update(x0, x, gx0, dx_unet, noise, c2, d2, scale4, -1.0)
    xe, ge = torch.empty_like(x), torch.empty_like(x)
    ops.ddim_update(x0, x, gx0, dx_unet, noise, c2, d2, scale4, -1.0, xe, ge, B, H * W)
    assert torch.equal(xd, xe) and torch.equal(gd_, ge)
    torch.library.opcheck(torch.ops.osmosis.ddim_update.default, (x0, x, gx0, dx_unet, noise, c2, d2, scale4, -1.0))
    ps = CM.get_conditioning_method("ps", M.get_operator("rgb_guidance", device=DEV, batch_size=B), M.get_noise("gaussian", sigma=0.05),
                                    scale="0.6,0.5,0.4,0.0")
    g_ps, l_ps = ps.loss_grad_x0(x0s, y)
    icfg3, fcfg3 = torch_ops.phys_config(ps._states[(B, H * W, DEV)]["desc"])
    l3, g3, _phi3 = torch.ops.osmosis.phys_loss_grad(x0s, y, phi0, icfg3, fcfg3, 1, True)
    assert torch.equal(g3, g_ps) and torch.equal(l3, l_ps)
    with pytest.raises(Exception, match="no parameters"):
        torch.ops.osmosis.phys_loss_grad(x0s, y, phi0, icfg3, fcfg3, 1, False)
    # clip_denoised of the shipped rgb-guidance config: clamped posterior + the clamp's backward as functional operators
    xc, mc, lc, raw = torch.ops.osmosis.posterior_clip(model_out, x, coef)
    assert torch.equal(raw, x0) and torch.equal(xc, x0.clamp(-1, 1)) and torch.equal(lc, lv) and float(xc.abs().max()) == 1.0
    gm = torch.ops.osmosis.clamp_bwd(gx0, raw)
    assert torch.equal(gm, torch.where(raw.abs() <= 1.0, gx0, torch.zeros_like(gx0))) and not torch.equal(gm, gx0)
    torch.library.opcheck(torch.ops.osmosis.posterior_clip.default, (model_out, x, coef))
    torch.library.opcheck(torch.ops.osmosis.clamp_bwd.default, (gx0, raw))
