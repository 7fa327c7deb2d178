"""Sampler-step kernels (posterior, physical model + loss + analytic gradients, phi SGD, guidance
update) against the CPU oracle's autograd on the same seeded inputs.  fp32, tolerances stated."""
import numpy as np
import pytest
import torch

from oracle import diffusion_ref as D

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

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


@pytest.fixture(scope="module")
def mods():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from osmosis_diffusion_code_amd import ops
    from osmosis_diffusion_code_amd.guided_diffusion import measurements as M
    from osmosis_diffusion_code_amd.guided_diffusion import condition_methods as CM
    return ops, M, CM


def test_posterior_kernel(mods):
    ops, _, _ = mods
    tb = D.make_tables(1000, "linear", "250")
    g = torch.Generator().manual_seed(0)
    B, HW, t = 2, 48, 137
    mo = torch.randn(B, 8, HW, generator=g)
    x = torch.randn(B, 4, HW, generator=g)
    ref = D.p_mean_variance(tb, mo.reshape(B, 8, 6, 8), x.reshape(B, 4, 6, 8), t)
    coef = torch.tensor([np.float32(tb.sqrt_recip_alphas_cumprod[t]), np.float32(tb.sqrt_recipm1_alphas_cumprod[t]),
                         np.float32(tb.posterior_mean_coef1[t]), np.float32(tb.posterior_mean_coef2[t]),
                         np.float32(tb.posterior_log_variance_clipped[t]), np.float32(tb.log_betas[t]), 1.0, t],
                        dtype=torch.float32, device=DEV)
    x0, mean, lv = (torch.empty(B, 4, HW, device=DEV) for _ in range(3))
    ops.posterior(mo.to(DEV), x.to(DEV), coef, x0, mean, lv, B, HW)
    assert torch.allclose(x0.cpu().reshape(B, 4, 6, 8), ref["pred_xstart"], atol=1e-5, rtol=1e-6)
    assert torch.allclose(mean.cpu().reshape(B, 4, 6, 8), ref["mean"], atol=1e-5, rtol=1e-6)
    assert torch.allclose(lv.cpu().reshape(B, 4, 6, 8), ref["log_variance"], atol=1e-5, rtol=1e-6)


@pytest.mark.parametrize("var_type", ["learned_range", "fixed_small", "fixed_large", "learned"])
@pytest.mark.parametrize("mean_type", ["epsilon", "start_x", "previous_x"])
def test_posterior_typed_every_processor_pair(mods, mean_type, var_type):
    """osm_posterior_typed fed by the processors' own `kernel_kind` / `kernel_coefs` vs the oracle's p_mean_variance for every
    registered (mean, variance) pair (posterior_mean_variance.py:53-136, :171-258), and the ROW's chain coefficients -- c0 = d x0/d x
    read by the update kernels, c1 = -d x0/d out read by osm_posterior_bwd -- vs autograd through the oracle."""
    ops, _, _ = mods
    from osmosis_diffusion_code_amd.guided_diffusion import gaussian_diffusion as gd
    sampler = gd.create_sampler(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type=mean_type, model_var_type=var_type,
                                dynamic_threshold=False, clip_denoised=False, rescale_timesteps=False, timestep_respacing="250")
    tb = D.make_tables(1000, "linear", "250")
    table = sampler.coef_table()
    mk, vk = sampler.mean_processor.kernel_kind, sampler.var_processor.kernel_kind
    assert (mk, vk) == ({"epsilon": 0, "start_x": 1, "previous_x": 2}[mean_type],
                        {"learned_range": 0, "fixed_small": 1, "fixed_large": 1, "learned": 2}[var_type])
    g = torch.Generator().manual_seed(5)
    B, HW = 2, 48
    for t in (249, 137, 1, 0):
        mo = torch.randn(B, 8, 6, 8, generator=g).requires_grad_(True)
        x = torch.randn(B, 4, 6, 8, generator=g).requires_grad_(True)
        ref = D.p_mean_variance(tb, mo, x, t, mean_type, var_type)
        coef = torch.from_numpy(table[t].copy()).to(DEV)
        x0, mean, lv = (torch.empty(B, 4, 6, 8, device=DEV) for _ in range(3))
        ops.posterior(mo.detach().to(DEV), x.detach().to(DEV), coef, x0, mean, lv, B, HW, mk, vk)
        scale = max(1.0, float(ref["pred_xstart"].detach().abs().max()))     # previous_x at t = 249: 1 / coef1 ~ 1e3
        assert torch.allclose(x0.cpu(), ref["pred_xstart"].detach(), atol=2e-6 * scale, rtol=1e-6), (t, "x0")
        assert torch.allclose(mean.cpu(), ref["mean"].detach(), atol=2e-6, rtol=1e-6), (t, "mean")
        if mean_type == "start_x":
            assert torch.equal(x0.cpu(), mo.detach()[:, :4])
        if mean_type == "previous_x":
            assert torch.equal(mean.cpu(), mo.detach()[:, :4])
        want_lv = ref["log_variance"].detach()
        if var_type == "fixed_small" and t == 0:
            assert torch.isinf(want_lv).all() and torch.equal(lv.cpu(), want_lv)       # log 0 = -inf, as in the reference
        elif var_type != "learned_range":
            assert torch.equal(lv.cpu(), want_lv), (t, "logvar")
        else:
            assert torch.allclose(lv.cpu(), want_lv, atol=1e-5, rtol=1e-6)
        # chain rule of pred_xstart: d_out = -c1 gx0 on the first four channels, direct term c0 gx0
        gx0 = torch.randn(B, 4, 6, 8, generator=g)
        d_mo, d_x = torch.autograd.grad((ref["pred_xstart"] * gx0).sum(), [mo, x], allow_unused=True)
        d_x = torch.zeros_like(gx0) if d_x is None else d_x
        d_out = torch.empty(B, 8, 6, 8, device=DEV)
        ops.posterior_bwd(gx0.to(DEV), coef, d_out, B, HW)
        assert torch.allclose(d_out.cpu(), d_mo, atol=1e-6 * scale, rtol=1e-6) and float(d_out[:, 4:].abs().max()) == 0.0
        assert torch.allclose(float(coef[0]) * gx0, d_x, atol=1e-6 * scale, rtol=1e-6)


@pytest.mark.parametrize("mean_type", ["epsilon", "start_x", "previous_x"])
def test_posterior_clip_denoised_and_its_backward(mods, mean_type):
    """`clip_denoised` inside osm_posterior_typed (x0 clamped, mean from the clamped x0, raw prediction kept) and osm_clamp_bwd vs
    autograd through the oracle's p_mean_variance(..., clip_denoised=True): the masked gradient through osm_posterior_bwd and the
    direct term c0 * g.  Elements exactly ON the bound pass the gradient, NaN does not."""
    ops, _, _ = mods
    from osmosis_diffusion_code_amd.guided_diffusion import gaussian_diffusion as gd
    sampler = gd.create_sampler(sampler="ddpm", steps=1000, noise_schedule="linear", model_mean_type=mean_type,
                                model_var_type="learned_range", dynamic_threshold=False, clip_denoised=True, rescale_timesteps=False,
                                timestep_respacing="250")
    tb = D.make_tables(1000, "linear", "250")
    table, mk = sampler.coef_table(), sampler.mean_processor.kernel_kind
    g = torch.Generator().manual_seed(6)
    B, HW, t = 2, 48, 60
    mo = torch.randn(B, 8, 6, 8, generator=g).requires_grad_(True)
    x = torch.randn(B, 4, 6, 8, generator=g).requires_grad_(True)
    ref = D.p_mean_variance(tb, mo, x, t, mean_type, "learned_range", clip_denoised=True)
    coef = torch.from_numpy(table[t].copy()).to(DEV)
    x0, mean, lv, raw = (torch.empty(B, 4, 6, 8, device=DEV) for _ in range(4))
    ops.posterior(mo.detach().to(DEV), x.detach().to(DEV), coef, x0, mean, lv, B, HW, mk, 0, x0_raw=raw)
    clamped = float((x0.abs() == 1.0).float().mean())
    assert 0.05 < clamped < 0.99 and float(x0.abs().max()) == 1.0          # (previous_x: 1 / coef1 ~ 8 at t = 60: 98 % clamped)
    assert torch.equal(x0, raw.clamp(-1, 1))
    assert torch.allclose(x0.cpu(), ref["pred_xstart"].detach(), atol=2e-6) and torch.allclose(mean.cpu(), ref["mean"].detach(), atol=2e-6)
    gx0 = torch.randn(B, 4, 6, 8, generator=g)
    d_mo, d_x = torch.autograd.grad((ref["pred_xstart"] * gx0).sum(), [mo, x], allow_unused=True)
    d_x = torch.zeros_like(gx0) if d_x is None else d_x
    gm = gx0.to(DEV)
    ops.clamp_bwd(gm, raw)
    assert torch.equal(gm == 0, (raw.abs() > 1.0))                   # masked exactly where the prediction left [-1, 1]
    d_out = torch.empty(B, 8, 6, 8, device=DEV)
    ops.posterior_bwd(gm, coef, d_out, B, HW)
    assert torch.allclose(d_out.cpu(), d_mo, atol=2e-6, rtol=1e-6)
    assert torch.allclose(float(coef[0]) * gm.cpu(), d_x, atol=2e-6, rtol=1e-6)
    # bounds pass, NaN does not (ATen clamp_backward)
    edge = torch.tensor([-1.0, 1.0, float("nan"), 1.0000001, -0.5], device=DEV)
    ge = torch.ones(5, device=DEV)
    ops.clamp_bwd(ge, edge)
    er = edge.cpu().clone().requires_grad_(True)
    (de,) = torch.autograd.grad(er.clamp(-1, 1).sum(), er)
    assert torch.equal(ge.cpu(), de) and ge.tolist() == [1.0, 1.0, 0.0, 0.0, 1.0]


@pytest.mark.parametrize("opname", list(OPS))
@pytest.mark.parametrize("loss_function", ["norm", "mse"])
def test_physics_loss_grad_and_phi_sgd(mods, opname, loss_function):
    """20 inner iterations of (loss, d/dphi, SGD) + the final d/dx0 -- versus autograd on the oracle."""
    ops, M, CM = mods
    okw, ckw = OPS[opname]
    H = W = 24
    g = torch.Generator().manual_seed(11)
    x0 = (0.6 * torch.randn(1, 4, H, W, generator=g)).requires_grad_(True)
    y = torch.rand(1, 3, H, W, generator=g) * 1.6 - 0.8

    # oracle: x0 is the leaf here (chain through the UNet is tested elsewhere)
    op = D.PhysOperator(opname, batch_size=1, **{**okw, **{k + "_eta": 1e-3 for k in ("phi_a", "phi_b", "phi_ab", "phi_inf")}})
    guide = D.OsmosisGuidance(op, n_iter=20, loss_function=loss_function, **ckw)
    op.set_requires_grad(True)
    for it in range(20):
        sep, loss = guide.loss(x0, y)
        total = loss + D.aux_loss(x0, guide.aux)
        if it == 19:
            total.backward(inputs=[x0] + list(op.phi.values()))
        else:
            total.backward(inputs=list(op.phi.values()))
        op.sgd_step()
    ref_g = x0.grad.clone()

    # HIP path through the drop-in operator / conditioning classes
    oper = M.get_operator(opname, device=DEV, batch_size=1,
                          **{**okw, **{k + "_eta": 1e-3 for k in ("phi_a", "phi_b", "phi_ab", "phi_inf")
                                       if k in ("phi_inf",) or k in okw}}, optimizer="sgd")
    cond = CM.get_conditioning_method("osmosis", oper, M.get_noise("clean"), loss_function=loss_function,
                                      loss_weight="depth", weight_function="gamma,1.4,1.4,1",
                                      scale=ckw["scale"], gradient_x_prev=True, gradient_clip=ckw["gradient_clip"],
                                      n_iter=20, aux_loss=ckw["aux"], pattern="pcgs")
    gx0, sep_loss = cond.loss_grad_x0(x0.detach().to(DEV), y.to(DEV), freeze_phi=False)
    assert np.allclose(sep_loss.cpu().numpy(), sep, rtol=2e-5), (sep_loss, sep)
    for n, v in oper.variables().items():
        assert torch.allclose(v.cpu(), op.phi[n].detach(), atol=2e-6), n
    scale = float(ref_g.abs().max())
    assert float((gx0.cpu() - ref_g).abs().max()) < 2e-5 * scale + 1e-9


def _random_case(seed):
    """One seeded point of the option space of the operator / conditioning configs (measurements.py, condition_methods.py:63-107,
    utils.py:544-566, :674-700, losses.py): operator, depth type and values, loss function, loss weight and its function, auxiliary
    losses, learn flags, step sizes, inner iterations, image size."""
    r = np.random.RandomState(seed)
    opname = ["underwater_physical_revised", "underwater_physical", "haze_physical"][r.randint(3)]
    depth_type = ["gamma", "original", "move"][r.randint(3)]
    value = {"gamma": f"{r.uniform(1.1, 1.6):.3f},{r.uniform(0.8, 1.5):.3f},{[1, 1, 1.5][r.randint(3)]}", "original": "1.4,1.4,1",
             "move": round(float(r.uniform(1.05, 1.5)), 3)}[depth_type]      # (a YAML number: the reference adds a numpy array to a
    #                                                                   graph tensor when `value` is a string here, which raises)
    f3 = lambda lo, hi: ",".join(f"{v:.3f}" for v in r.uniform(lo, hi, 3))   # noqa: E731
    okw = dict(depth_type=depth_type, value=value, phi_inf=f3(0.1, 0.7), phi_inf_eta=f"{r.choice([1e-3, 1e-4, 5e-3])}",
               phi_inf_learn_flag=bool(r.rand() > 0.2))
    if opname == "underwater_physical_revised":
        okw.update(phi_a=f3(0.7, 1.2), phi_b=f3(0.6, 1.1), phi_a_eta=f"{r.choice([1e-3, 2e-3])}", phi_b_eta=f"{r.choice([1e-3, 5e-4])}",
                   phi_a_learn_flag=bool(r.rand() > 0.2), phi_b_learn_flag=bool(r.rand() > 0.2))
    elif opname == "underwater_physical":
        okw.update(phi_ab=f3(0.7, 1.2), phi_ab_eta=f"{r.choice([1e-3, 2e-3])}", phi_ab_learn_flag=bool(r.rand() > 0.2))
    else:
        okw.update(phi_ab=f"{r.uniform(0.6, 1.3):.3f}", phi_ab_eta=f"{r.choice([1e-3, 2e-3])}", phi_ab_learn_flag=bool(r.rand() > 0.2))
    wtype = ["gamma", "original", "move"][r.randint(3)]
    wfun = {"gamma": f"gamma,{r.uniform(1.1, 1.6):.3f},{r.uniform(0.8, 1.5):.3f},1", "original": "original,0", "move": f"move,{r.uniform(1.05, 1.5):.3f}"}[wtype]
    aux = [None, {"avrg_loss": 0.5, "val_loss": 20}, {"val_loss": 40}, {"avrg_loss": 1.5}][r.randint(4)]
    ckw = dict(loss_function=["norm", "mse"][r.randint(2)], loss_weight=["depth", "none"][int(r.rand() > 0.75)], weight_function=wfun)
    hw = [(24, 24), (16, 40), (32, 20), (8, 8)][r.randint(4)]
    return opname, okw, ckw, aux, int(r.choice([1, 3, 20])), hw


@pytest.mark.parametrize("seed", list(range(24)))
def test_physics_random_configurations_vs_oracle(mods, seed):
    """24 seeded points of the configuration space (every operator x depth type x loss x weight function x auxiliary-loss subset x
    learn flags x inner iterations x non-square sizes): loss, phi after the inner SGD iterations and dL/dx0 of the device path vs
    autograd through the oracle.  The parametrised tests above walk the shipped configurations; this one walks what a user can
    write into the YAML."""
    ops, M, CM = mods
    opname, okw, ckw, aux, n_iter, (H, W) = _random_case(seed)
    g = torch.Generator().manual_seed(100 + seed)
    x0 = (0.6 * torch.randn(1, 4, H, W, generator=g)).clamp(-1.0, 1.0).requires_grad_(True)   # depth >= -1: the gamma bases stay positive
    y = torch.rand(1, 3, H, W, generator=g) * 1.6 - 0.8
    op = D.PhysOperator(opname, batch_size=1, **okw)
    guide = D.OsmosisGuidance(op, n_iter=n_iter, scale="7,7,7,0.9", gradient_clip="False,0", aux=aux, **ckw)
    op.set_requires_grad(True)
    for it in range(n_iter):
        sep, loss = guide.loss(x0, y)
        a = D.aux_loss(x0, guide.aux)
        total = loss if a is None else loss + a
        total.backward(inputs=([x0] if it == n_iter - 1 else []) + list(op.phi.values()))
        op.sgd_step()
    ref_g = x0.grad.clone()
    oper = M.get_operator(opname, device=DEV, batch_size=1, optimizer="sgd", **okw)
    cond = CM.get_conditioning_method("osmosis", oper, M.get_noise("clean"), scale="7,7,7,0.9", gradient_x_prev=True,
                                      gradient_clip="False,0", n_iter=n_iter, aux_loss=aux, pattern="pcgs", **ckw)
    gx0, sep_loss = cond.loss_grad_x0(x0.detach().to(DEV), y.to(DEV), freeze_phi=False)
    tag = (seed, opname, okw["depth_type"], ckw, aux, n_iter, (H, W))
    assert np.allclose(sep_loss.cpu().numpy(), sep, rtol=3e-5), (tag, sep_loss, sep)
    for n, v in oper.variables().items():
        assert torch.allclose(v.cpu(), op.phi[n].detach(), atol=3e-6), (tag, n, v.cpu().flatten(), op.phi[n].detach().flatten())
    scale = float(ref_g.abs().max())
    assert float((gx0.cpu() - ref_g).abs().max()) < 3e-5 * scale + 1e-9, (tag, float((gx0.cpu() - ref_g).abs().max()), scale)


TORCH_OPTIMIZERS = {"adam": torch.optim.Adam, "adamw": torch.optim.AdamW, "adamax": torch.optim.Adamax,
                    "rmsprop": torch.optim.RMSprop, "adagrad": torch.optim.Adagrad, "adadelta": torch.optim.Adadelta,
                    "asgd": torch.optim.ASGD, "rprop": torch.optim.Rprop}


@pytest.mark.parametrize("opname,optimizer", [(o, "adam") for o in OPS] +
                         [("underwater_physical_revised", n) for n in TORCH_OPTIMIZERS if n != "adam"] +
                         [("haze_physical", "rmsprop"), ("underwater_physical", "rprop")])
def test_physics_phi_adam(mods, opname, optimizer):
    """`optimizer: <name>` (utils.py:494-524 -> the torch.optim class with its defaults, one parameter group per phi with
    lr = eta, measurements.py:132-136): 20 inner iterations on device vs that torch optimizer stepping the oracle's parameters,
    twice in a row (the optimizer state -- moments, step count, step sizes -- carries over from one guided step to the next).
    Round 6: every elementwise optimizer of the reference's factory, not Adam alone."""
    ops, M, CM = mods
    okw, ckw = OPS[opname]
    H = W = 24
    g = torch.Generator().manual_seed(12)
    x0 = 0.6 * torch.randn(1, 4, H, W, generator=g)       # B = 1: the reference's norm loss is joint over a batch
    y = torch.rand(1, 3, H, W, generator=g) * 1.6 - 0.8
    eta = {"phi_a": 2e-3, "phi_b": 1e-3, "phi_ab": 2e-3, "phi_inf": 5e-4}
    if optimizer in ("adadelta", "asgd"):      # (lr multiplies a unit-free step / a raw gradient of O(10): keep phi in its physical range)
        eta = {k: v * (50.0 if optimizer == "adadelta" else 0.02) for k, v in eta.items()}
    op = D.PhysOperator(opname, batch_size=1, **{**okw, **{k + "_eta": v for k, v in eta.items()}})
    guide = D.OsmosisGuidance(op, n_iter=20, loss_function="norm", **ckw)
    op.set_requires_grad(True)
    opt = TORCH_OPTIMIZERS[optimizer]([{"params": op.phi[n], "lr": eta[n]} for n in op.names])
    oper = M.get_operator(opname, device=DEV, batch_size=1,
                          **{**okw, **{k + "_eta": v for k, v in eta.items() if k == "phi_inf" or k in okw}}, optimizer=optimizer)
    cond = CM.get_conditioning_method("osmosis", oper, M.get_noise("clean"), loss_function="norm",
                                      loss_weight="depth", weight_function="gamma,1.4,1.4,1",
                                      scale=ckw["scale"], gradient_x_prev=True, gradient_clip=ckw["gradient_clip"],
                                      n_iter=20, aux_loss=ckw["aux"], pattern="pcgs")
    for rnd in range(2):
        xr = (x0 * (1.0 - 0.1 * rnd)).requires_grad_(True)
        for it in range(20):
            sep, loss = guide.loss(xr, y)
            total = loss + D.aux_loss(xr, guide.aux)
            opt.zero_grad()
            total.backward(inputs=([xr] if it == 19 else []) + list(op.phi.values()))
            opt.step()
        gx0, sep_loss = cond.loss_grad_x0(xr.detach().to(DEV), y.to(DEV), freeze_phi=False)
        assert np.allclose(sep_loss.cpu().numpy(), sep, rtol=2e-5), (rnd, sep_loss, sep)
        for n, v in oper.variables().items():
            assert torch.allclose(v.cpu(), op.phi[n].detach(), atol=5e-6), (rnd, n, v.cpu().flatten(), op.phi[n].flatten())
        scale = float(xr.grad.abs().max())
        assert float((gx0.cpu() - xr.grad).abs().max()) < 2e-5 * scale + 1e-9
    moved = float((oper.variables()["phi_inf"].cpu() - torch.tensor([float(u) for u in okw["phi_inf"].split(",")])[None, :, None, None]).abs().max())
    print(opname, optimizer, "phi_inf moved by", moved)
    assert moved > (5e-3 if optimizer == "adam" else 2e-4)          # the optimizer really stepped


@pytest.mark.parametrize("opname", list(OPS))
def test_conditioning_five_tuple_of_the_reference_api(mods, opname):
    """`PosteriorSamplingOsmosis.conditioning(x_prev=, x_t=, x_0_hat=, measurement=, freeze_phi=)` as third-party code calls it
    (condition_methods.py:146-231): x_t updated in place, sep_loss ndarray[B], the variables dict, the unclipped gradient on the
    CPU, and the auxiliary-loss dictionary (losses.py:67-83: un-weighted terms, detached, on the CPU; avrg_loss :29-45 = sum_c
    |mean_hw rgb_c|, val_loss :50-62 = mean(max(|rgb| - 0.7, 0)^2)) -- vs the oracle's conditioning and those two formulas."""
    ops, M, CM = mods
    okw, ckw = OPS[opname]
    g = torch.Generator().manual_seed(13)
    H = W = 24
    xp = 0.7 * torch.randn(1, 4, H, W, generator=g)
    y = torch.rand(1, 3, H, W, generator=g) * 1.6 - 0.8
    mean0 = 0.5 * torch.randn(1, 4, H, W, generator=g)
    a, b = 0.83, 0.21                                                # pred_xstart as a differentiable function of x_prev
    # oracle
    op = D.PhysOperator(opname, batch_size=1, **okw)
    guide = D.OsmosisGuidance(op, n_iter=20, **ckw)
    xr = xp.clone().requires_grad_(True)
    x0r = a * xr + b * torch.tanh(xr)
    xt_ref, sep_ref, var_ref, g_ref = guide.conditioning(xr, mean0.clone(), x0r, y, freeze_phi=False)
    # product
    oper = M.get_operator(opname, device=DEV, batch_size=1, **okw)
    cond = CM.get_conditioning_method("osmosis", oper, M.get_noise("clean"), loss_function="norm", loss_weight="depth",
                                      weight_function="gamma,1.4,1.4,1", scale=ckw["scale"], gradient_x_prev=True,
                                      gradient_clip=ckw["gradient_clip"], n_iter=20, aux_loss=ckw["aux"], pattern="pcgs")
    xd = xp.to(DEV).requires_grad_(True)
    x0d = a * xd + b * torch.tanh(xd)
    x_t = mean0.to(DEV).clone()
    ret = cond.conditioning(x_prev=xd, x_t=x_t, x_0_hat=x0d, measurement=y.to(DEV), noisy_measurement=None, freeze_phi=False,
                            time_index=0.3)
    assert len(ret) == 5 and ret[0] is x_t                                      # in place, and returned
    assert torch.allclose(x_t.cpu(), xt_ref, atol=2e-6)
    assert isinstance(ret[1], np.ndarray) and ret[1].shape == (1,) and np.allclose(ret[1], sep_ref, rtol=1e-5)
    assert set(ret[2]) == set(var_ref)
    for n in var_ref:
        assert ret[2][n].shape == var_ref[n].shape and torch.allclose(ret[2][n].cpu(), var_ref[n], atol=1e-6), n
    assert ret[3].device.type == "cpu" and torch.allclose(ret[3], g_ref, atol=2e-6 * float(g_ref.abs().max()) + 1e-7)
    rgb = x0r.detach()[:, 0:3]
    want = {"avrg_loss": rgb.mean(dim=(2, 3)).abs().sum(), "val_loss": (torch.clamp(rgb.abs() - 0.7, min=0) ** 2).mean()}
    assert set(ret[4]) == set(ckw["aux"])
    for n in ret[4]:
        assert ret[4][n].device.type == "cpu" and ret[4][n].dim() == 0 and abs(float(ret[4][n]) - float(want[n])) < 1e-6 * max(1.0, float(want[n])), n


def test_third_party_auxiliary_loss_takes_the_autograd_step(mods):
    """An auxiliary loss added with `register_loss` (losses.py:13-26) has no kernel slot: the 'osmosis' conditioning method reports
    `hip_ok() == False` (the fused loop is not entered) and takes its torch.autograd step -- through the BUILT-IN operator's autograd API
    (parameter tensors as leaves, `optimize()` applying the caller-produced gradients) -- vs the same step written out on the CPU."""
    ops, M, CM = mods
    from osmosis_diffusion_code_amd.osmosis_utils import losses as L
    if "tv_loss" not in L.__LOSS__:
        @L.register_loss(name="tv_loss")
        class TV(torch.nn.Module):
            def forward(self, x):
                rgb = x[:, 0:3]
                return (rgb[:, :, 1:] - rgb[:, :, :-1]).abs().mean() + (rgb[:, :, :, 1:] - rgb[:, :, :, :-1]).abs().mean()
    tv = L.get_loss("tv_loss")
    opname = "underwater_physical_revised"
    okw, ckw = OPS[opname]
    eta = {k + "_eta": 1e-3 for k in ("phi_a", "phi_b", "phi_inf")}
    g = torch.Generator().manual_seed(17)
    H = W = 24
    xp = (0.6 * torch.randn(1, 4, H, W, generator=g)).clamp(-1, 1)
    y = torch.rand(1, 3, H, W, generator=g) * 1.6 - 0.8
    mean0 = 0.4 * torch.randn(1, 4, H, W, generator=g)
    # CPU: the oracle's data term + 0.3 tv + 20 val, 5 inner iterations of plain descent, then the update of x_t
    op = D.PhysOperator(opname, batch_size=1, **okw, **eta)
    guide = D.OsmosisGuidance(op, n_iter=5, **{**ckw, "aux": {"val_loss": 20}})
    xr = xp.clone().requires_grad_(True)
    x0r = 0.9 * xr + 0.1 * torch.tanh(xr)
    op.set_requires_grad(True)
    for it in range(5):
        sep, loss = guide.loss(x0r, y)
        total = loss + D.aux_loss(x0r, guide.aux) + 0.3 * tv(x0r)
        total.backward(inputs=([xr] if it == 4 else []) + list(op.phi.values()), retain_graph=it < 4)
        op.sgd_step()
    clipv = float(ckw["gradient_clip"].split(",")[1])
    scale = torch.tensor([float(v) for v in ckw["scale"].split(",")])
    xt_ref = mean0 - scale[None, :, None, None] * torch.clamp(xr.grad, -clipv, clipv)
    # device
    oper = M.get_operator(opname, device=DEV, batch_size=1, optimizer="sgd", **okw, **eta)
    cond = CM.get_conditioning_method("osmosis", oper, M.get_noise("clean"), loss_function="norm", loss_weight="depth",
                                      weight_function="gamma,1.4,1.4,1", scale=ckw["scale"], gradient_x_prev=True,
                                      gradient_clip=ckw["gradient_clip"], n_iter=5, aux_loss={"tv_loss": 0.3, "val_loss": 20}, pattern="pcgs")
    assert cond.hip_ok() is False
    xd = xp.to(DEV).requires_grad_(True)
    x0d = 0.9 * xd + 0.1 * torch.tanh(xd)
    x_t = mean0.to(DEV).clone()
    ret = cond.conditioning(x_prev=xd, x_t=x_t, x_0_hat=x0d, measurement=y.to(DEV), freeze_phi=False, time_index=0.2)
    assert torch.allclose(x_t.cpu(), xt_ref, atol=2e-6) and np.allclose(ret[1], sep, rtol=1e-5)
    assert torch.allclose(ret[3], xr.grad, atol=2e-6 * float(xr.grad.abs().max()) + 1e-7)
    for n, v in ret[2].items():
        assert torch.allclose(v.cpu(), op.phi[n].detach(), atol=1e-6), n
        assert float((v.cpu() - torch.tensor([float(u) for u in okw[n].split(",")])[None, :, None, None]).abs().max()) > 1e-5   # phi moved
    assert set(ret[4]) == {"tv_loss", "val_loss"}


@pytest.mark.parametrize("optimizer,freeze", [("sgd", False), ("adam", False), ("sgd", True)])
def test_phys_optimize_equals_the_launch_by_launch_loop(mods, monkeypatch, optimizer, freeze):
    """osm_phys_optimize (the inner phi loop enqueued by ONE C call) issues exactly the launches of the per-launch entry points
    (osm_phys_reduce / finalize / grad, still reachable with OSM_PHYS_PY_LOOP=1): loss, dL/dx0, phi and the Adam state bit-identical."""
    ops, M, CM = mods
    opname = "underwater_physical_revised"
    okw, ckw = OPS[opname]
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 24, 40
    x0 = (0.6 * torch.randn(B, 4, H, W, generator=g)).to(DEV)
    y = (torch.rand(B, 3, H, W, generator=g) * 1.6 - 0.8).to(DEV)
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("OSM_PHYS_PY_LOOP", mode)
        oper = M.get_operator(opname, device=DEV, batch_size=B, **{**okw, "phi_inf_eta": 1e-3, "phi_a_eta": 1e-3, "phi_b_eta": 1e-3},
                              optimizer=optimizer)
        cond = CM.get_conditioning_method("osmosis", oper, M.get_noise("clean"), loss_function="norm", loss_weight="depth",
                                          weight_function="gamma,1.4,1.4,1", scale=ckw["scale"], gradient_x_prev=True,
                                          gradient_clip=ckw["gradient_clip"], n_iter=20, aux_loss=ckw["aux"], pattern="pcgs")
        out = []
        for _ in range(2):      # two guided steps: the optimizer state carries over
            gx0, loss = cond.loss_grad_x0(x0, y, freeze_phi=freeze)
            out += [gx0.clone(), loss.clone(), oper.phi.clone()]
        if cond._opt is not None:
            out.append(cond._opt.clone())
        res[mode] = out
    assert len(res["0"]) == len(res["1"])
    for a, b in zip(res["0"], res["1"]):
        assert torch.equal(a, b)


def test_unsupported_optimizers_are_refused(mods):
    ops, M, CM = mods
    okw, _ = OPS["haze_physical"]
    for name in ("lbfgs", "sparseadam"):        # these cannot step phi in the reference either (no closure / dense gradients)
        with pytest.raises(NotImplementedError):
            M.get_operator("haze_physical", device=DEV, batch_size=1, **okw, optimizer=name)
    with pytest.raises(ValueError):
        M.get_operator("haze_physical", device=DEV, batch_size=1, **okw, optimizer="nonsense")


def test_guide_update_kernel(mods):
    ops, _, _ = mods
    g = torch.Generator().manual_seed(4)
    B, HW = 2, 40
    mean, lv, gg, dxu, nz = (torch.randn(B, 4, HW, generator=g) * s for s in (1, 0.3, 0.01, 0.01, 1))
    coef = torch.tensor([1.7, 0.3, 0, 0, 0, 0, 1.0, 5.0], device=DEV)
    scale4 = torch.tensor([7.0, 7.0, 7.0, 0.9])
    out = torch.empty(B, 4, HW, device=DEV)
    gout = torch.empty(B, 4, HW, device=DEV)
    ops.guide_update(mean.to(DEV), lv.to(DEV), gg.to(DEV), dxu.to(DEV), nz.to(DEV), coef, scale4.to(DEV), 0.005,
                     out, gout, B, HW)
    grad = 1.7 * gg + dxu
    ref = mean - scale4[None, :, None] * grad.clamp(-0.005, 0.005) + torch.exp(0.5 * lv) * nz
    assert torch.allclose(gout.cpu(), grad, atol=1e-7)
    assert torch.allclose(out.cpu(), ref, atol=1e-6)
    coef[6] = 0.0   # t == 0: no noise
    ops.guide_update(mean.to(DEV), lv.to(DEV), gg.to(DEV), dxu.to(DEV), nz.to(DEV), coef, scale4.to(DEV), 0.005,
                     out, None, B, HW)
    assert torch.allclose(out.cpu(), mean - scale4[None, :, None] * grad.clamp(-0.005, 0.005), atol=1e-6)


def test_guide_update_rng_equals_guide_update_with_the_same_draws(mods):
    """osm_guide_update_rng = osm_guide_update fed with what osm_randn draws for the same (seed, image, step): the fused kernel's
    noise is exactly the tensor form's, the step offset and image stride mean what the header says, index 0 adds no noise."""
    ops, _, _ = mods
    g = torch.Generator().manual_seed(14)
    B, HW = 3, 48
    mean, lv, gg, dxu = (torch.randn(B, 4, HW, generator=g).to(DEV) * s for s in (1, 0.3, 0.01, 0.01))
    coef = torch.tensor([1.7, 0.3, 0, 0, 0, 0, 1.0, 5.0], device=DEV)
    scale4 = torch.tensor([7.0, 7.0, 7.0, 0.9], device=DEV)
    step = torch.tensor([41], device=DEV, dtype=torch.int32)
    seed = 0x1234_5678_9abc_def0
    for img0, stride in ((0, 1), (5, 1), (2, 0)):
        nz = torch.empty(B, 4, HW, device=DEV)
        ops.randn(nz, B, 4 * HW, seed, step_const=42, img0=img0, img_stride=stride)
        ref, gref = torch.empty(B, 4, HW, device=DEV), torch.empty(B, 4, HW, device=DEV)
        ops.guide_update(mean, lv, gg, dxu, nz, coef, scale4, 0.005, ref, gref, B, HW)
        out, gout, used = (torch.empty(B, 4, HW, device=DEV) for _ in range(3))
        ops.guide_update_rng(mean, lv, gg, dxu, coef, scale4, 0.005, out, gout, used, B, HW, seed, step, step_offset=1, img0=img0,
                             img_stride=stride)
        assert torch.equal(used, nz) and torch.equal(gout, gref)
        assert torch.allclose(out, ref, atol=1e-6)
        if stride == 0:
            assert torch.equal(used[0], used[1]) and torch.equal(used[1], used[2])
        else:
            assert not torch.equal(used[0], used[1])
    nz2 = torch.empty(B, 4, HW, device=DEV)
    ops.randn(nz2, B, 4 * HW, seed, step=step, img0=0)             # the device counter (41) instead of the constant
    ops.randn(nz, B, 4 * HW, seed, step_const=41, img0=0)
    assert torch.equal(nz, nz2)
    tail = torch.full((13,), 7.0, device=DEV)                        # B = 1, n % 4 != 0: the scalar tail, nothing past n
    ops.randn(tail[:10], 1, 10, seed, step_const=3)
    full = torch.empty(12, device=DEV)
    ops.randn(full, 1, 12, seed, step_const=3)
    assert torch.equal(tail[:10], full[:10]) and float(tail[10:].min()) == 7.0
    coef[6] = 0.0                                                    # index 0 of a chain: no noise (gaussian_diffusion.py:267)
    ops.guide_update_rng(mean, lv, gg, dxu, coef, scale4, 0.005, out, None, used, B, HW, seed, step)
    assert float(used.abs().max()) == 0.0
    assert torch.allclose(out, mean - scale4[None, :, None] * (1.7 * gg + dxu).clamp(-0.005, 0.005), atol=1e-6)


@pytest.mark.parametrize("eta", [0.0, 0.7])
def test_ddim_update_kernel(mods, eta):
    """osm_ddim_update vs DDIM.p_sample's statements (gaussian_diffusion.py:505-528) in fp32 torch, + the guidance subtraction of
    condition_methods.py:247-251, for eta = 0 (what p_sample_loop uses) and eta > 0 (noise term), and at index 0 (no noise)."""
    ops, _, _ = mods
    g = torch.Generator().manual_seed(8)
    B, HW = 2, 40
    x, x0, gg, dxu, nz = (torch.randn(B, 4, HW, generator=g) * s for s in (1, 0.8, 0.01, 0.01, 1))
    ab, abp = 0.37, 0.52
    r0, r1 = float(np.float32(np.sqrt(1.0 / ab))), float(np.float32(np.sqrt(1.0 / ab - 1.0)))
    # coef[0] = d x0 / d x of the MEAN PROCESSOR (the chain coefficient of the guidance gradient), dcoef[4:6] = the sampler's own
    # sqrt_recip / sqrt_recipm1 tables of predict_eps_from_x_start: equal for 'epsilon', different otherwise (here: different)
    c0 = -0.731
    coef = torch.tensor([c0, 123.0, 0, 0, 0, 0, 1.0, 5.0], device=DEV)
    scale4 = torch.tensor([0.6, 0.5, 0.4, 0.0])
    for noise_on in (1.0, 0.0):
        dcoef = torch.tensor([ab, abp, eta, noise_on, r0, r1, 0, 5.0], device=DEV)
        out, gout = torch.empty(B, 4, HW, device=DEV), torch.empty(B, 4, HW, device=DEV)
        xd = x.to(DEV)
        ops.ddim_update(x0.to(DEV), xd, gg.to(DEV), dxu.to(DEV), nz.to(DEV), coef, dcoef, scale4.to(DEV), -1.0, out, gout, B, HW)
        abt, abpt = torch.tensor(ab), torch.tensor(abp)
        eps = (torch.tensor(r0) * x - x0) / torch.tensor(r1)
        sigma = eta * torch.sqrt((1 - abpt) / (1 - abt)) * torch.sqrt(1 - abt / abpt)
        ref = x0 * torch.sqrt(abpt) + torch.sqrt(1 - abpt - sigma ** 2) * eps
        if noise_on:
            ref = ref + sigma * nz
        grad = c0 * gg + dxu
        ref = ref - grad * scale4[None, :, None]
        assert torch.allclose(gout.cpu(), grad, atol=1e-7)
        assert torch.allclose(out.cpu(), ref, atol=2e-6), float((out.cpu() - ref).abs().max())
        ops.ddim_update(x0.to(DEV), xd, gg.to(DEV), dxu.to(DEV), nz.to(DEV), coef, dcoef, scale4.to(DEV), -1.0, xd, None, B, HW)
        assert torch.equal(xd, out)                                  # x_next may alias x


def test_identity_operator_loss_and_gradient(mods):
    """osm_phys_desc.kind 3 (the `noise` / `rgb_guidance` operators of the `ps` path): loss[b] = ||y[b] - x0[b, 0:3]||, g = d loss / d x0
    (zero on depth), per image, vs torch autograd; parameters cannot be stepped."""
    ops, M, CM = mods
    from osmosis_diffusion_code_amd._lib import OsmosisHipError
    g = torch.Generator().manual_seed(2)
    B, H, W = 2, 24, 40
    x0 = torch.randn(B, 4, H, W, generator=g)
    y = torch.rand(B, 3, H, W, generator=g) * 1.6 - 0.8
    y[1] *= 3.0                                                      # very different norms per image
    cond = CM.get_conditioning_method("ps", M.get_operator("rgb_guidance", device=DEV, batch_size=B),
                                      M.get_noise("gaussian", sigma=0.05), scale="0.6,0.5,0.4,0.0")
    assert cond.hip_ok()
    gx, loss = cond.loss_grad_x0(x0.to(DEV), y.to(DEV))
    xr = x0.clone().requires_grad_(True)
    per = torch.stack([torch.linalg.norm(y[b] - xr[b, 0:3]) for b in range(B)])
    per.sum().backward()
    assert torch.allclose(loss.cpu(), per.detach(), rtol=2e-6)
    assert torch.allclose(gx.cpu(), xr.grad, atol=2e-7 + 1e-5 * float(xr.grad.abs().max()))
    assert float(gx[:, 3].abs().max()) == 0.0
    st = cond._states[(B, H * W, DEV)]
    with pytest.raises(OsmosisHipError, match="no parameters"):
        ops.phys_finalize(st["desc"], st["part"], st["red"], st["phi"], True, None)
    assert not CM.get_conditioning_method("ps", M.get_operator("rgb_guidance", device=DEV, batch_size=B),
                                          M.get_noise("poisson", rate=1.0), scale="1").hip_ok()


@pytest.mark.parametrize("t", [999, 500, 3])
def test_posterior_round_trip_at_full_size(mods, t):
    """Size-independent property at BASELINE size (B = 8, 256x256): noising a clean image with q(x_t | x_0) and handing
    the TRUE noise to the posterior kernel must give x_0 back (pred_xstart), and the posterior mean must equal
    coef1 x_0 + coef2 x_t; the learned-range log-variance stays between its two bounds."""
    ops, _, _ = mods
    tb = D.make_tables(1000, "linear", 1000)
    g = torch.Generator(device=DEV).manual_seed(t)
    B, HW = 8, 256 * 256
    x0 = torch.rand(B, 4, HW, device=DEV, generator=g) * 2 - 1
    eps = torch.randn(B, 4, HW, device=DEV, generator=g)
    v = torch.rand(B, 4, HW, device=DEV, generator=g) * 2 - 1
    ab = float(tb.alphas_cumprod[t])
    x_t = np.float32(np.sqrt(ab)) * x0 + np.float32(np.sqrt(1 - ab)) * eps
    coef = torch.tensor([np.float32(tb.sqrt_recip_alphas_cumprod[t]), np.float32(tb.sqrt_recipm1_alphas_cumprod[t]),
                         np.float32(tb.posterior_mean_coef1[t]), np.float32(tb.posterior_mean_coef2[t]),
                         np.float32(tb.posterior_log_variance_clipped[t]), np.float32(tb.log_betas[t]), 1.0, t],
                        dtype=torch.float32, device=DEV)
    px0, mean, lv = (torch.empty(B, 4, HW, device=DEV) for _ in range(3))
    ops.posterior(torch.cat([eps, v], dim=1).contiguous(), x_t, coef, px0, mean, lv, B, HW)
    amp = float(tb.sqrt_recip_alphas_cumprod[t])                    # error amplification of the inversion (158 at t = 999)
    assert float((px0 - x0).abs().max()) < 4e-6 * amp * 5
    want = np.float32(tb.posterior_mean_coef1[t]) * px0 + np.float32(tb.posterior_mean_coef2[t]) * x_t
    assert float((mean - want).abs().max()) < 1e-5
    lo, hi = float(tb.posterior_log_variance_clipped[t]), float(tb.log_betas[t])
    assert float(lv.min()) >= min(lo, hi) - 1e-5 and float(lv.max()) <= max(lo, hi) + 1e-5
