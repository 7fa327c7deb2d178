"""Guided p_sample_loop (fused HIP path) vs the per-step traces captured from the REAL reference
(tests/golden/loop_*.npz, same weights, same x_T, same measurement, same injected noise).
Tolerance: the north-star bar, 1e-3 max-abs on images; observed errors are printed."""
import os

import numpy as np
import pytest
import torch

from oracle import unet_ref as U

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TINY_KW = dict(image_size=256, num_channels=32, num_res_blocks=1, channel_mult="1,2,2",
               attention_resolutions="128,64", num_head_channels=16, num_heads=4,
               learn_sigma=True, use_scale_shift_norm=True, resblock_updown=True,
               pretrain_model="osmosis")
OPERATORS = {
    "underwater_physical_revised": dict(
        operator=dict(optimizer="sgd", depth_type="gamma", value="1.4,1.4,1",
                      phi_a="1.1,0.95,0.95", phi_a_eta="1e-5", phi_a_learn_flag=True,
                      phi_b="0.95, 0.8, 0.8", phi_b_eta="1e-5", phi_b_learn_flag=True,
                      phi_inf="0.14, 0.29, 0.49", phi_inf_eta="1e-5", phi_inf_learn_flag=True),
        cond=dict(loss_function="norm", loss_weight="depth", weight_function="gamma,1.4,1.4,1",
                  scale="7,7,7,0.9", gradient_x_prev=True, gradient_clip="True,0.005"),
        aux=dict(aux_loss={"avrg_loss": 0.5, "val_loss": 20})),
    "underwater_physical": dict(
        operator=dict(optimizer="sgd", depth_type="original", value="1.4,1.4,1",
                      phi_ab="1.1,0.95,0.95", phi_ab_eta="1e-5", phi_ab_learn_flag=True,
                      phi_inf="0.2,0.4,0.7", phi_inf_eta="1e-5", phi_inf_learn_flag=True),
        cond=dict(loss_function="norm", loss_weight="depth", weight_function="gamma,1.4,1.4,1",
                  scale="4,4,4,1", gradient_x_prev=True, gradient_clip="True,0.001"),
        aux=dict(aux_loss={"val_loss": 40})),
    "haze_physical": dict(
        operator=dict(optimizer="sgd", depth_type="gamma", value="1.4,1.4,1",
                      phi_ab="1.0", phi_ab_eta="1e-5", phi_ab_learn_flag=True,
                      phi_inf="0.14, 0.29, 0.49", phi_inf_eta="1e-5", phi_inf_learn_flag=True),
        cond=dict(loss_function="norm", loss_weight="depth", weight_function="gamma,1.4,1.4,1",
                  scale="7,7,7,0.9", gradient_x_prev=True, gradient_clip="True,0.005"),
        aux=dict(aux_loss={"avrg_loss": 0.5, "val_loss": 20})),
}
PATTERN = dict(pattern="pcgs", update_start=0.7, update_end=0, global_N=1, local_M=1, s_start=1, s_end=0,
               n_iter=20, start_guidance=1, stop_guidance=0)


@pytest.fixture(scope="module")
def pkg():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from osmosis_diffusion_code_amd.guided_diffusion import unet, gaussian_diffusion, measurements, condition_methods
    return unet, gaussian_diffusion, measurements, condition_methods


def make_model(unet):
    cfg = U.UNetConfig.from_create_model_kwargs(**TINY_KW)
    sd = U.seeded_state_dict(cfg, 1234)
    m = unet.create_model(**TINY_KW)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval()


def make_sampler(gd):
    return gd.get_sampler("ddpm")(use_timesteps=range(0, 100, 10),
                                  betas=gd.get_named_beta_schedule("linear", 1000),
                                  model_mean_type="epsilon", model_var_type="learned_range",
                                  dynamic_threshold=False, clip_denoised=False, rescale_timesteps=False)


@pytest.mark.parametrize("conv_mode", ["f32", "bf16x6", "f16x3"])
@pytest.mark.parametrize("opname", list(OPERATORS))
def test_fused_loop_matches_reference_trace(pkg, opname, conv_mode):
    unet, gd, M, CM = pkg
    g = dict(np.load(os.path.join(GOLD, f"loop_{opname}.npz")))
    spec = OPERATORS[opname]
    model = make_model(unet)
    model.conv_mode = conv_mode
    operator = M.get_operator(opname, device=DEV, batch_size=1, **spec["operator"])
    cond = CM.get_conditioning_method("osmosis", operator, M.get_noise("clean"), **spec["cond"], **PATTERN,
                                      **spec["aux"])
    sampler = make_sampler(gd)
    assert sampler.timestep_map == list(g["timestep_map"])
    noise = torch.from_numpy(g["noise"]).to(DEV)
    trace = []
    img, variables, loss, x0 = sampler.p_sample_loop(
        model=model, x_start=torch.from_numpy(g["x_T"]).to(DEV), measurement=torch.from_numpy(g["y"]).to(DEV),
        measurement_cond_fn=cond.conditioning, record=False, save_root=None, pretrain_model="osmosis",
        rgb_guidance=False, sample_pattern=PATTERN, noise_fn=lambda k, shape: noise[k], trace=trace)
    assert len(trace) == 10
    worst = {}
    for k, rec in enumerate(trace):
        for key, gk in (("x_in", "trace.x_in"), ("x0", "trace.x0"), ("mean", "trace.mean"), ("grad", "trace.grad")):
            e = float((rec[key].cpu() - torch.from_numpy(g[gk][k])).abs().max())
            worst[key] = max(worst.get(key, 0.0), e)
        assert np.allclose(rec["loss"].cpu().numpy(), g["trace.loss"][k], rtol=1e-4), (k, rec["loss"], g["trace.loss"][k])
    print(opname, conv_mode, "worst max-abs errors over 10 free-running steps:", worst)
    assert worst["x_in"] < 1e-3 and worst["x0"] < 1e-3 and worst["mean"] < 1e-3
    assert worst["grad"] < 1e-3 * max(1.0, float(np.abs(g["trace.grad"]).max()))
    assert float((img.cpu() - torch.from_numpy(g["final_img"])).abs().max()) < 1e-3
    assert float((x0 - torch.from_numpy(g["final_x0"])).abs().max()) < 1e-3
    assert np.allclose(loss, g["final_loss"], rtol=1e-4)
    for n, v in variables.items():
        assert torch.allclose(v.cpu(), torch.from_numpy(g[f"final.{n}"]), atol=2e-6), n


@pytest.mark.parametrize("optimizer", ["adam", "adamw", "adamax", "rmsprop", "adagrad", "adadelta", "asgd", "rprop"])
def test_fused_loop_with_every_phi_optimizer_matches_the_reference(pkg, optimizer):
    """Round 6: `optimizer: <name>` of the operator config (utils.py:494-524) through the fused loop vs the REAL reference's guided
    loop with that torch optimizer (tests/golden/loop_optimizers.npz: 10 steps, 7 of them with 20 inner iterations; phi moves by
    1e-2 ... 3e-1): per-step loss and phi, final image and pred_xstart."""
    unet, gd, M, CM = pkg
    g = np.load(os.path.join(GOLD, "loop_optimizers.npz"))
    spec = OPERATORS["underwater_physical_revised"]
    eta = repr(float(g[f"{optimizer}.eta"]))
    model = make_model(unet)
    operator = M.get_operator("underwater_physical_revised", device=DEV, batch_size=1,
                              **{**spec["operator"], "optimizer": optimizer, "phi_a_eta": eta, "phi_b_eta": eta, "phi_inf_eta": eta})
    cond = CM.get_conditioning_method("osmosis", operator, M.get_noise("clean"), **spec["cond"], **PATTERN, **spec["aux"])
    noise = torch.from_numpy(g["noise"]).to(DEV)
    trace = []
    img, variables, loss, x0 = make_sampler(gd).p_sample_loop(
        model=model, x_start=torch.from_numpy(g["x_T"]).to(DEV), measurement=torch.from_numpy(g["y"]).to(DEV),
        measurement_cond_fn=cond.conditioning, record=False, save_root=None, pretrain_model="osmosis",
        rgb_guidance=False, sample_pattern=PATTERN, noise_fn=lambda k, shape: noise[k], trace=trace)
    moved = max(float(np.abs(g[f"{optimizer}.{n}"][-1] - g[f"{optimizer}.{n}"][0]).max()) for n in ("phi_a", "phi_b", "phi_inf"))
    e_loss = max(abs(float(rec["loss"][0]) - float(g[f"{optimizer}.loss"][k].reshape(-1)[0])) / float(g[f"{optimizer}.loss"][k].reshape(-1)[0])
                 for k, rec in enumerate(trace))
    e_phi = 0.0
    for k, rec in enumerate(trace):
        ph = rec["phi"].cpu().reshape(9)
        ref = np.concatenate([g[f"{optimizer}.{n}"][k].reshape(-1) for n in ("phi_a", "phi_b", "phi_inf")])
        e_phi = max(e_phi, float(np.abs(ph.numpy() - ref).max()))
    e_img = float((img.cpu() - torch.from_numpy(g[f"{optimizer}.final_img"])).abs().max())
    e_x0 = float((x0 - torch.from_numpy(g[f"{optimizer}.final_x0"])).abs().max())
    print(f"{optimizer}: phi moved {moved:.3f}; errors vs the reference: loss(rel) {e_loss:.1e} phi {e_phi:.1e} img {e_img:.1e} x0 {e_x0:.1e}")
    assert moved > 5e-3
    # measured (round 6): loss 1.0e-7 ... 2.1e-7, phi 0 ... 3.6e-7, image / x0 7.7e-7 ... 9.8e-7; bars at ~5x.  Adadelta's first updates
    # are ~ sqrt(eps) * grad / |grad| (sign-like wherever a gradient component is small): the REFERENCE's own phi moves by 7.6e-6 when
    # x_T is perturbed by 1e-6 (`<optimizer>.phi_drift_1e-6`, <= 6e-8 for the other seven), and a one-ulp change of pred_xstart moved
    # this comparison from 3.6e-7 to 2.4e-5 -- its phi bar is 10 x that drift
    phi_bar = max(2e-6, 10.0 * float(g[f"{optimizer}.phi_drift_1e-6"]))
    assert e_loss < 1e-6 and e_phi < phi_bar and e_img < 5e-6 and e_x0 < 5e-6


def test_reference_api_generic_path_matches_fused(pkg):
    """The reference call pattern (model(x,t) -> p_mean_variance -> conditioning(...)) through
    torch.autograd gives the same step as the fused path."""
    unet, gd, M, CM = pkg
    opname = "underwater_physical_revised"
    g = dict(np.load(os.path.join(GOLD, f"loop_{opname}.npz")))
    spec = OPERATORS[opname]
    model = make_model(unet)
    operator = M.get_operator(opname, device=DEV, batch_size=1, **spec["operator"])
    cond = CM.get_conditioning_method("osmosis", operator, M.get_noise("clean"), **spec["cond"], **PATTERN,
                                      **spec["aux"])
    sampler = make_sampler(gd)
    x = torch.from_numpy(g["trace.x_in"][0]).to(DEV).requires_grad_(True)
    t = torch.tensor([9], device=DEV)
    out = sampler.p_mean_variance(model=model, x=x, t=t)
    assert float((out["pred_xstart"].detach().cpu() - torch.from_numpy(g["trace.x0"][0])).abs().max()) < 1e-4
    x_t, loss, variables, grads, aux = cond.conditioning(x_prev=x, x_t=out["mean"], x_0_hat=out["pred_xstart"],
                                                         measurement=torch.from_numpy(g["y"]).to(DEV),
                                                         freeze_phi=True, time_index=0.9)  # idx 9 > 0.7*T
    assert float((grads - torch.from_numpy(g["trace.grad"][0])).abs().max()) < 1e-4 * max(1.0, float(np.abs(g["trace.grad"][0]).max()))
    assert float((x_t.detach().cpu() - torch.from_numpy(g["trace.x_guided"][0])).abs().max()) < 1e-4
    assert np.allclose(loss, g["trace.loss"][0], rtol=1e-4)


@pytest.mark.parametrize("opname,optimizer", [("underwater_physical_revised", "sgd"), ("haze_physical", "GD"),
                                              ("underwater_physical", "sgd"), ("underwater_physical_revised", "adam")])
def test_third_party_conditioner_on_the_reference_api(pkg, opname, optimizer):
    """A conditioning method written the way the reference's is (condition_methods.py:146-231) -- registered through
    `register_conditioning_method`, using ONLY the reference-level API of the package's objects under torch.autograd: `model(x, t)`
    differentiable w.r.t. x, `operator.forward` differentiable w.r.t. its input AND its parameter tensors,
    `operator.set_variable_gradients` / `get_variable_list` / `optimize`, `utils.set_loss_weight`, `AuxiliaryLoss.forward` -- runs the
    whole 10-step chain through `_generic_loop` and lands on the REAL reference's trace (loop_<operator>.npz; Adam: loop_optimizers.npz)."""
    unet, gd, M, CM = pkg
    from osmosis_diffusion_code_amd.osmosis_utils import losses as L
    from osmosis_diffusion_code_amd.osmosis_utils import utils as OU
    name = f"third_party_{opname}_{optimizer}"

    @CM.register_conditioning_method(name=name)
    class ThirdParty(CM.ConditioningMethod):
        def __init__(self, operator, noiser, **kw):
            super().__init__(operator, noiser)
            self.scale = torch.tensor([float(v) for v in kw["scale"].split(",")])
            self.clip = float(kw["gradient_clip"].split(",")[1])
            self.n_iter, self.weight_function = kw["n_iter"], kw["weight_function"]
            self.aux = L.AuxiliaryLoss(kw["aux_loss"])
            self.calls = 0

        def conditioning(self, x_prev, x_t, x_0_hat, measurement, **kw):
            self.calls += 1
            freeze = kw.get("freeze_phi", False)
            self.operator.set_variable_gradients(value=not freeze)
            phis = self.operator.get_variable_list()
            n = 1 if freeze else self.n_iter
            for it in range(n):
                image = self.operator.forward(x_0_hat)
                w = OU.set_loss_weight("depth", self.weight_function, degraded_image=image.detach(), x_0_hat=x_0_hat.detach())
                diff = (measurement - (2 * image - 1)) * w
                loss = torch.linalg.norm(diff)
                total = loss + self.aux.forward(x_0_hat)[0]
                last = it == n - 1
                total.backward(inputs=([x_prev] if last else []) + ([] if freeze else phis), retain_graph=not last)
                variables = self.operator.optimize(freeze_phi=freeze)
            with torch.no_grad():
                x_t -= self.scale[None, :, None, None].to(x_t.device) * torch.clamp(x_prev.grad, -self.clip, self.clip)
            return x_t, np.array([float(loss.detach())]), variables, x_prev.grad.cpu(), None

    spec = OPERATORS[opname]
    if optimizer == "adam":
        g = np.load(os.path.join(GOLD, "loop_optimizers.npz"))
        eta = repr(float(g["adam.eta"]))
        okw = {**spec["operator"], "optimizer": "adam", "phi_a_eta": eta, "phi_b_eta": eta, "phi_inf_eta": eta}
        want_img, want_loss = g["adam.final_img"], g["adam.loss"].reshape(10)
        want_phi = {n: g[f"adam.{n}"][-1] for n in ("phi_a", "phi_b", "phi_inf")}
    else:
        g = np.load(os.path.join(GOLD, f"loop_{opname}.npz"))
        okw = {**spec["operator"], "optimizer": optimizer}
        want_img, want_loss = g["final_img"], g["trace.loss"].reshape(10)
        want_phi = {k[len("final."):]: g[k] for k in g.files if k.startswith("final.phi")}
    model = make_model(unet)
    operator = M.get_operator(opname, device=DEV, batch_size=1, **okw)
    cond = CM.get_conditioning_method(name, operator, M.get_noise("clean"), **spec["cond"], **PATTERN, **spec["aux"])
    sampler = make_sampler(gd)
    noise = iter(torch.from_numpy(g["noise"]).to(DEV))
    losses = []
    orig = cond.conditioning

    def traced(**kw):
        ret = orig(**kw)
        losses.append(float(ret[1][0]))
        return ret
    real_randn_like = torch.randn_like
    torch.randn_like = lambda t, **kw: next(noise) if t.shape[1] == 4 else real_randn_like(t, **kw)      # the reference's used draws
    try:
        img, variables, loss, x0 = sampler.p_sample_loop(
            model=model, x_start=torch.from_numpy(g["x_T"]).to(DEV).requires_grad_(), measurement=torch.from_numpy(g["y"]).to(DEV),
            measurement_cond_fn=traced, record=False, save_root=None, pretrain_model="osmosis", rgb_guidance=False, sample_pattern=PATTERN)
    finally:
        torch.randn_like = real_randn_like
    assert cond.calls == 10                                              # the generic loop drove the third-party method
    assert np.allclose(losses, want_loss, rtol=1e-4), (losses, want_loss)
    e_img = float((img.detach().cpu() - torch.from_numpy(want_img)).abs().max())
    print(f"third-party conditioner, {opname} / {optimizer}: final image error {e_img:.1e}")
    assert e_img < 1e-4
    for n, v in variables.items():
        assert torch.allclose(v.cpu(), torch.from_numpy(want_phi[n]).reshape(v.shape), atol=5e-6), (n, v.flatten(), want_phi[n].flatten())
        assert float((v.cpu() - operator.variables()[n].cpu()).abs().max()) == 0.0


def test_third_party_operator_with_the_builtin_conditioner(pkg):
    """The other half of the registry contract: an operator the kernels do not know -- registered with `register_operator`, a plain
    torch object with the reference's `LearnableOperator` methods (here: the revised underwater model again, so that the reference's
    trace is the answer) -- under the package's OWN 'osmosis' conditioning method: no `fill_desc`, so `p_sample_loop` takes
    `_generic_loop` and the conditioner its torch.autograd restatement of condition_methods.py:109-231."""
    unet, gd, M, CM = pkg
    from osmosis_diffusion_code_amd.osmosis_utils import utils as OU
    name = "third_party_revised_model"
    if name not in M.__OPERATOR__:
        @M.register_operator(name=name)
        class Mine(M.LearnableOperator):
            def __init__(self, device, batch_size=1, **kw):
                vec = lambda s: torch.tensor([float(v) for v in s.split(",")], device=device).repeat(batch_size, 1)[..., None, None]   # noqa: E731
                self.p = {n: vec(kw[n]) for n in ("phi_a", "phi_b", "phi_inf")}
                self.eta = {n: float(kw[n + "_eta"]) for n in self.p}
                self.value = OU.get_depth_value(kw["value"])

            def forward(self, data, **kw):
                d = OU.convert_depth(depth=data[:, -1:], depth_type="gamma", value=self.value)
                return 0.5 * (data[:, :-1] + 1) * torch.exp(-self.p["phi_a"] * d) + self.p["phi_inf"] * (1 - torch.exp(-self.p["phi_b"] * d))

            def set_variable_gradients(self, value=None, **kw):
                for v in self.p.values():
                    v.requires_grad_(bool(value))

            def get_variable_list(self, **kw):
                return list(self.p.values())

            def optimize(self, freeze_phi=False, **kw):
                if not freeze_phi:
                    with torch.no_grad():
                        for n, v in self.p.items():
                            v.add_(v.grad, alpha=-self.eta[n])
                            v.grad.zero_()
                return {n: v.detach() for n, v in self.p.items()}
    opname = "underwater_physical_revised"
    g = np.load(os.path.join(GOLD, f"loop_{opname}.npz"))
    spec = OPERATORS[opname]
    model = make_model(unet)
    operator = M.get_operator(name, device=DEV, batch_size=1, **spec["operator"])
    assert not hasattr(operator, "fill_desc")
    cond = CM.get_conditioning_method("osmosis", operator, M.get_noise("clean"), **spec["cond"], **PATTERN, **spec["aux"])
    noise = iter(torch.from_numpy(g["noise"]).to(DEV))
    losses, aux_seen = [], []
    orig = cond.conditioning

    def traced(**kw):
        ret = orig(**kw)
        losses.append(float(ret[1][0]))
        aux_seen.append(ret[4])
        return ret
    real_randn_like = torch.randn_like
    torch.randn_like = lambda t, **kw: next(noise) if t.shape[1] == 4 else real_randn_like(t, **kw)
    try:
        img, variables, loss, x0 = make_sampler(gd).p_sample_loop(
            model=model, x_start=torch.from_numpy(g["x_T"]).to(DEV).requires_grad_(), measurement=torch.from_numpy(g["y"]).to(DEV),
            measurement_cond_fn=traced, record=False, save_root=None, pretrain_model="osmosis", rgb_guidance=False, sample_pattern=PATTERN)
    finally:
        torch.randn_like = real_randn_like
    assert len(losses) == 10 and np.allclose(losses, g["trace.loss"].reshape(10), rtol=1e-4)
    assert set(aux_seen[-1]) == {"avrg_loss", "val_loss"}
    e_img = float((img.detach().cpu() - torch.from_numpy(g["final_img"])).abs().max())
    print(f"third-party operator under the built-in conditioner: final image error {e_img:.1e}")
    assert e_img < 1e-4 and float((x0 - torch.from_numpy(g["final_x0"])).abs().max()) < 1e-4
    for n, v in variables.items():
        assert torch.allclose(v.cpu(), torch.from_numpy(g[f"final.{n}"]), atol=5e-6), n


def test_third_party_processors_and_sampler_through_the_registries(pkg):
    """`register_mean_processor` / `register_var_processor` / `register_sampler` additions (the decorators of
    posterior_mean_variance.py:15-28, :146-159 and gaussian_diffusion.py:24-35): processors the kernels do not know (no `hip_kernel`: plain
    torch `get_mean_and_xstart` / `get_variance`, here the epsilon / learned-range formulas again so that the reference's trace is the
    answer) in a sampler class of the user's own send `p_sample_loop` through `_generic_loop` + `p_mean_variance`, with the package's
    'osmosis' conditioner and its kernels for the guidance part."""
    unet, gd, M, CM = pkg
    from osmosis_diffusion_code_amd.guided_diffusion import posterior_mean_variance as PMV
    if "my_eps" not in PMV.__MODEL_MEAN_PROCESSOR__:
        @PMV.register_mean_processor(name="my_eps")
        class MyMean(PMV.MeanProcessor):
            def __init__(self, betas, dynamic_threshold, clip_denoised):
                super().__init__(betas, dynamic_threshold, clip_denoised)
                ac = np.cumprod(1.0 - betas)
                self.a, self.b = np.sqrt(1.0 / ac), np.sqrt(1.0 / ac - 1)

            def get_mean_and_xstart(self, x, t, model_output):
                x0 = PMV.extract_and_expand(self.a, t, x) * x - PMV.extract_and_expand(self.b, t, x) * model_output
                return self.q_posterior_mean(x0, x, t), x0

        @PMV.register_var_processor(name="my_range")
        class MyVar(PMV.VarianceProcessor):
            def __init__(self, betas):
                pv = betas * (1.0 - np.append(1.0, np.cumprod(1.0 - betas)[:-1])) / (1.0 - np.cumprod(1.0 - betas))
                self.lo, self.hi = np.log(np.append(pv[1], pv[1:])), np.log(betas)

            def get_variance(self, x, t):
                f = (x + 1.0) / 2.0
                lv = f * PMV.extract_and_expand(self.hi, t, x) + (1 - f) * PMV.extract_and_expand(self.lo, t, x)
                return torch.exp(lv), lv

        @gd.register_sampler(name="my_ddpm")
        class MySampler(gd.SpacedDiffusion):
            def p_sample(self, model, x, t):
                out = self.p_mean_variance(model, x, t)
                return {"sample": out["mean"], "pred_xstart": out["pred_xstart"]}
    with pytest.raises(NameError):
        PMV.register_mean_processor(name="my_eps")(object)                 # duplicate names raise, as in the reference
    opname = "underwater_physical_revised"
    g = np.load(os.path.join(GOLD, f"loop_{opname}.npz"))
    spec = OPERATORS[opname]
    model = make_model(unet)
    operator = M.get_operator(opname, device=DEV, batch_size=1, **spec["operator"])
    cond = CM.get_conditioning_method("osmosis", operator, M.get_noise("clean"), **spec["cond"], **PATTERN, **spec["aux"])
    sampler = gd.get_sampler("my_ddpm")(use_timesteps=range(0, 100, 10), betas=gd.get_named_beta_schedule("linear", 1000),
                                        model_mean_type="my_eps", model_var_type="my_range", dynamic_threshold=False,
                                        clip_denoised=False, rescale_timesteps=False)
    assert sampler.mean_processor.hip_kernel is None and sampler._fast_path_ok(model, cond.conditioning, "osmosis", False, PATTERN) is None
    noise = iter(torch.from_numpy(g["noise"]).to(DEV))
    real_randn_like = torch.randn_like
    torch.randn_like = lambda t, **kw: next(noise) if t.shape[1] == 4 else real_randn_like(t, **kw)
    try:
        img, variables, loss, x0 = sampler.p_sample_loop(
            model=model, x_start=torch.from_numpy(g["x_T"]).to(DEV).requires_grad_(), measurement=torch.from_numpy(g["y"]).to(DEV),
            measurement_cond_fn=cond.conditioning, record=False, save_root=None, pretrain_model="osmosis", rgb_guidance=False,
            sample_pattern=PATTERN)
    finally:
        torch.randn_like = real_randn_like
    e_img = float((img.detach().cpu() - torch.from_numpy(g["final_img"])).abs().max())
    print(f"third-party processors + sampler: final image error {e_img:.1e}")
    assert e_img < 1e-4 and np.allclose(loss, g["final_loss"], rtol=1e-4)
    for n, v in variables.items():
        assert torch.allclose(v.cpu(), torch.from_numpy(g[f"final.{n}"]), atol=5e-6), n


def test_batched_images_equal_single_image_runs(pkg):
    """B=2 (two different images) == two B=1 runs (per-image reductions)."""
    unet, gd, M, CM = pkg
    opname = "underwater_physical_revised"
    spec = OPERATORS[opname]
    model = make_model(unet)
    gen = torch.Generator().manual_seed(21)
    xT = 0.5 * torch.randn(2, 4, 32, 32, generator=gen)
    y = torch.rand(2, 3, 32, 32, generator=gen) * 1.6 - 0.8
    nz = torch.randn(10, 2, 4, 32, 32, generator=gen).to(DEV)

    def run(sl):
        B = sl.stop - sl.start
        operator = M.get_operator(opname, device=DEV, batch_size=B, **spec["operator"])
        cond = CM.get_conditioning_method("osmosis", operator, M.get_noise("clean"), **spec["cond"], **PATTERN,
                                          **spec["aux"])
        return make_sampler(gd).p_sample_loop(
            model=model, x_start=xT[sl].to(DEV), measurement=y[sl].to(DEV), measurement_cond_fn=cond.conditioning,
            record=False, save_root=None, pretrain_model="osmosis", rgb_guidance=False, sample_pattern=PATTERN,
            noise_fn=lambda k, shape: nz[k, sl])

    both = run(slice(0, 2))
    for i in range(2):
        one = run(slice(i, i + 1))
        assert float((both[0][i:i + 1] - one[0]).abs().max()) < 1e-5
        assert np.allclose(both[2][i], one[2][0], rtol=1e-5)
        for n in both[1]:
            assert torch.allclose(both[1][n][i:i + 1], one[1][n], atol=1e-6)


@pytest.mark.parametrize("optimizer", ["sgd", "adam"])
def test_ragged_chunks_equal_whole_batch(pkg, monkeypatch, optimizer):
    """A batch that does not fit at once is walked in chunks of (at most) two sizes through two engines
    (GaussianDiffusion.chunk_sizes: B = 5 with room for 4 -> [2, 2, 1]); images are independent chains, so every image
    must come out as in the one-pass run -- including the Adam state of its phi rows, which lives with the operator and
    not with the chunk."""
    unet, gd, M, CM = pkg
    opname = "underwater_physical_revised"
    spec = OPERATORS[opname]
    model = make_model(unet)
    gen = torch.Generator().manual_seed(33)
    B = 5
    xT = 0.5 * torch.randn(B, 4, 32, 32, generator=gen)
    y = torch.rand(B, 3, 32, 32, generator=gen) * 1.6 - 0.8
    nz = torch.randn(10, B, 4, 32, 32, generator=gen).to(DEV)

    def run():
        operator = M.get_operator(opname, device=DEV, batch_size=B, **dict(spec["operator"], optimizer=optimizer))
        cond = CM.get_conditioning_method("osmosis", operator, M.get_noise("clean"), **spec["cond"], **PATTERN,
                                          **spec["aux"])
        return make_sampler(gd).p_sample_loop(
            model=model, x_start=xT.to(DEV), measurement=y.to(DEV), measurement_cond_fn=cond.conditioning,
            record=False, save_root=None, pretrain_model="osmosis", rgb_guidance=False, sample_pattern=PATTERN,
            noise_fn=lambda k, shape: nz[k])

    whole = run()
    assert sorted(e.B for e in model._engines.values()) == [5]
    monkeypatch.setenv("OSM_MAX_BATCH", "4")
    assert gd.GaussianDiffusion.chunk_sizes(5, 4) == [2, 2, 1]
    ragged = run()
    assert sorted(e.B for e in model._engines.values()) == [1, 2]          # two live engines, both within the cap
    assert float((whole[0] - ragged[0]).abs().max()) < 1e-5
    assert np.allclose(whole[2], ragged[2], rtol=1e-5)
    for n in whole[1]:
        assert torch.allclose(whole[1][n], ragged[1][n], atol=1e-6), n


@pytest.mark.parametrize("name", ["ddpm", "ddim"])
def test_rgb_guidance_ps_chain_matches_reference(pkg, name, monkeypatch):
    """`ps` conditioning + `rgb_guidance` operator + gaussian noiser through DDPM.p_sample and
    DDIM.p_sample (generic autograd path over the HIP UNet) vs the chain the reference produced with the same draws
    (tests/golden/loop_ps.npz; torch.randn_like replayed in the reference's call order)."""
    unet, gd, M, CM = pkg
    g = np.load(os.path.join(GOLD, "loop_ps.npz"))
    model = make_model(unet)
    operator = M.get_operator("rgb_guidance", device=DEV, batch_size=1)
    cond = CM.get_conditioning_method("ps", operator, M.get_noise("gaussian", sigma=0.05), scale="0.6,0.5,0.4,0.0")
    sampler = gd.get_sampler(name)(use_timesteps=range(0, 100, 10), betas=gd.get_named_beta_schedule("linear", 1000),
                                   model_mean_type="epsilon", model_var_type="learned_range", dynamic_threshold=False,
                                   clip_denoised=False, rescale_timesteps=False)
    is_x = list(g[f"{name}.draw_is_x"])
    dx, dy = iter(g[f"{name}.draws_x"]), iter(g[f"{name}.draws_y"])
    order = iter(is_x)

    def replay(t, **kw):
        want_x = next(order)
        d = torch.from_numpy(next(dx) if want_x else next(dy))
        assert tuple(d.shape) == tuple(t.shape), "draw order differs from the reference's"
        return d.to(t.device)

    monkeypatch.setattr(torch, "randn_like", replay)
    losses = []
    orig = cond.conditioning

    def traced(**kw):
        ret = orig(**kw)
        losses.append(float(ret[1]))
        return ret

    img = sampler.p_sample_loop(model=model, x_start=torch.from_numpy(g[f"{name}.x_T"]).to(DEV).requires_grad_(),
                                measurement=torch.from_numpy(g[f"{name}.y"]).to(DEV), measurement_cond_fn=traced,
                                record=False, save_root=None, pretrain_model="osmosis", rgb_guidance=True,
                                sample_pattern=PATTERN)
    monkeypatch.undo()
    assert next(order, None) is None                      # every reference draw was consumed, in order
    assert np.allclose(losses, g[f"{name}.loss"], rtol=1e-4)
    err = float((img.detach().cpu() - torch.from_numpy(g[f"{name}.final_img"])).abs().max())
    print(name, "rgb-guidance chain max-abs error", err)
    assert err < 1e-3


def _ps_setup(pkg, name):
    unet, gd, M, CM = pkg
    g = np.load(os.path.join(GOLD, "loop_ps.npz"))
    model = make_model(unet)
    operator = M.get_operator("rgb_guidance", device=DEV, batch_size=1)
    cond = CM.get_conditioning_method("ps", operator, M.get_noise("gaussian", sigma=0.05), scale="0.6,0.5,0.4,0.0")
    sampler = gd.get_sampler(name)(use_timesteps=range(0, 100, 10), betas=gd.get_named_beta_schedule("linear", 1000),
                                   model_mean_type="epsilon", model_var_type="learned_range", dynamic_threshold=False,
                                   clip_denoised=False, rescale_timesteps=False)
    return g, model, cond, sampler


@pytest.mark.parametrize("name", ["ddpm", "ddim"])
def test_rgb_guidance_ps_chain_fused_matches_reference(pkg, name, monkeypatch):
    """SURVEY a22 on the FUSED kernels (round 6): the same reference chain as above through `_fused_loop` -- osm_phys_* with
    the identity operator (kind 3) for ||y - x0[:, 0:3]|| and its gradient, osm_guide_update (DDPM) / osm_ddim_update (DDIM)
    for the step; the generic autograd loop must not be entered.  Noise: the reference's x-shaped draws, injected."""
    unet, gd, M, CM = pkg
    g, model, cond, sampler = _ps_setup(pkg, name)

    def no_generic(*a, **k):
        raise AssertionError("the rgb-guidance chain fell back to the generic loop")
    monkeypatch.setattr(type(sampler), "_generic_loop", no_generic)
    draws = torch.from_numpy(g[f"{name}.draws_x"]).to(DEV)
    trace = []
    img = sampler.p_sample_loop(model=model, x_start=torch.from_numpy(g[f"{name}.x_T"]).to(DEV),
                                measurement=torch.from_numpy(g[f"{name}.y"]).to(DEV), measurement_cond_fn=cond.conditioning,
                                record=False, save_root=None, pretrain_model="osmosis", rgb_guidance=True,
                                sample_pattern=PATTERN, noise_fn=lambda k, shape: draws[k], trace=trace)
    assert isinstance(img, torch.Tensor) and len(trace) == 10
    losses = [float(r["loss"][0]) for r in trace]
    assert np.allclose(losses, g[f"{name}.loss"], rtol=1e-4), (losses, g[f"{name}.loss"])
    err = float((img.detach().cpu() - torch.from_numpy(g[f"{name}.final_img"])).abs().max())
    print(name, "fused rgb-guidance chain max-abs error", err)
    assert err < 2e-5


def _no_generic(monkeypatch, sampler):
    def no_generic(*a, **k):
        raise AssertionError("the chain fell back to the generic loop")
    monkeypatch.setattr(type(sampler), "_generic_loop", no_generic)


def _free_running_bar(drift):
    """Bar on a free-running 10-step chain from the REFERENCE's own sensitivity (`<tag>.drift_1e-6` of loop_processors.npz: how far its
    final image moves when x_T is perturbed by 1e-6 N(0,1)): well-conditioned chains are held tight, mildly amplifying ones to the
    north-star 1e-3, and chains the reference itself cannot reproduce to 1e-3 are compared teacher-forced instead (None)."""
    if drift <= 1e-4:
        return max(2e-5, 10.0 * drift)
    return 1e-3 if drift <= 1e-3 else None


@pytest.mark.parametrize("mean_type,var_type", [("start_x", "fixed_small"), ("epsilon", "fixed_large"), ("epsilon", "learned"),
                                                ("start_x", "learned")])
def test_fused_loop_with_the_other_processors_matches_the_reference(pkg, monkeypatch, mean_type, var_type):
    """Round 6 (VERDICT r05 missing 4): the mean / variance processors no shipped config names (posterior_mean_variance.py:53-101
    previous_x / start_x, :171-222 fixed_small / fixed_large / learned) run in osm_posterior_typed inside the fused loop; vs the REAL
    reference's 10-step guided Osmosis loop built with them (tests/golden/loop_processors.npz; x_T, y, noise of
    loop_underwater_physical_revised.npz).  start_x / fixed_small amplifies (the reference's own drift from a 1e-6 perturbation is
    2.6e-3): that chain is compared step by step from the reference's inputs."""
    unet, gd, M, CM = pkg
    g, base = np.load(os.path.join(GOLD, "loop_processors.npz")), np.load(os.path.join(GOLD, "loop_underwater_physical_revised.npz"))
    spec = OPERATORS["underwater_physical_revised"]
    model = make_model(unet)
    operator = M.get_operator("underwater_physical_revised", device=DEV, batch_size=1, **spec["operator"])
    cond = CM.get_conditioning_method("osmosis", operator, M.get_noise("clean"), **spec["cond"], **PATTERN, **spec["aux"])
    sampler = gd.get_sampler("ddpm")(use_timesteps=range(0, 100, 10), betas=gd.get_named_beta_schedule("linear", 1000),
                                     model_mean_type=mean_type, model_var_type=var_type, dynamic_threshold=False,
                                     clip_denoised=False, rescale_timesteps=False)
    _no_generic(monkeypatch, sampler)
    noise = torch.from_numpy(base["noise"]).to(DEV)
    y = torch.from_numpy(base["y"]).to(DEV)
    tag = f"osmosis.{mean_type}.{var_type}"
    scale = max(1.0, float(np.abs(g[f"{tag}.final_img"]).max()))
    bar = _free_running_bar(float(g[f"{tag}.drift_1e-6"]))
    kw = dict(model=model, measurement=y, measurement_cond_fn=cond.conditioning, record=False, save_root=None, pretrain_model="osmosis",
              rgb_guidance=False, sample_pattern=PATTERN)
    trace = []
    if bar is None:                                          # teacher-forced: step k from the reference's input of step k
        x_in, worst = g[f"{tag}.x_in"], 0.0
        for k in range(10):
            img, variables, loss, x0 = sampler.p_sample_loop(x_start=torch.from_numpy(x_in[k]).to(DEV), index_range=(9 - k, 9 - k),
                                                             noise_fn=lambda kk, shape, k=k: noise[k], trace=trace, **kw)
            want = x_in[k + 1] if k < 9 else g[f"{tag}.final_img"]
            worst = max(worst, float((img.cpu() - torch.from_numpy(want)).abs().max()))
        print(f"{tag}: teacher-forced, worst one-step error {worst:.1e}")
        assert worst < 2e-5
    else:
        img, variables, loss, x0 = sampler.p_sample_loop(x_start=torch.from_numpy(base["x_T"]).to(DEV),
                                                         noise_fn=lambda k, shape: noise[k], trace=trace, **kw)
    e_first = float((trace[0]["x0"].cpu() - torch.from_numpy(g[f"{tag}.x0_first"])).abs().max())
    e_mean = float((trace[0]["mean"].cpu() - torch.from_numpy(g[f"{tag}.mean_first"])).abs().max())
    e_loss = max(abs(float(rec["loss"][0]) - float(g[f"{tag}.loss"][k].reshape(-1)[0])) / float(g[f"{tag}.loss"][k].reshape(-1)[0])
                 for k, rec in enumerate(trace))
    e_img = float((img.cpu() - torch.from_numpy(g[f"{tag}.final_img"])).abs().max()) / scale
    e_x0 = float((x0 - torch.from_numpy(g[f"{tag}.final_x0"])).abs().max()) / scale
    print(f"{tag}: first x0 {e_first:.1e} mean {e_mean:.1e}; loss(rel) {e_loss:.1e}; final img {e_img:.1e} x0 {e_x0:.1e} (of max |img| {scale:.1f}); bar {bar}")
    assert len(trace) == 10 and e_first < 5e-6 and e_mean < 5e-6 and e_loss < 2e-5
    assert e_img < (bar or 2e-5) and e_x0 < (bar or 2e-5)
    for n, v in variables.items():
        assert torch.allclose(v.cpu(), torch.from_numpy(g[f"{tag}.{n}"]), atol=2e-6), n


@pytest.mark.parametrize("name,mean_type,var_type", [("ddpm", "start_x", "fixed_large"), ("ddpm", "epsilon", "fixed_small"),
                                                     ("ddpm", "start_x", "learned"), ("ddim", "previous_x", "fixed_small"),
                                                     ("ddim", "start_x", "learned_range")])
def test_rgb_guidance_chains_with_the_other_processors(pkg, monkeypatch, name, mean_type, var_type):
    """The rgb-guidance chains (DDPM.p_sample / DDIM.p_sample + `ps`) with the other processors, on the fused kernels, vs the REAL
    reference's chains (loop_processors.npz; x_T, y and the draws of loop_ps.npz).  DDIM takes eps from the sampler's own tables
    (predict_eps_from_x_start :533-536) whatever the mean processor: dcoef[4:6] of osm_ddim_update.  Chains the reference itself
    cannot reproduce to 1e-3 from a 1e-6 perturbation (previous_x: drift 0.22; start_x / fixed_large: 1.3e-3) are compared step by
    step from the reference's inputs."""
    unet, gd, M, CM = pkg
    g, ps = np.load(os.path.join(GOLD, "loop_processors.npz")), np.load(os.path.join(GOLD, "loop_ps.npz"))
    model = make_model(unet)
    cond = CM.get_conditioning_method("ps", M.get_operator("rgb_guidance", device=DEV, batch_size=1),
                                      M.get_noise("gaussian", sigma=0.05), scale="0.6,0.5,0.4,0.0")
    sampler = gd.get_sampler(name)(use_timesteps=range(0, 100, 10), betas=gd.get_named_beta_schedule("linear", 1000),
                                   model_mean_type=mean_type, model_var_type=var_type, dynamic_threshold=False,
                                   clip_denoised=False, rescale_timesteps=False)
    _no_generic(monkeypatch, sampler)
    draws = torch.from_numpy(ps[f"{name}.draws_x"]).to(DEV)
    tag = f"ps.{name}.{mean_type}.{var_type}"
    bar = _free_running_bar(float(g[f"{tag}.drift_1e-6"]))
    kw = dict(model=model, measurement=torch.from_numpy(ps[f"{name}.y"]).to(DEV), measurement_cond_fn=cond.conditioning, record=False,
              save_root=None, pretrain_model="osmosis", rgb_guidance=True, sample_pattern=PATTERN)
    trace = []
    if bar is None:
        x_in, worst = g[f"{tag}.x_in"], 0.0
        for k in range(10):
            img = sampler.p_sample_loop(x_start=torch.from_numpy(x_in[k]).to(DEV), index_range=(9 - k, 9 - k),
                                        noise_fn=lambda kk, shape, k=k: draws[k], trace=trace, **kw)
            want = torch.from_numpy(x_in[k + 1] if k < 9 else g[f"{tag}.final_img"])
            worst = max(worst, float((img.cpu() - want).abs().max()) / max(1.0, float(want.abs().max())))
        print(f"{tag}: teacher-forced, worst one-step error {worst:.1e} (of the step's max |x|)")
        assert worst < 2e-5
    else:
        img = sampler.p_sample_loop(x_start=torch.from_numpy(ps[f"{name}.x_T"]).to(DEV), noise_fn=lambda k, shape: draws[k],
                                    trace=trace, **kw)
        err = float((img.detach().cpu() - torch.from_numpy(g[f"{tag}.final_img"])).abs().max())
        print(f"{tag}: free-running chain max-abs error {err:.1e}, bar {bar:.1e}")
        assert err < bar
    losses = [float(r["loss"][0]) for r in trace]
    assert len(losses) == 10 and np.allclose(losses, g[f"{tag}.loss"], rtol=2e-5), (losses, g[f"{tag}.loss"])


@pytest.mark.parametrize("name", ["ddpm", "ddim"])
def test_rgb_guidance_chain_with_clip_denoised_is_fused(pkg, monkeypatch, name):
    """`clip_denoised: True` is what configs/rgb_guidance_sample_config.yaml SHIPS (ddpm, `ps`, scale 3,3,3,0.1, gaussian noiser with
    sigma 0): until round 6 that switch sent the chain to the generic ATen loop.  Now: x0 clamped inside osm_posterior_typed, the
    guidance gradient masked by osm_clamp_bwd, the rest of the fused rgb-guidance step unchanged -- vs the REAL reference's chains
    (tests/golden/loop_clip.npz; x_T, y, draws of loop_ps.npz)."""
    unet, gd, M, CM = pkg
    g, ps = np.load(os.path.join(GOLD, "loop_clip.npz")), np.load(os.path.join(GOLD, "loop_ps.npz"))
    model = make_model(unet)
    ckw = dict(scale="3,3,3,0.1", sigma=0) if name == "ddpm" else dict(scale="0.6,0.5,0.4,0.0", sigma=0.05)
    cond = CM.get_conditioning_method("ps", M.get_operator("rgb_guidance", device=DEV, batch_size=1),
                                      M.get_noise("gaussian", sigma=ckw["sigma"]), scale=ckw["scale"])
    sampler = gd.get_sampler(name)(use_timesteps=range(0, 100, 10), betas=gd.get_named_beta_schedule("linear", 1000),
                                   model_mean_type="epsilon", model_var_type="learned_range", dynamic_threshold=False,
                                   clip_denoised=True, rescale_timesteps=False)
    _no_generic(monkeypatch, sampler)
    draws = torch.from_numpy(ps[f"{name}.draws_x"]).to(DEV)
    trace = []
    img = sampler.p_sample_loop(model=model, x_start=torch.from_numpy(ps[f"{name}.x_T"]).to(DEV),
                                measurement=torch.from_numpy(ps[f"{name}.y"]).to(DEV), measurement_cond_fn=cond.conditioning,
                                record=False, save_root=None, pretrain_model="osmosis", rgb_guidance=True, sample_pattern=PATTERN,
                                noise_fn=lambda k, shape: draws[k], trace=trace)
    losses = [float(r["loss"][0]) for r in trace]
    assert np.allclose(losses, g[f"ps.{name}.loss"], rtol=2e-5), (losses, g[f"ps.{name}.loss"])
    assert max(float(r["x0"].abs().max()) for r in trace) == 1.0            # the clamp was active
    err, bar = float((img.cpu() - torch.from_numpy(g[f"ps.{name}.final_img"])).abs().max()), _free_running_bar(float(g[f"ps.{name}.drift_1e-6"]))
    print(f"ps.{name} clip_denoised: free-running chain max-abs error {err:.1e}, bar {bar:.1e}")
    assert err < bar


@pytest.mark.parametrize("conv_mode", ["f32", "f16x3"])
def test_fused_loop_with_clip_denoised_matches_the_reference(pkg, monkeypatch, conv_mode):
    """The Osmosis loop with `clip_denoised: True` (6.4 % of the pred_xstart elements on the clamp) on the fused kernels vs the REAL
    reference: per-step pred_xstart, guidance gradient (masked through the clamp) and loss, final image and phi."""
    unet, gd, M, CM = pkg
    g, base = np.load(os.path.join(GOLD, "loop_clip.npz")), np.load(os.path.join(GOLD, "loop_underwater_physical_revised.npz"))
    spec = OPERATORS["underwater_physical_revised"]
    model = make_model(unet)
    model.conv_mode = conv_mode
    operator = M.get_operator("underwater_physical_revised", device=DEV, batch_size=1, **spec["operator"])
    cond = CM.get_conditioning_method("osmosis", operator, M.get_noise("clean"), **spec["cond"], **PATTERN, **spec["aux"])
    sampler = gd.get_sampler("ddpm")(use_timesteps=range(0, 100, 10), betas=gd.get_named_beta_schedule("linear", 1000),
                                     model_mean_type="epsilon", model_var_type="learned_range", dynamic_threshold=False,
                                     clip_denoised=True, rescale_timesteps=False)
    _no_generic(monkeypatch, sampler)
    noise = torch.from_numpy(base["noise"]).to(DEV)
    trace = []
    img, variables, loss, x0 = sampler.p_sample_loop(
        model=model, x_start=torch.from_numpy(base["x_T"]).to(DEV), measurement=torch.from_numpy(base["y"]).to(DEV),
        measurement_cond_fn=cond.conditioning, record=False, save_root=None, pretrain_model="osmosis", rgb_guidance=False,
        sample_pattern=PATTERN, noise_fn=lambda k, shape: noise[k], trace=trace)
    e_x0 = max(float((r["x0"].cpu() - torch.from_numpy(g["osmosis.x0"][k])).abs().max()) for k, r in enumerate(trace))
    e_g = max(float((r["grad"].cpu() - torch.from_numpy(g["osmosis.grad"][k])).abs().max()) / float(np.abs(g["osmosis.grad"][k]).max())
              for k, r in enumerate(trace))
    e_loss = max(abs(float(r["loss"][0]) - float(g["osmosis.loss"][k].reshape(-1)[0])) / float(g["osmosis.loss"][k].reshape(-1)[0])
                 for k, r in enumerate(trace))
    e_img = float((img.cpu() - torch.from_numpy(g["osmosis.final_img"])).abs().max())
    print(f"osmosis clip_denoised {conv_mode}: x0 {e_x0:.1e} grad(rel) {e_g:.1e} loss(rel) {e_loss:.1e} final img {e_img:.1e}")
    assert max(float(r["x0"].abs().max()) for r in trace) == 1.0
    bar = _free_running_bar(float(g["osmosis.drift_1e-6"]))
    assert e_x0 < bar and e_g < 1e-4 and e_loss < 2e-5 and e_img < bar
    for n, v in variables.items():
        assert torch.allclose(v.cpu(), torch.from_numpy(g[f"osmosis.{n}"]), atol=2e-6), n


def test_shipped_rgb_guidance_config_runs_on_the_fused_kernels(pkg, monkeypatch):
    """configs/rgb_guidance_sample_config.yaml AS THE REFERENCE PARSES IT (tests/golden/configs.json: ddpm, 1000 steps, `ps` with scale
    3,3,3,0.1, gaussian noiser sigma 0, clip_denoised True, rgb_guidance True) through the per-image driver (`sampling.restore_image`)
    on the tiny seeded network: all 1000 steps on the fused kernels (the generic loop must not be entered), bounded by the clamp,
    reproducible from the seed."""
    import json
    unet, gd, M, CM = pkg
    from osmosis_diffusion_code_amd import sampling
    with open(os.path.join(GOLD, "configs.json")) as f:
        cfg = json.load(f)["rgb_guidance_sample_config.yaml"]
    assert cfg["rgb_guidance"] is True and cfg["diffusion"]["clip_denoised"] is True and cfg["conditioning"]["method"] == "ps"
    assert cfg["diffusion"]["sampler"] == "ddpm" and str(cfg["diffusion"]["timestep_respacing"]) == "1000"

    def no_generic(*a, **k):
        raise AssertionError("the shipped rgb-guidance configuration fell back to the generic loop")
    monkeypatch.setattr(gd.GaussianDiffusion, "_generic_loop", no_generic)
    model = make_model(unet)
    ref = (torch.rand(1, 3, 32, 32, generator=torch.Generator().manual_seed(4)) * 1.6 - 0.8).to(DEV)
    a = sampling.restore_image(model, ref, cfg, noise_seed=11)[0]
    b = sampling.restore_image(model, ref, cfg, noise_seed=11)[0]
    c = sampling.restore_image(model, ref, cfg, noise_seed=12)[0]
    assert a["sample"].shape == (1, 4, 32, 32) and a["rgb"].shape == (3, 32, 32) and bool(torch.isfinite(a["sample"]).all())
    assert torch.equal(a["sample"], b["sample"]) and not torch.equal(a["sample"], c["sample"])
    # the last step's mean is coef1 * clamp(x0) + coef2 * x: the chain cannot leave the neighbourhood of [-1, 1]
    print("shipped rgb-guidance config, 1000 fused steps: max |sample|", float(a["sample"].abs().max()),
          " ||y - rgb|| / ||y||", float((ref.cpu()[0] - a["rgb"]).norm() / ref.cpu().norm()))
    assert float(a["sample"].abs().max()) < 1.5
    assert float((ref.cpu()[0] - a["rgb"]).norm()) < 0.5 * float(ref.cpu().norm())      # the guidance pulled the RGB channels to y


def test_shipped_rgb_guidance_config_batched_and_sharded_driver(pkg, tmp_path):
    """`sampling.restore_images` with the shipped rgb-guidance configuration (respaced to 40 steps to keep the test short): three
    images as one batch of two + one single give, image by image, what three batch-1 runs give (independent chains: per-image norm,
    per-image noise stream), and every result -- batched ones included -- is written by `save_outputs` as the reference's second
    driver branch writes it."""
    import copy
    import json
    unet, gd, M, CM = pkg
    from osmosis_diffusion_code_amd import sampling
    with open(os.path.join(GOLD, "configs.json")) as f:
        cfg = copy.deepcopy(json.load(f)["rgb_guidance_sample_config.yaml"])
    cfg["diffusion"]["timestep_respacing"] = "40"
    model = make_model(unet)
    g = torch.Generator().manual_seed(31)
    images = [(torch.rand(1, 3, 32, 32, generator=g) * 1.6 - 0.8).to(DEV) for _ in range(3)]
    one = sampling.restore_images(model, images, cfg, batch_size=1, noise_seed=3)
    two = sampling.restore_images(model, images, cfg, batch_size=2, noise_seed=3)
    assert sorted(one) == sorted(two) == [0, 1, 2]
    for i in range(3):
        err = float((one[i]["sample"] - two[i]["sample"]).abs().max())
        assert err < 2e-5, (i, err)
        assert set(two[i]) >= {"sample", "rgb", "rgb_01_clip", "depth_mm", "depth_pmm", "measurement"}
        paths = sampling.save_outputs(two[i], images[i], str(tmp_path), f"img{i}")
        assert paths["grid"].endswith(f"img{i}.png") and all(os.path.getsize(p) > 0 for p in paths.values())
    assert not torch.equal(one[0]["sample"], one[1]["sample"])
    # ranks: images[rank::world], no collective
    r1 = sampling.restore_images(model, images, cfg, rank=1, world=2, batch_size=1, noise_seed=3)
    assert sorted(r1) == [1] and torch.equal(r1[1]["sample"], one[1]["sample"])


@pytest.mark.parametrize("mode", ["osmosis", "ps.ddpm"])
def test_previous_x_raises_where_the_reference_raises(pkg, mode):
    """`previous_x` returns the network's split output as the mean, and the Osmosis branch / DDPM.p_sample add to it in place: the
    reference raises autograd's view error there (recorded by the generator in loop_processors.npz `previous_x.raises`)."""
    unet, gd, M, CM = pkg
    g = np.load(os.path.join(GOLD, "loop_processors.npz"))
    assert mode in list(g["previous_x.raises"])
    model = make_model(unet)
    sampler = gd.get_sampler("ddpm")(use_timesteps=range(0, 100, 10), betas=gd.get_named_beta_schedule("linear", 1000),
                                     model_mean_type="previous_x", model_var_type="learned_range", dynamic_threshold=False,
                                     clip_denoised=False, rescale_timesteps=False)
    if mode == "osmosis":
        spec = OPERATORS["underwater_physical_revised"]
        operator = M.get_operator("underwater_physical_revised", device=DEV, batch_size=1, **spec["operator"])
        cond = CM.get_conditioning_method("osmosis", operator, M.get_noise("clean"), **spec["cond"], **PATTERN, **spec["aux"])
    else:
        cond = CM.get_conditioning_method("ps", M.get_operator("rgb_guidance", device=DEV, batch_size=1),
                                          M.get_noise("gaussian", sigma=0.05), scale="0.6,0.5,0.4,0.0")
    x = torch.zeros(1, 4, 32, 32, device=DEV)
    with pytest.raises(RuntimeError, match="modified inplace"):
        sampler.p_sample_loop(model=model, x_start=x, measurement=x[:, :3], measurement_cond_fn=cond.conditioning, record=False,
                              save_root=None, pretrain_model="osmosis", rgb_guidance=(mode != "osmosis"), sample_pattern=PATTERN)


def test_generic_loop_honours_record(pkg, tmp_path):
    """VERDICT r05 missing 4: `record=True` is honoured by every loop (reference gaussian_diffusion.py:308-333), also by the
    generic autograd loop a third-party conditioner runs through."""
    unet, gd, M, CM = pkg
    g, model, cond, sampler = _ps_setup(pkg, "ddpm")
    calls = []

    def third_party(**kw):                      # a plain function: no `__self__`, so the fused loop does not recognise it
        calls.append(1)
        return cond.conditioning(**kw)
    recs = []
    torch.manual_seed(0)
    img = sampler.p_sample_loop(model=model, x_start=torch.from_numpy(g["ddpm.x_T"]).to(DEV).requires_grad_(),
                                measurement=torch.from_numpy(g["ddpm.y"]).to(DEV), measurement_cond_fn=third_party,
                                record=True, save_root=None, pretrain_model="osmosis", rgb_guidance=True,
                                sample_pattern=PATTERN, record_every=3, record_out=recs,
                                save_grids_path=str(tmp_path), original_file_name="img7")
    assert len(calls) == 10 and torch.isfinite(img).all()
    assert [i for i, _ in recs] == [9, 6, 3, 0]
    assert os.path.exists(os.path.join(str(tmp_path), "img7_process.png"))


def test_record_process_grid_matches_the_reference(pkg, tmp_path):
    """`record=True` of the fused Osmosis loop: `<name>_process.png` (gaussian_diffusion.py:308-333: clipped RGB of pred_xstart on
    the first row, viridis of its percentile-normalised depth on the second, make_grid(nrow = snapshots) -> to_pil_image) vs the array
    the REAL reference hands to to_pil_image for the same 10-step chain (tests/golden/loop_record.npz; record_every = 3)."""
    from PIL import Image
    unet, gd, M, CM = pkg
    g, want = np.load(os.path.join(GOLD, "loop_underwater_physical_revised.npz")), np.load(os.path.join(GOLD, "loop_record.npz"))["process_png"]
    spec = OPERATORS["underwater_physical_revised"]
    model = make_model(unet)
    operator = M.get_operator("underwater_physical_revised", device=DEV, batch_size=1, **spec["operator"])
    cond = CM.get_conditioning_method("osmosis", operator, M.get_noise("clean"), **spec["cond"], **PATTERN, **spec["aux"])
    noise = torch.from_numpy(g["noise"]).to(DEV)
    make_sampler(gd).p_sample_loop(
        model=model, x_start=torch.from_numpy(g["x_T"]).to(DEV), measurement=torch.from_numpy(g["y"]).to(DEV),
        measurement_cond_fn=cond.conditioning, record=True, save_root=None, pretrain_model="osmosis", rgb_guidance=False,
        sample_pattern=PATTERN, noise_fn=lambda k, shape: noise[k], record_every=3, save_grids_path=str(tmp_path),
        original_file_name="frame")
    png = np.asarray(Image.open(os.path.join(str(tmp_path), "frame_process.png")))
    assert png.shape == want.shape == (2 * 34 + 2, 4 * 34 + 2, 3)
    diff = np.abs(png.astype(np.int32) - want.astype(np.int32))
    print("process grid: fraction of differing pixels", float((diff > 0).mean()), "max difference", int(diff.max()))
    # truncation to uint8 of values that differ by ~1e-6, and one-entry moves of the colour map: a handful of pixels, small steps
    assert float((diff > 0).mean()) < 0.01 and int(diff.max()) <= 6
    assert int(png[0, 0, 0]) == 0                                                  # make_grid's default pad_value


def test_library_step_noise(pkg):
    """Round 6: the per-step noise is drawn inside osm_guide_update_rng (Philox-4x32-10, counter = (element / 4, image, step)).
    (i) the raw generator reproduces the Random123 known-answer vectors; (ii) osm_randn has the moments of N(0, 1) and
    independent per-image / per-step streams; (iii) the fused loop's chain is reproducible from its seed, differs between
    seeds, does not depend on how a batch is chunked, and `shared_noise` gives every image of a batch the batch-1 chain."""
    from osmosis_diffusion_code_amd import ops
    unet, gd, M, CM = pkg
    # (i) Random123 kat_vectors: philox4x32-10, counter / key all zero, all ones, and the pi digits
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        n4 = (ctr[0] & 0xffffffff) + 1                       # word 0 of the counter is the element-quad index
        out = torch.zeros(4 * n4 if n4 <= 4096 else 4, device=DEV, dtype=torch.int32)
        if n4 > 4096:
            continue                                        # (quad index 0xffffffff / 0x243f6a88: checked through numpy below)
        ops.philox_raw(out, n4, ctr[1], ctr[2], ctr[3], key[0], key[1])
        got = tuple(int(v) & 0xffffffff for v in out[-4:].cpu())
        assert got == want, (hex(got[0]), hex(want[0]))

    def philox_np(c, k):                                    # the published algorithm, for the vectors a launch cannot reach
        c, k = [int(v) for v in c], [int(v) for v in k]
        for _ in range(10):
            p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
            c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xffffffff, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xffffffff]
            k = [(k[0] + 0x9E3779B9) & 0xffffffff, (k[1] + 0xBB67AE85) & 0xffffffff]
        return tuple(c)
    for ctr, key, want in kat:
        assert philox_np(ctr, key) == want
    out = torch.zeros(4 * 1000, device=DEV, dtype=torch.int32)
    ops.philox_raw(out, 1000, 7, 8, 9, 10, 11)
    ref = np.array([philox_np((q, 7, 8, 9), (10, 11)) for q in (0, 1, 999)], dtype=np.uint32)
    got = out.view(1000, 4)[[0, 1, 999]].cpu().numpy().astype(np.uint32)
    assert (got == ref).all()
    # (ii) moments, independence
    n = 4 * 256 * 256
    z = torch.empty(3, n, device=DEV)
    ops.randn(z, 3, n, seed=1234, step_const=5, img0=0)
    z64 = z.double()
    assert abs(float(z64.mean())) < 4e-3 and abs(float(z64.var()) - 1.0) < 6e-3
    assert abs(float((z64 ** 3).mean())) < 2e-2 and abs(float((z64 ** 4).mean()) - 3.0) < 5e-2
    assert torch.isfinite(z).all() and float(z.abs().max()) < 6.5
    for a, b in ((0, 1), (0, 2), (1, 2)):
        assert abs(float((z64[a] * z64[b]).mean())) < 6e-3       # distinct images: uncorrelated
    z2 = torch.empty(1, n, device=DEV)
    ops.randn(z2, 1, n, seed=1234, step_const=5, img0=2)
    assert torch.equal(z2[0], z[2])                              # an image's stream does not depend on the batch it is drawn in
    ops.randn(z2, 1, n, seed=1234, step_const=6, img0=2)
    assert abs(float((z2[0].double() * z64[2]).mean())) < 6e-3   # another step: another stream
    ops.randn(z2, 1, n, seed=1235, step_const=5, img0=2)
    assert abs(float((z2[0].double() * z64[2]).mean())) < 6e-3   # another seed: another stream
    # (iii) through the fused loop
    g = dict(np.load(os.path.join(GOLD, "loop_underwater_physical_revised.npz")))
    spec = OPERATORS["underwater_physical_revised"]
    model = make_model(unet)
    x_T = torch.from_numpy(g["x_T"]).to(DEV)
    y = torch.from_numpy(g["y"]).to(DEV)

    def chain(B, seed, monkey_cap=None, **kw):
        operator = M.get_operator("underwater_physical_revised", device=DEV, batch_size=B, **spec["operator"])
        cond = CM.get_conditioning_method("osmosis", operator, M.get_noise("clean"), **spec["cond"], **PATTERN, **spec["aux"])
        trace = []
        if monkey_cap is not None:
            os.environ["OSM_MAX_BATCH"] = str(monkey_cap)
        try:
            img = make_sampler(gd).p_sample_loop(model=model, x_start=x_T.repeat(B, 1, 1, 1), measurement=y.repeat(B, 1, 1, 1),
                                                 measurement_cond_fn=cond.conditioning, record=False, save_root=None,
                                                 pretrain_model="osmosis", sample_pattern=PATTERN, noise_seed=seed, trace=trace, **kw)[0]
        finally:
            os.environ.pop("OSM_MAX_BATCH", None)
        return img, trace
    a, tr = chain(1, 77)
    assert "noise" in tr[0] and abs(float(tr[0]["noise"].double().var()) - 1.0) < 0.1
    assert float(tr[-1]["noise"].abs().max()) == 0.0            # index 0: no noise (gaussian_diffusion.py:267)
    b, _ = chain(1, 77)
    assert torch.equal(a, b)                                    # reproducible from the seed
    c, _ = chain(1, 78)
    assert float((a - c).abs().max()) > 1e-3                    # another seed: another chain
    d3, tr3 = chain(3, 77)
    assert torch.allclose(d3[0:1], a, atol=2e-5)                # image 0 of a batch = the batch-1 chain (same index, same stream)
    assert float((d3[1] - d3[0]).abs().max()) > 1e-3            # images of a batch get different noise
    e3, _ = chain(3, 77, monkey_cap=2)                          # the batch walked in chunks of two sizes
    assert torch.allclose(e3, d3, atol=2e-5)
    s3, _ = chain(3, 77, shared_noise=True)
    assert torch.allclose(s3[1:2], a, atol=2e-5) and torch.allclose(s3[2:3], a, atol=2e-5)


def test_full_size_batch_equals_independent_images(pkg):
    """BASELINE-size property (no oracle needed): a B = 2 batch through the fused guided loop at 256x256 on the
    552.8 M-parameter network equals two B = 1 chains (images are independent Markov chains: per-image reductions,
    per-image phi), including phi after the SGD updates."""
    unet, gd, M, CM = pkg
    import contextlib
    import io
    import bench
    with contextlib.redirect_stdout(io.StringIO()):
        model = unet.create_model(**bench.UNET_KW)
    model.reset_parameters(1234)
    model = model.to(DEV).eval()
    sampler = gd.create_sampler(**bench.DIFFUSION)
    g = torch.Generator().manual_seed(5)
    x_T = 0.5 * torch.randn(2, 4, 256, 256, generator=g)
    y = torch.rand(2, 3, 256, 256, generator=g) * 1.6 - 0.8
    noise = torch.randn(3, 2, 4, 256, 256, generator=g)

    def chain(sl):
        b = sl.stop - sl.start
        op = M.get_operator("underwater_physical_revised", device=DEV, batch_size=b, **bench.OPERATOR)
        cond = CM.get_conditioning_method("osmosis", op, M.get_noise("clean"), **bench.COND, **bench.PATTERN,
                                          aux_loss=bench.AUX)
        nd = noise[:, sl].to(DEV)
        img, variables, loss, x0 = sampler.p_sample_loop(
            model=model, x_start=x_T[sl].to(DEV), measurement=y[sl].to(DEV), measurement_cond_fn=cond.conditioning,
            record=False, save_root=None, pretrain_model="osmosis", rgb_guidance=False, sample_pattern=bench.PATTERN,
            index_range=(202, 200), noise_fn=lambda k, shape: nd[k])
        return img.cpu(), {k: v.cpu() for k, v in variables.items()}, np.asarray(loss), x0

    both = chain(slice(0, 2))
    singles = [chain(slice(i, i + 1)) for i in range(2)]
    for i in range(2):
        assert torch.isfinite(both[0][i]).all()
        assert float((both[0][i] - singles[i][0][0]).abs().max()) < 1e-5
        assert float((both[3][i] - singles[i][3][0]).abs().max()) < 1e-5
        assert abs(both[2][i] - singles[i][2][0]) < 1e-4 * abs(singles[i][2][0])
        for k in both[1]:
            assert torch.allclose(both[1][k][i], singles[i][1][k][0], atol=1e-7), k
