"""Full-size parity at the timesteps that matter.

Every earlier full-size comparison with the oracle sat at t <= 37, where `sqrt_recipm1_alphas_cumprod` (the factor
that carries the UNet's error into pred_xstart, reference posterior_mean_variance.py:127-130) is 0.01-0.06.  Here ONE
teacher-forced guided step of BASELINE config 2 (B = 1, 256 x 256, 552.8 M parameters, n_iter = 20) is run from a
bounded x_t (the q-sample of a synthetic ground truth at that index) at

  idx = 299   (where bench.py's timed window starts),
  idx = 500,
  idx = the highest index at which the oracle's guided step stays finite with the seeded weights (SURVEY F10:
        pred_xstart leaves the physical model's range at high t and exp(-phi * depth) overflows -- in the reference too),
  idx = 999   (posterior only: guidance is non-finite there in the reference as well, so the guidance window is closed
        for this case; factor 157),

in the exact-fp32, bf16x6 and default f16x3 conv arithmetics, against oracle/diffusion_ref.py on the same weights /
x_t / y / noise.  Asserted: x_(t-1) and pred_xstart within north_star's 1e-3 max-abs.  Printed: the UNet-output
(eps / v) error next to the amplification factor, i.e. the error-vs-idx table DESIGN.md quotes."""
import contextlib
import io
import json
import os

import numpy as np
import pytest
import torch

import baseline_configs as BC
from oracle import diffusion_ref as D
from oracle import unet_ref as U
from test_configs_gpu import noised_start, synthetic_scene

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MODES = ("f32", "bf16x6", "f16x3")
TOL = 1e-3                      # north_star: outputs within 1e-3 max-abs of the reference


@pytest.fixture(scope="module")
def setup():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from osmosis_diffusion_code_amd.guided_diffusion import unet
    ucfg = U.UNetConfig.from_create_model_kwargs(**BC.UNET)
    sd = U.seeded_state_dict(ucfg, 1234)
    with contextlib.redirect_stdout(io.StringIO()):
        model = unet.create_model(**BC.UNET)
    model.load_state_dict(sd, strict=True)
    model = model.to(DEV).eval()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))   # oneDNN is fastest at ~16 threads on the 256-thread hosts
    gt, y = synthetic_scene(1, seed=5, phi_ab=(1.1, 0.95, 0.95), phi_inf=(0.14, 0.29, 0.49), depth_type="gamma")
    return model, ucfg, sd, gt, y


def hip_step(model, cfg, x_t, y, idx, noise, guided=True):
    from osmosis_diffusion_code_amd.guided_diffusion import condition_methods as CM
    from osmosis_diffusion_code_amd.guided_diffusion import gaussian_diffusion as gd
    from osmosis_diffusion_code_amd.guided_diffusion import measurements as M
    sampler = gd.create_sampler(**cfg["diffusion"])
    opc = dict(cfg["measurement"]["operator"])
    name = opc.pop("name")
    op = M.get_operator(name, device=DEV, batch_size=1, **opc)
    pattern = dict(cfg["sample_pattern"])
    if not guided:
        pattern["start_guidance"] = 0.0          # guidance window closed: the step is UNet + posterior + noise
    cond = CM.get_conditioning_method("osmosis", op, M.get_noise("clean"), **cfg["conditioning"]["params"],
                                      **pattern, **cfg["aux_loss"])
    trace = []
    nd = noise.to(DEV)
    img, variables, loss, x0 = sampler.p_sample_loop(
        model=model, x_start=x_t.to(DEV), measurement=y.to(DEV), measurement_cond_fn=cond.conditioning,
        record=False, save_root=None, pretrain_model="osmosis", rgb_guidance=False,
        sample_pattern=pattern, index_range=(idx, idx), noise_fn=lambda k, shape: nd[k], trace=trace)
    return dict(img=img.cpu(), x0=x0, loss=loss, variables={k: v.cpu() for k, v in variables.items()},
                model_out=trace[0]["model_out"].cpu(), grad=trace[0]["grad"].cpu() if guided else None)


def oracle_step(cfg, ucfg, sd, x_t, y, idx, noise, guided=True):
    tb = D.make_tables(1000, "linear", 1000)
    opc = dict(cfg["measurement"]["operator"])
    name = opc.pop("name")
    xi = x_t.clone().requires_grad_(guided)
    with torch.set_grad_enabled(guided):
        mo = U.unet_forward(sd, ucfg, xi, torch.tensor([float(idx)]))
        out = D.p_mean_variance(tb, mo, xi, idx)
    res = dict(model_out=mo.detach(), x0=out["pred_xstart"].detach())
    if guided:
        rop = D.PhysOperator(name, batch_size=1, depth_type=opc["depth_type"], value=opc["value"], phi_a=opc["phi_a"],
                             phi_b=opc["phi_b"], phi_inf=opc["phi_inf"])
        p = cfg["conditioning"]["params"]
        rg = D.OsmosisGuidance(rop, n_iter=20, scale=p["scale"], gradient_clip=p["gradient_clip"],
                               aux=cfg["aux_loss"]["aux_loss"])
        r_xt, r_loss, r_vars, r_grad = rg.conditioning(xi, out["mean"], out["pred_xstart"], y,
                                                       D.is_freeze_phi(cfg["sample_pattern"], idx, 1000))
        res.update(loss=float(np.asarray(r_loss).ravel()[0]), grad=r_grad.detach(),
                   variables={k: v.detach() for k, v in r_vars.items()})
        mean = r_xt.detach()
    else:
        mean = out["mean"].detach()
    res["img"] = mean + torch.exp(0.5 * out["log_variance"].detach()) * noise[0]
    res["finite"] = bool(torch.isfinite(res["img"]).all() and torch.isfinite(res["x0"]).all()
                         and (not guided or (np.isfinite(res["loss"]) and torch.isfinite(res["grad"]).all())))
    return res


def compare(tag, idx, factor, ref, got, rows, guided=True):
    e_mo = float((got["model_out"][:, :4] - ref["model_out"][:, :4]).abs().max())
    e_v = float((got["model_out"][:, 4:] - ref["model_out"][:, 4:]).abs().max())
    e_x0 = float((got["x0"] - ref["x0"]).abs().max())
    e_img = float((got["img"] - ref["img"]).abs().max())
    row = dict(mode=tag, idx=idx, sqrt_recipm1=round(factor, 4), eps_err=e_mo, v_err=e_v, pred_xstart_err=e_x0,
               x_prev_err=e_img, x0_scale=float(ref["x0"].abs().max()), guided=guided)
    if guided:
        gmax = float(ref["grad"].abs().max())
        # the guidance update is x -= scale * clamp(grad, +-clip): continuous, but where the gradient is astronomically large
        # (loss 1e16 at the top index) a gradient error far below 2e-4 of its maximum still exceeds the clip bound at pixels
        # whose own gradient is small, and the clamped value flips sign there (2 * 7 * 0.005 = 0.07).  `x_prev_err_determined`
        # is the error over the pixels whose gradient sign is beyond that error (|grad| > 2e-4 max |grad|)
        det = ref["grad"].abs() > 2e-4 * gmax
        d_img = (got["img"] - ref["img"]).abs()
        row.update(grad_err=float((got["grad"] - ref["grad"]).abs().max()), grad_max=gmax,
                   loss=float(got["loss"][0]), loss_ref=ref["loss"],
                   x_prev_err_determined=float(d_img[det].max()) if bool(det.any()) else 0.0,
                   undetermined_pixels=int((~det).sum()), flipped_pixels=int((d_img > TOL).sum()))
    rows.append(row)
    print(json.dumps(row))
    return row


SCALES = (1.0, 0.5, 0.25, 0.1, 0.03)


def bounded_x_t(gt, sampler, idx, scale):
    """The q-sample of the synthetic ground truth at idx, shrunk by `scale`: seeded weights do not denoise (eps is not the
    noise that was added), so pred_xstart = x_t / sqrt(abar) - sqrt(1 / abar - 1) eps leaves the physical model's range as
    idx grows and the reference's own 20-iteration phi SGD diverges (exp(-phi depth) overflows: measured here,
    the ORACLE is non-finite from idx 299 up for the un-shrunk q-sample).  Shrinking x_t keeps what it can bounded; the eps
    term (factor sqrt_recipm1) cannot be shrunk -- it is what limits the highest finite index."""
    return scale * noised_start(gt, sampler, idx)


def find_finite(model, cfg, ucfg, sd, gt, y, sampler, noise, idx):
    """Largest scale in SCALES at which the guided step at idx is finite on the HIP path (20 ms per probe) AND on the oracle
    (the claim is about the oracle).  Returns (x_t, scale, oracle result) or None."""
    for sc in SCALES:
        x_t = bounded_x_t(gt, sampler, idx, sc)
        g = hip_step(model, cfg, x_t, y, idx, noise)
        if not (torch.isfinite(g["img"]).all() and np.isfinite(g["loss"]).all() and torch.isfinite(g["grad"]).all()
                and all(torch.isfinite(v).all() for v in g["variables"].values())):
            continue
        r = oracle_step(cfg, ucfg, sd, x_t, y, idx, noise)
        print(f"idx {idx} scale {sc}: HIP finite, oracle finite = {r['finite']}")
        if r["finite"]:
            return x_t, sc, r
    return None


def test_teacher_forced_guided_step_at_high_t(setup):
    from osmosis_diffusion_code_amd.guided_diffusion import gaussian_diffusion as gd
    model, ucfg, sd, gt, y = setup
    cfg = BC.SAMPLE
    sampler = gd.create_sampler(**cfg["diffusion"])
    noise = torch.randn(1, 1, 4, 256, 256, generator=torch.Generator().manual_seed(4))
    rows = []

    model.conv_mode = "f32"
    cases = {}
    for idx in (299, 500):
        cases[idx] = find_finite(model, cfg, ucfg, sd, gt, y, sampler, noise, idx)
        assert cases[idx] is not None, f"no finite guided step at idx {idx} at any scale in {SCALES}"
    # ---- the highest index with a finite guided step (25-index scan from the top; phi is frozen above 0.7 T, what overflows
    # there is the squared residual itself)
    top = None
    for idx in range(975, 500, -25):
        x_t = bounded_x_t(gt, sampler, idx, SCALES[-1])
        g = hip_step(model, cfg, x_t, y, idx, noise)
        if torch.isfinite(g["img"]).all() and np.isfinite(g["loss"]).all() and torch.isfinite(g["grad"]).all():
            found = find_finite(model, cfg, ucfg, sd, gt, y, sampler, noise, idx)
            if found is not None:
                top, cases[idx] = idx, found
                break
    assert top is not None and top > 500, "no finite guided step above idx 500"
    print("highest idx with a finite guided oracle step (25-index scan):", top)

    for idx in (299, 500, top):
        x_t, sc, ref = cases[idx]
        factor = float(sampler.sqrt_recipm1_alphas_cumprod[idx])
        for mode in MODES:
            model.conv_mode = mode
            got = hip_step(model, cfg, x_t, y, idx, noise)
            row = compare(mode, idx, factor, ref, got, rows)
            row["x_t_scale"] = sc
            # Bars = <= 5x what this test measures on MI355X (profiles/r04_high_t_parity.json, re-measured in round 5; VERDICT r04
            # weak 1: the old bars -- 1e-3 everywhere -- let a 10x accuracy regression pass).  The network's error (eps: 2-7e-6 in
            # every arithmetic) enters pred_xstart times sqrt(1 / alphabar - 1): measured 7.1e-6 / 1.7e-5 / 1.9e-4 at idx 299 /
            # 500 / 800 (exact-fp32 MFMA, the least accurate of the three)
            assert row["pred_xstart_err"] < 2.5e-5 * max(1.0, factor), row
            if idx == top:      # loss ~1e16 there: see `x_prev_err_determined` in compare()   (measured: 2.1e-7, 3-5 pixels flip)
                assert row["x_prev_err_determined"] < 1.2e-6 and row["flipped_pixels"] <= 25, row
                assert row["flipped_pixels"] <= row["undetermined_pixels"], row
            else:               # measured 1.7e-5 (idx 299), 1.9e-4 (idx 500: the +-clip amplifies a 1e-5 gradient error)
                assert row["x_prev_err"] < (8e-5 if idx == 299 else TOL), row
            # gradient vs the oracle's, relative to its maximum: measured 4.6e-6 / 1.2e-5 / 1.1e-4
            assert row["grad_err"] < {299: 2.5e-5, 500: 6e-5}.get(idx, 2e-4) * row["grad_max"] + 1e-9, row
            assert abs(row["loss"] - row["loss_ref"]) < 5e-5 * abs(row["loss_ref"]), row
            for k, v in ref["variables"].items():
                assert torch.allclose(got["variables"][k].reshape(-1), v.reshape(-1), atol=5e-6), (k, mode, idx)

    # ---- idx 999: UNet + posterior + noise only (amplification 157), the un-shrunk q-sample
    idx = 999
    x_t = noised_start(gt, sampler, idx)
    ref = oracle_step(cfg, ucfg, sd, x_t, y, idx, noise, guided=False)
    assert ref["finite"]
    factor = float(sampler.sqrt_recipm1_alphas_cumprod[idx])
    for mode in MODES:
        model.conv_mode = mode
        got = hip_step(model, cfg, x_t, y, idx, noise, guided=False)
        row = compare(mode, idx, factor, ref, got, rows, guided=False)
        # pred_xstart is O(100) here: 1e-3 absolute is 1e-5 of its scale; the fp32 oracle itself is no better than that
        # measured 6.6e-4 (exact-fp32 MFMA) / 3.4e-4 (bf16x6) / 2.9e-4 (f16x3) of a pred_xstart that reaches 758: north_star's 1e-3
        # holds at the most amplified index of the chain; x_(t-1) itself: 4.8e-7
        assert row["pred_xstart_err"] < TOL, row
        assert row["x_prev_err"] < 2.5e-6, row
    model.conv_mode = "f16x3"

    out = os.environ.get("OSM_HIGH_T_TABLE")
    if out:
        with open(out, "w") as f:
            json.dump(rows, f, indent=1)


@pytest.mark.parametrize("fname,cfg_name", [("full_step.npz", "SAMPLE"), ("full_step_underwater_physical.npz", "SIMULATION"),
                                            ("full_step_haze_physical.npz", "HAZE")])
def test_full_size_guided_steps_vs_the_real_reference(setup, fname, cfg_name):
    """Two guided steps with the FULL network against the REAL reference's p_sample_loop (tests/golden/full_step.npz, produced on
    CPU by oracle/tools/gen_golden.py full_step; no oracle in between): sampler on use_timesteps = {0, 299} -- the first step runs
    the network at t = 299 with the 20-iteration phi update, the second at t = 0 --, x_T = 0.5 randn(seed 0), y ~ U(-0.8, 0.8),
    the reference's own noise draws (torch.manual_seed(0), measurement-shaped then image-shaped per step).  Every 4th pixel of
    x_t / pred_xstart / the unclipped gradient per step, loss and phi, in three arithmetics; tolerance: north_star's 1e-3."""
    from osmosis_diffusion_code_amd.guided_diffusion import condition_methods as CM
    from osmosis_diffusion_code_amd.guided_diffusion import gaussian_diffusion as gd
    from osmosis_diffusion_code_amd.guided_diffusion import measurements as M
    model, ucfg, sd, _gt, _y = setup
    cfg = getattr(BC, cfg_name)          # operator / guidance settings of BASELINE config 2 | 3 | 5 (B = 1, fp32 storage)
    g = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", fname)))
    st = int(g["stride"])
    x_T = 0.5 * torch.randn(1, 4, 256, 256, generator=torch.Generator().manual_seed(0))
    y = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(7)) * 1.6 - 0.8
    assert abs(float(x_T.double().abs().sum()) - float(g["x_T_abs_sum"])) < 1e-6 * float(g["x_T_abs_sum"])
    torch.manual_seed(0)
    noises = []
    for _ in range(2):
        torch.randn_like(y)                              # q_sample's unused draw
        noises.append(torch.randn_like(x_T))
    nd = torch.stack(noises).to(DEV)
    for mode in MODES:
        model.conv_mode = mode
        opc = dict(cfg["measurement"]["operator"])
        name = opc.pop("name")
        op = M.get_operator(name, device=DEV, batch_size=1, **opc)
        cond = CM.get_conditioning_method("osmosis", op, M.get_noise("clean"), **cfg["conditioning"]["params"],
                                          **cfg["sample_pattern"], **cfg["aux_loss"])
        sampler = gd.get_sampler("ddpm")(use_timesteps=(0, 299), betas=gd.get_named_beta_schedule("linear", 1000),
                                         model_mean_type="epsilon", model_var_type="learned_range", dynamic_threshold=False,
                                         clip_denoised=False, rescale_timesteps=False)
        assert list(sampler.timestep_map) == [int(v) for v in g["timestep_map"]]
        trace = []
        img, variables, loss, x0 = sampler.p_sample_loop(
            model=model, x_start=x_T.to(DEV), measurement=y.to(DEV), measurement_cond_fn=cond.conditioning, record=False,
            save_root=None, pretrain_model="osmosis", rgb_guidance=False, sample_pattern=cfg["sample_pattern"],
            noise_fn=lambda k, shape: nd[k], trace=trace)
        sub = lambda t: t.detach().cpu()[:, :, ::st, ::st]      # noqa: E731
        e = {k: max(float((sub(trace[i][tk]) - torch.from_numpy(g[f"trace.{k}_sub"][i])).abs().max()) for i in range(2))
             for k, tk in (("x_in", "x_in"), ("x0", "x0"), ("grad", "grad"))}
        e_fin = float((sub(img) - torch.from_numpy(g["final_img_sub"])).abs().max())
        e_x0f = float((sub(x0) - torch.from_numpy(g["final_x0_sub"])).abs().max())
        print(f"{mode}: full-size guided steps vs the real reference: x_t {e['x_in']:.2e}  pred_xstart {e['x0']:.2e}  grad {e['grad']:.2e} "
              f"(max {float(g['trace.grad_max'].max()):.1f})  final x {e_fin:.2e}  final pred_xstart {e_x0f:.2e}  loss {float(loss[0]):.4f} vs "
              f"{float(g['final_loss'][0]):.4f}")
        # measured vs the REAL reference: 2.2e-5 (f32) / 6e-6 (bf16x6) / 5.3e-6 (f16x3); gradient 1.5e-6 of its maximum; loss 3e-6
        assert e["x_in"] < 1e-4 and e["x0"] < 1e-4 and e_fin < 1e-4 and e_x0f < 1e-4
        assert e["grad"] < 8e-6 * float(g["trace.grad_max"].max())
        assert abs(float(loss[0]) - float(g["final_loss"][0])) < 1.5e-5 * abs(float(g["final_loss"][0]))
        for k in variables:
            assert np.allclose(variables[k].cpu().numpy().ravel(), g["final." + k].ravel(), atol=5e-6), (k, mode)
    model.conv_mode = "f16x3"
