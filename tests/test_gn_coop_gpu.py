"""Cooperative single-read GroupNorm (csrc/gn_coop.inc.h): one launch, x (and dy) read once, per-group partial sums exchanged
between the workgroups of an image through tagged 8-byte slots.

Checked: against torch-CPU fp64 GroupNorm (+FiLM, +SiLU) forward and input gradient at the tolerances of the chunked path's own
test; against the chunked path; repeated launches on one workspace (the launch counter in the slots); the BOUNDED wait -- with
the timeout at zero every workgroup takes the slow path (recomputes what is missing itself), with the residency limit ignored
the grid cannot be resident at once: both must give the fast path's bits."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = 32


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from osmosis_diffusion_code_amd import ops as o
    return o


@pytest.fixture(autouse=True)
def knobs(ops):
    ops.gn_coop_set(on=1, kb=32, min_kb=0, force=0, timeout_us=2000)       # min_kb 0: the small test shapes take the kernel
    yield
    ops.gn_coop_set(on=0, kb=32, min_kb=512, force=0, timeout_us=2000)        # the library's defaults


def reference(x, gamma, beta, e, dy, silu, B, C, HW):
    xr = x.double().clone().requires_grad_(True)
    y = F.group_norm(xr, G, gamma.double(), beta.double(), eps=1e-5)
    if e is not None:
        y = y * (1 + e[:, :C, None].double()) + e[:, C:, None].double()
    if silu:
        y = F.silu(y)
    (dx,) = torch.autograd.grad(y, xr, dy.double())
    return y.detach(), dx


def make(B, C, HW, film, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, HW, generator=g) * 1.7 + 0.4
    gamma = 1 + 0.1 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    e = 0.3 * torch.randn(B, 2 * C, generator=g) if film else None
    dy = torch.randn(B, C, HW, generator=g)
    add = torch.randn(B, C, HW, generator=g)
    return x, gamma, beta, e, dy, add


def nhwc(t, B, C, HW, dtype=torch.float32):
    return t.permute(0, 2, 1).reshape(B * HW, C).contiguous().to(DEV, dtype)


def back(t, B, C, HW):
    return t.float().cpu().reshape(B, HW, C).permute(0, 2, 1)


def run_coop(ops, B, C, HW, x, gamma, beta, e, dy, add, silu, ws_f=None, ws_b=None, dtype=torch.float32, maxabs=False):
    half = dtype == torch.float16
    xm = ops.Mat.of(nhwc(x, B, C, HW, dtype))
    gd, bd = gamma.to(DEV), beta.to(DEV)
    ed = e.to(DEV) if e is not None else None
    stats = torch.empty(B * G * 2, device=DEV)
    gstats = torch.empty(B * G * 2, device=DEV)
    y = torch.empty(B * HW, C, device=DEV, dtype=dtype)
    dx = torch.empty(B * HW, C, device=DEV, dtype=dtype)
    ws_f = ws_f if ws_f is not None else ops.gn_coop_workspace(B, HW, C, G, 0, DEV, half=half)
    ws_b = ws_b if ws_b is not None else ops.gn_coop_workspace(B, HW, C, G, 1, DEV, half=half)
    P = ops.MAXABS_PARTS
    mo = torch.full((B * P,), float("nan"), device=DEV) if maxabs else None
    mi = torch.full((B * P,), float("nan"), device=DEV) if maxabs else None
    mb = torch.full((B * P,), float("nan"), device=DEV) if maxabs else None
    ops.gn_fwd_coop(xm, ops.Mat.of(y), B, HW, G, stats, gd, bd, ws_f, film=ed, silu=silu, maxabs=mo, maxabs_in=mi)
    ops.gn_bwd_coop(xm, ops.Mat.of(nhwc(dy, B, C, HW, dtype)), ops.Mat.of(dx), B, HW, G, stats, gd, bd, gstats, ws_b,
                    film=ed, silu=silu, addend=ops.Mat.of(nhwc(add, B, C, HW, dtype)), maxabs=mb)
    torch.cuda.synchronize()
    return dict(y=y, dx=dx, stats=stats, gstats=gstats, mo=mo, mi=mi, mb=mb, ws_f=ws_f, ws_b=ws_b, xm=xm)


SHAPES = [  # B, C, HW, film, silu
    (1, 256, 4096, True, True),        # 64 x 64, 256 channels
    (2, 512, 1024, True, True),        # 32 x 32, 512 channels, two images
    (1, 1024, 1024, False, True),
    (2, 128, 900, True, False),        # ragged: the last workgroup's chunk is cut by the end of the image
    (1, 768, 1024, True, True),        # 192 channel vectors per row: 384 of the 512 threads work
    (1, 2048, 256, False, True),       # one row per sweep
    (1, 1536, 256, True, True),
    (2, 256, 16384, False, True),      # 128 x 128, two images: forward 16 vectors per thread, backward does not fit
    (1, 256, 16384, True, True),       # 128 x 128: x and dy resident (8 + 8 vectors per thread)
    (1, 512, 16384, True, True),       # forward: 16 vectors per thread; backward: not resident -> no plan
]


@pytest.mark.parametrize("B,C,HW,film,silu", SHAPES)
def test_coop_group_norm_vs_fp64_and_chunked(ops, B, C, HW, film, silu):
    x, gamma, beta, e, dy, add = make(B, C, HW, film, C + HW)
    y64, dx64 = reference(x, gamma, beta, e, dy, silu, B, C, HW)
    dx64 = dx64 + add.double()
    nf, nb = ops.gn_coop_plan(B, HW, C, G, 0), ops.gn_coop_plan(B, HW, C, G, 1)
    print(f"B={B} C={C} HW={HW}: workgroups per image forward {nf} backward {nb}")
    assert nf > 0, "every shape of this table has a forward plan"
    xm = ops.Mat.of(nhwc(x, B, C, HW))
    gd, bd = gamma.to(DEV), beta.to(DEV)
    ed = e.to(DEV) if film else None
    part = torch.empty(B * ops.gn_nchunk(HW) * G * 2, device=DEV)
    st_c, gst_c = torch.empty(B * G * 2, device=DEV), torch.empty(B * G * 2, device=DEV)
    y_c, dx_c = torch.empty(B * HW, C, device=DEV), torch.empty(B * HW, C, device=DEV)
    ops.gn_coop_set(on=0)              # the chunked / one-launch kernels of rounds 1-3
    ops.gn_fwd(xm, ops.Mat.of(y_c), B, HW, G, part, st_c, gd, bd, film=ed, silu=silu)
    ops.gn_bwd(xm, ops.Mat.of(nhwc(dy, B, C, HW)), ops.Mat.of(dx_c), B, HW, G, st_c, gd, bd, part, gst_c, film=ed, silu=silu,
               addend=ops.Mat.of(nhwc(add, B, C, HW)))
    ops.gn_coop_set(on=1)
    if nb == 0:
        with pytest.raises(Exception, match="no cooperative plan"):
            run_coop(ops, B, C, HW, x, gamma, beta, e, dy, add, silu)
        stats = torch.empty(B * G * 2, device=DEV)
        y = torch.empty(B * HW, C, device=DEV)
        ops.gn_fwd_coop(xm, ops.Mat.of(y), B, HW, G, stats, gd, bd, ops.gn_coop_workspace(B, HW, C, G, 0, DEV), film=ed, silu=silu)
        assert float((back(y, B, C, HW).double() - y64).abs().max()) < 2e-5
        return
    r = run_coop(ops, B, C, HW, x, gamma, beta, e, dy, add, silu, maxabs=True)
    assert float((back(r["y"], B, C, HW).double() - y64).abs().max()) < 2e-5
    assert float((back(r["dx"], B, C, HW).double() - dx64).abs().max()) < 5e-5 * max(1.0, float(dx64.abs().max()))
    assert float((r["y"] - y_c).abs().max()) < 2e-5
    assert float((r["dx"] - dx_c).abs().max()) < 5e-5 * max(1.0, float(dx64.abs().max()))
    assert float((r["stats"] - st_c).abs().max()) < 1e-5 * float(st_c.abs().max())
    assert float((r["gstats"] - gst_c).abs().max()) < 1e-5 * max(1e-3, float(gst_c.abs().max()))
    P = ops.MAXABS_PARTS
    for parts, t in ((r["mo"], r["y"]), (r["mb"], r["dx"]), (r["mi"], r["xm"].t)):      # exact, every slot rewritten
        assert torch.equal(parts.view(B, P).max(1).values, t.abs().view(B, HW * C).amax(1))


def test_coop_workspace_counts_launches(ops):
    """Ten launches on one workspace: the tag in the slots advances by one per launch and every launch gives the same bits."""
    B, C, HW = 2, 512, 1024
    x, gamma, beta, e, dy, add = make(B, C, HW, True, 5)
    first = run_coop(ops, B, C, HW, x, gamma, beta, e, dy, add, True)
    for k in range(9):
        r = run_coop(ops, B, C, HW, x, gamma, beta, e, dy, add, True, ws_f=first["ws_f"], ws_b=first["ws_b"])
        assert torch.equal(r["y"], first["y"]) and torch.equal(r["dx"], first["dx"])
        assert torch.equal(r["stats"], first["stats"]) and torch.equal(r["gstats"], first["gstats"])
    tags = (first["ws_f"] >> 32).cpu()
    assert int(tags.min()) == 10 and int(tags.max()) == 10


@pytest.mark.parametrize("B,C,HW", [(2, 512, 1024), (1, 256, 4096), (2, 128, 900)])
def test_coop_slow_path_gives_the_same_bits(ops, B, C, HW):
    """timeout 0: a workgroup whose first poll finds a slot missing stops waiting at once, recomputes the missing partial sums
    from HBM in workgroup order, and applies its chunk by re-reading it.  Same per-thread arithmetic, same combination order:
    the result must equal the fast path bit for bit (statistics included)."""
    x, gamma, beta, e, dy, add = make(B, C, HW, True, 11)
    fast = run_coop(ops, B, C, HW, x, gamma, beta, e, dy, add, True, maxabs=True)
    ops.gn_coop_set(timeout_us=0)
    slow = run_coop(ops, B, C, HW, x, gamma, beta, e, dy, add, True, maxabs=True)
    for k in ("y", "dx", "stats", "gstats"):
        assert torch.equal(fast[k], slow[k]), k
    P = ops.MAXABS_PARTS
    for k in ("mo", "mi", "mb"):
        assert torch.equal(fast[k].view(B, P).max(1).values, slow[k].view(B, P).max(1).values), k


def test_coop_oversubscribed_grid_still_finishes(ops):
    """force = 1 ignores the residency limit: 32 images x 128 workgroups of 512 threads are not on the device at once; images are
    independent exchange domains and start in order, the bounded wait (50 us here) covers whatever is cut at a boundary.  Same
    bits as the images one at a time."""
    B, C, HW = 32, 512, 1024
    x, gamma, beta, e, dy, add = make(B, C, HW, True, 3)
    ops.gn_coop_set(force=1, kb=16, timeout_us=50)
    assert ops.gn_coop_plan(B, HW, C, G, 0) * B > 2048
    big = run_coop(ops, B, C, HW, x, gamma, beta, e, dy, add, True)
    for b in (0, 17, 31):
        one = run_coop(ops, 1, C, HW, x[b:b + 1], gamma, beta, e[b:b + 1], dy[b:b + 1], add[b:b + 1], True)
        assert torch.equal(big["y"][b * HW:(b + 1) * HW], one["y"])
        assert torch.equal(big["dx"][b * HW:(b + 1) * HW], one["dx"])


def test_two_spinning_grids_on_one_device_do_not_deadlock(ops):
    """What two processes sharing a GPU do to each other: two streams, each launching a cooperative GroupNorm whose grid needs
    the WHOLE device (128 x 128 x 512 forward: 256 workgroups of 233 registers, one per CU).  Each grid gets part of the CUs and
    waits for partners that cannot start -- the bounded wait must break the tie (timeout 100 us), and the results must be the
    bits of the undisturbed launches."""
    B, C, HW = 1, 512, 16384
    x, gamma, beta, e, dy, add = make(B, C, HW, True, 21)
    assert ops.gn_coop_plan(B, HW, C, G, 0) == 256
    xm = ops.Mat.of(nhwc(x, B, C, HW))
    gd, bd, ed = gamma.to(DEV), beta.to(DEV), e.to(DEV)

    def launch(ws, y, st):
        ops.gn_fwd_coop(xm, ops.Mat.of(y), B, HW, G, st, gd, bd, ws, film=ed, silu=True)

    ref_y, ref_st = torch.empty(B * HW, C, device=DEV), torch.empty(B * G * 2, device=DEV)
    launch(ops.gn_coop_workspace(B, HW, C, G, 0, DEV), ref_y, ref_st)
    torch.cuda.synchronize()
    ops.gn_coop_set(timeout_us=100)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    wss = [ops.gn_coop_workspace(B, HW, C, G, 0, DEV) for _ in streams]
    ys = [torch.empty(B * HW, C, device=DEV) for _ in streams]
    sts = [torch.empty(B * G * 2, device=DEV) for _ in streams]
    torch.cuda.synchronize()
    for rep in range(25):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                launch(wss[i], ys[i], sts[i])
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.equal(ys[i], ref_y) and torch.equal(sts[i], ref_st)
        assert int((wss[i] >> 32).min()) == 25 and int((wss[i] >> 32).max()) == 25


def test_coop_half_storage_family(ops):
    B, C, HW = 2, 512, 1024
    x, gamma, beta, e, dy, add = make(B, C, HW, True, 9)
    xh, dyh, addh = x.half().float(), dy.half().float(), add.half().float()
    y64, dx64 = reference(xh, gamma, beta, e, dyh, True, B, C, HW)
    dx64 = dx64 + addh.double()
    assert ops.gn_coop_plan(B, HW, C, G, 1, half=True) > 0
    r = run_coop(ops, B, C, HW, xh, gamma, beta, e, dyh, addh, True, dtype=torch.float16)
    assert float((back(r["y"], B, C, HW).double() - y64).abs().max()) < 1.5 * 2 ** -11 * max(1.0, float(y64.abs().max()))
    assert float((back(r["dx"], B, C, HW).double() - dx64).abs().max()) < 1.5 * 2 ** -11 * max(1.0, float(dx64.abs().max()))


def test_coop_plan_limits(ops):
    ops.gn_coop_set(min_kb=512)
    assert ops.gn_coop_plan(1, 64, 1024, G, 0) == 0            # 256 KB image: the one-launch kernels keep it
    assert ops.gn_coop_plan(1, 65536, 512, G, 0) == 0          # 134 MB do not fit the register file
    assert ops.gn_coop_plan(1, 65536, 256, G, 1) == 0          # x + dy of a 256^2 x 256 tensor neither
    assert ops.gn_coop_plan(1, 1024, 512, 16, 0) == 0          # GroupNorm32 only
    assert ops.gn_coop_plan(1, 1024, 100, G, 0) == 0           # C not a multiple of 128
    assert ops.gn_coop_plan(1, 16384, 256, G, 1) > 0 and ops.gn_coop_plan(1, 4096, 512, G, 0) > 0
    ops.gn_coop_set(on=0)
    assert ops.gn_coop_plan(1, 16384, 256, G, 1) == 0


def test_engine_with_cooperative_group_norm_matches_the_default_path(ops):
    """The UNet engine routes its GroupNorms through the cooperative kernel where it has a plan (OSM_GN_COOP=1; off by default:
    a measured net loss in the step): forward and input gradient must agree with the default path to rounding."""
    from oracle import unet_ref as U
    from osmosis_diffusion_code_amd.guided_diffusion.unet import create_model
    kw = dict(image_size=256, num_channels=128, num_res_blocks=1, channel_mult="1,2,4", learn_sigma=True,
              attention_resolutions="64", num_heads=4, num_head_channels=64, use_scale_shift_norm=True,
              resblock_updown=True, pretrain_model="osmosis")
    cfg = U.UNetConfig.from_create_model_kwargs(**kw)
    sd = U.seeded_state_dict(cfg, 5)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 4, 128, 128, generator=g)
    t = torch.tensor([300.0])
    w = torch.randn(1, 8, 128, 128, generator=g)
    outs = []
    for on in (0, 1):
        ops.gn_coop_set(on=on, min_kb=512)
        m = create_model(**kw)
        m.load_state_dict(sd, strict=True)
        m = m.to(DEV).eval()
        xd = x.to(DEV).requires_grad_(True)
        yd = m(xd, t.to(DEV))
        (dxd,) = torch.autograd.grad((yd * w.to(DEV)).sum(), xd)
        eng = next(iter(m._engines.values()))
        names = {c[0].__name__ for c in eng._fwd_plan.calls} | {c[0].__name__ for c in eng._bwd_plan.calls}
        assert ("osm_gn_fwd_coop" in names) == bool(on) and ("osm_gn_bwd_coop" in names) == bool(on)
        yd2 = m(xd, t.to(DEV))                      # graph replay: the workspaces' launch counters keep counting
        assert torch.equal(yd2, yd)
        outs.append((yd.detach().cpu(), dxd.cpu()))
    assert float((outs[0][0] - outs[1][0]).abs().max()) < 2e-5 * max(1.0, float(outs[0][0].abs().max()))
    assert float((outs[0][1] - outs[1][1]).abs().max()) < 2e-5 * max(1.0, float(outs[0][1].abs().max()))
