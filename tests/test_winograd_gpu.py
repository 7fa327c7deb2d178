"""Winograd F(2x2, 3x3) convolution kernel (csrc/conv3_wino.inc.h) through the C ABI, against fp64 torch convolutions
and against the direct halo-tile kernel: forward and data-gradient images, ragged patches, several images, channel
tails, split-K, residual / accumulate epilogues, the fused GroupNorm+SiLU input transform, column sums through the
split-K combine, and what the entry point refuses."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from osmosis_diffusion_code_amd import ops as o
    return o


def to_nhwc(x):
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(DEV)


def from_nhwc(m, B, H, W):
    return m.view(B, H, W, -1).permute(0, 3, 1, 2).cpu()


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


CASES = [  # B, Cin, Cout, H, W, splitk
    (1, 64, 64, 16, 16, 1), (2, 32, 96, 16, 32, 1), (1, 96, 160, 24, 40, 1), (1, 40, 64, 17, 19, 1),   # ragged patch, Cin tail
    (2, 64, 128, 32, 32, 2), (1, 256, 64, 16, 16, 4), (1, 128, 192, 40, 24, 3), (1, 16, 64, 64, 64, 1),
    (1, 512, 512, 32, 32, 8), (3, 64, 64, 18, 30, 1),
    # >= 4 slabs of 16 channels per workgroup (also (1, 256, 64, 16, 16, 4), (1, 512, 512, 32, 32, 8) above): the software-pipelined K loop of the f16x3 kernel (odd / even slab counts, ragged
    # patches, split-K with a shorter last share)
    (1, 144, 64, 16, 16, 1), (1, 272, 96, 20, 36, 1), (2, 288, 64, 16, 16, 2), (1, 400, 64, 17, 16, 3), (1, 256, 256, 32, 48, 1)]


@pytest.mark.parametrize("mode,tol", [("bf16x6", 4e-6), ("bf16x3", 3e-4)])
@pytest.mark.parametrize("B,Cin,Cout,H,W,splitk", CASES)
def test_winograd_fwd_and_dgrad(ops, mode, tol, B, Cin, Cout, H, W, splitk):
    wfmt = ops.WFMT[mode]
    g = torch.Generator().manual_seed(B * 977 + Cin + 3 * Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    bias = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    assert ops.conv_winograd_ok(H, W, Cin, Cout, 3, wfmt)
    ref = (F.conv2d(x.double(), w.double(), bias.double(), padding=1) + res.double()).float()
    wf, wd = ops.pack_conv_weight_winograd(w.to(DEV), wfmt=wfmt)
    y = torch.full((B * H * W, Cout), float("nan"), device=DEV)
    ws = torch.empty(splitk * B * H * W * Cout, device=DEV) if splitk > 1 else None
    ops.conv2d(ops.Mat.of(to_nhwc(x)), wf, bias.to(DEV), ops.Mat.of(y), B, H, W, 3, res=ops.Mat.of(to_nhwc(res)),
               splitk=splitk, splitk_ws=ws, wfmt=wfmt | ops.WINOGRAD)
    e = relerr(from_nhwc(y, B, H, W), ref)
    assert e < tol, (mode, e)
    if not ops.conv_winograd_ok(H, W, Cout, Cin, 3, wfmt):     # the data-gradient has Cin output columns
        return
    dy = torch.randn(B, Cout, H, W, generator=g)
    xr = x.double().requires_grad_(True)
    (dref,) = torch.autograd.grad(F.conv2d(xr, w.double(), None, padding=1), xr, dy.double())
    dx = torch.full((B * H * W, Cin), float("nan"), device=DEV)
    ws2 = torch.empty(splitk * B * H * W * Cin, device=DEV) if splitk > 1 else None
    ops.conv2d(ops.Mat.of(to_nhwc(dy)), wd, None, ops.Mat.of(dx), B, H, W, 3, splitk=splitk, splitk_ws=ws2,
               wfmt=wfmt | ops.WINOGRAD)
    e = relerr(from_nhwc(dx, B, H, W), dref.float())
    assert e < tol, (mode, "dgrad", e)


def _f16x3_conv(ops, xm, img, bias, y, B, H, W, **kw):
    parts = torch.full((B * ops.MAXABS_PARTS,), float("nan"), device=DEV)
    ops.maxabs(xm, B, parts)
    ops.conv2d(xm, img, bias, y, B, H, W, 3, wfmt=ops.WFMT["f16x3"] | ops.WINOGRAD, x_maxabs=parts, **kw)


@pytest.mark.parametrize("B,Cin,Cout,H,W,splitk", CASES)
def test_f16x3_fwd_and_dgrad(ops, B, Cin, Cout, H, W, splitk):
    """"f16x3" Winograd images (round 3): operands scaled into the fp16 range by powers of two, two IEEE-half planes each,
    three fp16 MFMAs per product.  Same cases and the same fp64-referenced tolerance as bf16x6."""
    tol = 4e-6
    g = torch.Generator().manual_seed(B * 977 + Cin + 3 * Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    bias = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    ref = (F.conv2d(x.double(), w.double(), bias.double(), padding=1) + res.double()).float()
    wf, wd = ops.pack_conv_weight_winograd(w.to(DEV), wfmt=4)
    y = torch.full((B * H * W, Cout), float("nan"), device=DEV)
    ws = torch.empty(splitk * B * H * W * Cout, device=DEV) if splitk > 1 else None
    _f16x3_conv(ops, ops.Mat.of(to_nhwc(x)), wf, bias.to(DEV), ops.Mat.of(y), B, H, W, res=ops.Mat.of(to_nhwc(res)),
                splitk=splitk, splitk_ws=ws)
    e = relerr(from_nhwc(y, B, H, W), ref)
    assert e < tol, e
    if not ops.conv_winograd_ok(H, W, Cout, Cin, 3, 4):
        return
    dy = torch.randn(B, Cout, H, W, generator=g)
    xr = x.double().requires_grad_(True)
    (dref,) = torch.autograd.grad(F.conv2d(xr, w.double(), None, padding=1), xr, dy.double())
    dx = torch.full((B * H * W, Cin), float("nan"), device=DEV)
    ws2 = torch.empty(splitk * B * H * W * Cin, device=DEV) if splitk > 1 else None
    _f16x3_conv(ops, ops.Mat.of(to_nhwc(dy)), wd, None, ops.Mat.of(dx), B, H, W, splitk=splitk, splitk_ws=ws2)
    e = relerr(from_nhwc(dx, B, H, W), dref.float())
    assert e < tol, ("dgrad", e)


@pytest.mark.parametrize("xs,wsc", [(1e-6, 1.0), (3e4, 1.0), (1.0, 1e-5), (1.0, 2e3), (1e-20, 1e-12), (1e12, 1e9), (1e-30, 1e3)])
def test_f16x3_is_scale_invariant(ops, xs, wsc):
    """fp16 has 5 exponent bits; the kernel's power-of-two scaling of both operands must make the result independent of
    their magnitudes (activations of 1e-6 as in late-chain gradients, weights of 1e-5 ... 1e3): same relative error."""
    B, Cin, Cout, H, W = 1, 96, 64, 32, 16
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, Cin, H, W, generator=g) * xs
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9) * wsc
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    wf, _ = ops.pack_conv_weight_winograd(w.to(DEV), wfmt=4)
    y = torch.full((B * H * W, Cout), float("nan"), device=DEV)
    _f16x3_conv(ops, ops.Mat.of(to_nhwc(x)), wf, None, ops.Mat.of(y), B, H, W)
    assert relerr(from_nhwc(y, B, H, W), ref) < 4e-6


def test_f16x3_per_image_scale_outliers_zero_and_nan(ops):
    """The scale is per image: image 0 ~ 1e-4, image 1 ~ 1e3 with a 1e3 x outlier pixel, image 2 all zeros.  Every image
    keeps its own relative accuracy (error measured per image against ITS maximum); an all-zero image gives exactly the bias;
    a NaN in one image poisons that image's output only."""
    B, Cin, Cout, H, W = 3, 64, 64, 32, 32
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, Cin, H, W, generator=g)
    x[0] *= 1e-4
    x[1] *= 1e3
    x[1, 5, 7, 9] = 1e6
    x[2] = 0
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    bias = torch.randn(Cout, generator=g) * 1e-5
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    wf, _ = ops.pack_conv_weight_winograd(w.to(DEV), wfmt=4)
    y = torch.full((B * H * W, Cout), float("nan"), device=DEV)
    _f16x3_conv(ops, ops.Mat.of(to_nhwc(x)), wf, bias.to(DEV), ops.Mat.of(y), B, H, W)
    out = from_nhwc(y, B, H, W).double() - bias.double()[None, :, None, None]
    for b in range(2):
        # image 1: the outlier is 1000 x the rest, so elements 2^10 below the maximum keep ~12 + 11 bits: 1e-6 of the max
        assert float((out[b] - ref[b]).abs().max() / ref[b].abs().max()) < 4e-6, b
    far = torch.ones(H, W, dtype=torch.bool)
    far[5:10, 7:12] = False                      # away from the outlier the output is O(1e3), the error budget is relative
    e_far = float((out[1] - ref[1])[:, far].abs().max() / ref[1][:, far].abs().max())   # to the IMAGE maximum 1e6 x 2^-23
    assert e_far < 1e-3, e_far
    assert torch.equal(out[2].float() + bias[:, None, None], from_nhwc(y, B, H, W)[2]) and float(out[2].abs().max()) < 1e-12
    xn = x.clone()
    xn[0, 3, 4, 5] = float("nan")
    _f16x3_conv(ops, ops.Mat.of(to_nhwc(xn)), wf, bias.to(DEV), ops.Mat.of(y), B, H, W)
    o2 = from_nhwc(y, B, H, W)
    assert torch.isnan(o2[0]).all() and torch.isfinite(o2[1]).all() and torch.isfinite(o2[2]).all()


def test_f16x3_denormal_inputs_stay_finite(ops):
    """An image whose every activation is an fp32 denormal (max |x| = 1e-40): the power-of-two scale is clamped, the
    output is finite and what an fp32 convolution gives to its own resolution (~0 next to the bias)."""
    B, Cin, Cout, H, W = 1, 64, 64, 16, 16
    g = torch.Generator().manual_seed(8)
    x = torch.rand(B, Cin, H, W, generator=g) * 1e-40
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    bias = torch.randn(Cout, generator=g)
    wf, _ = ops.pack_conv_weight_winograd(w.to(DEV), wfmt=4)
    y = torch.full((B * H * W, Cout), float("nan"), device=DEV)
    _f16x3_conv(ops, ops.Mat.of(to_nhwc(x)), wf, bias.to(DEV), ops.Mat.of(y), B, H, W)
    out = from_nhwc(y, B, H, W)
    assert torch.isfinite(out).all()
    assert float((out - bias[None, :, None, None]).abs().max()) < 1e-30


def test_maxabs_partials(ops):
    """osm_maxabs: OSM_MAXABS_PARTS partial maxima per image (every slot rewritten), dense and strided inputs."""
    g = torch.Generator().manual_seed(1)
    for B, rows, C, ld in [(1, 256, 64, 64), (3, 1000, 36, 36), (2, 4096, 128, 160), (1, 65536, 256, 256)]:
        big = torch.randn(B * rows, ld, generator=g).to(DEV)
        big[:, C:] = 1e9                                        # columns outside the view must not be read
        m = ops.Mat.of(big).cols_slice(0, C) if ld != C else ops.Mat.of(big)
        parts = torch.full((B * ops.MAXABS_PARTS,), float("nan"), device=DEV)
        ops.maxabs(m, B, parts)
        got = parts.view(B, ops.MAXABS_PARTS).max(1).values
        want = big[:, :C].abs().view(B, rows, C).amax((1, 2))
        assert torch.equal(got, want), (B, rows, C, ld)


@pytest.mark.parametrize("B,Cin,Cout,H,W,splitk", [(1, 64, 64, 16, 16, 1), (2, 96, 160, 16, 24, 1), (1, 256, 512, 64, 64, 1),
                                                    (3, 40, 36, 8, 16, 1), (1, 512, 128, 32, 32, 4), (2, 1024, 64, 16, 8, 3)])
def test_f16x3_1x1_fwd_and_dgrad(ops, B, Cin, Cout, H, W, splitk):
    """The 1x1 layers' f16x3 image (late round 3: the tap-chunked kernel with two half planes, igemm_bf16s_kernel<1, 2, true>):
    forward + residual + bias and the data-gradient against fp64, the tolerance of bf16x6; per-image scales (image 1 is
    1e4 x image 0); channel tails; split-K."""
    tol = 4e-6
    g = torch.Generator().manual_seed(B * 31 + Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    if B > 1:
        x[1] *= 1e4
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin) * 3e-3
    bias = torch.randn(Cout, generator=g) * 1e-3
    res = torch.randn(B, Cout, H, W, generator=g) * 1e-3
    ref = F.conv2d(x.double(), w.double(), bias.double()) + res.double()
    wf, wd = ops.pack_conv_weight(w.to(DEV), wfmt=4)
    xm = ops.Mat.of(to_nhwc(x))
    parts = torch.full((B * ops.MAXABS_PARTS,), float("nan"), device=DEV)
    ops.maxabs(xm, B, parts)
    y = torch.full((B * H * W, Cout), float("nan"), device=DEV)
    ws = torch.empty(splitk * B * H * W * Cout, device=DEV) if splitk > 1 else None
    ops.conv2d(xm, wf, bias.to(DEV), ops.Mat.of(y), B, H, W, 1, res=ops.Mat.of(to_nhwc(res)), splitk=splitk, splitk_ws=ws,
               wfmt=4, x_maxabs=parts)
    out = from_nhwc(y, B, H, W).double()
    for b in range(B):          # every image against ITS OWN maximum
        assert float((out[b] - ref[b]).abs().max() / ref[b].abs().max()) < tol, b
    dy = torch.randn(B, Cout, H, W, generator=g) * 1e-5
    dref = F.conv2d(dy.double(), w.double().permute(1, 0, 2, 3))
    dym = ops.Mat.of(to_nhwc(dy))
    ops.maxabs(dym, B, parts)
    dx = torch.full((B * H * W, Cin), float("nan"), device=DEV)
    ws2 = torch.empty(splitk * B * H * W * Cin, device=DEV) if splitk > 1 else None
    ops.conv2d(dym, wd, None, ops.Mat.of(dx), B, H, W, 1, splitk=splitk, splitk_ws=ws2, wfmt=4, x_maxabs=parts)
    assert relerr(from_nhwc(dx, B, H, W), dref.float()) < tol
    # accumulate epilogue + a NaN poisons its own image only
    c0 = torch.randn(B, Cin, H, W, generator=g) * float(dref.abs().max())
    acc = to_nhwc(c0).clone()
    ops.conv2d(dym, wd, None, ops.Mat.of(acc), B, H, W, 1, splitk=splitk, splitk_ws=ws2, wfmt=4, x_maxabs=parts, accumulate=True)
    assert relerr(from_nhwc(acc, B, H, W), (dref + c0.double()).float()) < tol
    if B > 1:
        xn = x.clone()
        xn[0, 1, 2, 3] = float("nan")
        xnm = ops.Mat.of(to_nhwc(xn))
        ops.maxabs(xnm, B, parts)
        ops.conv2d(xnm, wf, None, ops.Mat.of(y), B, H, W, 1, splitk=splitk, splitk_ws=ws, wfmt=4, x_maxabs=parts)
        o2 = from_nhwc(y, B, H, W)
        assert torch.isnan(o2[0]).any() and torch.isfinite(o2[1:]).all()


@pytest.mark.parametrize("B,Cin,Cout,H,W,splitk", [(1, 1024, 1024, 8, 8, 32), (2, 256, 320, 8, 8, 4), (1, 96, 72, 12, 9, 1),
                                                    (3, 64, 160, 10, 24, 2), (1, 256, 8, 16, 32, 1), (2, 128, 128, 8, 15, 3),
                                                    (1, 2048, 1024, 8, 8, 16)])
def test_f16x3_direct_3x3_fwd_and_dgrad(ops, B, Cin, Cout, H, W, splitk):
    """Round 5: 3x3 layers on images smaller than the Winograd tile (the sampler's 8 x 8 level) -- the halo-tile kernel with two
    half planes (conv3_halo_bf16s_kernel<2, ..., HP>): forward + residual + bias and the data-gradient against fp64 at the tolerance
    of bf16x6; per-image scales (image 1 is 1e4 x image 0); 8-wide and 16-wide patches, ragged edges, channel tails, the narrow
    (<= 32 columns) instance, split-K partials."""
    tol = 4e-6
    g = torch.Generator().manual_seed(B * 131 + Cin + Cout + H + W)
    x = torch.randn(B, Cin, H, W, generator=g)
    if B > 1:
        x[1] *= 1e4
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9) * 3e-3
    bias = torch.randn(Cout, generator=g) * 1e-3
    res = torch.randn(B, Cout, H, W, generator=g) * 1e-3
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1) + res.double()
    wf, wd = ops.pack_conv_weight(w.to(DEV), wfmt=4)
    xm = ops.Mat.of(to_nhwc(x))
    parts = torch.full((B * ops.MAXABS_PARTS,), float("nan"), device=DEV)
    ops.maxabs(xm, B, parts)
    y = torch.full((B * H * W, Cout), float("nan"), device=DEV)
    ws = torch.empty(splitk * B * H * W * Cout, device=DEV) if splitk > 1 else None
    ops.conv2d(xm, wf, bias.to(DEV), ops.Mat.of(y), B, H, W, 3, res=ops.Mat.of(to_nhwc(res)), splitk=splitk, splitk_ws=ws,
               wfmt=4, x_maxabs=parts)
    out = from_nhwc(y, B, H, W).double()
    for b in range(B):          # every image against ITS OWN maximum
        assert float((out[b] - ref[b]).abs().max() / ref[b].abs().max()) < tol, b
    dy = torch.randn(B, Cout, H, W, generator=g) * 1e-5
    xr = x.double().requires_grad_(True)
    (dref,) = torch.autograd.grad(F.conv2d(xr, w.double(), None, padding=1), xr, dy.double())
    dym = ops.Mat.of(to_nhwc(dy))
    ops.maxabs(dym, B, parts)
    dx = torch.full((B * H * W, Cin), float("nan"), device=DEV)
    ws2 = torch.empty(splitk * B * H * W * Cin, device=DEV) if splitk > 1 else None
    ops.conv2d(dym, wd, None, ops.Mat.of(dx), B, H, W, 3, splitk=splitk, splitk_ws=ws2, wfmt=4, x_maxabs=parts)
    assert relerr(from_nhwc(dx, B, H, W), dref.float()) < tol
    if B > 1:                   # a NaN poisons its own image only
        xn = x.clone()
        xn[0, 1, 2, 3] = float("nan")
        xnm = ops.Mat.of(to_nhwc(xn))
        ops.maxabs(xnm, B, parts)
        ops.conv2d(xnm, wf, None, ops.Mat.of(y), B, H, W, 3, splitk=splitk, splitk_ws=ws, wfmt=4, x_maxabs=parts)
        o2 = from_nhwc(y, B, H, W)
        assert torch.isnan(o2[0]).any() and torch.isfinite(o2[1:]).all()


def test_f16x3_1x1_refusals(ops):
    from osmosis_diffusion_code_amd._lib import OsmosisHipError
    w = torch.randn(64, 64, 1, 1, device=DEV)
    wf, _ = ops.pack_conv_weight(w, wfmt=4)
    parts = torch.zeros(ops.MAXABS_PARTS, device=DEV)
    x, y = torch.randn(256, 64, device=DEV), torch.empty(256, 64, device=DEV)
    with pytest.raises(OsmosisHipError):        # no x_maxabs
        ops.conv2d(ops.Mat.of(x), wf, None, ops.Mat.of(y), 1, 16, 16, 1, wfmt=4)
    x2, y2 = torch.randn(100, 64, device=DEV), torch.empty(100, 64, device=DEV)
    with pytest.raises(OsmosisHipError, match="one image per tile"):      # 10 x 10 pixels: a 128-row tile would straddle two images
        ops.conv2d(ops.Mat.of(x2), wf, None, ops.Mat.of(y2), 1, 10, 10, 1, wfmt=4, x_maxabs=parts)
    w3, _ = ops.pack_conv_weight(torch.randn(64, 64, 3, 3, device=DEV), wfmt=4)
    x3, y3 = torch.randn(16, 64, device=DEV), torch.empty(16, 64, device=DEV)
    with pytest.raises(OsmosisHipError, match="H, W >= 8"):      # the direct 3x3 f16x3 kernel is the halo-tile kernel
        ops.conv2d(ops.Mat.of(x3), w3, None, ops.Mat.of(y3), 1, 4, 4, 3, wfmt=4, x_maxabs=parts)


def test_f16x3_refusals(ops):
    from osmosis_diffusion_code_amd._lib import OsmosisHipError
    w = torch.randn(64, 64, 3, 3, device=DEV)
    wf, _ = ops.pack_conv_weight_winograd(w, wfmt=4)
    x = torch.randn(256, 64, device=DEV)
    y = torch.empty(256, 64, device=DEV)
    with pytest.raises(OsmosisHipError):        # no x_maxabs
        ops.conv2d(ops.Mat.of(x), wf, None, ops.Mat.of(y), 1, 16, 16, 3, wfmt=4 | ops.WINOGRAD)
    with pytest.raises(OsmosisHipError):        # the direct f16x3 kernel needs x_maxabs as well
        ops.conv2d(ops.Mat.of(x), wf, None, ops.Mat.of(y), 1, 16, 16, 3, wfmt=4)
    parts = torch.zeros(ops.MAXABS_PARTS, device=DEV)
    with pytest.raises(OsmosisHipError, match="fused GroupNorm"):      # the range handed over is x's, not GN(x)'s
        ops.conv2d(ops.Mat.of(x), wf, None, ops.Mat.of(y), 1, 16, 16, 3, wfmt=4 | ops.WINOGRAD, x_maxabs=parts,
                   gn_table=torch.zeros(4 * 64, device=DEV))


def test_winograd_matches_direct_kernel_and_accumulates(ops):
    """Same layer through both kernels: they agree to fp32 rounding; accumulate adds into y; strided output view."""
    B, Cin, Cout, H, W = 1, 128, 128, 48, 32
    g = torch.Generator().manual_seed(5)
    x = to_nhwc(torch.randn(B, Cin, H, W, generator=g))
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    wf, _ = ops.pack_conv_weight(w, wfmt=3)
    uf, _ = ops.pack_conv_weight_winograd(w, wfmt=3)
    y0 = torch.empty(B * H * W, Cout, device=DEV)
    ops.conv2d(ops.Mat.of(x), wf, b, ops.Mat.of(y0), B, H, W, 3, wfmt=3)
    big = torch.full((B * H * W, Cout + 64), 7.0, device=DEV)          # y is a column slice of a wider tensor
    y1 = ops.Mat.of(big).cols_slice(32, 32 + Cout)
    ops.conv2d(ops.Mat.of(x), uf, b, y1, B, H, W, 3, wfmt=3 | ops.WINOGRAD)
    assert relerr(big[:, 32:32 + Cout], y0) < 2e-6
    assert torch.all(big[:, :32] == 7.0) and torch.all(big[:, 32 + Cout:] == 7.0)
    ops.conv2d(ops.Mat.of(x), uf, None, y1, B, H, W, 3, wfmt=3 | ops.WINOGRAD, accumulate=True)
    ref = 2 * y0 - b
    assert relerr(big[:, 32:32 + Cout], ref) < 2e-6


@pytest.mark.parametrize("film,B,C,Cout,H,W,splitk", [(True, 2, 64, 96, 16, 24, 1), (False, 1, 96, 64, 19, 17, 1),
                                                     (True, 1, 256, 128, 32, 32, 2)])
def test_winograd_with_fused_group_norm_input(ops, film, B, C, Cout, H, W, splitk):
    g = torch.Generator().manual_seed(C + H)
    G, HW = 32, H * W
    x = torch.randn(B, C, H, W, generator=g) * 1.3 + 0.2
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    e = 0.3 * torch.randn(B, 2 * C, generator=g) if film else None
    w = torch.randn(Cout, C, 3, 3, generator=g) / math.sqrt(9 * C)
    bias = torch.randn(Cout, generator=g)
    y = F.group_norm(x.double(), G, gamma.double(), beta.double(), eps=1e-5)
    if film:
        y = y * (1 + e[:, :C, None, None].double()) + e[:, C:, None, None].double()
    ref = F.conv2d(F.silu(y), w.double(), bias.double(), padding=1).float()
    xm = ops.Mat.of(to_nhwc(x))
    part = torch.empty(B * ops.gn_nchunk(HW) * G * 2, device=DEV)
    stats = torch.empty(B * G * 2, device=DEV)
    table = torch.empty(B * 4 * C, device=DEV)
    ops.gn_prep(xm, B, HW, G, part, stats, gamma.to(DEV), beta.to(DEV), table, film=e.to(DEV) if film else None)
    uf, _ = ops.pack_conv_weight_winograd(w.to(DEV), wfmt=3)
    out = torch.full((B * HW, Cout), float("nan"), device=DEV)
    ws = torch.empty(splitk * B * HW * Cout, device=DEV) if splitk > 1 else None
    ops.conv2d(xm, uf, bias.to(DEV), ops.Mat.of(out), B, H, W, 3, wfmt=3 | ops.WINOGRAD, gn_table=table, gn_silu=True,
               splitk=splitk, splitk_ws=ws)
    assert relerr(from_nhwc(out, B, H, W), ref) < 6e-6


@pytest.mark.parametrize("B,Cin,Cout,H,W,sk", [(1, 256, 64, 32, 32, 4), (2, 64, 96, 24, 40, 1), (1, 128, 128, 64, 64, 1)])
def test_winograd_column_sums(ops, B, Cin, Cout, H, W, sk):
    """Column sums of the output (stat_mode 1) from the kernel's own epilogue (no split-K: one chunk per 16 x 16 patch)
    or from the split-K combine -- same contract as the direct kernel."""
    wfmt = 3 | ops.WINOGRAD
    g = torch.Generator().manual_seed(9 + H)
    x = to_nhwc(torch.randn(B, Cin, H, W, generator=g))
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).to(DEV)
    bias = torch.randn(Cout, generator=g).to(DEV)
    res = to_nhwc(torch.randn(B, Cout, H, W, generator=g))
    uf, _ = ops.pack_conv_weight_winograd(w, wfmt=3)
    nch = ops.conv_stat_chunks(B, H, W, Cin, Cout, 3, wfmt, sk)
    assert nch == (H * W // 8 if sk > 1 else ((H + 15) // 16) * ((W + 15) // 16))
    y = torch.empty(B * H * W, Cout, device=DEV)
    cs = torch.full((B * nch * 2 * Cout,), float("nan"), device=DEV)
    ws = torch.empty(sk * B * H * W * Cout, device=DEV) if sk > 1 else None
    ops.conv2d(ops.Mat.of(x), uf, bias, ops.Mat.of(y), B, H, W, 3, res=ops.Mat.of(res), splitk=sk, splitk_ws=ws, wfmt=wfmt,
               colsum=cs, stat_mode=1)
    c = cs.view(B, nch, 2, Cout).double().sum(1).cpu()
    yb = y.view(B, H * W, Cout).double().cpu()
    assert relerr(c[:, 0], yb.sum(1)) < 1e-5
    assert relerr(c[:, 1], (yb ** 2).sum(1)) < 1e-5


def test_winograd_backward_sums_for_group_norm(ops):
    """stat_mode 2: y is the gradient w.r.t. SiLU(GN(x)); the epilogue emits sum(dxh), sum(dxh xh) per column."""
    B, Cin, Cout, H, W, G = 2, 64, 64, 32, 16, 32
    wfmt = 3 | ops.WINOGRAD
    g = torch.Generator().manual_seed(3)
    dy = to_nhwc(torch.randn(B, Cin, H, W, generator=g))
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).to(DEV)
    uf, _ = ops.pack_conv_weight_winograd(w, wfmt=3)
    xg = torch.randn(B, Cout, H, W, generator=g) * 1.5 + 0.3           # the GroupNorm input whose backward is reduced
    xm = ops.Mat.of(to_nhwc(xg))
    gamma, beta = (1 + 0.1 * torch.randn(Cout, generator=g)).to(DEV), (0.1 * torch.randn(Cout, generator=g)).to(DEV)
    HW = H * W
    part = torch.empty(B * ops.gn_nchunk(HW) * G * 2, device=DEV)
    stats = torch.empty(B * G * 2, device=DEV)
    table = torch.empty(B * 4 * Cout, device=DEV)
    ops.gn_prep(xm, B, HW, G, part, stats, gamma, beta, table)
    nch = ops.conv_stat_chunks(B, H, W, Cin, Cout, 3, wfmt, 1)
    y = torch.empty(B * HW, Cout, device=DEV)
    cs = torch.full((B * nch * 2 * Cout,), float("nan"), device=DEV)
    ops.conv2d(ops.Mat.of(dy), uf, None, ops.Mat.of(y), B, H, W, 3, wfmt=wfmt, colsum=cs, stat_mode=2, stat_x=xm,
               stat_table=table, stat_silu=True)
    t = table.view(B, 4, Cout).double().cpu()
    mean, rstd, gg, bb = t[:, 0:1], t[:, 1:2], t[:, 2:3], t[:, 3:4]
    xb = xm.t.view(B, HW, Cout).double().cpu()
    xh = (xb - mean) * rstd
    z = xh * gg + bb
    sig = torch.sigmoid(z)
    dxh = y.view(B, HW, Cout).double().cpu() * (sig * (1 + z * (1 - sig))) * gg
    c = cs.view(B, nch, 2, Cout).double().sum(1).cpu()
    assert relerr(c[:, 0], dxh.sum(1)) < 2e-5
    assert relerr(c[:, 1], (dxh * xh).sum(1)) < 2e-5


def test_winograd_refusals(ops):
    from osmosis_diffusion_code_amd._lib import OsmosisHipError
    assert not ops.conv_winograd_ok(8, 16, 64, 64, 3, 3)        # H < 16
    assert not ops.conv_winograd_ok(16, 16, 64, 32, 3, 3)       # Cout < 64
    assert not ops.conv_winograd_ok(16, 16, 64, 72, 3, 3)       # Cout % 32
    assert not ops.conv_winograd_ok(16, 16, 4, 64, 3, 3)        # Cin < 16
    assert not ops.conv_winograd_ok(16, 16, 64, 64, 1, 3)       # 1x1
    assert not ops.conv_winograd_ok(16, 16, 64, 64, 3, 0)       # exact-f32 images
    x = torch.randn(64, 64, device=DEV)
    y = torch.empty(64, 64, device=DEV)
    w = torch.randn(64, 64, 3, 3, device=DEV)
    uf, _ = ops.pack_conv_weight_winograd(w, wfmt=3)
    with pytest.raises(OsmosisHipError, match="Winograd"):
        ops.conv2d(ops.Mat.of(x), uf, None, ops.Mat.of(y), 1, 8, 8, 3, wfmt=3 | ops.WINOGRAD)


# ----------------------------------------------------------------------------- fp16 family (use_fp16)
def to_nhwc_h(x):
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(DEV, torch.float16)


@pytest.mark.parametrize("B,Cin,Cout,H,W,splitk,gn", [(1, 64, 64, 16, 16, 1, False), (2, 32, 96, 16, 32, 1, False),
                                                     (1, 96, 160, 24, 40, 1, True), (1, 256, 64, 32, 32, 4, False),
                                                     (1, 128, 128, 64, 64, 1, True), (1, 40, 64, 17, 19, 1, False)])
def test_winograd_fp16_family(ops, B, Cin, Cout, H, W, splitk, gn):
    """Half activations, one fp16 plane per operand (wfmt 1 | WINOGRAD), transforms in fp32: against an fp64 convolution
    of the SAME half-rounded inputs / weights.  Stated tolerance: V and U are each rounded once more to half (2^-11
    relative) and the output transform adds 9 of the 16 products, so the result is within ~3x the rounding of one
    half-stored tensor (measured 3-6e-4 of the output's max-abs; the direct fp16 kernel: 3e-4)."""
    g = torch.Generator().manual_seed(B * 31 + Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    bias = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    xh, rh = to_nhwc_h(x), to_nhwc_h(res)
    xin = xh.float().cpu().view(B, H, W, Cin).permute(0, 3, 1, 2).double()
    G = 32
    table = None
    if gn:
        gamma, beta = 1 + 0.1 * torch.randn(Cin, generator=g), 0.1 * torch.randn(Cin, generator=g)
        part = torch.empty(B * ops.gn_nchunk(H * W) * G * 2, device=DEV)
        stats = torch.empty(B * G * 2, device=DEV)
        table = torch.empty(B * 4 * Cin, device=DEV)
        ops.gn_prep(ops.Mat.of(xh), B, H * W, G, part, stats, gamma.to(DEV), beta.to(DEV), table)
        xin = F.silu(F.group_norm(xin, G, gamma.double(), beta.double(), eps=1e-5))
    ref = F.conv2d(xin, w.half().double(), bias.double(), padding=1) + \
        rh.float().cpu().view(B, H, W, Cout).permute(0, 3, 1, 2).double()
    assert ops.conv_winograd_ok(H, W, Cin, Cout, 3, 1)
    uf, ud = ops.pack_conv_weight_winograd(w.to(DEV), wfmt=1)
    y = torch.full((B * H * W, Cout), float("nan"), device=DEV, dtype=torch.float16)
    ws = torch.empty(splitk * B * H * W * Cout, device=DEV) if splitk > 1 else None
    ops.conv2d(ops.Mat.of(xh), uf, bias.to(DEV), ops.Mat.of(y), B, H, W, 3, res=ops.Mat.of(rh), splitk=splitk, splitk_ws=ws,
               wfmt=1 | ops.WINOGRAD, gn_table=table, gn_silu=True)
    e = relerr(from_nhwc(y.float(), B, H, W), ref.float())
    assert e < 1.5e-3, e
    if gn or not ops.conv_winograd_ok(H, W, Cout, Cin, 3, 1):
        return
    dy = torch.randn(B, Cout, H, W, generator=g)
    dyh = to_nhwc_h(dy)
    xr = torch.zeros(B, Cin, H, W, dtype=torch.float64, requires_grad=True)
    (dref,) = torch.autograd.grad(F.conv2d(xr, w.half().double(), None, padding=1), xr,
                                  dyh.float().cpu().view(B, H, W, Cout).permute(0, 3, 1, 2).double())
    dx = torch.full((B * H * W, Cin), float("nan"), device=DEV, dtype=torch.float16)
    ws2 = torch.empty(splitk * B * H * W * Cin, device=DEV) if splitk > 1 else None
    ops.conv2d(ops.Mat.of(dyh), ud, None, ops.Mat.of(dx), B, H, W, 3, splitk=splitk, splitk_ws=ws2, wfmt=1 | ops.WINOGRAD)
    e = relerr(from_nhwc(dx.float(), B, H, W), dref.float())
    assert e < 1.5e-3, ("dgrad", e)


def test_full_size_unet_winograd_vs_direct_kernel(monkeypatch):
    """The real 552.8 M-parameter UNet at 1x4x256x256, forward and input gradient, with the Winograd kernel on its 3x3
    layers (default) and with the direct halo-tile kernel everywhere (OSM_WINOGRAD=0): same network, two algorithms for
    88 of its convolutions -- they must agree to fp32 rounding, in both Winograd arithmetics: bf16x6 (6 bf16 MFMAs per
    product) and f16x3 (round 3, the default: operands scaled into the fp16 range, 3 fp16 MFMAs per product)."""
    from oracle import unet_ref as U
    from osmosis_diffusion_code_amd.guided_diffusion.unet import create_model
    kw = dict(image_size=256, num_channels=256, num_res_blocks=2, channel_mult="", learn_sigma=True,
              class_cond=False, use_checkpoint=False, attention_resolutions="32, 16, 8", num_heads=4,
              num_head_channels=64, num_heads_upsample=-1, use_scale_shift_norm=True, dropout=0.0,
              resblock_updown=True, use_fp16=False, use_new_attention_order=False, model_path="",
              pretrain_model="osmosis")
    cfg = U.UNetConfig.from_create_model_kwargs(**kw)
    sd = U.seeded_state_dict(cfg, 1234)
    g = torch.Generator().manual_seed(0)
    x = 0.7 * torch.randn(1, 4, 256, 256, generator=g)
    t = torch.tensor([37.0])
    w = torch.randn(1, 8, 256, 256, generator=g)
    outs = {}
    for mode, flag in (("f16x3", "1"), ("bf16x6", "1"), ("bf16x6", "0")):
        monkeypatch.setenv("OSM_WINOGRAD", flag)
        m = create_model(**kw)
        m.load_state_dict(sd, strict=True)
        m.conv_mode = mode
        m = m.to(DEV).eval()
        xd = x.to(DEV).requires_grad_(True)
        yd = m(xd, t.to(DEV))
        (dxd,) = torch.autograd.grad((yd * w.to(DEV)).sum(), xd)
        outs[(mode, flag)] = (yd.detach().cpu(), dxd.cpu())
        del m
        torch.cuda.empty_cache()
    direct = outs[("bf16x6", "0")]
    for key in (("f16x3", "1"), ("bf16x6", "1")):
        ey = relerr(outs[key][0], direct[0])
        ed = relerr(outs[key][1], direct[1])
        print("Winograd", key[0], "vs direct bf16x6, full size: y", ey, "dx", ed)
        assert ey < 2e-5 and ed < 2e-5, key


def test_f16x3_range_handover_is_checked_launch_by_launch(monkeypatch):
    """OSM_CHECK_XMAX=1: while the plans are recorded (kernels execute), every f16x3 convolution that takes the per-image max |x|
    from the pass that produced x compares that bound with max |x| of the tensor it really reads (engine._xmax_debug_check).
    A model with attention blocks between ResBlocks (a second kind of writer of the gradient buffers), skip 1x1 convolutions
    at >= 64 x 64 and up / down blocks; forward + input gradient must pass the check and equal the unchecked run."""
    from oracle import unet_ref as U
    from osmosis_diffusion_code_amd.guided_diffusion.unet import create_model
    kw = dict(image_size=256, num_channels=64, num_res_blocks=2, channel_mult="1,2,2", learn_sigma=True,
              attention_resolutions="128,64", num_heads=4, num_head_channels=64, use_scale_shift_norm=True,
              resblock_updown=True, pretrain_model="osmosis")
    cfg = U.UNetConfig.from_create_model_kwargs(**kw)
    sd = U.seeded_state_dict(cfg, 77)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 4, 128, 128, generator=g)
    x[1] *= 30.0                                         # per-image ranges differ
    t = torch.tensor([11.0, 640.0])
    w = torch.randn(2, 8, 128, 128, generator=g)
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("OSM_CHECK_XMAX", flag)
        m = create_model(**kw)
        m.load_state_dict(sd, strict=True)
        m.conv_mode = "f16x3"
        m = m.to(DEV).eval()
        xd = x.to(DEV).requires_grad_(True)
        yd = m(xd, t.to(DEV))
        (dxd,) = torch.autograd.grad((yd * w.to(DEV)).sum(), xd)
        eng = next(iter(m._engines.values()))
        assert eng._check_xmax == (flag == "1")
        outs.append((yd.detach().cpu(), dxd.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][1]).all()
