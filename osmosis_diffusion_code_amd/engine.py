"""UNetEngine: the reference UNet (guided_diffusion/unet.py:713-742) as a recorded plan of gfx950
kernel launches over persistent NHWC buffers, forward and data-gradient.

Design (MI355X-first, not a module-by-module translation):
  * activations are NHWC matrices [B*H*W][C] in HBM; every kernel takes a row stride, so the
    UNet's skip concatenations (`th.cat([h, hs.pop()], dim=1)`, unet.py:739) are ZERO-COPY: an
    input block writes its output straight into the right-hand columns of the buffer the matching
    output block will read, and the split of the concat gradient in backward is a pair of views;
  * weights are repacked once per engine into two images per conv -- [tap][Cout][Cin] for forward
    and the flipped/transposed [tap][Cin][Cout] for the data gradient -- so forward and backward
    run the SAME implicit-GEMM kernel (csrc/igemm.hip); 288 GB of HBM makes the 2x weight copy and
    the un-recomputed activation stash (~8 GB / image at 256x256) a non-issue;
  * guidance needs d(out)/d(x) only (condition_methods.py:186-194): no weight gradients exist;
  * attention (T <= 1024, 64-wide heads) keeps P = softmax(QK^T) instead of re-running the block
    in backward (the reference checkpoints it, unet.py:376 / nn.py:142-170 -- same values);
  * the first forward/backward executes the kernels while a Recorder captures the (function, args)
    list; later steps replay that list (no Python per-op work) and can be captured in a hipGraph.

Per ResBlock (unet.py:315-335):   GN+SiLU -> [pool|upsample] -> conv3x3 -> GN*(1+s)+t, SiLU -> conv3x3 (+skip)
Per AttentionBlock (:378-433):    GN -> qkv 1x1 -> per-head softmax(q k^T / sqrt(ch)) v -> proj 1x1 (+x)
"""
import contextlib
import math
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import ops
from ._lib import OsmosisHipError, Recorder, current_stream_ptr
from .ops import Mat

G = 32  # GroupNorm32 groups (nn.py:93-100)


def _winograd_on() -> bool:
    """OSM_WINOGRAD=0: direct halo-tile kernel on every 3x3 layer (A/B measurements)."""
    return os.environ.get("OSM_WINOGRAD", "1") != "0"


class _Conv:
    def __init__(self, slot, dev, wfmt=0):
        """wfmt: ops.WFMT code of the model's conv arithmetic.  "f16x3" (4) exists for Winograd images only: such a model keeps
        bf16x6 (3) images for every layer / resolution the Winograd kernel does not serve."""
        w = slot.weight.detach().to(dev, torch.float32)
        self.cout, self.cin = w.shape[0], w.shape[1]
        self.k = w.shape[2] if w.dim() == 4 else 1
        self.wwfmt = wfmt                          # format of the Winograd-domain images
        wfmt = 3 if wfmt == 4 else wfmt
        self.wfmt = wfmt
        self.wf, self.wd = ops.pack_conv_weight(w, wfmt=wfmt)
        # Winograd F(2x2, 3x3) images next to the direct ones (the layer picks per (H, W): 16 x 16 and larger)
        self.wwf = self.wwd = None
        # (the fp16 family's instance exists and is tested, but loses to the direct fp16 kernel -- one MFMA per product leaves
        # the transform as the whole cost: 10.1 vs 7.3 ms of 3x3 time per step -- so it is opt-in: OSM_WINOGRAD_F16=1)
        if self.k == 3 and (wfmt in (2, 3) or (wfmt == 1 and os.environ.get("OSM_WINOGRAD_F16", "0") == "1")) \
                and _winograd_on() and ops.conv_winograd_ok(16, 16, self.cin, self.cout, 3, wfmt) \
                and ops.conv_winograd_ok(16, 16, self.cout, self.cin, 3, wfmt):
            self.wwf, self.wwd = ops.pack_conv_weight_winograd(w, wfmt=self.wwfmt)
        # 1x1 layers of an f16x3 model: a two-half-plane image next to the bf16x6 one (used at H W >= 4096, where the kernel is
        # bound by its matrix work and the range of the input comes for free from the pass that wrote or normalised it)
        self.wf16 = self.wd16 = None
        if self.k == 1 and self.wwfmt == 4 and os.environ.get("OSM_F16X3_1X1", "1") != "0":
            self.wf16, self.wd16 = ops.pack_conv_weight(w, wfmt=4)
        # 3x3 layers of an f16x3 model on images SMALLER than the Winograd tile (the 8 x 8 level): the direct halo-tile kernel with
        # two-half-plane images (3 instead of 6 MFMAs per product, 4 instead of 6 bytes of weight stream per weight).  Which layers
        # meet such an image depends on the input size, so the image is packed on first use (direct_f16x3)
        self._slot = slot if (self.k == 3 and self.wwfmt == 4 and os.environ.get("OSM_F16X3_DIRECT", "1") != "0") else None
        self.wf3 = self.wd3 = None
        self.b = slot.bias.detach().to(dev, torch.float32).contiguous()

    def direct_f16x3(self, dgrad: bool):
        """The direct (non-Winograd) f16x3 image of a 3x3 layer, or None when the layer has none (other arithmetic / switched off)."""
        if self._slot is None:
            return None
        if self.wf3 is None:
            w = self._slot.weight.detach().to(self.b.device, torch.float32)
            with Recorder.suspended():               # one-off: not part of the launch plan being recorded
                self.wf3, self.wd3 = ops.pack_conv_weight(w, wfmt=4)
        return self.wd3 if dgrad else self.wf3


class _Norm:
    def __init__(self, slot, dev):
        self.g = slot.weight.detach().to(dev, torch.float32).contiguous()
        self.b = slot.bias.detach().to(dev, torch.float32).contiguous()


class _Res:
    def __init__(self, p, dev, wfmt=0):
        self.cin, self.cout, self.up, self.down = p.cin, p.cout, p.up, p.down
        self.scale_shift = bool(getattr(p, "scale_shift", True))     # False: h + emb_out before the second GroupNorm (unet.py:331-332)
        self.n1 = _Norm(p.in_layers.at(0), dev)
        self.c1 = _Conv(p.in_layers.at(2), dev, wfmt)
        e = p.emb_layers.at(1)
        self.ew = e.weight.detach().to(dev, torch.float32).contiguous()
        self.eb = e.bias.detach().to(dev, torch.float32).contiguous()
        self.n2 = _Norm(p.out_layers.at(0), dev)
        self.c2 = _Conv(p.out_layers.at(3), dev, wfmt)
        self.skip = _Conv(p.skip_connection, dev, wfmt) if p.skip_connection is not None else None


class _Down:
    """Downsample as a layer of its own (resblock_updown=False; unet.py:192-219): 3x3 conv at stride 2, or 2x2 average pooling."""

    def __init__(self, p, dev, wfmt=0):
        self.ch = p.ch
        self.conv = _Conv(p.op, dev, wfmt) if p.use_conv else None


class _Up:
    """Upsample as a layer of its own (unet.py:160-189): nearest 2x, then an optional 3x3 conv."""

    def __init__(self, p, dev, wfmt=0):
        self.ch = p.ch
        self.conv = _Conv(p.conv, dev, wfmt) if p.use_conv else None


class _Attn:
    def __init__(self, p, dev, wfmt=0):
        self.ch, self.heads, self.new_order = p.ch, p.heads, p.new_order
        self.norm = _Norm(p.norm, dev)
        self.qkv = _Conv(p.qkv, dev, wfmt)
        self.proj = _Conv(p.proj_out, dev, wfmt)


class UNetWeights:
    """Device images of one model's parameters in one conv arithmetic: packed ONCE (2 x 3.3 GB of split-bf16
    fragment images for the 552.8 M-parameter net) and shared by every engine of that model, whatever its
    (B, H, W) -- engines own activations and launch plans only."""

    def __init__(self, model, dev, conv_mode: str = "f32"):
        if conv_mode not in ops.WFMT:
            raise ValueError(f"conv_mode must be one of {sorted(ops.WFMT)}, got {conv_mode!r}")
        self.dev, self.conv_mode = dev, conv_mode
        wfmt = ops.WFMT[conv_mode]
        self.mc = model.model_channels
        self.ted = 4 * self.mc
        self.cin, self.cout = model.in_channels, model.out_channels
        self.nlev = len(model.channel_mult)
        self.arch = describe_architecture(model)

        def wrap(m):
            from .guided_diffusion.unet import AttentionParams, DownsampleParams, ResBlockParams, UpsampleParams
            if isinstance(m, ResBlockParams):
                return _Res(m, dev, wfmt)
            if isinstance(m, AttentionParams):
                return _Attn(m, dev, wfmt)
            if isinstance(m, DownsampleParams):
                return _Down(m, dev, wfmt)
            if isinstance(m, UpsampleParams):
                return _Up(m, dev, wfmt)
            return _Conv(m, dev, wfmt)

        def seq(s):
            return [wrap(m) for _, m in sorted(((int(k), v) for k, v in s._modules.items()), key=lambda kv: kv[0])]

        te = model.time_embed
        self.te0 = (te.at(0).weight.detach().to(dev).contiguous(), te.at(0).bias.detach().to(dev).contiguous())
        self.te2 = (te.at(2).weight.detach().to(dev).contiguous(), te.at(2).bias.detach().to(dev).contiguous())
        self.num_classes = getattr(model, "num_classes", None)      # class-conditional: the engine adds label_emb rows to emb
        self.inp = [seq(s) for s in model.input_blocks]
        self.mid = seq(model.middle_block)
        self.outb = [seq(s) for s in model.output_blocks]
        self.out_norm = _Norm(model.out.at(0), dev)
        # the reference's convert_to_fp16 leaves self.out in fp32 and casts h back before it (unet.py:697-703, 743-744):
        # in the f16 arithmetic the head's weights are bf16x6 images (fp32-class) and the head runs in the fp32 family
        self.out_conv = _Conv(model.out.at(2), dev, 3 if wfmt == 1 else wfmt)

        # every ResBlock's FiLM projection Linear(SiLU(emb)) depends on emb only: one stacked weight, ONE launch
        res_blocks = [m for seqs in (self.inp, [self.mid], self.outb) for sq in seqs for m in sq if isinstance(m, _Res)]
        off = 0
        for m in res_blocks:
            m.film_off = off
            m.film_cols = (2 if m.scale_shift else 1) * m.cout     # (scale | shift), or the additive emb_out alone
            off += m.film_cols
        self.film_cols = off
        self.ew_all = torch.cat([m.ew for m in res_blocks], 0).contiguous()
        self.eb_all = torch.cat([m.eb for m in res_blocks], 0).contiguous()
        for m in res_blocks:
            m.ew = m.eb = None


def describe_architecture(model) -> dict:
    """Shapes-only description of a UNetModel (no device work): per block sequence, ('res', cin, cout, up, down) /
    ('attn', ch, heads) entries; what `activation_bytes_per_image` walks."""
    from .guided_diffusion.unet import AttentionParams, DownsampleParams, ResBlockParams, UpsampleParams

    def seq(s):
        out = []
        for _, m in sorted(((int(k), v) for k, v in s._modules.items()), key=lambda kv: kv[0]):
            if isinstance(m, ResBlockParams):
                out.append(("res", m.cin, m.cout, m.up, m.down))
            elif isinstance(m, (DownsampleParams, UpsampleParams)):      # costed like an up / down ResBlock (an upper bound)
                out.append(("res", m.ch, m.ch, isinstance(m, UpsampleParams), isinstance(m, DownsampleParams)))
            elif isinstance(m, AttentionParams):
                out.append(("attn", m.ch, m.heads))
            else:
                out.append(("conv", m.weight.shape[1], m.weight.shape[0]))
        return out

    return dict(inp=[seq(s) for s in model.input_blocks], mid=seq(model.middle_block),
                outb=[seq(s) for s in model.output_blocks], cin=model.in_channels, cout=model.out_channels)


def activation_bytes_per_image(arch: dict, H: int, W: int, itemsize: int = 4) -> int:
    """HBM bytes one image keeps resident in an engine (forward activations kept for the data-gradient pass,
    the gradient buffers of the recorded backward plan, attention probabilities, scratch): a dry walk over the
    same allocation logic as `UNetEngine._forward_impl / _backward_impl`.  Used to size the number of images
    processed per pass (`UNetModel.images_in_flight`)."""
    fixed = 0          # persistent buffers, elements of the activation type
    f32 = 0            # fp32 elements (attention logits / probabilities; qkv in the half family)
    scratch = {"a": 0, "b": 0, "c": 0, "s": 0}

    def scr(slot, n):
        scratch[slot] = max(scratch[slot], n)

    def seq_cost(layers, hw, cin, last_has_dst):
        nonlocal fixed, f32
        c = cin
        for i, l in enumerate(layers):
            last = i == len(layers) - 1
            if l[0] == "res":
                _, lcin, lcout, up, down = l
                ho = (hw[0] * 2, hw[1] * 2) if up else ((hw[0] // 2, hw[1] // 2) if down else hw)
                M, Mo = hw[0] * hw[1], ho[0] * ho[1]
                fixed += Mo * lcout                        # h1
                if not (last and last_has_dst):
                    fixed += Mo * lcout                    # block output
                fixed += M * lcin if i > 0 else 0          # backward: d/d(input) of a non-first layer
                scr("a", max(M * lcin, Mo * lcout))
                scr("b", max(Mo * lcin, Mo * lcout, M * lcin))
                scr("c", max(Mo * lcin, M * lcin))
                hw, c = ho, lcout
            else:
                _, ch, heads = l
                T = hw[0] * hw[1]
                if itemsize == 4:
                    fixed += T * 3 * ch                    # qkv
                else:
                    f32 += T * 3 * ch                      # half family: the attention core keeps qkv in fp32
                if T > 64:
                    f32 += 2 * heads * T * T               # P, P^T
                    scr("s", 3 * heads * T * T)            # S / dP, dS, dS^T
                if not (last and last_has_dst):
                    fixed += T * ch
                fixed += T * ch if i > 0 else 0
                scr("a", T * ch)
                scr("b", T * 3 * ch)
                f32 += 0 if itemsize == 4 else 0
        return hw, c

    stem_cout = arch["inp"][0][0][2]
    chans, hws = [stem_cout], [(H, W)]
    hw, c = (H, W), stem_cout
    fixed += H * W * (arch["cin"] + arch["cout"]) * 3      # x / out / gradients in NHWC
    for layers in arch["inp"][1:]:
        hw, c = seq_cost(layers, hw, c, True)
        chans.append(c)
        hws.append(hw)
    hw, c = seq_cost(arch["mid"], hw, c, True)
    n_in = len(arch["inp"])
    for i, layers in enumerate(arch["outb"]):
        j = n_in - 1 - i
        fixed += 2 * hws[j][0] * hws[j][1] * (c + chans[j])   # concat buffer + its gradient
        hw, c = seq_cost(layers, hws[j], c + chans[j], i + 1 < len(arch["outb"]))
    fixed += 2 * H * W * c
    scr("a", H * W * c)
    splitk = 4 * 1024 * 1024                               # split-K partials (bounded by the ~1 workgroup / CU target)
    return int(itemsize * (fixed + scratch["a"] + scratch["b"] + scratch["c"]) + 4 * (f32 + scratch["s"] + splitk))


class UNetEngine:
    def __init__(self, weights: UNetWeights, B: int, H: int, W: int):
        dev = weights.dev
        self.B, self.H, self.W, self.dev = B, H, W, dev
        self.weights = weights
        self.conv_mode = weights.conv_mode
        # storage type of every NHWC activation / gradient buffer: IEEE half in the "f16" arithmetic (the
        # reference's use_fp16), fp32 otherwise.  Sampler-side tensors (x_in, out, d_out, dx) are always fp32 NCHW.
        self.adt = torch.float16 if weights.conv_mode == "f16" else torch.float32
        self.mc, self.ted, self.cin, self.cout = weights.mc, weights.ted, weights.cin, weights.cout
        nlev = weights.nlev
        if H % (1 << (nlev - 1)) or W % (1 << (nlev - 1)):
            raise ValueError(f"H, W must be divisible by {1 << (nlev - 1)}")
        self.ticket = 0
        self._saved: Dict[int, dict] = {}       # per-block activations kept for the data-gradient pass
        self._xmax_reg: Dict[Tuple, Tuple] = {}      # (ptr, rows, ld) -> (slot, byte range): see _xmax_register
        self.params_version = None
        self._scratch: Dict[str, torch.Tensor] = {}
        self._fwd_plan: Optional[Recorder] = None
        self._bwd_plan: Optional[Recorder] = None
        self._plan_stream = None
        self._splitk_ws = None
        # the recorded plans are replayed as hipGraphs (one graph launch instead of ~600 host-side launches per pass:
        # 0.4 ms instead of 12 ms of host time per step); OSM_GRAPH=0 replays launch by launch
        self.use_graph = os.environ.get("OSM_GRAPH", "1") != "0"
        self._fwd_graph = self._bwd_graph = None
        # GN apply inside the consuming 3x3 conv (needs the halo-tile kernel, which OSM_CONV_HALO=0 switches off)
        self.fuse_gn = os.environ.get("OSM_FUSE_GN", "1") != "0" and os.environ.get("OSM_CONV_HALO", "1") != "0"
        # GroupNorm reductions as column sums from the epilogue of the producing conv.  OSM_FUSE_STATS = "fwd" (default of the
        # fp16-storage family): the forward statistics of a ResBlock's second GroupNorm come from its first convolution; "wino"
        # (default of the fp32-storage family, round 4): also the two backward reductions, from the data-gradient convolutions
        # the WINOGRAD kernel serves -- its epilogue owns 16-byte column groups and requests the GroupNorm input before its LDS
        # exchange (same-box A/B of the step: -0.13 ... -0.15 ms at B = 1, -1.4 ms at B = 8; with the four dependent loads inside
        # the finishing phase it was +0.18 ms); "all": from the direct halo-tile kernel as well (4-byte accesses in the MFMA C
        # layout: +1.3 ms of convolution for -1.5 ms of GroupNorm at B = 1, round 2; fp16 family +0.9 ms at B = 32); "0": neither
        self.winograd_min_hw = int(os.environ.get("OSM_WINOGRAD_MIN_HW", "16"))   # smallest H, W served by the Winograd kernel
        self.f16x3_1x1_min_hw = int(os.environ.get("OSM_F16X3_1X1_MIN_HW", "4096"))   # smallest H * W whose 1x1 layers take the f16x3 image (>= 1024: the
        # range hand-over needs the chunked statistics pass; measured round 6: 1024 instead of 4096 moves the step by +0.001 ms)
        fs = os.environ.get("OSM_FUSE_STATS", "fwd" if self.adt == torch.float16 else "wino")
        self.fuse_gn_wino = os.environ.get("OSM_FUSE_GN_WINO", "0") == "1"
        self.fuse_stats = self.fuse_gn and fs != "0"
        self.fuse_stats_bwd = self.fuse_stats and fs in ("all", "wino")
        self.fuse_stats_bwd_direct = self.fuse_stats and fs == "all"
        # smallest H * W whose Winograd data-gradients carry the backward reductions (below 64 x 64 the one-launch GroupNorm is cheaper)
        self.stats_bwd_min_hw = int(os.environ.get("OSM_STATS_BWD_MIN_HW", "1025"))
        self._check_xmax = os.environ.get("OSM_CHECK_XMAX", "0") == "1"
        self._attn_half = self.adt == torch.float16 and os.environ.get("OSM_ATTN_F16", "1") != "0"
        # f16x3 models: the attention cores in the same arithmetic as the convolutions (two half terms per operand, three fp16
        # MFMAs per product, ranges found in the kernels: csrc/flash.hip); OSM_ATTN_F16X3=0: bf16x6 (A/B, round-5 behaviour)
        self._attn_f16x3 = weights.conv_mode == "f16x3" and os.environ.get("OSM_ATTN_F16X3", "1") != "0"

        w = weights
        self.te0, self.te2, self.inp, self.mid, self.outb = w.te0, w.te2, w.inp, w.mid, w.outb
        self.out_norm, self.out_conv = w.out_norm, w.out_conv
        self.film_cols, self.ew_all, self.eb_all = w.film_cols, w.ew_all, w.eb_all

        f32 = dict(device=dev, dtype=torch.float32)
        self.x_in = torch.zeros(B, self.cin, H, W, **f32)
        self.t_dev = torch.zeros(B, **f32)
        self.out = torch.zeros(B, self.cout, H, W, **f32)
        self.d_out = torch.zeros(B, self.cout, H, W, **f32)
        self.dx = torch.zeros(B, self.cin, H, W, **f32)
        self.gn_part = torch.empty(B * ops.gn_nchunk(H * W) * G * 2, **f32)
        self.num_classes = getattr(weights, "num_classes", None)
        self.label_rows = torch.zeros(B, self.ted, **f32) if self.num_classes is not None else None   # label_emb[y], set per call

    # ------------------------------------------------------------------ buffers
    def _buf(self, rows, cols, dtype=None) -> Mat:
        return Mat.of(torch.empty(rows, cols, device=self.dev, dtype=dtype or self.adt))

    def _small(self, n) -> torch.Tensor:
        return torch.empty(n, device=self.dev, dtype=torch.float32)

    def _scr(self, slot: str, rows: int, cols: int, dtype=None) -> Mat:
        """Scratch matrix (contents live only until the next use of the same slot)."""
        need = rows * cols
        dtype = dtype or self.adt
        key = slot if dtype == self.adt else slot + "/32"
        t = self._scratch.get(key)
        if t is None or t.numel() < need:
            t = torch.empty(max(need, 1), device=self.dev, dtype=dtype)
            self._scratch[key] = t
        return Mat.of(t[:need].view(rows, cols))

    def _scr_flat(self, slot: str, n: int) -> torch.Tensor:
        t = self._scratch.get(slot)
        if t is None or t.numel() < n:
            t = torch.empty(max(n, 1), device=self.dev, dtype=torch.float32)
            self._scratch[slot] = t
        return t[:n]

    def _conv(self, x: Mat, cv: _Conv, y: Mat, hw: Tuple[int, int], dgrad=False, res: Optional[Mat] = None,
              accumulate=False, gn_table=None, gn_silu=True, stat=None, ws_slot="splitk", xmax=None):
        """stat: None, ("fwd",) -- also emit the per-column (sum, sum of squares) of y for the GroupNorm that reads it --
        or ("bwd", x_gn, table) -- y is the gradient w.r.t. SiLU(GN(x_gn)): emit that GroupNorm's two backward
        reductions.  Returns (colsum, chunks per image) when the layer's kernel produced them, else None (the caller
        then runs the GroupNorm's own reduction pass).
        xmax: [B][MAXABS_PARTS] partial max |x| the PRODUCER of x left behind (a bound is enough: the pooled / upsampled copy
        of a tensor may use the tensor's); None: an f16x3 layer runs ops.maxabs over x itself."""
        H, W = hw
        M = self.B * H * W
        cin = cv.cout if dgrad else cv.cin
        cout = cv.cin if dgrad else cv.cout
        assert x.cols == cin and y.cols == cout and x.rows == M and y.rows == M, (x.cols, cin, y.cols, cout)
        wfmt, wimg = cv.wfmt, (cv.wd if dgrad else cv.wf)
        xm = None
        # (the Winograd kernel addresses an image by 32-bit byte offsets: a view with a very wide row stride takes the direct kernel)
        if cv.wwf is not None and H >= self.winograd_min_hw and W >= self.winograd_min_hw and \
                ops.conv_winograd_ok(H, W, cin, cout, cv.k, cv.wfmt) and H * W * x.ld * x.t.element_size() < (1 << 32):
            wfmt, wimg = cv.wwfmt | ops.WINOGRAD, (cv.wwd if dgrad else cv.wwf)
            if cv.wwfmt == 4:       # f16x3: the kernel scales its input into the fp16 range from the per-image max |x|
                xm = xmax
                if xm is None:
                    xm = self._xmax_slot(ws_slot)
                    ops.maxabs(x, self.B, xm)
        elif cv.wf16 is not None and xmax is not None and H * W >= self.f16x3_1x1_min_hw and (H * W) % 128 == 0:
            wfmt, wimg, xm = 4, (cv.wd16 if dgrad else cv.wf16), xmax     # 1x1 f16x3: only where the range is already known
        elif gn_table is None and self._is_direct_f16(cv, hw):
            wfmt, wimg, xm = 4, cv.direct_f16x3(dgrad), xmax
            if xm is None:
                xm = self._xmax_slot(ws_slot)
                ops.maxabs(x, self.B, xm)
        if xm is not None and xmax is not None and self._check_xmax:
            self._xmax_debug_check(x, xm, f"conv {cin}->{cout} k{cv.k} at {H}x{W}{' dgrad' if dgrad else ''}")
        self._xmax_invalidate(y)            # a convolution leaves no max |y| behind
        sk = ops.conv_splitk(self.B, H, W, cin, cout, cv.k, wfmt, gn_table is not None)
        ws = None
        if sk > 1:
            ws = self._scr_flat(ws_slot, sk * M * cout)
        cs, nch, skw = None, 0, {}
        if stat is not None and self.fuse_stats:
            nch = ops.conv_stat_chunks(self.B, H, W, cin, cout, cv.k, wfmt, sk, gn_table is not None)
            if nch > 0:
                cs = self._scr_flat("colsum", self.B * nch * 2 * cout)      # consumed by the finalize that follows
                skw = dict(colsum=cs, stat_mode=1)
                if stat[0] == "bwd":
                    skw = dict(colsum=cs, stat_mode=2, stat_x=stat[1], stat_table=stat[2], stat_silu=True)
        ops.conv2d(x, wimg, None if dgrad else cv.b, y, self.B, H, W, cv.k, res=res,
                   accumulate=accumulate, splitk=sk, splitk_ws=ws, wfmt=wfmt, gn_table=gn_table, gn_silu=gn_silu,
                   x_maxabs=xm, **skw)
        return (cs, nch) if cs is not None else None

    def _is_wino(self, cv: _Conv, hw, dgrad=False) -> bool:
        H, W = hw
        cin, cout = (cv.cout, cv.cin) if dgrad else (cv.cin, cv.cout)
        return cv.wwf is not None and H >= self.winograd_min_hw and W >= self.winograd_min_hw and \
            bool(ops.conv_winograd_ok(H, W, cin, cout, cv.k, cv.wfmt))

    def _is_direct_f16(self, cv: _Conv, hw) -> bool:
        """3x3 layer of an f16x3 model on an image the Winograd kernel does not serve (8 <= H, W and one of them below
        `winograd_min_hw`, 16 unless OSM_WINOGRAD_MIN_HW raises it): direct f16x3."""
        H, W = hw
        # (only where the halo-tile kernel serves the layer: with OSM_CONV_HALO=0, the documented A/B switch, the wfmt-4 direct
        # image has no kernel and the layer keeps its bf16x6 images -- ADVICE r05)
        small = H < self.winograd_min_hw or W < self.winograd_min_hw
        # the layers the Winograd kernel never serves (stem 4 -> 256, the head's data-gradient 8 -> 256, the stem's 256 -> 4) take
        # the same image on the halo-tile kernel (round 6: -0.04 ms per step, same-box tools/step_ab.py; OSM_F16X3_HEADSTEM=0: bf16x6)
        never = cv.wwf is None and os.environ.get("OSM_F16X3_HEADSTEM", "1") != "0"
        return cv.k == 3 and cv._slot is not None and min(H, W) >= 8 and (small or never) and \
            os.environ.get("OSM_CONV_HALO", "1") != "0"

    # ---- registry of "max |.| of this gradient buffer was left behind by its last writer" (f16x3 range hand-over)
    @staticmethod
    def _span(m: Mat):
        """Byte range [lo, hi) a matrix view may touch (its rows at its row stride: a column slice covers its own columns only
        per row, the interval is conservative)."""
        lo = m.t.data_ptr()
        return lo, lo + ((m.rows - 1) * m.ld + m.cols) * m.t.element_size()

    def _xmax_register(self, m: Mat, slot: torch.Tensor):
        self._xmax_invalidate(m)
        self._xmax_reg[(m.t.data_ptr(), m.rows, m.ld)] = (slot, *self._span(m), m.cols)

    def _xmax_lookup(self, m: Mat):
        """The bound registered for a matrix with this (pointer, rows, ld) serves `m` only if `m` is not WIDER than what
        was registered: a bound over columns [0, c) says nothing about a view [0, c') with c' > c."""
        e = self._xmax_reg.get((m.t.data_ptr(), m.rows, m.ld))
        return e[0] if e is not None and m.cols <= e[3] else None

    def _xmax_invalidate(self, m: Mat):
        """A pass that does NOT leave max |out| behind is about to write `m`: forget every bound registered for memory it
        overlaps (a stale, too-small bound would scale an f16x3 operand out of the fp16 range silently)."""
        if not self._xmax_reg:
            return
        lo, hi = self._span(m)
        for k in [k for k, e in self._xmax_reg.items() if e[1] < hi and lo < e[2]]:
            del self._xmax_reg[k]

    def _xmax_debug_check(self, x: Mat, xm: torch.Tensor, what: str):
        """OSM_CHECK_XMAX=1 (recording pass only: kernels execute there): the bound handed over must cover max |x|."""
        bound = xm.view(self.B, -1).view(torch.int32).max(dim=1).values.view(torch.float32)
        actual = x.t.reshape(self.B, -1, x.cols).abs().amax(dim=(1, 2))
        ok = bool(((actual <= bound) | torch.isnan(bound)).all())
        if not ok:
            raise OsmosisHipError(f"f16x3 range hand-over broken at {what}: max |x| {actual.tolist()} > bound {bound.tolist()}")

    def _xmax_slot(self, key: str) -> torch.Tensor:
        """[B][MAXABS_PARTS] partial max |x| of an f16x3 convolution's input (every entry is rewritten by its producer)."""
        return self._scr_flat("xmax/" + key, self.B * ops.MAXABS_PARTS)

    def _xmax_from_gn(self, cv: _Conv, hw_conv, hw_gn, key: str, dgrad=False):
        """The slot a GroupNorm pass should fill with max |output| because the f16x3 convolution `cv` (at resolution hw_conv)
        reads that output next; None when the layer is not f16x3 or the GroupNorm grid does not fit the slots."""
        if cv.wwfmt != 4 or not (self._is_wino(cv, hw_conv, dgrad=dgrad) or self._is_direct_f16(cv, hw_conv)):
            return None
        if ops.gn_nchunk(hw_gn[0] * hw_gn[1]) > ops.MAXABS_PARTS:
            return None
        return self._xmax_slot(key)

    def _gn_fusable(self, cv: _Conv, hw) -> bool:
        """GroupNorm apply inside the consuming 3x3 convolution: needs the halo-tile kernel (split-bf16 / fp16 weights,
        W >= 16, H >= 8) on a tensor that is not better served by the one-launch low-resolution GroupNorm.
        NOT where the Winograd kernel serves the layer (round 3): that kernel is bound by its non-MFMA instructions and
        the power cap, every one of its Cout / 64 column tiles redoes the normalisation + SiLU of its input patch
        (256^2 256 -> 256: 326 us fused vs 256 us plain; a separate apply pass costs ~30 us): OSM_FUSE_GN_WINO=1 restores it."""
        H, W = hw
        if not (self.fuse_gn and cv.wfmt != 0 and cv.k == 3 and W >= 16 and H >= 8 and H * W > 256):
            return False
        if not self._is_wino(cv, hw):
            return True
        return self.fuse_gn_wino and cv.wwfmt != 4      # f16x3 needs the range of the tensor it multiplies: never fused

    def _bwd_stats_ok(self, cv: _Conv, hw) -> bool:
        """OSM_FUSE_STATS=all: may the data-gradient convolution of `cv` emit the two GroupNorm-backward reductions of the
        GroupNorm in front of `cv`?  Needs the per-channel table of that GroupNorm kept from the forward pass."""
        H, W = hw
        return (self.fuse_stats_bwd_direct and self._gn_fusable(cv, hw)) or \
            (self._is_wino(cv, hw) and self._is_wino(cv, hw, dgrad=True) and H * W >= self.stats_bwd_min_hw)

    def _gn_stats_from_conv(self, cv: _Conv, hw) -> bool:
        """May the convolution that produces a tensor also emit the column sums for the GroupNorm `cv` reads it through?"""
        H, W = hw
        if not (self.fuse_stats and cv.wfmt != 0 and cv.k == 3 and W >= 16 and H >= 8 and H * W > 256):
            return False
        # not fused: finalize + apply are two launches, the one-launch GroupNorm (H W <= 1024) is cheaper than that
        return self._gn_fusable(cv, hw) or H * W > 1024

    def _gn_conv(self, x: Mat, norm: _Norm, st, cv: _Conv, y: Mat, hw, film=None, res: Optional[Mat] = None,
                 cs=None, table=None, stat=None, xin_max=None):
        """y = conv3x3(SiLU(GN(+FiLM)(x))) (+res).  Where the halo-tile kernel runs (split-bf16 weights, W >= 16,
        H >= 8) the normalised tensor is never materialised: statistics -> per-channel table -> applied by the
        convolution while it stages its input; otherwise GN writes a scratch tensor first.
        cs: (colsum, chunks) the convolution that produced x wrote next to it (then no reduction pass over x is
        needed); table: persistent [B][4][C] buffer for the per-channel table (kept for the backward); stat: forwarded
        to the convolution (column sums of y).  Returns the convolution's (colsum, chunks) or None."""
        B = self.B
        H, W = hw
        if self._gn_fusable(cv, hw):
            if table is None:
                table = self._scr_flat("gnt", B * 4 * x.cols)
            if cs is not None:
                ops.gn_finalize_cols(cs[0], cs[1], B, H * W, x.cols, G, st, mode=0, gamma=norm.g, beta=norm.b, film=film,
                                     table=table)
            else:
                ops.gn_prep(x, B, H * W, G, self.gn_part, st, norm.g, norm.b, table, film=film)
            return self._conv(x, cv, y, hw, res=res, gn_table=table, gn_silu=True, stat=stat)
        a = self._scr("a", B * H * W, x.cols, dtype=x.t.dtype)
        xm = self._xmax_from_gn(cv, hw, hw, "gn")
        if cs is not None:      # statistics from the producer's column sums, then the apply pass alone
            if table is not None:
                ops.gn_finalize_cols(cs[0], cs[1], B, H * W, x.cols, G, st, mode=0, gamma=norm.g, beta=norm.b, film=film,
                                     table=table)
            else:
                ops.gn_finalize_cols(cs[0], cs[1], B, H * W, x.cols, G, st, mode=0)
            ops.gn_apply(x, a, B, H * W, G, st, norm.g, norm.b, film=film, silu=True, maxabs=xm)
        elif table is not None:     # the table is kept for the backward (OSM_FUSE_STATS=wino / all)
            ops.gn_prep(x, B, H * W, G, self.gn_part, st, norm.g, norm.b, table, film=film, maxabs_in=xin_max)
            xin_max = None
            ops.gn_apply(x, a, B, H * W, G, st, norm.g, norm.b, film=film, silu=True, maxabs=xm)
        else:
            # xin_max: the caller asked for max |x| of the INPUT as well (it has checked that this branch is the one taken)
            self._gn_fwd(x, a, H * W, st, norm, film=film, silu=True, maxabs=xm, maxabs_in=xin_max)
            xin_max = None
        assert xin_max is None, "max |x| of the input was requested on a path that has no statistics pass over x"
        return self._conv(a, cv, y, hw, res=res, stat=stat, xmax=xm)

    # ------------------------------------------------------------------ GroupNorm dispatch
    def _gn_fwd(self, x: Mat, y: Mat, HW: int, st, norm: _Norm, film=None, silu=True, maxabs=None, maxabs_in=None):
        """statistics + normalise (+FiLM) (+SiLU): the one-launch kernels for small tensors, the chunked two-pass path above them.
        (A cooperative single-read kernel -- one launch, tensor resident in registers, cross-workgroup exchange of the partial sums --
        existed in round 4: correct, deadlock-free, and +0.16 ... +0.42 ms in the step; removed in round 5, see profiles/NOTES_r01_r04.md.)"""
        ops.gn_fwd(x, y, self.B, HW, G, self.gn_part, st, norm.g, norm.b, film=film, silu=silu, maxabs=maxabs, maxabs_in=maxabs_in)

    def _gn_bwd(self, x: Mat, dy: Mat, dx: Mat, HW: int, st, norm: _Norm, gst, film=None, silu=True, addend=None,
                addend2=None, maxabs=None):
        ops.gn_bwd(x, dy, dx, self.B, HW, G, st, norm.g, norm.b, self.gn_part, gst, film=film, silu=silu, addend=addend,
                   addend2=addend2, maxabs=maxabs)

    # ------------------------------------------------------------------ ResBlock
    def _res_fwd(self, blk: _Res, x: Mat, dst: Mat, hw):
        B = self.B
        H, W = hw
        HW = H * W
        M = B * HW
        st1 = self._small(B * G * 2)
        if blk.up or blk.down:
            a1 = self._scr("a", M, blk.cin)
            hwo = (2 * H, 2 * W) if blk.up else (H // 2, W // 2)
            xm1 = self._xmax_from_gn(blk.c1, hwo, hw, "gn")      # max |pool(a)|, max |upsample(a)| <= max |a|
            self._gn_fwd(x, a1, HW, st1, blk.n1, silu=True, maxabs=xm1)
            if blk.up:
                ho, wo = 2 * H, 2 * W
                a1r = self._scr("b", B * ho * wo, blk.cin)
                xs = self._scr("c", B * ho * wo, blk.cin)
                ops.resample_pair(True, a1, a1r, x, xs, B, H, W, 1.0)        # both tensors, one launch
            else:
                ho, wo = H // 2, W // 2
                a1r = self._scr("b", B * ho * wo, blk.cin)
                xs = self._scr("c", B * ho * wo, blk.cin)
                ops.resample_pair(False, a1, a1r, x, xs, B, H, W, 0.25)
            Mo = B * ho * wo
            h1 = self._buf(Mo, blk.cout)
            fuse2 = self._gn_fusable(blk.c2, (ho, wo))
            tab1 = None
            cs1 = self._conv(a1r, blk.c1, h1, (ho, wo),
                             stat=("fwd",) if (blk.scale_shift and self._gn_stats_from_conv(blk.c2, (ho, wo))) else None, xmax=xm1)
        else:
            ho, wo = H, W
            xs = x
            Mo = M
            h1 = self._buf(Mo, blk.cout)
            fuse2 = self._gn_fusable(blk.c2, (ho, wo))
            # the per-channel GroupNorm tables are kept: the data-gradient convolutions fold the GroupNorm-backward
            # reductions into their epilogues with them (see _res_bwd)
            tab1 = self._small(B * 4 * blk.cin) if (self.fuse_stats_bwd and self._bwd_stats_ok(blk.c1, hw)) else None
            # the skip connection's 1x1 convolution reads x itself: where it has an f16x3 image, the statistics pass of the first
            # GroupNorm (which reads every element of x anyway) leaves max |x| behind for it -- so it runs AFTER that pass
            xin = None
            if blk.skip is not None and blk.skip.wf16 is not None and HW >= self.f16x3_1x1_min_hw and HW % 128 == 0 and \
                    not self._gn_fusable(blk.c1, hw) and ops.gn_nchunk(HW) <= ops.MAXABS_PARTS:
                xin = self._xmax_slot("skip")
            cs1 = self._gn_conv(x, blk.n1, st1, blk.c1, h1, hw, table=tab1,
                                stat=("fwd",) if (blk.scale_shift and self._gn_stats_from_conv(blk.c2, (ho, wo))) else None,
                                xin_max=xin)
            if blk.skip is not None:
                self._conv(xs, blk.skip, dst, (ho, wo), ws_slot="splitk2", xmax=xin)
        film = self.film_all[:, blk.film_off:blk.film_off + blk.film_cols]
        if not blk.scale_shift:     # additive conditioning (unet.py:331-332): h = h + emb_out, then GroupNorm / SiLU / conv without FiLM
            ops.add_rowvec(h1, film, self.film_all.stride(0), B, ho * wo)
            film, cs1 = None, None
        st2 = self._small(B * G * 2)
        if blk.skip is not None:
            res = dst
        else:
            res = xs
        tab2 = self._small(B * 4 * blk.cout) if (self.fuse_stats_bwd and self._bwd_stats_ok(blk.c2, (ho, wo))) else None
        self._gn_conv(h1, blk.n2, st2, blk.c2, dst, (ho, wo), film=film, res=res, cs=cs1, table=tab2)
        self._saved[id(blk)] = dict(x=x, st1=st1, h1=h1, st2=st2, film=film, hw=hw, hwo=(ho, wo), tab1=tab1, tab2=tab2)
        return (ho, wo)

    def _res_bwd(self, blk: _Res, dy: Mat, dx_dst: Mat, accumulate: bool):
        s = self._saved[id(blk)]
        B = self.B
        H, W = s["hw"]
        ho, wo = s["hwo"]
        M, Mo = B * H * W, B * ho * wo
        if blk.skip is not None:      # skip-path gradient (f16x3 where max |dy| was left behind by the pass that wrote dy)
            self._conv(dy, blk.skip, dx_dst, (H, W), dgrad=True, accumulate=accumulate, ws_slot="splitk2",
                       xmax=self._xmax_lookup(dy))
        dh2 = self._scr("a", Mo, blk.cout)
        # GroupNorm backward = two reductions over (x, dy) + an apply pass.  Where the forward kept the per-channel
        # table, the reductions are folded into the epilogue of the data-gradient convolution that PRODUCES dy (it
        # holds dy in registers and reads x once); what remains is a tiny finalize and the apply pass.
        # max |dy| for an f16x3 data-gradient: left behind by the GroupNorm pass that wrote dy (the previous ResBlock's last
        # launch; a bound over the wider concat-gradient buffer serves its column slice), else a pass of its own
        cs = self._conv(dy, blk.c2, dh2, (ho, wo), dgrad=True,
                        stat=("bwd", s["h1"], s["tab2"]) if s["tab2"] is not None else None,
                        xmax=self._xmax_lookup(dy))
        dh1 = self._scr("b", Mo, blk.cout)
        gst = self._small(B * G * 2)
        xmh = self._xmax_from_gn(blk.c1, (ho, wo), (ho, wo), "gnb", dgrad=True)
        if cs is not None:
            ops.gn_finalize_cols(cs[0], cs[1], B, ho * wo, blk.cout, G, gst, mode=1)
            ops.gn_bwd_apply(s["h1"], dh2, dh1, B, ho * wo, G, s["st2"], gst, blk.n2.g, blk.n2.b, film=s["film"],
                             silu=True, maxabs=xmh)
        else:
            self._gn_bwd(s["h1"], dh2, dh1, ho * wo, s["st2"], blk.n2, gst, film=s["film"], silu=True, maxabs=xmh)
        da1r = self._scr("a", Mo, blk.cin)
        cs1 = self._conv(dh1, blk.c1, da1r, (ho, wo), dgrad=True,
                         stat=("bwd", s["x"], s["tab1"]) if s["tab1"] is not None else None, xmax=xmh)
        # forward: nearest 2x upsample -> backward: 2x2 sum;  forward: 2x2 average -> backward: replicate / 4.  The gradient of the
        # resampled block input (dy -> t, below) has the same shape: both in one launch
        if blk.up or blk.down:
            assert blk.skip is None
            da1 = self._scr("b", M, blk.cin)
            t = self._scr("c", M, blk.cin)
            ops.resample_pair(not blk.up, da1r, da1, dy, t, B, ho, wo, 1.0 if blk.up else 0.25)
        else:
            da1 = da1r
        # skip-path gradient + the gradient already sitting in dx_dst (concat / residual accumulation) are ADDENDS of the
        # GroupNorm-backward apply pass (up to two, one of which may be dx_dst itself): no accumulation pass of its own
        add2 = None
        if blk.up or blk.down:
            add = t
            add2 = dx_dst if accumulate else None
        elif blk.skip is not None:
            add = dx_dst
        else:
            add = dy
            add2 = dx_dst if accumulate else None
        gst1 = self._small(B * G * 2)
        # this pass writes dx_dst and leaves max |dx_dst| behind for the f16x3 data-gradient convolution of whichever ResBlock
        # reads the buffer as its dy; any OTHER later writer of that memory (attention backward, a convolution) drops the entry
        xmo = None
        if self.conv_mode == "f16x3" and ops.gn_nchunk(H * W) <= ops.MAXABS_PARTS:
            xmo = self._small(B * ops.MAXABS_PARTS)
            self._xmax_register(dx_dst, xmo)
        else:
            self._xmax_invalidate(dx_dst)
        if cs1 is not None:
            ops.gn_finalize_cols(cs1[0], cs1[1], B, H * W, blk.cin, G, gst1, mode=1)
            ops.gn_bwd_apply(s["x"], da1, dx_dst, B, H * W, G, s["st1"], gst1, blk.n1.g, blk.n1.b, silu=True, addend=add,
                             addend2=add2, maxabs=xmo)
        else:
            self._gn_bwd(s["x"], da1, dx_dst, H * W, s["st1"], blk.n1, gst1, silu=True, addend=add, addend2=add2, maxabs=xmo)

    # ------------------------------------------------------------------ Attention
    def _gemm(self, *a, **kw):
        """ops.gemm with split-K when the batch of (M, N) tiles alone would leave most CUs idle
        (P V / dq / dk / dv: 64-wide heads, K = T)."""
        M, N, K = a[6], a[7], a[8]
        nbatch = kw.get("nb1", 1) * kw.get("nb2", 1)
        sk = ops.splitk_hint(M, N, K, 1, nbatch)
        if sk > 1:
            kw.update(splitk=sk, splitk_ws=self._scr_flat("splitk", sk * nbatch * M * N))
        ops.gemm(*a, **kw)

    def _attn_offsets(self, blk: _Attn):
        C, nh = blk.ch, blk.heads
        ch = C // nh
        if blk.new_order:   # q,k,v = qkv.chunk(3) then heads       (unet.py:459-467)
            return ch, (0, C, 2 * C), ch
        return ch, (0, ch, 2 * ch), 3 * ch      # legacy: per head [q|k|v]  (unet.py:426)

    def _attn_fwd(self, blk: _Attn, x: Mat, dst: Mat, hw):
        B = self.B
        T = hw[0] * hw[1]
        M = B * T
        C, nh = blk.ch, blk.heads
        ch, (qo, ko, vo), hs = self._attn_offsets(blk)
        st = self._small(B * G * 2)
        xn = self._scr("a", M, C)
        self._gn_fwd(x, xn, T, st, blk.norm, silu=False)
        half = self.adt != torch.float32
        if half:     # the attention core is fp32 in both modes (the reference soft-maxes in fp32, unet.py:431)
            qkv_h = self._scr("b", M, 3 * C)
            self._conv(xn, blk.qkv, qkv_h, hw)
            qkv = self._buf(M, 3 * C, torch.float32)
            ops.convert(qkv_h, qkv)
        else:
            qkv = self._buf(M, 3 * C)
            self._conv(xn, blk.qkv, qkv, hw)
        nmat = B * nh
        alpha = 1.0 / math.sqrt(ch)     # (q*ch^-1/4)·(k*ch^-1/4)
        a = self._scr("c", M, C, torch.float32)
        # fused core: measured faster at T = 64 (3 launches instead of 14); at T = 256 its fp32 FMA work sits on
        # only 64 workgroups and the unfused GEMM pipeline wins (OSM_ATTN_FUSED=all / 0 to force either way)
        # 64-wide heads (every block of the 256-channel model): flash-style core on the matrix cores
        # (OSM_ATTN_FLASH=0: the round-1 paths -- FMA core at T = 64, GEMM pipeline above; =256: flash from T = 256 only)
        fl = os.environ.get("OSM_ATTN_FLASH", "1")
        try:
            fl_min = int(fl) if fl not in ("0", "1") else 64
        except ValueError:          # "on", "all", ...: flash wherever it is supported
            fl_min = 64
        flash = fl != "0" and ops.attn_flash_supported(T, ch) and T >= fl_min
        mode = os.environ.get("OSM_ATTN_FUSED", "64")
        fused = (not flash) and ops.attn_small_supported(T, ch) and (mode == "all" or (mode != "0" and T <= 64))
        P = PT = lse = None
        if flash:      # logits / probabilities stay in registers; the output and its log-sum-exp are kept for the backward
            a = self._buf(M, C, torch.float32)
            lse = self._small(nmat * T)
            # fp16-storage family: one fp16 MFMA per product, fp32 accumulation and softmax -- what the reference's half attention
            # computes (unet.py:426-433); OSM_ATTN_F16=0 keeps the fp32-class bf16x6 core there too (rounds 2-3)
            ops.attn_flash_fwd(qkv, a, lse, B, T, nh, ch, (qo, ko, vo), hs, alpha, half=self._attn_half, f16x3=self._attn_f16x3)
        elif fused:    # 8x8: logits stay on the CU, one launch, nothing kept for the backward
            ops.attn_small_fwd(qkv, a, B, T, nh, ch, (qo, ko, vo), hs, alpha)
        else:
            S = self._scr_flat("s0", nmat * T * T)
            self._gemm(qkv.t, 3 * C, qkv.t, 3 * C, S, T, T, T, ch, b_kn=False, alpha=alpha, nb1=nh, nb2=B,
                       sA=(hs, T * 3 * C), sB=(hs, T * 3 * C), sC=(T * T, nh * T * T), a_off=qo, b_off=ko)
            P = self._small(nmat * T * T)
            PT = self._small(nmat * T * T)
            ops.softmax_rows(S, P, PT, nmat, T)
            self._gemm(P, T, qkv.t, 3 * C, a.t, C, T, ch, T, b_kn=True, nb1=nh, nb2=B,
                       sA=(T * T, nh * T * T), sB=(hs, T * 3 * C), sC=(ch, T * C), b_off=vo)
        a32 = a
        if half:
            a_h = self._scr("a", M, C)
            ops.convert(a, a_h)
            a = a_h
        self._conv(a, blk.proj, dst, hw, res=x)
        self._saved[id(blk)] = dict(x=x, st=st, qkv=qkv, P=P, PT=PT, hw=hw, fused=fused, flash=flash, lse=lse,
                                    a=a32 if flash else None)
        return hw

    def _attn_bwd(self, blk: _Attn, dy: Mat, dx_dst: Mat, accumulate: bool):
        s = self._saved[id(blk)]
        B = self.B
        hw = s["hw"]
        T = hw[0] * hw[1]
        M = B * T
        C, nh = blk.ch, blk.heads
        ch, (qo, ko, vo), hs = self._attn_offsets(blk)
        nmat = B * nh
        qkv, P, PT = s["qkv"], s["P"], s["PT"]
        alpha = 1.0 / math.sqrt(ch)
        half = self.adt != torch.float32
        da = self._scr("a", M, C)
        self._conv(dy, blk.proj, da, hw, dgrad=True)
        if half:
            da32 = self._scr("c", M, C, torch.float32)
            ops.convert(da, da32)
            da = da32
        dqkv = self._scr("b", M, 3 * C, torch.float32)
        if s["flash"]:
            delta = self._scr_flat("s0", nmat * T)
            ops.attn_flash_bwd(qkv, s["a"], da, dqkv, s["lse"], delta, B, T, nh, ch, (qo, ko, vo), hs, alpha, half=self._attn_half,
                               f16x3=self._attn_f16x3)
        elif s["fused"]:
            ws = self._scr_flat("s0", 2 * nmat * T * T)
            ops.attn_small_bwd(qkv, da, dqkv, ws, B, T, nh, ch, (qo, ko, vo), hs, alpha)
        else:
            dP = self._scr_flat("s0", nmat * T * T)
            self._gemm(da.t, C, qkv.t, 3 * C, dP, T, T, T, ch, b_kn=False, nb1=nh, nb2=B,
                       sA=(ch, T * C), sB=(hs, T * 3 * C), sC=(T * T, nh * T * T), b_off=vo)
            dS = self._scr_flat("s1", nmat * T * T)
            dST = self._scr_flat("s2", nmat * T * T)
            ops.softmax_rows_bwd(P, dP, dS, dST, nmat, T)
            sQ = (hs, T * 3 * C)
            # dq = alpha * dS k ; dk = alpha * dS^T q ; dv = P^T da
            self._gemm(dS, T, qkv.t, 3 * C, dqkv.t, 3 * C, T, ch, T, b_kn=True, alpha=alpha, nb1=nh, nb2=B,
                       sA=(T * T, nh * T * T), sB=sQ, sC=sQ, b_off=ko, c_off=qo)
            self._gemm(dST, T, qkv.t, 3 * C, dqkv.t, 3 * C, T, ch, T, b_kn=True, alpha=alpha, nb1=nh, nb2=B,
                       sA=(T * T, nh * T * T), sB=sQ, sC=sQ, b_off=qo, c_off=ko)
            self._gemm(PT, T, da.t, C, dqkv.t, 3 * C, T, ch, T, b_kn=True, nb1=nh, nb2=B,
                       sA=(T * T, nh * T * T), sB=(ch, T * C), sC=sQ, c_off=vo)
        if half:
            dqkv_h = self._scr("b", M, 3 * C)
            ops.convert(dqkv, dqkv_h)
            dqkv = dqkv_h
        dxn = self._scr("a", M, C)
        self._conv(dqkv, blk.qkv, dxn, hw, dgrad=True)
        gst = self._small(B * G * 2)
        # this pass writes dx_dst: it leaves max |dx_dst| behind for the f16x3 data-gradient convolution of the ResBlock that reads
        # the buffer as its dy (as the ResBlock backward's last pass does)
        xmo = None
        if self.conv_mode == "f16x3" and ops.gn_nchunk(T) <= ops.MAXABS_PARTS and dx_dst.t.dtype == torch.float32:
            xmo = self._small(B * ops.MAXABS_PARTS)
            self._xmax_register(dx_dst, xmo)
        else:
            self._xmax_invalidate(dx_dst)
        self._gn_bwd(s["x"], dxn, dx_dst, T, s["st"], blk.norm, gst, silu=False, addend=dy,
                     addend2=dx_dst if accumulate else None, maxabs=xmo)

    # ------------------------------------------------------------------ Upsample / Downsample as layers (resblock_updown=False)
    def _resample_fwd(self, l, x: Mat, dst: Mat, hw):
        """Downsample: 3x3 conv at stride 2 = the stride-1 convolution (any of the 3x3 kernels) with every other pixel kept, or
        2x2 average pooling; Upsample: nearest 2x, then the 3x3 conv (unet.py:160-219).  These variants are not on any shipped
        config: the stride-2 form spends 4x the multiplies of a dedicated kernel on five layers."""
        B = self.B
        H, W = hw
        self._xmax_invalidate(dst)
        if isinstance(l, _Down):
            if H % 2 or W % 2:      # the reference's stride-2, pad-1 conv yields ceil(H / 2); this path keeps every other pixel of an EVEN
                                    # image (UNetEngine refuses other sizes at construction; a BlockEngine can be handed any) -- ADVICE r05
                raise ValueError(f"Downsample needs even H and W (got {H} x {W}): odd sizes are not supported")
            ho = (H // 2, W // 2)
            if l.conv is not None:
                full = self._scr("a", B * H * W, l.ch)
                self._conv(x, l.conv, full, hw)
                ops.stride2_pick(full, dst, B, H, W)
            else:
                ops.pool2x2(x, dst, B, H, W, 0.25)
        else:
            ho = (2 * H, 2 * W)
            if l.conv is not None:
                u = self._scr("a", B * ho[0] * ho[1], l.ch)
                ops.upsample2x(x, u, B, H, W, 1.0)
                self._conv(u, l.conv, dst, ho)
            else:
                ops.upsample2x(x, dst, B, H, W, 1.0)
        self._saved[id(l)] = dict(x=x, hw=hw, hwo=ho)
        return ho

    def _resample_bwd(self, l, dy: Mat, dx_dst: Mat, accumulate: bool):
        s = self._saved[id(l)]
        B = self.B
        H, W = s["hw"]
        ho, wo = s["hwo"]
        self._xmax_invalidate(dx_dst)
        if l.conv is not None and isinstance(l, _Down):       # adjoint of "keep every other pixel", then the stride-1 data gradient
            df = self._scr("a", B * H * W, l.ch)
            ops.stride2_place(dy, df, B, H, W)
            self._conv(df, l.conv, dx_dst, (H, W), dgrad=True, accumulate=accumulate)
            return
        tgt = dx_dst if not accumulate else self._scr("b", B * H * W, l.ch)
        if isinstance(l, _Down):                                # average pooling: replicate / 4
            ops.upsample2x(dy, tgt, B, ho, wo, 0.25)
        else:
            src = dy
            if l.conv is not None:
                src = self._scr("a", B * ho * wo, l.ch)
                self._conv(dy, l.conv, src, (ho, wo), dgrad=True)
            ops.pool2x2(src, tgt, B, ho, wo, 1.0)               # nearest 2x: the gradient is the 2x2 sum
        if accumulate:
            ops.copy2d(tgt, dx_dst, accumulate=True)

    # ------------------------------------------------------------------ whole network
    @staticmethod
    def _out_ch(layers, cin):
        c = cin
        for l in layers:
            if isinstance(l, _Res):
                c = l.cout
        return c

    def _run_layers_fwd(self, layers, h: Mat, hw, dst: Mat):
        for i, l in enumerate(layers):
            last = i == len(layers) - 1
            if isinstance(l, _Res):
                ho = (hw[0] * 2, hw[1] * 2) if l.up else ((hw[0] // 2, hw[1] // 2) if l.down else hw)
                d = dst if last else self._buf(self.B * ho[0] * ho[1], l.cout)
                hw = self._res_fwd(l, h, d, hw)
            elif isinstance(l, _Attn):
                d = dst if last else self._buf(h.rows, l.ch)
                hw = self._attn_fwd(l, h, d, hw)
            elif isinstance(l, (_Down, _Up)):
                ho = (hw[0] // 2, hw[1] // 2) if isinstance(l, _Down) else (hw[0] * 2, hw[1] * 2)
                d = dst if last else self._buf(self.B * ho[0] * ho[1], l.ch)
                hw = self._resample_fwd(l, h, d, hw)
            else:
                raise AssertionError
            h = d
        return h, hw

    def _run_layers_bwd(self, layers, dy: Mat, dx_dst: Mat, accumulate: bool):
        for i in range(len(layers) - 1, -1, -1):
            l = layers[i]
            first = i == 0
            x_saved = self._saved[id(l)]["x"]
            d = dx_dst if first else self._buf(x_saved.rows, x_saved.cols)
            acc = accumulate if first else False
            if isinstance(l, _Res):
                self._res_bwd(l, dy, d, acc)
            elif isinstance(l, _Attn):
                self._attn_bwd(l, dy, d, acc)
            else:
                self._resample_bwd(l, dy, d, acc)
            dy = d

    def _forward_impl(self):
        B, H, W = self.B, self.H, self.W
        f32 = dict(device=self.dev, dtype=torch.float32)
        # ---- timestep embedding MLP (nn.py:103-121, unet.py:550-554, 727)
        temb = torch.empty(B, self.mc, **f32)
        ops.timestep_embedding(self.t_dev, temb, B, self.mc)
        e1 = torch.empty(B, self.ted, **f32)
        ops.linear(temb, self.te0[0], self.te0[1], e1, B, self.mc, self.ted, silu_out=True)
        self.emb = torch.empty(B, self.ted, **f32)
        ops.linear(e1, self.te2[0], self.te2[1], self.emb, B, self.ted, self.ted)
        if self.label_rows is not None:          # class-conditional: emb = emb + label_emb(y) (unet.py:729-731)
            ops.add_rowvec(Mat.of(self.emb), self.label_rows, self.ted, B, 1)
        self.film_all = torch.empty(B, self.film_cols, **f32)
        ops.linear(self.emb, self.ew_all, self.eb_all, self.film_all, B, self.ted, self.film_cols, silu_in=True)

        # ---- channel bookkeeping for the zero-copy concatenations
        stem: _Conv = self.inp[0][0]
        chans = [stem.cout]
        c = stem.cout
        for layers in self.inp[1:]:
            c = self._out_ch(layers, c)
            chans.append(c)
        n_in = len(self.inp)
        c_mid = self._out_ch(self.mid, c)
        # output block i reads cat(h_i, hs[n_in-1-i]); h_0 = middle output
        hws = []           # resolution of each input block output
        hw = (H, W)
        for layers in self.inp:
            for l in layers:
                if (isinstance(l, _Res) and l.down) or isinstance(l, _Down):
                    hw = (hw[0] // 2, hw[1] // 2)
            hws.append(hw)
        c_h = c_mid
        self.cat: List[Mat] = []
        self.cat_split: List[int] = []
        for i, layers in enumerate(self.outb):
            j = n_in - 1 - i
            rows = B * hws[j][0] * hws[j][1]
            self.cat.append(self._buf(rows, c_h + chans[j]))
            self.cat_split.append(c_h)
            assert layers[0].cin == c_h + chans[j], (layers[0].cin, c_h, chans[j])
            c_h = self._out_ch(layers, c_h + chans[j])

        def skip_dst(j):
            i = n_in - 1 - j
            return self.cat[i].cols_slice(self.cat_split[i], self.cat[i].cols)

        # ---- input blocks
        x_nhwc = self._buf(B * H * W, self.cin)
        ops.nchw_to_nhwc(self.x_in, x_nhwc, B, self.cin, H * W)
        h = skip_dst(0)
        self._conv(x_nhwc, stem, h, (H, W))
        hw = (H, W)
        for j in range(1, n_in):
            h, hw = self._run_layers_fwd(self.inp[j], h, hw, skip_dst(j))
        # ---- middle, writing into the left columns of the first concat buffer
        h, hw = self._run_layers_fwd(self.mid, h, hw, self.cat[0].cols_slice(0, self.cat_split[0]))
        # ---- output blocks
        for i, layers in enumerate(self.outb):
            if i + 1 < len(self.outb):
                dst = self.cat[i + 1].cols_slice(0, self.cat_split[i + 1])
            else:
                dst = self._buf(B * H * W, c_h)
            h, hw = self._run_layers_fwd(layers, self.cat[i], hw, dst)
        assert hw == (H, W)
        # ---- head: GN, SiLU, conv3x3 -> NCHW.  Always in fp32 (use_fp16: `h = h.type(x.dtype)` before the fp32 self.out,
        # unet.py:743-744): half storage is converted once, eps / the variance logits are not rounded to half
        if self.adt != torch.float32:
            h32 = self._buf(B * H * W, h.cols, dtype=torch.float32)
            ops.convert(h, h32)
            h = h32
        self.h_last = h
        self.st_out = self._small(B * G * 2)
        o = self._buf(B * H * W, self.cout, dtype=torch.float32)
        self._gn_conv(h, self.out_norm, self.st_out, self.out_conv, o, (H, W))
        ops.nhwc_to_nchw(o, self.out, B, self.cout, H * W)
        self.x_nhwc = x_nhwc

    def _backward_impl(self):
        B, H, W = self.B, self.H, self.W
        n_in = len(self.inp)
        self._xmax_reg = {}         # (pointer, rows, ld) of a gradient buffer -> max |.| slot its last writer filled
        f32 = torch.float32         # the head's gradient runs in the fp32 family in every arithmetic (see _forward_impl)
        do = self._buf(B * H * W, self.cout, dtype=f32)
        ops.nchw_to_nhwc(self.d_out, do, B, self.cout, H * W)
        da = self._scr("a", B * H * W, self.h_last.cols, dtype=f32)
        self._conv(do, self.out_conv, da, (H, W), dgrad=True)
        dy = self._buf(B * H * W, self.h_last.cols, dtype=f32)
        gst = self._small(B * G * 2)
        xmo = None          # max |dy| for the f16x3 data-gradient convolutions of the last ResBlock (3x3 and skip)
        if self.conv_mode == "f16x3" and ops.gn_nchunk(H * W) <= ops.MAXABS_PARTS:
            xmo = self._small(B * ops.MAXABS_PARTS)
            self._xmax_register(dy, xmo)
        self._gn_bwd(self.h_last, da, dy, H * W, self.st_out, self.out_norm, gst, silu=True, maxabs=xmo)
        if self.adt != f32:
            dy_h = self._buf(B * H * W, self.h_last.cols)
            ops.convert(dy, dy_h)
            dy = dy_h
        # ---- output blocks in reverse; dcat[i] receives d/d(cat_i) at full width
        dcat: List[Optional[Mat]] = [None] * len(self.outb)
        for i in range(len(self.outb) - 1, -1, -1):
            dcat[i] = self._buf(self.cat[i].rows, self.cat[i].cols)
            self._run_layers_bwd(self.outb[i], dy, dcat[i], accumulate=False)
            dy = dcat[i].cols_slice(0, self.cat_split[i])

        def dskip(j):
            i = n_in - 1 - j
            return dcat[i].cols_slice(self.cat_split[i], dcat[i].cols)

        # ---- middle: its input is hs[n_in-1], which already holds the concat contribution
        self._run_layers_bwd(self.mid, dy, dskip(n_in - 1), accumulate=True)
        for j in range(n_in - 1, 0, -1):
            self._run_layers_bwd(self.inp[j], dskip(j), dskip(j - 1), accumulate=True)
        dxn = self._buf(B * H * W, self.cin)
        self._conv(dskip(0), self.inp[0][0], dxn, (H, W), dgrad=True)
        ops.nhwc_to_nchw(dxn, self.dx, B, self.cin, H * W)

    # ------------------------------------------------------------------ public
    def _check_stream(self):
        s = current_stream_ptr()
        if self._plan_stream is not None and s != self._plan_stream:
            # plans are bound to the stream they were recorded on
            self._fwd_plan = self._bwd_plan = None
            self._fwd_graph = self._bwd_graph = None
        self._plan_stream = s

    def load_inputs(self, x: torch.Tensor, timesteps: torch.Tensor):
        self.x_in.copy_(x.detach())
        self.t_dev.copy_(timesteps.detach().to(torch.float32))

    def run_forward(self):
        """Launch the forward plan on the current stream (inputs already in x_in / t_dev)."""
        self._check_stream()
        self.ticket += 1
        if self._fwd_plan is None:
            self._bwd_plan = None
            self._fwd_graph = self._bwd_graph = None
            with Recorder() as rec:
                self._forward_impl()
            self._fwd_plan = rec
            if self.use_graph:          # capture now (nothing executes) so that the next call is already a graph launch
                self._fwd_graph = self._fwd_plan.to_graph()
        elif self.use_graph:
            if self._fwd_graph is None:
                self._fwd_graph = self._fwd_plan.to_graph()
            self._fwd_graph.replay()
        else:
            self._fwd_plan.replay()

    def run_backward(self):
        """Launch the data-gradient plan (d_out holds dL/d(out), result lands in dx)."""
        self._check_stream()
        if self._fwd_plan is None:
            raise OsmosisHipError("backward before forward")
        if self._bwd_plan is None:
            with Recorder() as rec:
                self._backward_impl()
            self._bwd_plan = rec
            if self.use_graph:
                self._bwd_graph = self._bwd_plan.to_graph()
        elif self.use_graph:
            if self._bwd_graph is None:
                self._bwd_graph = self._bwd_plan.to_graph()
            self._bwd_graph.replay()
        else:
            self._bwd_plan.replay()

    def forward(self, x, timesteps, need_grad=True):
        if not x.is_cuda:
            raise OsmosisHipError("UNet input must live on the HIP device (no CPU fallback)")
        self.load_inputs(x, timesteps)
        self.run_forward()
        return self.out

    def backward(self, grad_out):
        self.d_out.copy_(grad_out)
        self.run_backward()
        return self.dx

    def n_launches(self):
        return (len(self._fwd_plan) if self._fwd_plan else 0, len(self._bwd_plan) if self._bwd_plan else 0)


class BlockEngine(UNetEngine):
    """ONE block of the UNet -- a ResBlock (unet.py:315-335; plain / 1x1 skip / up / down) or an AttentionBlock (:378-384,
    legacy or new head order) -- as a forward / input-gradient plan of its own: the same `_res_* / _attn_*` launch sequences,
    kernels and weight images the whole network uses, between an NCHW -> NHWC transpose and its inverse.  This is how the
    reference's block-level golden vectors (tests/golden/blocks.npz) reach the HIP path, so that a wrong kernel localises to
    a block instead of showing up as a whole-network mismatch.

        eng = BlockEngine(params, B, H, W, dev, conv_mode)     # params: ResBlockParams | AttentionParams (guided_diffusion/unet.py)
        y = eng.forward(x[B,cin,H,W], emb[B,emb_ch] | None);   dx = eng.backward(dy)"""

    def __init__(self, params, B: int, H: int, W: int, dev, conv_mode: str = "f32"):
        import types

        from .guided_diffusion.unet import ResBlockParams
        wfmt = ops.WFMT[conv_mode]
        is_res = isinstance(params, ResBlockParams)
        blk = _Res(params, dev, wfmt) if is_res else _Attn(params, dev, wfmt)
        cin = blk.cin if is_res else blk.ch
        cout = blk.cout if is_res else blk.ch
        ted = blk.ew.shape[1] if is_res else 1
        if is_res:
            blk.film_off = 0
            blk.film_cols = (2 if blk.scale_shift else 1) * blk.cout
        w = types.SimpleNamespace(dev=dev, conv_mode=conv_mode, mc=cin, ted=ted, cin=cin, cout=cout, nlev=1, te0=None, te2=None,
                                  inp=[], mid=[blk], outb=[], out_norm=None, out_conv=None,
                                  film_cols=blk.film_cols if is_res else 0,
                                  ew_all=blk.ew if is_res else None, eb_all=blk.eb if is_res else None, arch=None)
        super().__init__(w, B, H, W)
        self.block, self.is_res = blk, is_res
        up, down = is_res and blk.up, is_res and blk.down
        self.hwo = (2 * H, 2 * W) if up else ((H // 2, W // 2) if down else (H, W))
        f32 = dict(device=dev, dtype=torch.float32)
        self.out = torch.zeros(B, cout, *self.hwo, **f32)
        self.d_out = torch.zeros(B, cout, *self.hwo, **f32)
        self.emb = torch.zeros(B, ted, **f32)
        # (the GroupNorm partial-sum workspace is sized by the LARGEST tensor of the plan: an up block's output)
        self.gn_part = torch.empty(B * ops.gn_nchunk(max(H * W, self.hwo[0] * self.hwo[1])) * G * 2, **f32)

    def _to_nchw(self, m: Mat, dst, C, HW):
        if m.t.dtype != torch.float32:          # fp16-storage family: the sampler side of the boundary is fp32 NCHW
            m32 = self._buf(m.rows, m.cols, dtype=torch.float32)
            ops.convert(m, m32)
            m = m32
        ops.nhwc_to_nchw(m, dst, self.B, C, HW)

    def _forward_impl(self):
        B, H, W = self.B, self.H, self.W
        blk = self.block
        if self.is_res:
            self.film_all = torch.empty(B, self.film_cols, device=self.dev, dtype=torch.float32)
            ops.linear(self.emb, self.ew_all, self.eb_all, self.film_all, B, self.ted, self.film_cols, silu_in=True)
        x = self._buf(B * H * W, self.cin)
        ops.nchw_to_nhwc(self.x_in, x, B, self.cin, H * W)
        dst = self._buf(B * self.hwo[0] * self.hwo[1], self.cout)
        if self.is_res:
            self._res_fwd(blk, x, dst, (H, W))
        else:
            self._attn_fwd(blk, x, dst, (H, W))
        self._to_nchw(dst, self.out, self.cout, self.hwo[0] * self.hwo[1])

    def _backward_impl(self):
        B, H, W = self.B, self.H, self.W
        self._xmax_reg = {}
        dy = self._buf(B * self.hwo[0] * self.hwo[1], self.cout)
        ops.nchw_to_nhwc(self.d_out, dy, B, self.cout, self.hwo[0] * self.hwo[1])
        dx = self._buf(B * H * W, self.cin)
        if self.is_res:
            self._res_bwd(self.block, dy, dx, accumulate=False)
        else:
            self._attn_bwd(self.block, dy, dx, accumulate=False)
        self._to_nchw(dx, self.dx, self.cin, H * W)

    def forward(self, x, emb=None, need_grad=True):
        if not x.is_cuda:
            raise OsmosisHipError("block input must live on the HIP device (no CPU fallback)")
        self.x_in.copy_(x.detach())
        if emb is not None:
            self.emb.copy_(emb.detach())
        self.run_forward()
        return self.out
