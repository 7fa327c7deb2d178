"""Functional CPU restatement of the reference UNet (TEST INFRASTRUCTURE, see oracle/__init__.py).

Walks a reference-layout state_dict (keys as produced by guided_diffusion/unet.py UNetModel)
and evaluates the network with plain torch-CPU fp32 ops, keeping autograd so that
`torch.autograd.grad(out, x)` yields the input gradient the guidance step needs.

Follows (reference file:line):
  * UNetModel.forward                 guided_diffusion/unet.py:713-742
  * ResBlock._forward                 guided_diffusion/unet.py:315-335
  * AttentionBlock._forward           guided_diffusion/unet.py:378-384
  * QKVAttentionLegacy.forward        guided_diffusion/unet.py:416-433
  * QKVAttention.forward              guided_diffusion/unet.py:449-468
  * GroupNorm32 / timestep_embedding  guided_diffusion/nn.py:17-19, 103-121
  * Upsample / Downsample (conv-less) guided_diffusion/unet.py:179-189, 217-219
  * round 5, the variants no shipped config uses: Upsample / Downsample WITH convolution as layers of their own
    (`resblock_updown=False`, unet.py:160-219, 600-612, 667-679), `use_scale_shift_norm=False` (:329-332), the class
    embedding (`num_classes`, :556-557, 729-731)
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


@dataclass
class UNetConfig:
    """Architecture hyper-parameters, in the units UNetModel.__init__ uses (unet.py:506-527)."""
    in_channels: int = 4
    out_channels: int = 8
    model_channels: int = 256
    num_res_blocks: int = 2
    channel_mult: Tuple[int, ...] = (1, 1, 2, 2, 4, 4)
    attention_ds: Tuple[int, ...] = (8, 16, 32)      # downsample rates that get attention
    num_heads: int = 4
    num_head_channels: int = 64
    use_scale_shift_norm: bool = True
    resblock_updown: bool = True
    use_new_attention_order: bool = False
    conv_resample: bool = True          # UNetModel's default; create_model never changes it
    num_classes: int = 0                # > 0: class-conditional (label_emb)

    @staticmethod
    def from_create_model_kwargs(image_size, num_channels, num_res_blocks, channel_mult="",
                                 attention_resolutions="16", num_heads=1, num_head_channels=-1,
                                 use_scale_shift_norm=False, resblock_updown=False,
                                 use_new_attention_order=False, learn_sigma=False,
                                 pretrain_model="", class_cond=False, **_unused):
        """Mirror of create_model's argument digestion (unet.py:47-68, 91-92)."""
        if channel_mult == "":
            table = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4),
                     128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}
            if image_size not in table:
                raise ValueError(f"unsupported image size: {image_size}")
            cm = table[image_size]
        else:
            cm = tuple(int(c) for c in channel_mult.split(","))
        if isinstance(attention_resolutions, int):
            ads = (image_size // attention_resolutions,)
        else:
            ads = tuple(image_size // int(r) for r in attention_resolutions.split(","))
        osm = pretrain_model == "osmosis"
        return UNetConfig(in_channels=4 if osm else 3,
                          out_channels=8 if osm else (6 if learn_sigma else 3),
                          model_channels=num_channels, num_res_blocks=num_res_blocks,
                          channel_mult=cm, attention_ds=ads, num_heads=num_heads,
                          num_head_channels=num_head_channels,
                          use_scale_shift_norm=use_scale_shift_norm,
                          resblock_updown=resblock_updown,
                          use_new_attention_order=use_new_attention_order,
                          num_classes=1000 if class_cond else 0)      # NUM_CLASSES, unet.py:23


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """nn.py:103-121 -- cos first, then sin."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def group_norm32(x, w, b):
    """nn.py:17-19 : GroupNorm(32, C) evaluated in fp32, eps 1e-5."""
    return F.group_norm(x.float(), 32, w, b, eps=1e-5).type(x.dtype)


def res_block(sd: Dict[str, torch.Tensor], p: str, x, emb, up=False, down=False,
              scale_shift=True):
    """unet.py:315-335."""
    h = F.silu(group_norm32(x, sd[p + "in_layers.0.weight"], sd[p + "in_layers.0.bias"]))
    if up:
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif down:
        h = F.avg_pool2d(h, 2, 2)
        x = F.avg_pool2d(x, 2, 2)
    h = F.conv2d(h, sd[p + "in_layers.2.weight"], sd[p + "in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[p + "emb_layers.1.weight"], sd[p + "emb_layers.1.bias"])
    e = e[..., None, None]
    if scale_shift:
        scale, shift = torch.chunk(e, 2, dim=1)
        h = group_norm32(h, sd[p + "out_layers.0.weight"], sd[p + "out_layers.0.bias"]) * (1 + scale) + shift
        h = F.silu(h)
    else:
        h = h + e
        h = F.silu(group_norm32(h, sd[p + "out_layers.0.weight"], sd[p + "out_layers.0.bias"]))
    h = F.conv2d(h, sd[p + "out_layers.3.weight"], sd[p + "out_layers.3.bias"], padding=1)
    if (p + "skip_connection.weight") in sd:
        w = sd[p + "skip_connection.weight"]
        x = F.conv2d(x, w, sd[p + "skip_connection.bias"], padding=w.shape[-1] // 2)
    return x + h


def attention_block(sd, p: str, x, n_heads: int, new_order=False):
    """unet.py:378-384 + 416-433 (legacy) / 449-468 (new order)."""
    b, c, *spatial = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(group_norm32(xf, sd[p + "norm.weight"], sd[p + "norm.bias"]),
                   sd[p + "qkv.weight"], sd[p + "qkv.bias"])
    bs, width, length = qkv.shape
    ch = width // (3 * n_heads)
    scale = 1 / math.sqrt(math.sqrt(ch))
    if new_order:
        q, k, v = qkv.chunk(3, dim=1)
        q = q.reshape(bs * n_heads, ch, length)
        k = k.reshape(bs * n_heads, ch, length)
        v = v.reshape(bs * n_heads, ch, length)
    else:
        q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(bs, -1, length)
    h = F.conv1d(a, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])
    return (xf + h).reshape(b, c, *spatial)


@dataclass
class _Layout:
    """Which sub-modules each TimestepEmbedSequential holds; derived exactly as
    UNetModel.__init__ builds them (unet.py:559-695)."""
    input_blocks: List[List[Tuple[str, dict]]] = field(default_factory=list)
    middle: List[Tuple[str, dict]] = field(default_factory=list)
    output_blocks: List[List[Tuple[str, dict]]] = field(default_factory=list)


def _heads(cfg: UNetConfig, ch: int) -> int:
    return cfg.num_heads if cfg.num_head_channels == -1 else ch // cfg.num_head_channels


def build_layout(cfg: UNetConfig) -> _Layout:
    lay = _Layout()
    mc = cfg.model_channels
    ch = int(cfg.channel_mult[0] * mc)
    lay.input_blocks.append([("conv", {})])
    chans = [ch]
    ds = 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            layers = [("res", {})]
            ch = int(mult * mc)
            if ds in cfg.attention_ds:
                layers.append(("attn", {"heads": _heads(cfg, ch)}))
            lay.input_blocks.append(layers)
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            lay.input_blocks.append([("res", {"down": True})] if cfg.resblock_updown else [("down", {})])
            chans.append(ch)
            ds *= 2
    lay.middle = [("res", {}), ("attn", {"heads": _heads(cfg, ch)}), ("res", {})]
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            chans.pop()
            layers = [("res", {})]
            ch = int(mc * mult)
            if ds in cfg.attention_ds:
                layers.append(("attn", {"heads": _heads(cfg, ch)}))
            if level and i == cfg.num_res_blocks:
                layers.append(("res", {"up": True}) if cfg.resblock_updown else ("up", {}))
                ds //= 2
            lay.output_blocks.append(layers)
    return lay


def _run_seq(sd, cfg, prefix, layers, h, emb):
    for j, (kind, kw) in enumerate(layers):
        p = f"{prefix}{j}."
        if kind == "conv":
            h = F.conv2d(h, sd[p + "weight"], sd[p + "bias"], padding=1)
        elif kind == "res":
            h = res_block(sd, p, h, emb, scale_shift=cfg.use_scale_shift_norm, **kw)
        elif kind == "down":        # Downsample, unet.py:192-219: 3x3 conv with stride 2, or 2x2 average pooling
            h = F.conv2d(h, sd[p + "op.weight"], sd[p + "op.bias"], stride=2, padding=1) if cfg.conv_resample \
                else F.avg_pool2d(h, 2, 2)
        elif kind == "up":          # Upsample, unet.py:160-189: nearest 2x, then an optional 3x3 conv
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            if cfg.conv_resample:
                h = F.conv2d(h, sd[p + "conv.weight"], sd[p + "conv.bias"], padding=1)
        else:
            h = attention_block(sd, p, h, kw["heads"], cfg.use_new_attention_order)
    return h


def unet_forward(sd: Dict[str, torch.Tensor], cfg: UNetConfig, x: torch.Tensor,
                 timesteps: torch.Tensor, y: torch.Tensor = None) -> torch.Tensor:
    """unet.py:713-742.  x [B,Cin,H,W] fp32 NCHW, timesteps [B] -> [B,Cout,H,W]; y [B] class labels iff num_classes > 0."""
    assert (y is not None) == (cfg.num_classes > 0), "must specify y if and only if the model is class-conditional"
    lay = build_layout(cfg)
    emb = timestep_embedding(timesteps, cfg.model_channels)
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    if cfg.num_classes > 0:
        emb = emb + sd["label_emb.weight"][y]
    hs = []
    h = x
    for i, layers in enumerate(lay.input_blocks):
        h = _run_seq(sd, cfg, f"input_blocks.{i}.", layers, h, emb)
        hs.append(h)
    h = _run_seq(sd, cfg, "middle_block.", lay.middle, h, emb)
    for i, layers in enumerate(lay.output_blocks):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_seq(sd, cfg, f"output_blocks.{i}.", layers, h, emb)
    h = F.silu(group_norm32(h, sd["out.0.weight"], sd["out.0.bias"]))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


def param_shapes(cfg: UNetConfig) -> Dict[str, Tuple[int, ...]]:
    """All state_dict keys + shapes of the reference module for this config."""
    shapes: Dict[str, Tuple[int, ...]] = {}
    mc = cfg.model_channels
    ted = mc * 4
    shapes["time_embed.0.weight"] = (ted, mc)
    shapes["time_embed.0.bias"] = (ted,)
    shapes["time_embed.2.weight"] = (ted, ted)
    shapes["time_embed.2.bias"] = (ted,)
    if cfg.num_classes > 0:
        shapes["label_emb.weight"] = (cfg.num_classes, ted)

    def res(p, cin, cout):
        shapes[p + "in_layers.0.weight"] = (cin,)
        shapes[p + "in_layers.0.bias"] = (cin,)
        shapes[p + "in_layers.2.weight"] = (cout, cin, 3, 3)
        shapes[p + "in_layers.2.bias"] = (cout,)
        eo = 2 * cout if cfg.use_scale_shift_norm else cout
        shapes[p + "emb_layers.1.weight"] = (eo, ted)
        shapes[p + "emb_layers.1.bias"] = (eo,)
        shapes[p + "out_layers.0.weight"] = (cout,)
        shapes[p + "out_layers.0.bias"] = (cout,)
        shapes[p + "out_layers.3.weight"] = (cout, cout, 3, 3)
        shapes[p + "out_layers.3.bias"] = (cout,)
        if cin != cout:
            shapes[p + "skip_connection.weight"] = (cout, cin, 1, 1)
            shapes[p + "skip_connection.bias"] = (cout,)

    def attn(p, c):
        shapes[p + "norm.weight"] = (c,)
        shapes[p + "norm.bias"] = (c,)
        shapes[p + "qkv.weight"] = (3 * c, c, 1)
        shapes[p + "qkv.bias"] = (3 * c,)
        shapes[p + "proj_out.weight"] = (c, c, 1)
        shapes[p + "proj_out.bias"] = (c,)

    lay = build_layout(cfg)
    ch = int(cfg.channel_mult[0] * mc)
    shapes["input_blocks.0.0.weight"] = (ch, cfg.in_channels, 3, 3)
    shapes["input_blocks.0.0.bias"] = (ch,)
    chans = [ch]
    idx = 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            cout = int(mult * mc)
            res(f"input_blocks.{idx}.0.", ch, cout)
            ch = cout
            if len(lay.input_blocks[idx]) > 1:
                attn(f"input_blocks.{idx}.1.", ch)
            chans.append(ch)
            idx += 1
        if level != len(cfg.channel_mult) - 1:
            if cfg.resblock_updown:
                res(f"input_blocks.{idx}.0.", ch, ch)
            elif cfg.conv_resample:
                shapes[f"input_blocks.{idx}.0.op.weight"] = (ch, ch, 3, 3)
                shapes[f"input_blocks.{idx}.0.op.bias"] = (ch,)
            chans.append(ch)
            idx += 1
    res("middle_block.0.", ch, ch)
    attn("middle_block.1.", ch)
    res("middle_block.2.", ch, ch)
    idx = 0
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            cout = int(mc * mult)
            res(f"output_blocks.{idx}.0.", ch + ich, cout)
            ch = cout
            for j, (kind, _kw) in enumerate(lay.output_blocks[idx][1:], start=1):
                if kind == "attn":
                    attn(f"output_blocks.{idx}.{j}.", ch)
                elif kind == "res":
                    res(f"output_blocks.{idx}.{j}.", ch, ch)
                elif cfg.conv_resample:
                    shapes[f"output_blocks.{idx}.{j}.conv.weight"] = (ch, ch, 3, 3)
                    shapes[f"output_blocks.{idx}.{j}.conv.bias"] = (ch,)
            idx += 1
    shapes["out.0.weight"] = (ch,)
    shapes["out.0.bias"] = (ch,)
    shapes["out.2.weight"] = (cfg.out_channels, ch, 3, 3)
    shapes["out.2.bias"] = (cfg.out_channels,)
    return shapes


def seeded_state_dict(cfg: UNetConfig, seed: int = 1234, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Synthetic non-degenerate weights: every parameter,
    including the reference's zero-initialised modules, drawn from one seeded generator:
    conv/linear weights N(0, (0.5/sqrt(fan_in))^2), biases 0.05*N, GroupNorm weights 1+0.1*N.
    Iteration order = insertion order of param_shapes() (deterministic)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in param_shapes(cfg).items():
        if len(shape) == 1:
            v = torch.randn(shape, generator=g, dtype=torch.float32)
            is_norm = (".in_layers.0." in name or ".out_layers.0." in name or ".norm." in name
                       or name.startswith("out.0."))
            if name.endswith("weight") and is_norm:
                t = 1.0 + 0.1 * v
            else:
                t = 0.05 * v
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g, dtype=torch.float32) * (0.5 / math.sqrt(fan_in))
        sd[name] = t.to(dtype)
    return sd
