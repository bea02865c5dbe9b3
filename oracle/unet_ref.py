"""fp32 CPU restatement of the reference UNet forward pass.

Each function names the reference code it follows (paths relative to the
reference root).  Written against a flat state_dict (oracle.weights), not the
reference classes, so it runs where /root/reference does not exist (GPU box).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import math

import torch
import torch.nn.functional as F

from .plan import UNetConfig, build_plan


def timestep_embedding(t, dim, max_period=10000):
    """ldm/modules/diffusionmodules/util.py:151-171 (cos first, then sin)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn(sd, p, x, eps):
    return F.group_norm(x, 32, sd[p + '.weight'], sd[p + '.bias'], eps)


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + '.weight'], sd[p + '.bias'], stride=stride, padding=padding)


def _lin(sd, p, x):
    return F.linear(x, sd[p + '.weight'], sd.get(p + '.bias'))


def res_block(sd, p, x, emb):
    """ResBlock._forward, openaimodel.py:255-275 (no up/down, no scale-shift norm).
    GroupNorm32 = nn.GroupNorm(32, C) eps 1e-5 in fp32, util.py:199-216."""
    h = _conv(sd, p + '.in_layers.2', F.silu(_gn(sd, p + '.in_layers.0', x, 1e-5)))
    e = _lin(sd, p + '.emb_layers.1', F.silu(emb))
    h = h + e[:, :, None, None]
    h = _conv(sd, p + '.out_layers.3', F.silu(_gn(sd, p + '.out_layers.0', h, 1e-5)))
    if (p + '.skip_connection.weight') in sd:
        x = _conv(sd, p + '.skip_connection', x, padding=0)
    return x + h


def cross_attention(sd, p, x, context, heads):
    """CrossAttention.forward, ldm/modules/attention.py:170-193 (mask unused)."""
    q = _lin(sd, p + '.to_q', x)
    ctx = x if context is None else context
    k = _lin(sd, p + '.to_k', ctx)
    v = _lin(sd, p + '.to_v', ctx)
    b, n, c = q.shape
    d = c // heads

    def split(t):  # 'b n (h d) -> (b h) n d'
        return t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3).reshape(b * heads, t.shape[1], d)

    q, k, v = split(q), split(k), split(v)
    sim = torch.bmm(q, k.transpose(1, 2)) * (d ** -0.5)
    attn = sim.softmax(dim=-1)
    out = torch.bmm(attn, v)
    out = out.reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, c)  # '(b h) n d -> b n (h d)'
    return _lin(sd, p + '.to_out.0', out)


def feed_forward(sd, p, x):
    """FeedForward with GEGLU, attention.py:37-64: value = first half, gate = second half, erf GELU."""
    y = _lin(sd, p + '.net.0.proj', x)
    a, g = y.chunk(2, dim=-1)
    return _lin(sd, p + '.net.2', a * F.gelu(g))


def transformer_block(sd, p, x, context, heads):
    """BasicTransformerBlock._forward, attention.py:211-215; LayerNorm eps 1e-5."""
    c = x.shape[-1]
    ln = lambda name, t: F.layer_norm(t, (c,), sd[f'{p}.{name}.weight'], sd[f'{p}.{name}.bias'], 1e-5)
    x = cross_attention(sd, p + '.attn1', ln('norm1', x), None, heads) + x
    x = cross_attention(sd, p + '.attn2', ln('norm2', x), context, heads) + x
    x = feed_forward(sd, p + '.ff', ln('norm3', x)) + x
    return x


def spatial_transformer(sd, p, x, context, heads, depth):
    """SpatialTransformer.forward, attention.py:250-261; its GroupNorm uses eps 1e-6 (attention.py:76-77)."""
    b, c, h, w = x.shape
    x_in = x
    x = _gn(sd, p + '.norm', x, 1e-6)
    x = _conv(sd, p + '.proj_in', x, padding=0)
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
    for d in range(depth):
        x = transformer_block(sd, f'{p}.transformer_blocks.{d}', x, context, heads)
    x = x.reshape(b, h, w, c).permute(0, 3, 1, 2)
    x = _conv(sd, p + '.proj_out', x, padding=0)
    return x + x_in


def _run_layer(sd, cfg, L, h, emb, context):
    if L.kind == 'conv_in':
        return _conv(sd, L.prefix, h)
    if L.kind == 'res':
        return res_block(sd, L.prefix, h, emb)
    if L.kind == 'attn':
        return spatial_transformer(sd, L.prefix, h, context, L.heads, cfg.transformer_depth)
    if L.kind == 'down':   # Downsample: conv3x3 stride 2 pad 1, openaimodel.py:149-153
        return _conv(sd, L.prefix + '.op', h, stride=2)
    if L.kind == 'up':     # Upsample: nearest x2 then conv3x3, openaimodel.py:109-119
        return _conv(sd, L.prefix + '.conv', F.interpolate(h, scale_factor=2, mode='nearest'))
    raise ValueError(L.kind)


@torch.no_grad()
def unet_forward(sd, cfg: UNetConfig, x, t, context, taps=None):
    """UNetModel.forward, openaimodel.py:710-742.  `taps`: optional dict that
    receives intermediate activations keyed by layer prefix (for per-module parity)."""
    plan = build_plan(cfg)
    emb = timestep_embedding(t, cfg.model_channels)
    emb = _lin(sd, 'time_embed.2', F.silu(_lin(sd, 'time_embed.0', emb)))
    if taps is not None:
        taps['emb'] = emb
    hs = []
    h = x.float()
    for blk in plan.input_blocks:
        for L in blk:
            h = _run_layer(sd, cfg, L, h, emb, context)
            if taps is not None:
                taps[L.prefix] = h
        hs.append(h)
    for L in plan.middle_block:
        h = _run_layer(sd, cfg, L, h, emb, context)
        if taps is not None:
            taps[L.prefix] = h
    for blk in plan.output_blocks:
        h = torch.cat([h, hs.pop()], dim=1)   # current h first, skip second (openaimodel.py:735-737)
        for L in blk:
            h = _run_layer(sd, cfg, L, h, emb, context)
            if taps is not None:
                taps[L.prefix] = h
    h = F.silu(_gn(sd, 'out.0', h, 1e-5))
    return _conv(sd, 'out.2', h)
