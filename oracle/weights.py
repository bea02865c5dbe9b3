"""Seeded synthetic UNet weights under the reference's state_dict key names.

Key names / shapes follow SURVEY.md appendix B (= `UNetModel.state_dict()` of
ldm/modules/diffusionmodules/openaimodel.py:443-692 and ldm/modules/attention.py).
`oracle/make_golden.py` loads the result into the real reference `UNetModel`
with `strict=True`, which is what pins this list to the reference.

Default-initialised reference weights give eps == 0 exactly because of
`zero_module` (openaimodel.py:229-231,685; attention.py:244-248), so those
tensors are re-randomised here with std 0.5/sqrt(fan_in) (SURVEY.md 8c hygiene).
Norm affine parameters are randomised too so a swapped gamma/beta is caught.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import math
from collections import OrderedDict

import torch

from .plan import UNetConfig, build_plan


def param_specs(cfg: UNetConfig):
    """Ordered [(key, shape, kind)], kind in {'w','wz','b','gamma','beta'}.
    'wz' marks tensors the reference zero-initialises."""
    mc = cfg.model_channels
    te = 4 * mc
    specs = []

    def lin(prefix, cout, cin, bias=True, zero=False):
        specs.append((prefix + '.weight', (cout, cin), 'wz' if zero else 'w'))
        if bias:
            specs.append((prefix + '.bias', (cout,), 'b'))

    def conv(prefix, cout, cin, k, zero=False):
        specs.append((prefix + '.weight', (cout, cin, k, k), 'wz' if zero else 'w'))
        specs.append((prefix + '.bias', (cout,), 'b'))

    def norm(prefix, c):
        specs.append((prefix + '.weight', (c,), 'gamma'))
        specs.append((prefix + '.bias', (c,), 'beta'))

    lin('time_embed.0', te, mc)
    lin('time_embed.2', te, te)
    plan = build_plan(cfg)
    for L in plan.all_layers():
        p = L.prefix
        if L.kind == 'conv_in':
            conv(p, L.cout, L.cin, 3)
        elif L.kind == 'res':
            norm(p + '.in_layers.0', L.cin)
            conv(p + '.in_layers.2', L.cout, L.cin, 3)
            lin(p + '.emb_layers.1', L.cout, te)
            norm(p + '.out_layers.0', L.cout)
            conv(p + '.out_layers.3', L.cout, L.cout, 3, zero=True)
            if L.cin != L.cout:
                conv(p + '.skip_connection', L.cout, L.cin, 1)
        elif L.kind == 'attn':
            c = L.cin
            norm(p + '.norm', c)
            conv(p + '.proj_in', c, c, 1)
            for d in range(cfg.transformer_depth):
                t = f'{p}.transformer_blocks.{d}'
                for a, kd in (('attn1', c), ('attn2', cfg.context_dim)):
                    lin(f'{t}.{a}.to_q', c, c, bias=False)
                    lin(f'{t}.{a}.to_k', c, kd, bias=False)
                    lin(f'{t}.{a}.to_v', c, kd, bias=False)
                    lin(f'{t}.{a}.to_out.0', c, c)
                lin(f'{t}.ff.net.0.proj', 8 * c, c)
                lin(f'{t}.ff.net.2', c, 4 * c)
                norm(f'{t}.norm1', c)
                norm(f'{t}.norm2', c)
                norm(f'{t}.norm3', c)
            conv(p + '.proj_out', c, c, 1, zero=True)
        elif L.kind == 'down':
            conv(p + '.op', L.cout, L.cin, 3)
        elif L.kind == 'up':
            conv(p + '.conv', L.cout, L.cin, 3)
    norm('out.0', mc)
    conv('out.2', cfg.out_channels, mc, 3, zero=True)
    return specs


def make_state_dict(cfg: UNetConfig, seed: int = 0, dtype=torch.float32):
    """Deterministic (torch CPU generator) synthetic weights."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for key, shape, kind in param_specs(cfg):
        if kind in ('w', 'wz'):
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            if kind == 'w':
                bound = 1.0 / math.sqrt(fan_in)
                t = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
            else:
                t = torch.randn(shape, generator=g, dtype=torch.float32) * (0.5 / math.sqrt(fan_in))
        elif kind == 'b':
            t = torch.randn(shape, generator=g, dtype=torch.float32) * 0.02
        elif kind == 'gamma':
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
        elif kind == 'beta':
            t = 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
        else:
            raise ValueError(kind)
        sd[key] = t.to(dtype)
    return sd


def make_inputs(cfg: UNetConfig, batch: int, h: int, w: int, seed: int = 1, ctx_len: int = 77,
                timesteps=(981, 481, 1, 741)):
    """Seeded (x_t, t, context) triple; different context per batch row (SURVEY 8c hygiene 4)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, cfg.in_channels, h, w, generator=g)
    ctx = torch.randn(batch, ctx_len, cfg.context_dim, generator=g)
    t = torch.tensor([timesteps[i % len(timesteps)] for i in range(batch)], dtype=torch.int64)
    return x, t, ctx
