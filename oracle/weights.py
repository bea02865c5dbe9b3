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


def _student_t4(shape, g):
    """Student-t, nu = 4, unit variance: z / sqrt(chi2_4 / 4) / sqrt(2).  Only randn, +, *, sqrt and / (IEEE-exact ops on top
    of the generator), so the draw is reproducible on another host exactly like the randn-based tensors are."""
    z = torch.randn(shape, generator=g, dtype=torch.float32)
    c = torch.zeros(shape, dtype=torch.float32)
    for _ in range(4):
        c += torch.randn(shape, generator=g, dtype=torch.float32).square_()
    return z / (c * 0.25).sqrt_() * (1.0 / math.sqrt(2.0))


STYLES = ('uniform', 'realistic')
OUTLIER_FRACTION = 0.01     # 'realistic': share of output channels / norm channels with the gain below
OUTLIER_GAIN = 8.0


def make_state_dict(cfg: UNetConfig, seed: int = 0, dtype=torch.float32, style: str = 'uniform'):
    """Deterministic (torch CPU generator) synthetic weights.

    style 'uniform' (rounds 1-5): uniform +-1/sqrt(fan_in) weights, gamma = 1 + 0.1 N.
    style 'realistic' (round 6): what trained checkpoints have and the benign draw does not --
    heavy-tailed weights (Student-t, nu = 4, the SAME variance 1/(3 fan_in)), 1 % of the output channels of every
    conv / linear with a x8 gain on their weight rows, 1 % of every norm's channels with a x8 gamma."""
    assert style in STYLES, style
    g = torch.Generator().manual_seed(seed)
    real = style == 'realistic'
    sd = OrderedDict()
    for key, shape, kind in param_specs(cfg):
        if kind in ('w', 'wz'):
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            if real:
                std = math.sqrt(1.0 / (3.0 * fan_in)) if kind == 'w' else 0.5 / math.sqrt(fan_in)
                t = _student_t4(shape, g) * std
                rows = torch.rand(shape[0], generator=g) < OUTLIER_FRACTION
                if kind == 'w' and shape[0] >= 64:
                    t[rows] *= OUTLIER_GAIN
            elif kind == 'w':
                bound = 1.0 / math.sqrt(fan_in)
                t = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
            else:
                t = torch.randn(shape, generator=g, dtype=torch.float32) * (0.5 / math.sqrt(fan_in))
        elif kind == 'b':
            t = torch.randn(shape, generator=g, dtype=torch.float32) * 0.02
        elif kind == 'gamma':
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
            if real:
                t[torch.rand(shape[0], generator=g) < OUTLIER_FRACTION] *= OUTLIER_GAIN
        elif kind == 'beta':
            t = 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
        else:
            raise ValueError(kind)
        sd[key] = t.to(dtype)
    return sd


def make_inputs(cfg: UNetConfig, batch: int, h: int, w: int, seed: int = 1, ctx_len: int = 77,
                timesteps=(981, 481, 1, 741), style: str = 'uniform'):
    """Seeded (x_t, t, context) triple; different context per batch row (SURVEY 8c hygiene 4).

    style 'realistic': the context carries three outlier channels at |x| ~ 30 on every token (what CLIP ViT-L/14's
    last_hidden_state has) over unit-variance channels; rows whose timestep is < 500 get an x_t at the scale and
    smoothness of a nearly clean latent (a low-resolution field upsampled x4, std ~ 0.9, plus sqrt(1 - a_t)-sized noise)
    instead of white noise."""
    assert style in STYLES, style
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, cfg.in_channels, h, w, generator=g)
    ctx = torch.randn(batch, ctx_len, cfg.context_dim, generator=g)
    t = torch.tensor([timesteps[i % len(timesteps)] for i in range(batch)], dtype=torch.int64)
    if style == 'realistic':
        ch = torch.randperm(cfg.context_dim, generator=g)[:3]
        sign = torch.tensor([1.0, -1.0, 1.0])
        ctx[:, :, ch] = ctx[:, :, ch] * 1.5 + 30.0 * sign
        lo = torch.randn(batch, cfg.in_channels, (h + 3) // 4, (w + 3) // 4, generator=g) * 0.9
        smooth = lo.repeat_interleave(4, 2).repeat_interleave(4, 3)[:, :, :h, :w]
        for i in range(batch):
            if int(t[i]) < 500:
                x[i] = smooth[i] + 0.05 * x[i]
    return x, t, ctx
