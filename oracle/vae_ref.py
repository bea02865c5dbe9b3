"""CPU restatement (fp32, functional torch) of the AutoencoderKL first stage -- SURVEY.md 8(f-1).

Follows, op for op:
  * `AutoencoderKL.decode` / `.encode`            ldm/models/autoencoder.py:324-333
  * `Decoder.forward`                              ldm/modules/diffusionmodules/model.py:528-568 (ctor :462-526)
  * `Encoder.forward`                              model.py:427-460 (ctor :368-425)
  * `ResnetBlock.forward` (temb is None)           model.py:119-141
  * `AttnBlock.forward` (1 head, d = C)            model.py:172-202
  * `Upsample` / `Downsample` (pad (0,1,0,1))      model.py:41-79
  * `Normalize` = GroupNorm(32, eps=1e-6)          model.py:37-38
  * `LatentDiffusion.decode_first_stage` scaling   ldm/models/diffusion/ddpm.py:713

`oracle/make_golden.py` loads `make_vae_state_dict` into the real reference `Encoder` / `Decoder` (strict=True),
asserts this restatement equals them and freezes the reference outputs as tests/golden/vae_*.npz.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the product path (stable-diffusion_amd/) never imports this.
"""
import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Tuple

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class VAEConfig:
    ch: int = 128
    out_ch: int = 3
    ch_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    in_channels: int = 3
    z_channels: int = 4
    embed_dim: int = 4

    def ddconfig(self):
        """kwargs of the reference Encoder / Decoder (configs/stable-diffusion/v1-inference.yaml:51-65)."""
        return dict(double_z=True, z_channels=self.z_channels, resolution=256, in_channels=self.in_channels,
                    out_ch=self.out_ch, ch=self.ch, ch_mult=list(self.ch_mult), num_res_blocks=self.num_res_blocks,
                    attn_resolutions=[], dropout=0.0)


SD_VAE = VAEConfig()
TINY_VAE = VAEConfig(ch=64, ch_mult=(1, 2), num_res_blocks=1)
SMALL_VAE = VAEConfig(ch=64, ch_mult=(1, 2, 2), num_res_blocks=2)


# ------------------------------------------------------------------------------------------------------------------
# parameter list (= state_dict keys of AutoencoderKL minus `loss.*`)
# ------------------------------------------------------------------------------------------------------------------
def vae_param_specs(cfg: VAEConfig, encoder=True, decoder=True):
    specs = []

    def conv(p, co, ci, k):
        specs.append((p + '.weight', (co, ci, k, k), 'w'))
        specs.append((p + '.bias', (co,), 'b'))

    def norm(p, c):
        specs.append((p + '.weight', (c,), 'gamma'))
        specs.append((p + '.bias', (c,), 'beta'))

    def res(p, ci, co):
        norm(p + '.norm1', ci); conv(p + '.conv1', co, ci, 3)
        norm(p + '.norm2', co); conv(p + '.conv2', co, co, 3)
        if ci != co:
            conv(p + '.nin_shortcut', co, ci, 1)

    def attn(p, c):
        norm(p + '.norm', c)
        for n in ('q', 'k', 'v', 'proj_out'):
            conv(f'{p}.{n}', c, c, 1)

    n = len(cfg.ch_mult)
    if encoder:
        conv('encoder.conv_in', cfg.ch, cfg.in_channels, 3)
        in_mult = (1,) + tuple(cfg.ch_mult)
        block_in = cfg.ch
        for lvl in range(n):
            block_in = cfg.ch * in_mult[lvl]
            for i in range(cfg.num_res_blocks):
                res(f'encoder.down.{lvl}.block.{i}', block_in, cfg.ch * cfg.ch_mult[lvl])
                block_in = cfg.ch * cfg.ch_mult[lvl]
            if lvl != n - 1:
                conv(f'encoder.down.{lvl}.downsample.conv', block_in, block_in, 3)
        res('encoder.mid.block_1', block_in, block_in)
        attn('encoder.mid.attn_1', block_in)
        res('encoder.mid.block_2', block_in, block_in)
        norm('encoder.norm_out', block_in)
        conv('encoder.conv_out', 2 * cfg.z_channels, block_in, 3)
        conv('quant_conv', 2 * cfg.embed_dim, 2 * cfg.z_channels, 1)
    if decoder:
        conv('post_quant_conv', cfg.z_channels, cfg.embed_dim, 1)
        block_in = cfg.ch * cfg.ch_mult[-1]
        conv('decoder.conv_in', block_in, cfg.z_channels, 3)
        res('decoder.mid.block_1', block_in, block_in)
        attn('decoder.mid.attn_1', block_in)
        res('decoder.mid.block_2', block_in, block_in)
        for lvl in reversed(range(n)):
            for i in range(cfg.num_res_blocks + 1):
                res(f'decoder.up.{lvl}.block.{i}', block_in, cfg.ch * cfg.ch_mult[lvl])
                block_in = cfg.ch * cfg.ch_mult[lvl]
            if lvl != 0:
                conv(f'decoder.up.{lvl}.upsample.conv', block_in, block_in, 3)
        norm('decoder.norm_out', block_in)
        conv('decoder.conv_out', cfg.out_ch, block_in, 3)
    return specs


def make_vae_state_dict(cfg: VAEConfig, seed: int = 0, encoder=True, decoder=True):
    """Seeded synthetic weights (same recipe as oracle/weights.py): uniform weights, small biases,
    perturbed norm affines so a swapped gamma/beta is caught.  Weights are U(+-sqrt(3/fan_in)) here (unit gain)."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    keep = {k for k, _, _ in vae_param_specs(cfg, encoder, decoder)}
    for key, shape, kind in vae_param_specs(cfg):      # always draw the full list: a subset holds the same tensors
        if kind == 'w':
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            # sqrt(3) x the torch default bound: unit-variance-preserving, keeps the decoded image O(1)
            t = (torch.rand(shape, generator=g) * 2 - 1) * math.sqrt(3.0 / fan_in)
        elif kind == 'b':
            t = torch.randn(shape, generator=g) * 0.02
        elif kind == 'gamma':
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = 0.1 * torch.randn(shape, generator=g)
        if key in keep:
            sd[key] = t.float()
    return sd


# ------------------------------------------------------------------------------------------------------------------
# forward
# ------------------------------------------------------------------------------------------------------------------
def _gn(sd, p, x):
    return F.group_norm(x, 32, sd[p + '.weight'], sd[p + '.bias'], eps=1e-6)


def _conv(sd, p, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + '.weight'], sd[p + '.bias'], stride=stride, padding=padding)


def _swish(x):
    return x * torch.sigmoid(x)


def _res(sd, p, x):
    h = _conv(sd, p + '.conv1', _swish(_gn(sd, p + '.norm1', x)))
    h = _conv(sd, p + '.conv2', _swish(_gn(sd, p + '.norm2', h)))
    if p + '.nin_shortcut.weight' in sd:
        x = _conv(sd, p + '.nin_shortcut', x, padding=0)
    return x + h


def _attn(sd, p, x):
    b, c, hh, ww = x.shape
    hn = _gn(sd, p + '.norm', x)
    q = _conv(sd, p + '.q', hn, padding=0).reshape(b, c, hh * ww).permute(0, 2, 1)     # [b, hw, c]
    k = _conv(sd, p + '.k', hn, padding=0).reshape(b, c, hh * ww)                      # [b, c, hw]
    v = _conv(sd, p + '.v', hn, padding=0).reshape(b, c, hh * ww)
    w_ = torch.softmax(torch.bmm(q, k) * (int(c) ** (-0.5)), dim=2)                     # [b, hw_q, hw_k]
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, p + '.proj_out', h_, padding=0)


@torch.no_grad()
def vae_decode(sd, cfg: VAEConfig, z):
    """AutoencoderKL.decode: z [B, embed_dim, h, w] fp32 -> image [B, out_ch, 8h, 8w] (for 4 levels)."""
    h = _conv(sd, 'post_quant_conv', z.float(), padding=0)
    h = _conv(sd, 'decoder.conv_in', h)
    h = _res(sd, 'decoder.mid.block_1', h)
    h = _attn(sd, 'decoder.mid.attn_1', h)
    h = _res(sd, 'decoder.mid.block_2', h)
    for lvl in reversed(range(len(cfg.ch_mult))):
        for i in range(cfg.num_res_blocks + 1):
            h = _res(sd, f'decoder.up.{lvl}.block.{i}', h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode='nearest')
            h = _conv(sd, f'decoder.up.{lvl}.upsample.conv', h)
    return _conv(sd, 'decoder.conv_out', _swish(_gn(sd, 'decoder.norm_out', h)))


@torch.no_grad()
def vae_encode_moments(sd, cfg: VAEConfig, x):
    """AutoencoderKL.encode up to the posterior's parameters: image [B,3,H,W] -> moments [B, 2*embed_dim, H/8, W/8]
    (mean = first half, logvar = second half, clamped to [-30, 20] by DiagonalGaussianDistribution)."""
    n = len(cfg.ch_mult)
    h = _conv(sd, 'encoder.conv_in', x.float())
    for lvl in range(n):
        for i in range(cfg.num_res_blocks):
            h = _res(sd, f'encoder.down.{lvl}.block.{i}', h)
        if lvl != n - 1:
            h = F.pad(h, (0, 1, 0, 1), mode='constant', value=0)
            h = _conv(sd, f'encoder.down.{lvl}.downsample.conv', h, stride=2, padding=0)
    h = _res(sd, 'encoder.mid.block_1', h)
    h = _attn(sd, 'encoder.mid.attn_1', h)
    h = _res(sd, 'encoder.mid.block_2', h)
    h = _conv(sd, 'encoder.conv_out', _swish(_gn(sd, 'encoder.norm_out', h)))
    return _conv(sd, 'quant_conv', h, padding=0)


def decode_first_stage(sd, cfg: VAEConfig, z, scale_factor=0.18215):
    """LatentDiffusion.decode_first_stage (ddpm.py:713, :763): z / scale_factor -> decode."""
    return vae_decode(sd, cfg, z / scale_factor)


def make_vae_inputs(cfg: VAEConfig, batch, h, w, seed=1):
    """Seeded latent with the statistics of a finished sample divided by scale_factor (std ~ 5)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, cfg.embed_dim, h, w, generator=g) * 5.0
