"""AutoencoderKL *decoder* on stock PyTorch-ROCm (north_star: "the AutoencoderKL decode ... stays on PyTorch-ROCm").

Not part of the HIP hot path; it exists so bench.py can time the metric's full unit of work
(`sampler.sample` + `decode_first_stage`, scripts/txt2img.py:303-315) without the reference's
pytorch_lightning / taming imports.  Architecture and parameter names follow ldm/modules/diffusionmodules/model.py
`Decoder` (:462-568), `ResnetBlock` (:82-141), `AttnBlock` (:150-202) and `AutoencoderKL.decode`
(ldm/models/autoencoder.py:330-333) so a real `first_stage_model.*` state_dict loads into it.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _gn(c):
    return nn.GroupNorm(32, c, eps=1e-6, affine=True)


class ResnetBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1, self.conv1 = _gn(cin), nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2, self.conv2 = _gn(cout), nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (self.nin_shortcut(x) if hasattr(self, 'nin_shortcut') else x) + h


class AttnBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm = _gn(c)
        self.q, self.k, self.v, self.proj_out = (nn.Conv2d(c, c, 1) for _ in range(4))

    def forward(self, x):
        b, c, h, w = x.shape
        hn = self.norm(x)
        q, k, v = (m(hn).reshape(b, 1, c, h * w).transpose(2, 3) for m in (self.q, self.k, self.v))
        o = F.scaled_dot_product_attention(q, k, v)        # single head, scale c^-0.5 (model.py:186)
        return x + self.proj_out(o.transpose(2, 3).reshape(b, c, h, w))


class Upsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode='nearest'))


class Decoder(nn.Module):
    def __init__(self, ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4):
        super().__init__()
        n = len(ch_mult)
        block_in = ch * ch_mult[-1]
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, padding=1)
        self.mid = nn.Module()
        self.mid.block_1, self.mid.attn_1, self.mid.block_2 = ResnetBlock(block_in, block_in), AttnBlock(block_in), \
            ResnetBlock(block_in, block_in)
        self.up = nn.ModuleList()
        for lvl in reversed(range(n)):
            up = nn.Module()
            up.block = nn.ModuleList()
            up.attn = nn.ModuleList()
            for _ in range(num_res_blocks + 1):
                up.block.append(ResnetBlock(block_in, ch * ch_mult[lvl]))
                block_in = ch * ch_mult[lvl]
            if lvl != 0:
                up.upsample = Upsample(block_in)
            self.up.insert(0, up)
        self.norm_out, self.conv_out = _gn(block_in), nn.Conv2d(block_in, out_ch, 3, padding=1)

    def forward(self, z):
        h = self.conv_in(z)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for lvl in reversed(range(len(self.up))):
            for blk in self.up[lvl].block:
                h = blk(h)
            if lvl != 0:
                h = self.up[lvl].upsample(h)
        return self.conv_out(F.silu(self.norm_out(h)))


class AutoencoderKLDecoder(nn.Module):
    """decode_first_stage (ldm/models/diffusion/ddpm.py:705-763): z / scale_factor -> post_quant_conv -> Decoder."""

    def __init__(self, scale_factor=0.18215, embed_dim=4, z_channels=4):
        super().__init__()
        self.scale_factor = scale_factor
        self.post_quant_conv = nn.Conv2d(embed_dim, z_channels, 1)
        self.decoder = Decoder(z_channels=z_channels)

    @torch.no_grad()
    def decode_first_stage(self, z):
        return self.decoder(self.post_quant_conv(z / self.scale_factor))
