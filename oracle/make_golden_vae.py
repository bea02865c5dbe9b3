"""Generate tests/golden/vae_*.npz by running the REFERENCE first-stage modules (build container only).

    PYTHONPATH=/root/repo python oracle/make_golden_vae.py

Imports `/root/reference/ldm/modules/diffusionmodules/model.py` `Encoder` / `Decoder` (the `AutoencoderKL` Lightning
wrapper itself needs pytorch_lightning + taming, which are absent; its `decode` / `encode` are two lines --
autoencoder.py:324-333 -- and are restated here around the real Encoder / Decoder plus the two 1x1 convs), loads
`oracle.vae_ref.make_vae_state_dict` with strict=True, asserts the oracle restatement equals the reference and stores
the reference outputs as fixtures.  The GPU box has no /root/reference: tests there read the fixtures only.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get('SD_REFERENCE', '/root/reference')


class _RefAutoencoderKL(nn.Module):
    """AutoencoderKL.__init__ / encode / decode (autoencoder.py:285-333) without the Lightning base and the loss."""

    def __init__(self, Encoder, Decoder, ddconfig, embed_dim):
        super().__init__()
        with contextlib.redirect_stdout(io.StringIO()):
            self.encoder = Encoder(**ddconfig)
            self.decoder = Decoder(**ddconfig)
        self.quant_conv = nn.Conv2d(2 * ddconfig['z_channels'], 2 * embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig['z_channels'], 1)

    def encode_moments(self, x):
        return self.quant_conv(self.encoder(x))

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


def main():
    sys.path.insert(0, REF)
    from ldm.modules.diffusionmodules.model import Decoder, Encoder
    from oracle import vae_ref
    from oracle.vae_ref import SD_VAE, SMALL_VAE, TINY_VAE
    torch.set_num_threads(os.cpu_count())
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    os.makedirs(out_dir, exist_ok=True)

    cfgs = {'tiny': TINY_VAE, 'small': SMALL_VAE, 'sd': SD_VAE}
    # (name, cfg, weight seed, batch, latent h, latent w)
    dec_cases = [
        ('tiny_8x8', 'tiny', 0, 2, 8, 8),
        ('tiny_8x24', 'tiny', 0, 1, 8, 24),       # non-square
        ('small_16x16', 'small', 1, 2, 16, 16),
        ('sd_8x8', 'sd', 0, 2, 8, 8),
        ('sd_16x24', 'sd', 0, 1, 16, 24),
        ('sd_32x32', 'sd', 0, 1, 32, 32),
        ('sd_64x64', 'sd', 0, 1, 64, 64),         # BASELINE.json configs[1]: 512x512 decode
    ]
    enc_cases = [
        ('tiny_32x32', 'tiny', 0, 2, 32, 32),     # image sizes
        ('tiny_16x48', 'tiny', 0, 1, 16, 48),
        ('sd_64x64', 'sd', 0, 2, 64, 64),
        ('sd_128x192', 'sd', 0, 1, 128, 192),
        ('sd_256x256', 'sd', 0, 1, 256, 256),
    ]
    models = {}

    def model(cname, seed):
        key = (cname, seed)
        if key not in models:
            models.clear()
            cfg = cfgs[cname]
            sd = vae_ref.make_vae_state_dict(cfg, seed)
            m = _RefAutoencoderKL(Encoder, Decoder, cfg.ddconfig(), cfg.embed_dim).eval()
            m.load_state_dict(sd, strict=True)
            print(f'[{cname}] reference Encoder/Decoder loaded strict=True: {len(sd)} tensors, '
                  f'{sum(p.numel() for p in m.parameters())} params', flush=True)
            models[key] = (m, sd, cfg)
        return models[key]

    for name, cname, seed, b, h, w in dec_cases:
        m, sd, cfg = model(cname, seed)
        z = vae_ref.make_vae_inputs(cfg, b, h, w, seed=1)
        with torch.no_grad():
            ref = m.decode(z)
        orc = vae_ref.vae_decode(sd, cfg, z)
        err = (ref - orc).abs().max().item()
        print(f'[decode {name}] out {tuple(ref.shape)} |x| max {ref.abs().max():.4f} rms {ref.pow(2).mean().sqrt():.4f} '
              f'oracle-vs-reference {err:.3e}', flush=True)
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), err
        np.savez_compressed(os.path.join(out_dir, f'vae_dec_{name}.npz'), out=ref.numpy().astype(np.float32),
                            cfg=cname, weight_seed=seed, input_seed=1, batch=b, h=h, w=w,
                            absmax=float(ref.abs().max()), oracle_vs_reference=err)
    for name, cname, seed, b, h, w in enc_cases:
        m, sd, cfg = model(cname, seed)
        g = torch.Generator().manual_seed(2)
        x = torch.rand(b, cfg.in_channels, h, w, generator=g) * 2 - 1
        with torch.no_grad():
            ref = m.encode_moments(x)
        orc = vae_ref.vae_encode_moments(sd, cfg, x)
        err = (ref - orc).abs().max().item()
        print(f'[encode {name}] moments {tuple(ref.shape)} max {ref.abs().max():.4f} oracle-vs-reference {err:.3e}',
              flush=True)
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), err
        np.savez_compressed(os.path.join(out_dir, f'vae_enc_{name}.npz'), moments=ref.numpy().astype(np.float32),
                            cfg=cname, weight_seed=seed, input_seed=2, batch=b, h=h, w=w,
                            absmax=float(ref.abs().max()), oracle_vs_reference=err)
    print('VAE golden fixtures written to', out_dir)


if __name__ == '__main__':
    main()
