"""Reference-run goldens for the sampler variants beyond the script defaults (build container only):

    PYTHONPATH=/root/repo python oracle/make_golden_samplers2.py

  * DDIM with eta = 0.5 (ddim.py:195-203: sigma_t > 0, the `sigma_t * noise_like(...) * temperature` term is live);
  * the mask / x0 blend of PLMS and DDIM (plms.py:147-150, ddim.py:130-133: `img = q_sample(x0, ts) * mask + (1 - mask) * img`).

The reference draws both noises from the device RNG; to make the runs comparable across devices the draws are
fixed: `ldm.models.diffusion.ddim.noise_like` is replaced by a function that hands out a recorded sequence, and the
stub model's `q_sample` is the reference's DDPM.q_sample (ddpm.py:274-277) with its noise taken from a recorded sequence.
The sequences and the reference outputs go to tests/golden/samplers2.npz; the oracle restatement is asserted equal.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.make_golden import _StubLatentDiffusion, _import_reference  # noqa: E402
from oracle import samplers_ref  # noqa: E402


def stub_unet(x, t, c):
    return torch.tanh(0.7 * x + 0.001 * t.float()[:, None, None, None]) * 0.9 \
        + 0.05 * c.mean(dim=(1, 2))[:, None, None, None]


class _StubWithQSample(_StubLatentDiffusion):
    """+ DDPM.q_sample (ddpm.py:274-277) over the fp32 buffers of register_schedule (ddpm.py:134-136)."""

    def __init__(self, unet_fn, betas, ac, q_noises):
        super().__init__(unet_fn, betas, ac)
        self.sqrt_alphas_cumprod = torch.tensor(np.sqrt(ac.astype(np.float64)).astype(np.float32))
        self.sqrt_one_minus_alphas_cumprod = torch.tensor(np.sqrt(1.0 - ac.astype(np.float64)).astype(np.float32))
        self._q = list(q_noises)

    def q_sample(self, x_start, t, noise=None):
        from ldm.modules.diffusionmodules.util import extract_into_tensor
        noise = self._q.pop(0) if noise is None else noise
        return (extract_into_tensor(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start +
                extract_into_tensor(self.sqrt_one_minus_alphas_cumprod, t, x_start.shape) * noise)


def main():
    UNetModel, PLMSSampler, DDIMSampler, ref_util = _import_reference()
    import ldm.models.diffusion.ddim as ref_ddim
    betas, ac = samplers_ref.make_alphas_cumprod()
    S = 10
    g = torch.Generator().manual_seed(11)
    x_T = torch.randn(2, 4, 8, 8, generator=g)
    c = torch.randn(2, 77, 16, generator=g)
    uc = torch.randn(2, 77, 16, generator=g)
    noises = [torch.randn(2, 4, 8, 8, generator=g) for _ in range(S)]
    q_noises = [torch.randn(2, 4, 8, 8, generator=g) for _ in range(S)]
    x0 = torch.randn(2, 4, 8, 8, generator=g)
    mask = (torch.rand(2, 1, 8, 8, generator=g) > 0.5).float()

    def patch(s):
        s.register_buffer = lambda name, attr: setattr(s, name, attr)
        return s
    out = {}

    # ---- DDIM, eta = 0.5, the recorded noise sequence ------------------------------------------------------------------
    seq = list(noises)
    orig = ref_ddim.noise_like
    ref_ddim.noise_like = lambda shape, device, repeat=False: seq.pop(0)
    try:
        model = _StubLatentDiffusion(stub_unet, betas, ac)
        smp = patch(DDIMSampler(model))
        ref, _ = smp.sample(S=S, batch_size=2, shape=[4, 8, 8], conditioning=c, verbose=False, x_T=x_T,
                            unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.5)
    finally:
        ref_ddim.noise_like = orig
    assert not seq and float(smp.ddim_sigmas.max()) > 0.05
    mine = samplers_ref.ddim_sample(stub_unet, ac, S, x_T, c, 7.5, uc, eta=0.5, noises=noises)
    err = (ref - mine).abs().max().item()
    print(f'DDIM eta=0.5 S={S}: sigmas {smp.ddim_sigmas.numpy().round(4).tolist()} oracle-vs-reference {err:.3e}')
    assert err < 1e-5
    out['ddim_eta05'] = ref.numpy()
    # (same x_T with eta = 0 must differ: the noise term is really exercised)
    ref0 = samplers_ref.ddim_sample(stub_unet, ac, S, x_T, c, 7.5, uc)
    assert (ref - ref0).abs().max().item() > 0.05

    # ---- mask / x0 blend: PLMS and DDIM (eta = 0) ---------------------------------------------------------------------
    model = _StubWithQSample(stub_unet, betas, ac, q_noises)
    smp = patch(PLMSSampler(model))
    ref, _ = smp.sample(S=S, batch_size=2, shape=[4, 8, 8], conditioning=c, verbose=False, x_T=x_T, mask=mask, x0=x0,
                        unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0)
    assert not model._q
    mine = samplers_ref.plms_sample(stub_unet, ac, S, x_T, c, 7.5, uc, mask=mask, x0=x0, q_noises=q_noises)
    err = (ref - mine).abs().max().item()
    print(f'PLMS mask blend S={S}: {len(model.calls)} calls, oracle-vs-reference {err:.3e}')
    assert err < 1e-5 and len(model.calls) == S + 1
    out['plms_mask'] = ref.numpy()
    nomask = samplers_ref.plms_sample(stub_unet, ac, S, x_T, c, 7.5, uc)
    assert (ref - nomask).abs().max().item() > 0.05

    model = _StubWithQSample(stub_unet, betas, ac, q_noises)
    smp = patch(DDIMSampler(model))
    ref, _ = smp.sample(S=S, batch_size=2, shape=[4, 8, 8], conditioning=c, verbose=False, x_T=x_T, mask=mask, x0=x0,
                        unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0)
    assert not model._q
    mine = samplers_ref.ddim_sample(stub_unet, ac, S, x_T, c, 7.5, uc, mask=mask, x0=x0, q_noises=q_noises)
    err = (ref - mine).abs().max().item()
    print(f'DDIM mask blend S={S}: {len(model.calls)} calls, oracle-vs-reference {err:.3e}')
    assert err < 1e-5 and len(model.calls) == S
    out['ddim_mask'] = ref.numpy()

    path = os.path.join(ROOT, 'tests', 'golden', 'samplers2.npz')
    np.savez_compressed(path, S=S, x_T=x_T.numpy(), c=c.numpy(), uc=uc.numpy(), noises=torch.stack(noises).numpy(),
                        q_noises=torch.stack(q_noises).numpy(), x0=x0.numpy(), mask=mask.numpy(), alphas_cumprod=ac, betas=betas,
                        **out)
    print('written', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
