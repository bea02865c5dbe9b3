"""Generate tests/golden/*.npz by running the REFERENCE modules (build container only).

    PYTHONPATH=/root/repo python oracle/make_golden.py

Imports `/root/reference/ldm/...` (needs one stub: omegaconf.listconfig.ListConfig,
openaimodel.py:476-478), loads `oracle.weights.make_state_dict` into the real
`UNetModel` with strict=True, asserts the oracle restatement equals the
reference, and stores the reference outputs as small fixtures.  The GPU box has no
/root/reference: tests there read the fixtures only.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get('SD_REFERENCE', '/root/reference')


def _import_reference(root=None):
    """`root`: a reference checkout, or the sourceless bytecode bundle of oracle/build_ref_bundle.py (oracle/_ref/refbundle)."""
    sys.path.insert(0, root or REF)
    om = types.ModuleType('omegaconf')
    lc = types.ModuleType('omegaconf.listconfig')

    class ListConfig(list):
        pass
    lc.ListConfig = ListConfig
    om.listconfig = lc
    sys.modules.setdefault('omegaconf', om)
    sys.modules.setdefault('omegaconf.listconfig', lc)
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from ldm.models.diffusion.plms import PLMSSampler
    from ldm.models.diffusion.ddim import DDIMSampler
    from ldm.modules.diffusionmodules import util as ref_util
    return UNetModel, PLMSSampler, DDIMSampler, ref_util


class _StubLatentDiffusion:
    """The attributes the reference samplers touch on `model` (SURVEY.md 8b 'Sampler seam'),
    with apply_model restated from ddpm.py:893-900,1408-1410 (wrap cond, cat, call UNet)."""

    def __init__(self, unet_fn, betas, alphas_cumprod):
        self.num_timesteps = len(betas)
        self.betas = torch.tensor(betas)
        self.alphas_cumprod = torch.tensor(alphas_cumprod)
        self.alphas_cumprod_prev = torch.tensor(np.append(1.0, alphas_cumprod[:-1]).astype(np.float32))
        self.device = torch.device('cpu')
        self.unet_fn = unet_fn
        self.calls = []

    def apply_model(self, x, t, c):
        self.calls.append(int(t[0]))
        cc = torch.cat([c], 1)
        return self.unet_fn(x, t, cc)


def main():
    from oracle.plan import TINY, SMALL40, SD_V1
    from oracle.weights import make_state_dict, make_inputs, param_specs
    from oracle import unet_ref, samplers_ref
    UNetModel, PLMSSampler, DDIMSampler, ref_util = _import_reference()
    torch.set_num_threads(os.cpu_count())
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    os.makedirs(out_dir, exist_ok=True)

    # ---- UNet eps goldens --------------------------------------------------------------
    cases = [
        ('tiny_16x16', TINY, 0, 2, 16, 16, 77),
        ('tiny_8x24', TINY, 0, 2, 8, 24, 77),       # non-square latent (SURVEY 8c hygiene 5)
        ('tiny_b1_8x8', TINY, 3, 1, 8, 8, 77),
        ('small40_16x16', SMALL40, 0, 2, 16, 16, 77),
        ('sdv1_8x8', SD_V1, 0, 2, 8, 8, 77),
        ('sdv1_16x16', SD_V1, 0, 2, 16, 16, 77),
        ('sdv1_32x32', SD_V1, 0, 2, 32, 32, 77),
        ('sdv1_64x64', SD_V1, 0, 2, 64, 64, 77),
        ('sdv1_96x96', SD_V1, 0, 2, 96, 96, 77),     # BASELINE.json configs[3]: 768x768, 9216 tokens
        # SURVEY 8c hygiene (4): t in {1, 481, 981} must reach a whole-UNet reference comparison -- the batch-2 cases above
        # only see (981, 481); and txt2img's default n_samples = 3 is a CFG batch of 6 (scripts/txt2img.py:110-114)
        ('sdv1_t1_741_16x16', SD_V1, 0, 2, 16, 16, 77, (1, 741)),
        ('sdv1_b6_16x16', SD_V1, 0, 6, 16, 16, 77),          # t = (981, 481, 1, 741, 981, 481)
        ('tiny_b6_16x16', TINY, 0, 6, 16, 16, 77),
        ('tiny_b10_8x8', TINY, 0, 10, 8, 8, 77),             # > 8 rows: UNetModelHIP.forward chunks (n_samples = 5)
        # round 6: the parity claim must not rest on ONE weight draw from ONE benign distribution.
        # (a) two more draws of the uniform family at the three latent sizes that exercise every level's shapes
        ('sdv1_w1_16x16', SD_V1, 1, 2, 16, 16, 77), ('sdv1_w1_32x32', SD_V1, 1, 2, 32, 32, 77), ('sdv1_w1_64x64', SD_V1, 1, 2, 64, 64, 77),
        ('sdv1_w2_16x16', SD_V1, 2, 2, 16, 16, 77), ('sdv1_w2_32x32', SD_V1, 2, 2, 32, 32, 77), ('sdv1_w2_64x64', SD_V1, 2, 2, 64, 64, 77),
        # (b) the 'realistic' family (oracle/weights.py): Student-t weights, x8 outlier rows / gammas, context outlier channels at
        #     |x| ~ 30, x_t at the t = 981 (white noise) and t = 1 (nearly clean latent) scales in one batch
        ('sdv1_real_16x16', SD_V1, 0, 2, 16, 16, 77, (981, 1), 'realistic'),
        ('sdv1_real_32x32', SD_V1, 0, 2, 32, 32, 77, (981, 1), 'realistic'),
        ('sdv1_real_64x64', SD_V1, 0, 2, 64, 64, 77, (981, 1), 'realistic'),
        ('sdv1_real1_16x16', SD_V1, 1, 2, 16, 16, 77, (481, 1), 'realistic'),
        ('tiny_real_16x16', TINY, 0, 2, 16, 16, 77, (981, 1), 'realistic'),      # CPU-suite sized twin (oracle == reference)
    ]
    only = [a for a in sys.argv[1:] if not a.startswith('-')]
    if only:
        cases = [c for c in cases if c[0] in only]
    ref_models = {}
    for name, cfg, wseed, b, h, w, L, *rest in cases:
        tsteps = rest[0] if rest else (981, 481, 1, 741)
        style = rest[1] if len(rest) > 1 else 'uniform'
        key = (cfg, wseed, style)
        if key not in ref_models:
            ref_models.clear()  # keep memory bounded
            sd = make_state_dict(cfg, wseed, style=style)
            m = UNetModel(**cfg.ref_kwargs()).eval()
            missing = m.load_state_dict(sd, strict=True)
            n_params = sum(p.numel() for p in m.parameters())
            print(f'[{name}] reference UNetModel loaded strict=True: {len(sd)} tensors, {n_params} params', flush=True)
            ref_models[key] = (m, sd)
        m, sd = ref_models[key]
        x, t, ctx = make_inputs(cfg, b, h, w, seed=1, ctx_len=L, timesteps=tsteps, style=style)
        with torch.no_grad():
            eps_ref = m(x, t, context=ctx)
        taps = {}
        eps_orc = unet_ref.unet_forward(sd, cfg, x, t, ctx, taps=taps)
        err = (eps_ref - eps_orc).abs().max().item()
        print(f'[{name}] |eps| max {eps_ref.abs().max():.4f} rms {eps_ref.pow(2).mean().sqrt():.4f} '
              f'oracle-vs-reference max-abs {err:.3e}', flush=True)
        assert err < 2e-5, f'oracle restatement differs from reference: {err}'
        np.savez_compressed(os.path.join(out_dir, f'unet_{name}.npz'),
                            eps=eps_ref.numpy().astype(np.float32),
                            weight_seed=wseed, style=style, input_seed=1, batch=b, h=h, w=w, ctx_len=L,
                            t=t.numpy(), eps_absmax=float(eps_ref.abs().max()),
                            oracle_vs_reference=err)
    ref_models.clear()
    if only:
        print('golden fixtures written to', out_dir, '(subset:', only, ')')
        return

    # ---- schedule constants (SURVEY a19 goldens) ------------------------------------------
    betas_ref = ref_util.make_beta_schedule('linear', 1000, linear_start=0.00085, linear_end=0.0120)
    ac_ref = np.cumprod(1.0 - betas_ref, axis=0).astype(np.float32)
    betas, ac = samplers_ref.make_alphas_cumprod()
    assert np.array_equal(ac, ac_ref) and np.array_equal(betas, betas_ref.astype(np.float32))
    for S in (50, 10, 30):
        assert np.array_equal(samplers_ref.make_ddim_timesteps(S),
                              ref_util.make_ddim_timesteps('uniform', S, 1000, verbose=False))
    print('schedule: alphas_cumprod[0,1,981] =', ac[0], ac[1], ac[981])

    # ---- sampler trajectory goldens with a deterministic stub model ----------------------------
    # eps(x,t,c) = tanh(0.7 x + 0.001 t) * 0.9 + 0.05 * mean(c) -- cheap, nonlinear, t/c dependent.
    def stub_unet(x, t, c):
        return torch.tanh(0.7 * x + 0.001 * t.float()[:, None, None, None]) * 0.9 \
            + 0.05 * c.mean(dim=(1, 2))[:, None, None, None]

    def patch(s):
        s.register_buffer = lambda name, attr: setattr(s, name, attr)
        return s
    g = torch.Generator().manual_seed(7)
    x_T = torch.randn(2, 4, 8, 8, generator=g)
    c = torch.randn(2, 77, 16, generator=g)
    uc = torch.randn(2, 77, 16, generator=g)
    traj = {}
    for S in (50, 10):
        model = _StubLatentDiffusion(stub_unet, betas, ac)
        smp = patch(PLMSSampler(model))
        out, _ = smp.sample(S=S, batch_size=2, shape=[4, 8, 8], conditioning=c, verbose=False, x_T=x_T,
                            unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0)
        mine = samplers_ref.plms_sample(stub_unet, ac, S, x_T, c, 7.5, uc)
        err = (out - mine).abs().max().item()
        print(f'PLMS S={S}: {len(model.calls)} apply_model calls, first t {model.calls[:2]}, last {model.calls[-1]}, '
              f'oracle-vs-reference {err:.3e}')
        assert err < 1e-5 and len(model.calls) == S + 1
        traj[f'plms_{S}'] = out.numpy()
        model = _StubLatentDiffusion(stub_unet, betas, ac)
        smp = patch(DDIMSampler(model))
        out, _ = smp.sample(S=S, batch_size=2, shape=[4, 8, 8], conditioning=c, verbose=False, x_T=x_T,
                            unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0)
        mine = samplers_ref.ddim_sample(stub_unet, ac, S, x_T, c, 7.5, uc)
        err = (out - mine).abs().max().item()
        print(f'DDIM S={S}: {len(model.calls)} calls, oracle-vs-reference {err:.3e}')
        assert err < 1e-5 and len(model.calls) == S
        traj[f'ddim_{S}'] = out.numpy()
    # scale == 1 (no CFG) path
    model = _StubLatentDiffusion(stub_unet, betas, ac)
    smp = patch(PLMSSampler(model))
    out, _ = smp.sample(S=10, batch_size=2, shape=[4, 8, 8], conditioning=c, verbose=False, x_T=x_T)
    mine = samplers_ref.plms_sample(stub_unet, ac, 10, x_T, c)
    assert (out - mine).abs().max().item() < 1e-5
    traj['plms_10_nocfg'] = out.numpy()
    # img2img: stochastic_encode + decode (img2img.py:237-262), strength 0.75 -> t_enc 37
    model = _StubLatentDiffusion(stub_unet, betas, ac)
    smp = patch(DDIMSampler(model))
    smp.make_schedule(ddim_num_steps=50, ddim_eta=0.0, verbose=False)
    noise = torch.randn(2, 4, 8, 8, generator=g)
    x0 = torch.randn(2, 4, 8, 8, generator=g)
    t_enc = int(0.75 * 50)
    z = smp.stochastic_encode(x0, torch.tensor([t_enc] * 2), noise=noise)
    out = smp.decode(z, c, t_enc, unconditional_guidance_scale=5.0, unconditional_conditioning=uc)
    z2 = samplers_ref.ddim_stochastic_encode(ac, 50, x0, t_enc, noise)
    mine = samplers_ref.ddim_decode(stub_unet, ac, 50, z2, c, t_enc, 5.0, uc)
    err = (out - mine).abs().max().item()
    print(f'img2img t_enc={t_enc}: {len(model.calls)} calls, first t {model.calls[0]}, last {model.calls[-1]}, err {err:.3e}')
    assert err < 1e-5 and len(model.calls) == 37 and model.calls[0] == 721
    traj['img2img_z'] = z.numpy()
    traj['img2img_out'] = out.numpy()
    np.savez_compressed(os.path.join(out_dir, 'samplers.npz'), x_T=x_T.numpy(), c=c.numpy(), uc=uc.numpy(),
                        noise=noise.numpy(), x0=x0.numpy(), alphas_cumprod=ac, betas=betas, **traj)
    print('golden fixtures written to', out_dir)


if __name__ == '__main__':
    main()
