"""Generate tests/golden/dpm_solver.npz by running the REFERENCE DPMSolverSampler (build container only).

    PYTHONPATH=/root/repo python oracle/make_golden_dpm.py

Runs `ldm.models.diffusion.dpm_solver.sampler.DPMSolverSampler` (what `scripts/txt2img.py --dpm_solver` constructs,
txt2img.py:250-251) against the deterministic stub model of oracle/make_golden.py, asserts the oracle restatement
(`oracle.samplers_ref.dpm_solver_sample`) equals it and stores the trajectories' end points.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get('SD_REFERENCE', '/root/reference')


def stub_unet(x, t, c):
    return torch.tanh(0.7 * x + 0.001 * t.float()[:, None, None, None]) * 0.9 + 0.05 * c.mean(dim=(1, 2))[:, None, None, None]


class _StubLD:
    def __init__(self, betas, ac):
        self.betas = torch.tensor(betas)
        self.alphas_cumprod = torch.tensor(ac)
        self.device = torch.device('cpu')
        self.calls = []

    def apply_model(self, x, t, c):
        self.calls.append(float(t[0]))
        return stub_unet(x, t, c)


def main():
    sys.path.insert(0, REF)
    from ldm.models.diffusion.dpm_solver.sampler import DPMSolverSampler
    from oracle import samplers_ref
    DPMSolverSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)   # sampler.py:15-19 forces cuda
    betas, ac = samplers_ref.make_alphas_cumprod()
    g = torch.Generator().manual_seed(7)
    x_T = torch.randn(2, 4, 8, 8, generator=g)
    c = torch.randn(2, 77, 16, generator=g)
    uc = torch.randn(2, 77, 16, generator=g)
    out = {}
    for S, scale, ucond in ((20, 7.5, uc), (10, 7.5, uc), (50, 5.0, uc), (12, 1.0, None)):
        model = _StubLD(betas, ac)
        ref, _ = DPMSolverSampler(model).sample(S=S, batch_size=2, shape=[4, 8, 8], conditioning=c, verbose=False, x_T=x_T,
                                                unconditional_guidance_scale=scale, unconditional_conditioning=ucond)
        rec = []
        mine = samplers_ref.dpm_solver_sample(stub_unet, ac, S, x_T, c, scale, ucond, record=rec)
        err = (ref - mine).abs().max().item()
        print(f'DPM-Solver++(2M) S={S} scale={scale}: {len(model.calls)} apply_model calls, t_in first {model.calls[0]:.3f} '
              f'last {model.calls[-1]:.3f}; |x| max {ref.abs().max():.3f}; oracle-vs-reference {err:.3e}')
        assert err < 2e-5 and len(model.calls) == S and len(rec) == S
        assert np.allclose(rec, model.calls, atol=1e-3)
        out[f'dpm_{S}_{scale}'] = ref.numpy()
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'dpm_solver.npz'), x_T=x_T.numpy(), c=c.numpy(), uc=uc.numpy(),
                        alphas_cumprod=ac, **out)
    print('written tests/golden/dpm_solver.npz')


if __name__ == '__main__':
    main()
