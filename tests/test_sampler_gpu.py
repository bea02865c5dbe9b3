"""PLMS / DDIM loops on the GPU (fused CFG + update kernel) against the reference trajectories in
tests/golden/samplers.npz (produced by the reference PLMSSampler / DDIMSampler with a deterministic stub model),
and an end-to-end sampling run through UNetModelHIP compared with the oracle loop driving the same UNet."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def stub_unet(x, t, c):
    return torch.tanh(0.7 * x + 0.001 * t.float()[:, None, None, None]) * 0.9 \
        + 0.05 * c.mean(dim=(1, 2))[:, None, None, None]


class StubLD:
    def __init__(self, betas, ac, fn=stub_unet):
        self.num_timesteps = len(betas)
        self.betas = torch.tensor(betas).cuda()
        self.alphas_cumprod = torch.tensor(ac).cuda()
        self.alphas_cumprod_prev = torch.tensor(np.append(1.0, ac[:-1]).astype(np.float32)).cuda()
        self.device = torch.device('cuda')
        self.fn = fn
        self.calls = []

    def apply_model(self, x, t, c):
        self.calls.append(int(t[0]))
        return self.fn(x, t, c)


@pytest.fixture(scope='module')
def G(golden_dir):
    return np.load(os.path.join(golden_dir, 'samplers.npz'))


@pytest.mark.parametrize('S', [50, 10])
def test_plms_trajectory(G, S):
    from stable_diffusion_amd import PLMSSamplerHIP
    model = StubLD(G['betas'], G['alphas_cumprod'])
    smp = PLMSSamplerHIP(model)
    x_T, c, uc = (torch.from_numpy(G[k]).cuda() for k in ('x_T', 'c', 'uc'))
    out, inter = smp.sample(S=S, batch_size=2, shape=[4, 8, 8], conditioning=c, verbose=False, x_T=x_T,
                            unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0)
    err = (out.cpu() - torch.from_numpy(G[f'plms_{S}'])).abs().max().item()
    print(f'[plms S={S}] calls {len(model.calls)} first {model.calls[:2]} last {model.calls[-1]} max-abs {err:.3e}')
    assert len(model.calls) == S + 1 and model.calls[0] == (981 if S == 50 else 901) and model.calls[-1] == 1
    # the only difference from the reference run is GPU-vs-CPU tanh in the stub model (few ulp per step)
    assert err < 2e-4


@pytest.mark.parametrize('S', [50, 10])
def test_ddim_trajectory(G, S):
    from stable_diffusion_amd import DDIMSamplerHIP
    model = StubLD(G['betas'], G['alphas_cumprod'])
    smp = DDIMSamplerHIP(model)
    x_T, c, uc = (torch.from_numpy(G[k]).cuda() for k in ('x_T', 'c', 'uc'))
    out, _ = smp.sample(S=S, batch_size=2, shape=[4, 8, 8], conditioning=c, verbose=False, x_T=x_T,
                        unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0)
    err = (out.cpu() - torch.from_numpy(G[f'ddim_{S}'])).abs().max().item()
    print(f'[ddim S={S}] max-abs {err:.3e}')
    assert len(model.calls) == S and err < 2e-4


def test_plms_no_cfg(G):
    from stable_diffusion_amd import PLMSSamplerHIP
    model = StubLD(G['betas'], G['alphas_cumprod'])
    x_T, c = (torch.from_numpy(G[k]).cuda() for k in ('x_T', 'c'))
    out, _ = PLMSSamplerHIP(model).sample(S=10, batch_size=2, shape=[4, 8, 8], conditioning=c, verbose=False, x_T=x_T)
    assert (out.cpu() - torch.from_numpy(G['plms_10_nocfg'])).abs().max().item() < 2e-4


def test_img2img_encode_decode(G):
    """scripts/img2img.py:237-262: make_schedule(50) -> stochastic_encode(t_enc=37) -> decode; first t = 721."""
    from stable_diffusion_amd import DDIMSamplerHIP
    model = StubLD(G['betas'], G['alphas_cumprod'])
    smp = DDIMSamplerHIP(model)
    smp.make_schedule(ddim_num_steps=50, ddim_eta=0.0, verbose=False)
    x0, noise, c, uc = (torch.from_numpy(G[k]).cuda() for k in ('x0', 'noise', 'c', 'uc'))
    t_enc = int(0.75 * 50)
    z = smp.stochastic_encode(x0, torch.tensor([t_enc] * 2).cuda(), noise=noise)
    assert (z.cpu() - torch.from_numpy(G['img2img_z'])).abs().max().item() < 1e-6
    out = smp.decode(z, c, t_enc, unconditional_guidance_scale=5.0, unconditional_conditioning=uc)
    err = (out.cpu() - torch.from_numpy(G['img2img_out'])).abs().max().item()
    print(f'[img2img] calls {len(model.calls)} first t {model.calls[0]} max-abs {err:.3e}')
    assert len(model.calls) == 37 and model.calls[0] == 721 and err < 2e-4


# ---- sampler variants beyond the script defaults, against reference runs with recorded noise (tests/golden/samplers2.npz,
# oracle/make_golden_samplers2.py): DDIM eta > 0 (ddim.py:195-203) and the mask / x0 blend (plms.py:147-150, ddim.py:130-133)
@pytest.fixture(scope='module')
def G2(golden_dir):
    return np.load(os.path.join(golden_dir, 'samplers2.npz'))


class StubLDq(StubLD):
    """+ DDPM.q_sample (ddpm.py:274-277), its noise handed out from the recorded sequence of the reference run"""

    def __init__(self, betas, ac, q_noises):
        super().__init__(betas, ac)
        self.sqrt_alphas_cumprod = torch.tensor(np.sqrt(ac.astype(np.float64)).astype(np.float32)).cuda()
        self.sqrt_one_minus_alphas_cumprod = torch.tensor(np.sqrt(1.0 - ac.astype(np.float64)).astype(np.float32)).cuda()
        self._q = [torch.from_numpy(n).cuda() for n in q_noises]

    def q_sample(self, x_start, t, noise=None):
        noise = self._q.pop(0) if noise is None else noise
        sh = (t.shape[0],) + (1,) * (x_start.dim() - 1)
        return self.sqrt_alphas_cumprod.gather(-1, t).reshape(sh) * x_start + \
            self.sqrt_one_minus_alphas_cumprod.gather(-1, t).reshape(sh) * noise


def test_ddim_eta_trajectory(G2):
    """DDIM with eta = 0.5: sigma_t up to 0.39, the noise term of ddim.py:200-203 is live on every step."""
    from stable_diffusion_amd import DDIMSamplerHIP
    S = int(G2['S'])
    model = StubLD(G2['betas'], G2['alphas_cumprod'])
    smp = DDIMSamplerHIP(model)
    seq = [torch.from_numpy(n).cuda() for n in G2['noises']]
    smp._noise_like = lambda shape, device: seq.pop(0)
    x_T, c, uc = (torch.from_numpy(G2[k]).cuda() for k in ('x_T', 'c', 'uc'))
    out, _ = smp.sample(S=S, batch_size=2, shape=[4, 8, 8], conditioning=c, verbose=False, x_T=x_T,
                        unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.5)
    err = (out.cpu() - torch.from_numpy(G2['ddim_eta05'])).abs().max().item()
    print(f'[ddim eta=0.5 S={S}] calls {len(model.calls)} sigma max {float(smp.ddim_sigmas.max()):.3f} max-abs {err:.3e}')
    assert not seq and len(model.calls) == S and float(smp.ddim_sigmas.max()) > 0.3
    assert err < 2e-4


@pytest.mark.parametrize('kind', ['plms', 'ddim'])
def test_mask_blend_trajectory(G2, kind):
    """inpainting blend in front of every step: img = q_sample(x0, ts) * mask + (1 - mask) * img"""
    from stable_diffusion_amd import DDIMSamplerHIP, PLMSSamplerHIP
    S = int(G2['S'])
    model = StubLDq(G2['betas'], G2['alphas_cumprod'], G2['q_noises'])
    smp = (PLMSSamplerHIP if kind == 'plms' else DDIMSamplerHIP)(model)
    x_T, c, uc, x0, mask = (torch.from_numpy(G2[k]).cuda() for k in ('x_T', 'c', 'uc', 'x0', 'mask'))
    out, _ = smp.sample(S=S, batch_size=2, shape=[4, 8, 8], conditioning=c, verbose=False, x_T=x_T, mask=mask, x0=x0,
                        unconditional_guidance_scale=7.5, unconditional_conditioning=uc, eta=0.0)
    err = (out.cpu() - torch.from_numpy(G2[f'{kind}_mask'])).abs().max().item()
    print(f'[{kind} mask blend S={S}] calls {len(model.calls)} max-abs {err:.3e}')
    assert not model._q and len(model.calls) == S + (1 if kind == 'plms' else 0)
    assert err < 2e-4


def test_plms_end_to_end_with_hip_unet():
    """10-step PLMS, CFG 7.5, through LatentDiffusionHIP + UNetModelHIP vs the oracle loop driving the oracle UNet."""
    from oracle import samplers_ref, unet_ref
    from oracle.plan import TINY
    from oracle.weights import make_inputs, make_state_dict
    from stable_diffusion_amd import LatentDiffusionHIP, PLMSSamplerHIP, UNetModelHIP
    sd = make_state_dict(TINY, 0)
    kw = TINY.ref_kwargs()
    unet = UNetModelHIP(**kw)
    unet.load_state_dict(sd, strict=True)
    ld = LatentDiffusionHIP(unet).cuda()
    x_T, _, ctx = make_inputs(TINY, 1, 16, 16, seed=9)
    uc = torch.zeros_like(ctx) + 0.1
    out, _ = PLMSSamplerHIP(ld).sample(S=10, batch_size=1, shape=[4, 16, 16], conditioning=ctx.cuda(), verbose=False,
                                       x_T=x_T.cuda(), unconditional_guidance_scale=7.5,
                                       unconditional_conditioning=uc.cuda(), eta=0.0)
    _, ac = samplers_ref.make_alphas_cumprod()
    ref = samplers_ref.plms_sample(lambda x, t, c: unet_ref.unet_forward(sd, TINY, x, t, c), ac, 10, x_T, ctx, 7.5, uc)
    err = (out.cpu() - ref).abs().max().item()
    print(f'[plms e2e tiny] |x0|max {ref.abs().max():.3f} max-abs {err:.3e}')
    # eps errors (<=1e-3 each) are amplified by CFG 7.5 and accumulated over 11 UNet calls
    assert err < 5e-2


def test_img2img_end_to_end_with_hip_unet():
    """BASELINE.json configs[4] flow (scripts/img2img.py:237-262) on the HIP path: stochastic_encode at t_enc, then
    DDIM decode through UNetModelHIP, against the oracle loop driving the oracle UNet (DDIM; the reference's img2img
    rejects --plms, img2img.py:205-207)."""
    from oracle import samplers_ref, unet_ref
    from oracle.plan import TINY
    from oracle.weights import make_inputs, make_state_dict
    from stable_diffusion_amd import DDIMSamplerHIP, LatentDiffusionHIP, UNetModelHIP
    sd = make_state_dict(TINY, 0)
    unet = UNetModelHIP(**TINY.ref_kwargs())
    unet.load_state_dict(sd, strict=True)
    ld = LatentDiffusionHIP(unet).cuda()
    x0, _, ctx = make_inputs(TINY, 1, 16, 16, seed=11)
    g = torch.Generator().manual_seed(12)
    noise = torch.randn(x0.shape, generator=g)
    uc = torch.zeros_like(ctx) - 0.1
    S, t_enc = 20, 6                                   # strength 0.3 keeps the CPU oracle loop short
    smp = DDIMSamplerHIP(ld)
    smp.make_schedule(ddim_num_steps=S, ddim_eta=0.0, verbose=False)
    z = smp.stochastic_encode(x0.cuda(), torch.tensor([t_enc]).cuda(), noise=noise.cuda())
    out = smp.decode(z, ctx.cuda(), t_enc, unconditional_guidance_scale=5.0, unconditional_conditioning=uc.cuda())
    _, ac = samplers_ref.make_alphas_cumprod()
    z_ref = samplers_ref.ddim_stochastic_encode(ac, S, x0, t_enc, noise)
    ref = samplers_ref.ddim_decode(lambda x, t, c: unet_ref.unet_forward(sd, TINY, x, t, c), ac, S, z_ref, ctx, t_enc, 5.0, uc)
    assert (z.cpu() - z_ref).abs().max().item() < 1e-6
    err = (out.cpu() - ref).abs().max().item()
    print(f'[img2img e2e tiny] max-abs {err:.3e}')
    assert err < 3e-2


# ---- DPM-Solver++ (2M): scripts/txt2img.py --dpm_solver -------------------------------------------
@pytest.mark.parametrize('S,scale,cfg', [(20, 7.5, True), (10, 7.5, True), (50, 5.0, True), (12, 1.0, False)])
def test_dpm_solver_trajectory(golden_dir, S, scale, cfg):
    """DPMSolverSamplerHIP against the reference DPMSolverSampler's end point (tests/golden/dpm_solver.npz,
    oracle/make_golden_dpm.py); S = 10 / 12 exercise lower_order_final, S = 12 the guidance-free branch."""
    from stable_diffusion_amd import DPMSolverSamplerHIP
    D = np.load(os.path.join(golden_dir, 'dpm_solver.npz'))
    S0 = np.load(os.path.join(golden_dir, 'samplers.npz'))
    model = StubLD(S0['betas'], D['alphas_cumprod'])
    model.apply_model = lambda x, t, c: (model.calls.append(float(t[0])), stub_unet(x, t, c))[1]
    x_T, c, uc = (torch.from_numpy(D[k]).cuda() for k in ('x_T', 'c', 'uc'))
    out, none = DPMSolverSamplerHIP(model).sample(S=S, batch_size=2, shape=[4, 8, 8], conditioning=c, verbose=False, x_T=x_T,
                                                  unconditional_guidance_scale=scale,
                                                  unconditional_conditioning=uc if cfg else None)
    ref = torch.from_numpy(D[f'dpm_{S}_{scale}'])
    err = (out.cpu() - ref).abs().max().item()
    print(f'[dpm-solver S={S} scale={scale}] calls {len(model.calls)} t_in {model.calls[0]:.2f} .. {model.calls[-1]:.2f} '
          f'max-abs {err:.3e} (|x| max {ref.abs().max():.2f})')
    assert none is None and len(model.calls) == S and abs(model.calls[0] - 999.0) < 1e-3
    # GPU-vs-CPU tanh in the stub model (measured 1.5e-5 .. 2.7e-5 at |x| ~ 39)
    assert err < 2e-4


@pytest.mark.parametrize('order', [1, 2])
@pytest.mark.parametrize('cfg', [0, 1])
def test_dpm_step_bit_exact(order, cfg):
    """sdmi_dpm_solver_step == the reference's torch expressions evaluated on the GPU (dpm_solver.py:340-346, :386-399,
    :519-530, :776-790), bit for bit."""
    import ctypes  # noqa: F401
    from stable_diffusion_amd import _lib
    g = torch.Generator().manual_seed(5)
    n = 2 * 4 * 16 * 16
    eps = torch.randn(2 * n if cfg else n, generator=g).cuda()
    x = (torch.randn(n, generator=g) * 3).cuda()
    m1 = torch.randn(n, generator=g).cuda()
    alpha_s, sigma_s, cx, a, inv_r0, scale = 0.7341, 0.6790, 0.93127, -0.08123, 1.2345, 7.5
    T = lambda v: torch.tensor(v, device='cuda')
    e = eps[:n] + scale * (eps[n:] - eps[:n]) if cfg else eps
    m0_ref = (x - T(sigma_s) * e) / T(alpha_s)
    if order == 1:
        xt_ref = T(cx) * x - T(a) * m0_ref
    else:
        D1 = T(inv_r0) * (m0_ref - m1)
        xt_ref = T(cx) * x - T(a) * m0_ref - 0.5 * T(a) * D1
    m0, xt = torch.empty_like(x), torch.empty_like(x)
    _lib.check(_lib.load().sdmi_dpm_solver_step(eps.data_ptr(), cfg, scale, x.data_ptr(), m1.data_ptr() if order == 2 else None,
                                                alpha_s, sigma_s, cx, a, inv_r0, order, m0.data_ptr(), xt.data_ptr(), n,
                                                _lib.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(m0, m0_ref) and torch.equal(xt, xt_ref)


def test_image_postprocess_is_bit_identical_to_the_script(tmp_path):
    """scripts/txt2img.py:314-326: clamp((x + 1) / 2) -> NHWC -> 255 * x -> astype(uint8) -> PNG, on the device in one pass."""
    import numpy as np
    from PIL import Image
    from stable_diffusion_amd import postprocess
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 3, 24, 40, generator=g) * 0.9
    x[0, 0, 0, :8] = torch.tensor([-1.0, 1.0, -1.0000001, 0.9999999, 0.0, 1.5, -3.0, 0.99607843])     # range ends, 254/255
    ref = torch.clamp((x + 1.0) / 2.0, min=0.0, max=1.0).permute(0, 2, 3, 1).numpy()
    ref = (255. * ref).astype(np.uint8)
    out = postprocess.to_uint8_images(x.cuda())
    assert out.dtype == torch.uint8 and tuple(out.shape) == (3, 24, 40, 3)
    assert np.array_equal(out.cpu().numpy(), ref)
    path = postprocess.save_png(out[1], str(tmp_path / 'a.png'))
    assert np.array_equal(np.array(Image.open(path)), ref[1])
    with pytest.raises(RuntimeError, match='no CPU'):
        postprocess.to_uint8_images(x)
