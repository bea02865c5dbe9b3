"""`PLMSSamplerHIP` / `DDIMSamplerHIP` / `DPMSolverSamplerHIP` -- drop-ins for the reference samplers
(ldm/models/diffusion/plms.py PLMSSampler, ddim.py DDIMSampler, dpm_solver/sampler.py DPMSolverSampler): same
constructor, `make_schedule`, `sample(...) -> (samples, intermediates)`, and for DDIM `stochastic_encode` / `decode`
(scripts/img2img.py:237-262).

Per step the reference issues ~15 elementwise launches plus four `torch.full` from host scalars
(plms.py:178-236); here the classifier-free-guidance combine and the PLMS/DDIM latent update are one fused
gfx950 kernel (`sdmi_sampler_step`, csrc/sampler.hip) evaluated in the reference's fp32 operation order,
the duplicated `x_in`/`t_in`/`c_in` are built once, and the context tensor is kept identical across steps so
the UNet reuses its cross-attention K/V.
"""
import numpy as np
import torch

from . import _lib

try:                                    # progress bar only; the reference prints the same bars
    from tqdm import tqdm
except Exception:                       # pragma: no cover
    def tqdm(it, **kw):
        return it


# ---- schedule tables (host side; ldm/modules/diffusionmodules/util.py:46-74) --------------------------------------
def make_ddim_timesteps(num_ddim, num_ddpm, discretize='uniform'):
    if discretize == 'uniform':
        c = num_ddpm // num_ddim
        ts = np.asarray(list(range(0, num_ddpm, c)))
    elif discretize == 'quad':
        ts = ((np.linspace(0, np.sqrt(num_ddpm * .8), num_ddim)) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{discretize}"')
    return ts + 1


def make_tables(alphas_cumprod, ddim_timesteps, eta):
    """alphas_cumprod: 1-D fp32 numpy.  Returns fp32 numpy tables (alphas_prev[0] = alphas_cumprod[0], util.py:66)."""
    ac = np.asarray(alphas_cumprod, dtype=np.float32)
    alphas = ac[ddim_timesteps]
    alphas_prev = np.asarray([ac[0]] + ac[ddim_timesteps[:-1]].tolist(), dtype=np.float32)
    sigmas = (eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))).astype(np.float32)
    return dict(alphas=alphas, alphas_prev=alphas_prev, sigmas=sigmas,
                sqrt_one_minus_alphas=np.sqrt(1.0 - alphas).astype(np.float32))


def plms_plan(timesteps):
    """[(i, index, t, t_next, mode)] for the PLMS loop (plms.py:142-162,218-232); mode = multistep order code of
    sdmi_sampler_step (4 = first step: Euler predictor + second model evaluation)."""
    time_range = np.flip(timesteps)
    total = timesteps.shape[0]
    plan = []
    for i, step in enumerate(time_range):
        t_next = time_range[min(i + 1, len(time_range) - 1)]
        plan.append((i, total - i - 1, int(step), int(t_next), 4 if i == 0 else min(i, 3)))
    return plan


class _SamplerBase(object):
    def __init__(self, model, schedule='linear', **kwargs):
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule
        self._lib = None

    def register_buffer(self, name, attr):   # kept for API compatibility (plms.py:18-22)
        setattr(self, name, attr)

    @staticmethod
    def _noise_like(shape, device):
        """util.py noise_like (ddim.py:200): one draw from the device RNG per step.  A separate method so that a parity test
        can hand out the recorded sequence of a reference run (tests/test_sampler_gpu.py)."""
        return torch.randn(shape, device=device)

    def _tables(self, ddim_num_steps, ddim_discretize, ddim_eta):
        self.ddim_timesteps = make_ddim_timesteps(ddim_num_steps, self.ddpm_num_timesteps, ddim_discretize)
        ac = self.model.alphas_cumprod
        assert ac.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        ac_np = ac.detach().to(torch.float32).cpu().numpy()
        t = make_tables(ac_np, self.ddim_timesteps, ddim_eta)
        self._tab = t
        to_t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        self.alphas_cumprod = ac.detach().to(torch.float32)
        self.ddim_sigmas, self.ddim_alphas = to_t(t['sigmas']), to_t(t['alphas'])
        self.ddim_alphas_prev = to_t(t['alphas_prev'])
        self.ddim_sqrt_one_minus_alphas = to_t(t['sqrt_one_minus_alphas'])

    # ---- one fused CFG + update launch ------------------------------------------------------------------
    def _step(self, eps_model, cfg, scale, x, mode, old, index, e_t_out, x_prev, pred_x0=None, noise=None):
        if not x.is_cuda:
            raise RuntimeError('the HIP sampler step runs on MI355X device tensors only (no CPU fallback)')
        if self._lib is None:
            self._lib = _lib.load()
        t = self._tab
        o = list(old) + [None, None, None]
        _lib.check(self._lib.sdmi_sampler_step(
            eps_model.data_ptr(), int(cfg), float(scale), x.data_ptr(), int(mode),
            _lib.ptr(o[0]), _lib.ptr(o[1]), _lib.ptr(o[2]),
            float(t['alphas'][index]), float(t['alphas_prev'][index]), float(t['sigmas'][index]),
            float(t['sqrt_one_minus_alphas'][index]),
            _lib.ptr(noise), _lib.ptr(e_t_out), x_prev.data_ptr(), _lib.ptr(pred_x0), x.numel(), _lib.stream_ptr()))

    def _hip_unet(self):
        """The UNetModelHIP behind model.apply_model, if any (LatentDiffusion.model.diffusion_model, ddpm.py:1398)."""
        from .unet import UNetModelHIP
        u = getattr(getattr(self.model, 'model', None), 'diffusion_model', None)
        return u if isinstance(u, UNetModelHIP) else None

    def _prepare(self, cond, shape, x_T, uc, scale):
        device = self.model.betas.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T
        img = img.detach().float().contiguous()
        cfg = not (uc is None or scale == 1.)
        if cfg:
            c_in = torch.cat([uc, cond])                         # [uncond, cond] order, plms.py:184
            x_in = torch.empty((2 * b,) + tuple(shape[1:]), device=device, dtype=torch.float32)
        else:
            c_in, x_in = cond, torch.empty(tuple(shape), device=device, dtype=torch.float32)
        return device, b, img, cfg, c_in, x_in

    def _model_eps(self, x_in, img, t_val, c_in, cfg, b, t_dtype=torch.long):
        """apply_model on the (duplicated) latent; returns the raw model output [2b or b, C, H, W] fp32."""
        x_in[:b].copy_(img)
        if cfg:
            x_in[b:].copy_(img)
        t_in = torch.full((x_in.shape[0],), t_val, device=x_in.device, dtype=t_dtype)
        unet = self._hip_unet() if t_dtype == torch.long else None
        if unet is not None:
            unet.hint_timestep(int(t_val))          # every row of t_in is this int: rows of the timestep table, if cached
        try:
            out = self.model.apply_model(x_in, t_in, c_in)
        finally:
            if unet is not None:
                unet.clear_timestep_hint()          # (consumed by forward(); still set only if apply_model raised before it)
        return out.float().contiguous()


class PLMSSamplerHIP(_SamplerBase):
    def make_schedule(self, ddim_num_steps, ddim_discretize='uniform', ddim_eta=0., verbose=True):
        if ddim_eta != 0:
            raise ValueError('ddim_eta must be 0 for PLMS')
        self._tables(ddim_num_steps, ddim_discretize, ddim_eta)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, **kwargs):
        if conditioning is not None and not isinstance(conditioning, dict) and conditioning.shape[0] != batch_size:
            print(f'Warning: Got {conditioning.shape[0]} conditionings but batch-size is {batch_size}')
        if isinstance(conditioning, dict):
            raise NotImplementedError('dict conditioning (hybrid models) is outside the SD-v1 txt2img path')
        if score_corrector is not None or quantize_x0:
            raise NotImplementedError('score_corrector / quantize_x0 are not used by SD v1 and not implemented here')
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        print(f'Data shape for PLMS sampling is {size}')
        return self.plms_sampling(conditioning, size, callback=callback, img_callback=img_callback, mask=mask, x0=x0,
                                  x_T=x_T, log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning)

    @torch.no_grad()
    def plms_sampling(self, cond, shape, x_T=None, callback=None, img_callback=None, mask=None, x0=None,
                      log_every_t=100, unconditional_guidance_scale=1., unconditional_conditioning=None):
        scale, uc = unconditional_guidance_scale, unconditional_conditioning
        device, b, img, cfg, c_in, x_in = self._prepare(cond, shape, x_T, uc, scale)
        plan = plms_plan(self.ddim_timesteps)
        total_steps = len(plan)
        print(f'Running PLMS Sampling with {total_steps} timesteps')
        intermediates = {'x_inter': [img], 'pred_x0': [img]}
        old_eps = []            # newest first
        unet = self._hip_unet()
        if unet is not None:
            unet.pin_context(c_in)
            unet.cache_timesteps([t for p in plan for t in (p[2], p[3]) if t is not None])
        try:
            img = self._plms_loop(plan, total_steps, device, b, img, cfg, c_in, x_in, scale, mask, x0, callback,
                                  img_callback, log_every_t, intermediates, old_eps)
        finally:
            if unet is not None:
                unet.unpin_context()
        return img, intermediates

    def _plms_loop(self, plan, total_steps, device, b, img, cfg, c_in, x_in, scale, mask, x0, callback, img_callback,
                   log_every_t, intermediates, old_eps):
        for (i, index, t, t_next, mode) in tqdm(plan, desc='PLMS Sampler', total=total_steps):
            if mask is not None:
                assert x0 is not None
                ts = torch.full((b,), t, device=device, dtype=torch.long)
                img = (self.model.q_sample(x0, ts) * mask + (1. - mask) * img).float().contiguous()
            eps = self._model_eps(x_in, img, t, c_in, cfg, b)
            e_t = torch.empty_like(img)
            x_prev = torch.empty_like(img)
            pred_x0 = torch.empty_like(img)
            if mode == 4:
                # Pseudo Improved Euler: predictor with e_t, second evaluation at t_next, then (e_t + e_t_next)/2
                self._step(eps, cfg, scale, img, 0, [], index, e_t, x_prev)
                eps_next = self._model_eps(x_in, x_prev, t_next, c_in, cfg, b)
                x_prev2 = torch.empty_like(img)
                self._step(eps_next, cfg, scale, img, 4, [e_t], index, None, x_prev2, pred_x0)
                x_prev = x_prev2
            else:
                self._step(eps, cfg, scale, img, mode, old_eps, index, e_t, x_prev, pred_x0)
            img = x_prev
            old_eps.insert(0, e_t)
            if len(old_eps) >= 4:
                old_eps.pop()
            if callback: callback(i)
            if img_callback: img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates['x_inter'].append(img)
                intermediates['pred_x0'].append(pred_x0)
        return img


class DDIMSamplerHIP(_SamplerBase):
    def make_schedule(self, ddim_num_steps, ddim_discretize='uniform', ddim_eta=0., verbose=True):
        self._tables(ddim_num_steps, ddim_discretize, ddim_eta)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, **kwargs):
        if conditioning is not None and not isinstance(conditioning, dict) and conditioning.shape[0] != batch_size:
            print(f'Warning: Got {conditioning.shape[0]} conditionings but batch-size is {batch_size}')
        if isinstance(conditioning, dict):
            raise NotImplementedError('dict conditioning (hybrid models) is outside the SD-v1 txt2img path')
        if score_corrector is not None or quantize_x0:
            raise NotImplementedError('score_corrector / quantize_x0 are not used by SD v1 and not implemented here')
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        print(f'Data shape for DDIM sampling is {size}, eta {eta}')
        return self._loop(conditioning, size, self.ddim_timesteps, x_T=x_T, callback=callback,
                          img_callback=img_callback, mask=mask, x0=x0, log_every_t=log_every_t,
                          temperature=temperature, noise_dropout=noise_dropout,
                          scale=unconditional_guidance_scale, uc=unconditional_conditioning, desc='DDIM Sampler')

    @torch.no_grad()
    def _loop(self, cond, shape, timesteps, x_T=None, callback=None, img_callback=None, mask=None, x0=None,
              log_every_t=100, temperature=1., noise_dropout=0., scale=1., uc=None, desc='DDIM Sampler'):
        device, b, img, cfg, c_in, x_in = self._prepare(cond, shape, x_T, uc, scale)
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        print(f'Running DDIM Sampling with {total_steps} timesteps')
        intermediates = {'x_inter': [img], 'pred_x0': [img]}
        unet = self._hip_unet()
        if unet is not None:
            unet.pin_context(c_in)
            unet.cache_timesteps([int(t) for t in time_range])
        try:
            img = self._ddim_loop(time_range, total_steps, device, b, img, cfg, c_in, x_in, scale, mask, x0, callback,
                                  img_callback, log_every_t, temperature, noise_dropout, intermediates, desc)
        finally:
            if unet is not None:
                unet.unpin_context()
        return img, intermediates

    def _ddim_loop(self, time_range, total_steps, device, b, img, cfg, c_in, x_in, scale, mask, x0, callback,
                   img_callback, log_every_t, temperature, noise_dropout, intermediates, desc):
        for i, step in enumerate(tqdm(time_range, desc=desc, total=total_steps)):
            index = total_steps - i - 1
            if mask is not None:
                assert x0 is not None
                ts = torch.full((b,), int(step), device=device, dtype=torch.long)
                img = (self.model.q_sample(x0, ts) * mask + (1. - mask) * img).float().contiguous()
            eps = self._model_eps(x_in, img, int(step), c_in, cfg, b)
            # ddim.py:200 draws noise_like() on EVERY step, also at eta = 0 where sigma_t = 0 and the term vanishes: draw it
            # too, so the device RNG stream stays where the reference's is (the x_T of a later n_iter comes from it);
            # the kernel only reads it when sigma_t != 0.  The reference multiplies sigma_t * noise first, then the
            # temperature: the step kernel takes the product noise * temperature (identical at the default temperature 1).
            noise = self._noise_like(img.shape, device) * temperature
            if noise_dropout > 0.:
                noise = torch.nn.functional.dropout(noise, p=noise_dropout)
            noise = noise.float().contiguous() if float(self._tab['sigmas'][index]) != 0.0 else None
            x_prev = torch.empty_like(img)
            pred_x0 = torch.empty_like(img)
            self._step(eps, cfg, scale, img, 0, [], index, None, x_prev, pred_x0, noise)
            img = x_prev
            if callback: callback(i)
            if img_callback: img_callback(pred_x0, i)
            if index % log_every_t == 0 or index == total_steps - 1:
                intermediates['x_inter'].append(img)
                intermediates['pred_x0'].append(pred_x0)
        return img

    @torch.no_grad()
    def stochastic_encode(self, x0, t, use_original_steps=False, noise=None):
        """ddim.py:206-220 (runs once per image; plain torch, not on the hot loop)."""
        if use_original_steps:
            raise NotImplementedError('use_original_steps is not used by scripts/img2img.py')
        a = torch.sqrt(self.ddim_alphas).to(x0.device)
        s = self.ddim_sqrt_one_minus_alphas.to(x0.device)
        if noise is None:
            noise = torch.randn_like(x0)
        sh = (t.shape[0],) + (1,) * (x0.dim() - 1)
        return a.gather(-1, t).reshape(sh) * x0 + s.gather(-1, t).reshape(sh) * noise

    @torch.no_grad()
    def decode(self, x_latent, cond, t_start, unconditional_guidance_scale=1.0, unconditional_conditioning=None,
               use_original_steps=False):
        """ddim.py:222-241: timesteps[:t_start], flipped."""
        if use_original_steps:
            raise NotImplementedError('use_original_steps is not used by scripts/img2img.py')
        img, _ = self._loop(cond, tuple(x_latent.shape), self.ddim_timesteps[:t_start], x_T=x_latent,
                            scale=unconditional_guidance_scale, uc=unconditional_conditioning, desc='Decoding image')
        return img


# ---- DPM-Solver++ (2M): scripts/txt2img.py --dpm_solver ----------------------------------------------
class _DiscreteVP:
    """Host-side subset of NoiseScheduleVP('discrete', alphas_cumprod=...) (dpm_solver.py:96-156): log alpha_t is the
    piecewise-linear interpolant of 0.5 log(alphas_cumprod) over t_n = n / N, n = 1..N; fp32 torch ops on the CPU, in the
    reference's order, so the step coefficients equal the reference's."""

    def __init__(self, alphas_cumprod):
        ac = alphas_cumprod.detach().to(torch.float32).cpu()
        self.N = ac.shape[0]
        self.t_array = torch.linspace(0., 1., self.N + 1)[1:]
        self.log_alpha_array = 0.5 * torch.log(ac)

    def log_alpha(self, t):
        xp, yp = self.t_array, self.log_alpha_array
        i = int(torch.clamp(torch.searchsorted(xp, t.reshape(1)) - 1, 0, self.N - 2))
        return yp[i] + (t - xp[i]) * (yp[i + 1] - yp[i]) / (xp[i + 1] - xp[i])

    def alpha(self, t):
        return torch.exp(self.log_alpha(t))

    def sigma(self, t):
        return torch.sqrt(1. - torch.exp(2. * self.log_alpha(t)))

    def lam(self, t):
        la = self.log_alpha(t)
        return la - 0.5 * torch.log(1. - torch.exp(2. * la))


def dpm_plan(ns, S):
    """Per model evaluation k = 0..S-1 (at time t_k, stepping to t_{k+1}; time-uniform from 1 to 1/N):
    (t_input, alpha_s, sigma_s, order, cx, a, inv_r0) for sdmi_dpm_solver_step -- DPM_Solver.sample(method='multistep',
    order=2, lower_order_final=True), dpm_solver.py:1068-1096."""
    if S < 2:
        raise ValueError('multistep DPM-Solver of order 2 needs at least 2 steps')      # `assert steps >= order`
    ts = torch.linspace(1., 1. / ns.N, S + 1)
    plan = []
    for k in range(S):
        s_, t_ = ts[k], ts[k + 1]
        step = k + 1
        order = 1 if step == 1 else (min(2, S + 1 - step) if S < 15 else 2)
        cx = ns.sigma(t_) / ns.sigma(s_)
        if order == 1:
            h = ns.lam(t_) - ns.lam(s_)
            a = ns.alpha(t_) * torch.expm1(-h)
            inv_r0 = torch.zeros(())
        else:
            l1, l0, lt = ns.lam(ts[k - 1]), ns.lam(s_), ns.lam(t_)
            h0, h = l0 - l1, lt - l0
            inv_r0 = 1. / (h0 / h)
            a = ns.alpha(t_) * (torch.exp(-h) - 1.)
        t_input = (s_ - 1. / ns.N) * 1000.
        plan.append((float(t_input), float(ns.alpha(s_)), float(ns.sigma(s_)), order, float(cx), float(a), float(inv_r0)))
    return plan


class DPMSolverSamplerHIP(_SamplerBase):
    """dpm_solver/sampler.py:9-82: `sample(...)` returns (x, None)."""

    def __init__(self, model, **kwargs):
        super().__init__(model, **kwargs)
        self.alphas_cumprod = model.alphas_cumprod.detach().to(torch.float32)

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None, img_callback=None,
               quantize_x0=False, eta=0., mask=None, x0=None, temperature=1., noise_dropout=0., score_corrector=None,
               corrector_kwargs=None, verbose=True, x_T=None, log_every_t=100, unconditional_guidance_scale=1.,
               unconditional_conditioning=None, **kwargs):
        if isinstance(conditioning, dict):
            raise NotImplementedError('dict conditioning (hybrid models) is outside the SD-v1 txt2img path')
        if conditioning is not None and conditioning.shape[0] != batch_size:
            print(f'Warning: Got {conditioning.shape[0]} conditionings but batch-size is {batch_size}')
        C, H, W = shape
        scale, uc = unconditional_guidance_scale, unconditional_conditioning
        device, b, img, cfg, c_in, x_in = self._prepare(conditioning, (batch_size, C, H, W), x_T, uc, scale)
        if not img.is_cuda:
            raise RuntimeError('the HIP sampler step runs on MI355X device tensors only (no CPU fallback)')
        plan = dpm_plan(_DiscreteVP(self.alphas_cumprod), S)
        lib = _lib.load()
        unet = self._hip_unet()
        if unet is not None:
            unet.pin_context(c_in)
        try:
            m_prev = None
            for k, (t_input, alpha_s, sigma_s, order, cx, a, inv_r0) in enumerate(plan):
                eps = self._model_eps(x_in, img, t_input, c_in, cfg, b, t_dtype=torch.float32)
                m_new, x_next = torch.empty_like(img), torch.empty_like(img)
                _lib.check(lib.sdmi_dpm_solver_step(eps.data_ptr(), int(cfg), float(scale), img.data_ptr(), _lib.ptr(m_prev),
                                                    alpha_s, sigma_s, cx, a, inv_r0, order, m_new.data_ptr(),
                                                    x_next.data_ptr(), img.numel(), _lib.stream_ptr()))
                m_prev, img = m_new, x_next
                if callback: callback(k)
                if img_callback: img_callback(m_new, k)
        finally:
            if unet is not None:
                unet.unpin_context()
        return img, None
