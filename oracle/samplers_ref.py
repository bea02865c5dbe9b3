"""CPU restatement of the reference schedules and PLMS / DDIM sampler loops.

`apply_model(x, t, c)` is any callable with the contract of
`LatentDiffusion.apply_model` (ldm/models/diffusion/ddpm.py:891-900,986-992):
x [B,4,h,w] fp32, t [B] int64, c [B,L,D] -> eps [B,4,h,w].

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import numpy as np
import torch


def make_alphas_cumprod(n_timestep=1000, linear_start=0.00085, linear_end=0.0120):
    """make_beta_schedule('linear') util.py:21-25 + DDPM.register_schedule ddpm.py:117-137:
    betas = linspace(sqrt(s), sqrt(e), n, float64)**2; alphas_cumprod = cumprod(1-betas) (float64 numpy),
    stored as float32 buffers."""
    betas = np.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=np.float64) ** 2
    ac = np.cumprod(1.0 - betas, axis=0)
    return betas.astype(np.float32), ac.astype(np.float32)


def make_ddim_timesteps(num_ddim, num_ddpm=1000):
    """util.py:46-60, 'uniform': arange(0, N, N // S) + 1 (S=30 gives 31 steps; reference quirk)."""
    c = num_ddpm // num_ddim
    return np.asarray(list(range(0, num_ddpm, c))) + 1


def make_sampling_tables(alphas_cumprod, ddim_timesteps, eta=0.0):
    """util.py:63-74 + plms.py:44-51.  alphas_prev[0] = alphas_cumprod[0] (not 1.0)."""
    ac = np.asarray(alphas_cumprod)
    alphas = ac[ddim_timesteps]
    alphas_prev = np.asarray([ac[0]] + ac[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    return dict(alphas=alphas, alphas_prev=alphas_prev, sigmas=sigmas,
                sqrt_one_minus_alphas=np.sqrt(1.0 - alphas))


def _cfg_eps(apply_model, x, t, c, scale, uc):
    """plms.py:178-186 / ddim.py:171-178: batch order [uncond, cond]."""
    if uc is None or scale == 1.0:
        return apply_model(x, t, c)
    e_u, e_c = apply_model(torch.cat([x] * 2), torch.cat([t] * 2), torch.cat([uc, c])).chunk(2)
    return e_u + scale * (e_c - e_u)


def _x_prev(x, e, tabs, index, noise=None, temperature=1.0):
    """plms.py:199-216 / ddim.py:189-204, evaluated the way the reference does:
    python-float table entries broadcast into fp32 tensors."""
    b = x.shape[0]
    a_t = torch.full((b, 1, 1, 1), float(tabs['alphas'][index]))
    a_prev = torch.full((b, 1, 1, 1), float(tabs['alphas_prev'][index]))
    sigma = torch.full((b, 1, 1, 1), float(tabs['sigmas'][index]))
    s1m = torch.full((b, 1, 1, 1), float(tabs['sqrt_one_minus_alphas'][index]))
    pred_x0 = (x - s1m * e) / a_t.sqrt()
    dir_xt = (1.0 - a_prev - sigma ** 2).sqrt() * e
    n = 0.0 if noise is None else sigma * noise * temperature
    return a_prev.sqrt() * pred_x0 + dir_xt + n, pred_x0


def _q_sample(alphas_cumprod, x0, step, noise):
    """DDPM.q_sample (ddpm.py:274-277) with the buffers register_schedule stores as fp32 (ddpm.py:134-136):
    sqrt(alphas_cumprod)[t] * x0 + sqrt(1 - alphas_cumprod)[t] * noise."""
    ac = np.asarray(alphas_cumprod, dtype=np.float64)
    a = float(np.sqrt(ac).astype(np.float32)[step])
    s = float(np.sqrt(1.0 - ac).astype(np.float32)[step])
    return a * x0 + s * noise


def _mask_blend(alphas_cumprod, img, step, mask, x0, q_noise):
    """plms.py:147-150 / ddim.py:130-133: img = q_sample(x0, ts) * mask + (1 - mask) * img."""
    if mask is None:
        return img
    return _q_sample(alphas_cumprod, x0, int(step), q_noise) * mask + (1. - mask) * img


@torch.no_grad()
def plms_sample(apply_model, alphas_cumprod, S, x_T, c, scale=1.0, uc=None, record=None, mask=None, x0=None, q_noises=None):
    """PLMSSampler.sample/plms_sampling/p_sample_plms, plms.py:57-236 (eta = 0).
    `mask` / `x0` / `q_noises`: the inpainting blend in front of every step (plms.py:147-150); q_noises[i] is the noise
    DDPM.q_sample draws at step i."""
    ts = make_ddim_timesteps(S, len(alphas_cumprod))
    tabs = make_sampling_tables(alphas_cumprod, ts, 0.0)
    time_range = np.flip(ts)
    total = ts.shape[0]
    b = x_T.shape[0]
    img = x_T
    old_eps = []
    for i, step in enumerate(time_range):
        index = total - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        t_next = torch.full((b,), int(time_range[min(i + 1, len(time_range) - 1)]), dtype=torch.long)
        img = _mask_blend(alphas_cumprod, img, step, mask, x0, None if q_noises is None else q_noises[i])
        e_t = _cfg_eps(apply_model, img, t, c, scale, uc)
        if len(old_eps) == 0:
            x_p, _ = _x_prev(img, e_t, tabs, index)
            e_next = _cfg_eps(apply_model, x_p, t_next, c, scale, uc)
            e_prime = (e_t + e_next) / 2
        elif len(old_eps) == 1:
            e_prime = (3 * e_t - old_eps[-1]) / 2
        elif len(old_eps) == 2:
            e_prime = (23 * e_t - 16 * old_eps[-1] + 5 * old_eps[-2]) / 12
        else:
            e_prime = (55 * e_t - 59 * old_eps[-1] + 37 * old_eps[-2] - 9 * old_eps[-3]) / 24
        img, pred_x0 = _x_prev(img, e_prime, tabs, index)
        old_eps.append(e_t)
        if len(old_eps) >= 4:
            old_eps.pop(0)
        if record is not None:
            record.append(img.clone())
    return img


@torch.no_grad()
def ddim_sample(apply_model, alphas_cumprod, S, x_T, c, scale=1.0, uc=None, eta=0.0, noises=None,
                record=None, mask=None, x0=None, q_noises=None):
    """DDIMSampler.sample/ddim_sampling/p_sample_ddim, ddim.py:56-204.
    `noises`: list of per-step noise tensors when eta > 0 (the reference draws them on device); `mask` / `x0` /
    `q_noises`: the inpainting blend in front of every step (ddim.py:130-133)."""
    ts = make_ddim_timesteps(S, len(alphas_cumprod))
    tabs = make_sampling_tables(alphas_cumprod, ts, eta)
    return _ddim_loop(apply_model, tabs, ts, x_T, c, scale, uc, noises, record,
                      blend=None if mask is None else (alphas_cumprod, mask, x0, q_noises))


def _ddim_loop(apply_model, tabs, ts, x, c, scale, uc, noises=None, record=None, blend=None):
    time_range = np.flip(ts)
    total = ts.shape[0]
    b = x.shape[0]
    for i, step in enumerate(time_range):
        index = total - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        if blend is not None:
            x = _mask_blend(blend[0], x, step, blend[1], blend[2], blend[3][i])
        e_t = _cfg_eps(apply_model, x, t, c, scale, uc)
        x, _ = _x_prev(x, e_t, tabs, index, None if noises is None else noises[i])
        if record is not None:
            record.append(x.clone())
    return x


def ddim_stochastic_encode(alphas_cumprod, S, x0, t_enc, noise):
    """DDIMSampler.stochastic_encode, ddim.py:206-220: table index t_enc into the S-step tables."""
    ts = make_ddim_timesteps(S, len(alphas_cumprod))
    tabs = make_sampling_tables(alphas_cumprod, ts, 0.0)
    a = float(np.sqrt(tabs['alphas'][t_enc]))
    s = float(tabs['sqrt_one_minus_alphas'][t_enc])
    return a * x0 + s * noise


@torch.no_grad()
def ddim_decode(apply_model, alphas_cumprod, S, x_latent, c, t_start, scale=1.0, uc=None, record=None):
    """DDIMSampler.decode, ddim.py:222-241: timesteps[:t_start] flipped (first t = ts[t_start-1])."""
    ts = make_ddim_timesteps(S, len(alphas_cumprod))
    tabs = make_sampling_tables(alphas_cumprod, ts, 0.0)
    return _ddim_loop(apply_model, tabs, ts[:t_start], x_latent, c, scale, uc, None, record)


# ------------------------------------------------------------------------------------------------------------------
# DPM-Solver++ (2M) as `scripts/txt2img.py --dpm_solver` runs it -- SURVEY.md 8 f-3
#   DPMSolverSampler.sample            ldm/models/diffusion/dpm_solver/sampler.py:20-82
#   NoiseScheduleVP('discrete')        dpm_solver.py:96-108,125-156 (+ interpolate_fn :1132-1171)
#   model_wrapper (classifier-free)    dpm_solver.py:278-296,321-346
#   DPM_Solver(predict_x0=True).sample(steps=S, skip_type='time_uniform', method='multistep', order=2,
#                                      lower_order_final=True)          dpm_solver.py:386-399,504-530,755-790,1068-1096
# ------------------------------------------------------------------------------------------------------------------
def _interp(x, xp, yp):
    """interpolate_fn (dpm_solver.py:1132-1171) for one channel: piecewise linear through (xp, yp), the outermost
    segments extended beyond the key points; x [N], xp / yp [K] (xp ascending)."""
    K = xp.shape[0]
    idx = torch.searchsorted(xp.contiguous(), x.contiguous(), right=False)       # number of key points < x
    lo = torch.clamp(idx - 1, 0, K - 2)
    x0, x1, y0, y1 = xp[lo], xp[lo + 1], yp[lo], yp[lo + 1]
    return y0 + (x - x0) * (y1 - y0) / (x1 - x0)


class DiscreteVPSchedule:
    """The members of NoiseScheduleVP(schedule='discrete', alphas_cumprod=...) the sampler touches."""

    def __init__(self, alphas_cumprod):
        ac = torch.as_tensor(np.asarray(alphas_cumprod), dtype=torch.float32)
        self.total_N = ac.shape[0]
        self.T = 1.0
        self.t_array = torch.linspace(0., 1., self.total_N + 1)[1:]
        self.log_alpha_array = 0.5 * torch.log(ac)

    def log_alpha(self, t):
        return _interp(t.reshape(-1), self.t_array, self.log_alpha_array)

    def alpha(self, t):
        return torch.exp(self.log_alpha(t))

    def sigma(self, t):
        return torch.sqrt(1. - torch.exp(2. * self.log_alpha(t)))

    def lam(self, t):
        la = self.log_alpha(t)
        return la - 0.5 * torch.log(1. - torch.exp(2. * la))


def dpm_solver_sample(apply_model, alphas_cumprod, S, x_T, c, scale=1.0, uc=None, record=None):
    """x after S steps of multistep DPM-Solver++(2M), time-uniform steps from t = 1 to t = 1/N."""
    ns = DiscreteVPSchedule(alphas_cumprod)
    B = x_T.shape[0]
    ts = torch.linspace(ns.T, 1. / ns.total_N, S + 1)
    assert S >= 2

    def model(x, t):                                   # data prediction with classifier-free guidance
        tv = t.expand(B)
        t_in = (tv - 1. / ns.total_N) * 1000.          # float model time, dpm_solver.py:284-285
        if record is not None:
            record.append(float(t_in[0]))
        if uc is None or scale == 1.0:
            e = apply_model(x, t_in, c)
        else:
            e_u, e_c = apply_model(torch.cat([x] * 2), torch.cat([t_in] * 2), torch.cat([uc, c])).chunk(2)
            e = e_u + scale * (e_c - e_u)
        a, s = ns.alpha(tv).view(-1, 1, 1, 1), ns.sigma(tv).view(-1, 1, 1, 1)
        return (x - s * e) / a

    def first(x, s_, t_, m):                           # dpm_solver_first_update, predict_x0 branch
        h = ns.lam(t_) - ns.lam(s_)
        return (ns.sigma(t_) / ns.sigma(s_)).view(-1, 1, 1, 1) * x - (ns.alpha(t_) * torch.expm1(-h)).view(-1, 1, 1, 1) * m

    def second(x, m1, m0, t1, t0, t_):                 # multistep_dpm_solver_second_update, predict_x0 / 'dpm_solver'
        l1, l0, lt = ns.lam(t1), ns.lam(t0), ns.lam(t_)
        h0, h = l0 - l1, lt - l0
        r0 = h0 / h
        D1 = (1. / r0).view(-1, 1, 1, 1) * (m0 - m1)
        a = (ns.alpha(t_) * (torch.exp(-h) - 1.)).view(-1, 1, 1, 1)
        return (ns.sigma(t_) / ns.sigma(t0)).view(-1, 1, 1, 1) * x - a * m0 - 0.5 * a * D1

    x = x_T
    t_prev = [ts[0].expand(B)]
    m_prev = [model(x, ts[0])]
    x = first(x, t_prev[0], ts[1].expand(B), m_prev[0])
    t_prev.append(ts[1].expand(B))
    m_prev.append(model(x, ts[1]))
    for step in range(2, S + 1):
        vt = ts[step].expand(B)
        order = min(2, S + 1 - step) if S < 15 else 2   # lower_order_final
        if order == 2:
            x = second(x, m_prev[0], m_prev[1], t_prev[0], t_prev[1], vt)
        else:
            x = first(x, t_prev[1], vt, m_prev[1])
        t_prev = [t_prev[1], vt]
        m_prev = [m_prev[1], model(x, ts[step]) if step < S else m_prev[1]]
    return x
