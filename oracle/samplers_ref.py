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


@torch.no_grad()
def plms_sample(apply_model, alphas_cumprod, S, x_T, c, scale=1.0, uc=None, record=None):
    """PLMSSampler.sample/plms_sampling/p_sample_plms, plms.py:57-236 (eta = 0)."""
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
                record=None):
    """DDIMSampler.sample/ddim_sampling/p_sample_ddim, ddim.py:56-204.
    `noises`: list of per-step noise tensors when eta > 0 (the reference draws them on device)."""
    ts = make_ddim_timesteps(S, len(alphas_cumprod))
    tabs = make_sampling_tables(alphas_cumprod, ts, eta)
    return _ddim_loop(apply_model, tabs, ts, x_T, c, scale, uc, noises, record)


def _ddim_loop(apply_model, tabs, ts, x, c, scale, uc, noises=None, record=None):
    time_range = np.flip(ts)
    total = ts.shape[0]
    b = x.shape[0]
    for i, step in enumerate(time_range):
        index = total - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
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
